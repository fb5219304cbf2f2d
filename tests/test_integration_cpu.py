"""The drop-in boundary exercised with the REFERENCE's own plugin loader (needs /root/reference; CPU only):
`initialize_module(config["model"]["path"], args=...)` (audio_zen/utils.py:63-99, called from
base_inferencer.py:99) must find the HIP model class from config/inference_hip.toml, and a checkpoint written
from the reference model's state_dict must load with strict=True (base_inferencer.py:100-107)."""
import os
import sys

import pytest
import tomli
import torch

from oracle import ref_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference not present on this box")


def test_only_model_path_differs_from_reference_toml():
    ours = tomli.load(open(os.path.join(ROOT, "config", "inference_hip.toml"), "rb"))
    ref = tomli.load(open(os.path.join(ref_loader.REFERENCE_ROOT, "config", "inference.toml"), "rb"))
    assert ours["model"]["path"] == "fullsubnet_plus_amd.model.FullSubNet_Plus"
    ours["model"]["path"] = ref["model"]["path"]
    assert ours == ref


def test_reference_plugin_loader_builds_hip_model_and_loads_reference_checkpoint(tmp_path):
    RefModel = ref_loader.load_reference()
    from audio_zen.utils import initialize_module          # the reference's own loader
    cfg = tomli.load(open(os.path.join(ROOT, "config", "inference_hip.toml"), "rb"))
    model = initialize_module(cfg["model"]["path"], args=cfg["model"]["args"])
    from fullsubnet_plus_amd import FullSubNet_Plus
    assert isinstance(model, FullSubNet_Plus)
    torch.manual_seed(0)
    ref = RefModel(**cfg["model"]["args"])
    ckpt = tmp_path / "rand_ckpt.tar"
    torch.save({"model": ref.state_dict(), "epoch": 0}, ckpt)
    state = torch.load(ckpt, map_location="cpu")
    model.load_state_dict(state["model"])                  # strict, exactly base_inferencer.py:107
    model.to("cpu").eval()
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v)


def test_shims_cover_what_the_reference_cli_imports():
    sys.path.insert(0, os.path.join(ROOT, "shims"))
    try:
        import importlib
        toml = importlib.import_module("toml")
        d = toml.load(os.path.join(ROOT, "config", "inference_hip.toml"))
        assert toml.loads(toml.dumps(d)) == d
        sf = importlib.import_module("soundfile")
        assert hasattr(sf, "write")
    finally:
        sys.path.remove(os.path.join(ROOT, "shims"))
        for m in ("toml", "soundfile"):
            sys.modules.pop(m, None)


def test_reference_plugin_loader_builds_hip_fullsubnet_and_loads_reference_checkpoint(tmp_path):
    """config/inference_fullsubnet_hip.toml = the reference TOML with its own commented
    alternatives switched on (inference.toml:11,28) and [model].path pointing here."""
    RefModel = ref_loader.load_reference_fullsubnet()
    from audio_zen.utils import initialize_module
    cfg = tomli.load(open(os.path.join(ROOT, "config", "inference_fullsubnet_hip.toml"), "rb"))
    assert cfg["model"]["args"] == ref_loader.FULLSUBNET_MODEL_ARGS
    assert cfg["inferencer"]["type"] == "full_band_crm_mask"
    model = initialize_module(cfg["model"]["path"], args=cfg["model"]["args"])
    from fullsubnet_plus_amd import FullSubNet
    assert isinstance(model, FullSubNet)
    torch.manual_seed(0)
    ref = RefModel(**cfg["model"]["args"])
    ckpt = tmp_path / "rand_ckpt.tar"
    torch.save({"model": ref.state_dict(), "epoch": 0}, ckpt)
    model.load_state_dict(torch.load(ckpt, map_location="cpu")["model"])     # strict (base_inferencer.py:107)
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v)
