"""TEST INFRASTRUCTURE ONLY - imports the real reference (Python, CPU).

Works only where /root/reference exists (the build container).  The reference
needs two sys.path roots (``sequence_model.py:3`` imports
``speech_enhance.audio_zen...`` while everything else imports ``audio_zen...`` /
``utils.logger``) and a ``librosa`` module at import time
(``audio_zen/acoustics/feature.py:3``, ``mask.py:3``) that nothing on the
forward path calls.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FSNP_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "speech_enhance", "fullsubnet_plus"))


def _install_stub_modules():
    if "librosa" not in sys.modules:
        try:
            import librosa  # noqa: F401
        except Exception:
            stub = types.ModuleType("librosa")
            stub.util = types.ModuleType("librosa.util")
            stub.__dict__["__fsnp_stub__"] = True
            sys.modules["librosa"] = stub
            sys.modules["librosa.util"] = stub.util


def load_reference():
    """Return the reference's FullSubNet_Plus class (fullsubnet_plus.py:16)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_stub_modules()
    for p in (os.path.join(REFERENCE_ROOT, "speech_enhance"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from fullsubnet_plus.model.fullsubnet_plus import FullSubNet_Plus  # type: ignore
    return FullSubNet_Plus


def load_reference_fullsubnet():
    """Return the original FullSubNet ``Model`` class (fullsubnet/model/fullsubnet.py:12)."""
    load_reference()
    from fullsubnet.model.fullsubnet import Model  # type: ignore
    return Model


def reference_model_args():
    """[model.args] of the reference's config/inference.toml:29-44."""
    import tomli
    with open(os.path.join(REFERENCE_ROOT, "config", "inference.toml"), "rb") as f:
        return tomli.load(f)["model"]["args"]


# A frozen copy of config/inference.toml [model.args] so the GPU box (which has no
# /root/reference) builds the same network.  tests/test_oracle.py asserts it equals
# reference_model_args() whenever the reference is present.
DEFAULT_MODEL_ARGS = dict(
    sb_num_neighbors=15,
    fb_num_neighbors=0,
    num_freqs=257,
    look_ahead=2,
    sequence_model="LSTM",
    fb_output_activate_function="ReLU",
    sb_output_activate_function=False,
    channel_attention_model="TSSE",
    fb_model_hidden_size=512,
    sb_model_hidden_size=384,
    weight_init=False,
    norm_type="offline_laplace_norm",
    num_groups_in_drop_band=2,
    kersize=[3, 5, 10],
    subband_num=1,
)


# [model.args] of the original FullSubNet as the reference ships them (commented alternative of
# config/inference.toml:11,28 -> recipes' fullsubnet config): same values as above minus the FullSubNet+-only keys.
FULLSUBNET_MODEL_ARGS = dict(
    sb_num_neighbors=15,
    fb_num_neighbors=0,
    num_freqs=257,
    look_ahead=2,
    sequence_model="LSTM",
    fb_output_activate_function="ReLU",
    sb_output_activate_function=False,
    fb_model_hidden_size=512,
    sb_model_hidden_size=384,
    weight_init=False,
    norm_type="offline_laplace_norm",
    num_groups_in_drop_band=2,
)
