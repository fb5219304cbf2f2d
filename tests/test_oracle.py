"""CPU tests: pin the oracle restatements (oracle/fsnp_numpy.py, oracle/fsnp_torch.py)
against the golden vectors that oracle/make_golden.py produced from the REAL reference."""
import numpy as np
import pytest
import torch

from oracle import fsnp_numpy, fsnp_torch, ref_loader
from tests._util import Golden, golden_names, rel_err

SMALL = [n for n in golden_names() if "2s" not in n and "10s" not in n and "_att_" not in n and "_eca_" not in n and "_cbam_" not in n and "gru" not in n and "tcn" not in n and "fbn" not in n]


@pytest.mark.parametrize("name", golden_names())
def test_torch_port_matches_reference(name):
    g = Golden(name)
    mag, real, imag = g.inputs()
    out = fsnp_torch.forward(g.state_dict(), mag, real, imag, **g.fwd_kwargs()).numpy()[:, :, ::g.sub, :]
    assert out.shape == g.arrays["out"].shape
    # same ATen kernels as the reference -> agreement far inside the 1e-3 budget
    assert rel_err(out, g.arrays["out"]) < 2e-5
    if "full" in g.arrays:
        full = fsnp_torch.forward_full(g.state_dict(), mag, real, imag, **g.fwd_kwargs()).numpy()
        assert rel_err(full[:, :, ::g.sub, :], g.arrays["full"]) < 2e-5


@pytest.mark.parametrize("name", SMALL)
def test_numpy_fp64_matches_reference_fp64(name):
    g = Golden(name)
    mag, real, imag = (a.numpy() for a in g.inputs())
    sd = {k: v.numpy() for k, v in g.state_dict().items()}
    stages = {}
    kw = {k: v for k, v in g.fwd_kwargs().items() if k not in ("channel_attention_model", "subband_num")}
    out = fsnp_numpy.forward(sd, mag, real, imag, dtype=np.float64, stages=stages, **kw)
    # out64 is stored as fp32 -> 6e-8 quantisation
    assert rel_err(out, g.arrays["out64"]) < 5e-7
    for tag in ("att_mag", "att_real", "att_imag", "fb_mag", "fb_real", "fb_imag"):
        if "stage_" + tag in g.arrays:
            assert rel_err(stages[tag], g.arrays["stage_" + tag]) < 2e-4, tag


@pytest.mark.parametrize("name", ["b4_t16_default", "b5_t16_default", "b3_t20_harsh"])
def test_parity_mode_is_subselection_of_full(name):
    """out_parity[r] = out_full[s][:, p:256:2, :]."""
    g = Golden(name)
    out, full = g.arrays["out"], g.arrays["full"]
    B = full.shape[0]
    n0 = (B + 1) // 2
    scale = np.abs(full).max()
    for r in range(B):
        s, p = (2 * r, 0) if r < n0 else (2 * (r - n0) + 1, 1)
        assert np.abs(out[r] - full[s][:, p:256:2, :]).max() < 2e-4 * scale


def test_batch_two_raises_like_reference():
    g = Golden("b4_t16_default")
    mag, real, imag = g.inputs()
    with pytest.raises(AssertionError):
        fsnp_torch.forward(g.state_dict(), mag[:2], real[:2], imag[:2], **g.fwd_kwargs())


def test_frozen_model_args_match_reference_toml():
    if not ref_loader.reference_available():
        pytest.skip("reference not present on this box")
    assert ref_loader.reference_model_args() == ref_loader.DEFAULT_MODEL_ARGS


def test_state_dict_matches_reference_parameter_tree():
    if not ref_loader.reference_available():
        pytest.skip("reference not present on this box")
    cls = ref_loader.load_reference()
    model = cls(**ref_loader.DEFAULT_MODEL_ARGS)
    ref_sd = model.state_dict()
    from oracle.weights import make_state_dict
    sd = make_state_dict(0)
    assert set(sd) == set(ref_sd)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
    assert sum(v.numel() for v in sd.values()) == 8_675_102


# ---------------------------------------------------------------- the original FullSubNet
@pytest.mark.parametrize("name", golden_names("fullsubnet"))
def test_fullsubnet_torch_port_matches_reference(name):
    g = Golden(name)
    mag = g.inputs()[0]
    stages = {}
    out = fsnp_torch.forward_fullsubnet(g.state_dict(), mag, stages=stages, **g.fwd_kwargs()).numpy()
    assert out.shape == g.arrays["out"].shape
    assert rel_err(out, g.arrays["out"]) < 2e-5
    if "full" in g.arrays:
        full = fsnp_torch.forward_fullsubnet_full(g.state_dict(), mag, **g.fwd_kwargs()).numpy()
        assert rel_err(full, g.arrays["full"]) < 2e-5
    if "stage_fb_mag" in g.arrays:
        assert rel_err(stages["fb_mag"].numpy(), g.arrays["stage_fb_mag"]) < 2e-5


def test_fullsubnet_state_dict_matches_reference_parameter_tree():
    if not ref_loader.reference_available():
        pytest.skip("reference not mounted")
    from oracle.weights import make_state_dict_fullsubnet
    ref = ref_loader.load_reference_fullsubnet()(**ref_loader.FULLSUBNET_MODEL_ARGS)
    sd = make_state_dict_fullsubnet(0)
    rsd = ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(rsd[k].shape), k


def test_committed_golden_is_what_the_generator_produces():
    """tests/golden/*.npz are outputs of the REAL reference produced by oracle/make_golden.py: re-running the generator
    for one small case here (reference mounted) reproduces the committed arrays (up to thread-count rounding)."""
    if not ref_loader.reference_available():
        pytest.skip("reference not mounted")
    from oracle import make_golden
    case = next(c for c in make_golden.CASES if c["name"] == "b1_t8_min")
    payload, _ = make_golden.run_case(case, ref_loader.load_reference())
    g = Golden("b1_t8_min")
    for key in ("out", "out64"):
        assert rel_err(payload[key], g.arrays[key]) < 1e-6, key
    fcase = next(c for c in make_golden.FSN_CASES if c["name"] == "fsn_b1_t20_gaussian")
    fpayload, _ = make_golden.run_case_fsn(fcase, ref_loader.load_reference_fullsubnet())
    assert rel_err(fpayload["out"], Golden("fsn_b1_t20_gaussian").arrays["out"]) < 1e-6


def test_helper_restatements_match_the_reference_base_model():
    """oracle/fsnp_torch.py NORMS / unfold - the checkers of `model.norm` / `model.unfold` (fsnp_norm, fsnp_unfold) - against the reference's
    own BaseModel static functions (audio_zen/model/base_model.py:15-47, 210-316), on the shapes the GPU test uses."""
    if not ref_loader.reference_available():
        pytest.skip("reference not mounted")
    import numpy as np
    import torch
    ref_loader.load_reference()
    from audio_zen.model.base_model import BaseModel  # type: ignore
    rng = np.random.Generator(np.random.PCG64(31))
    for shape in [(3, 1, 257, 20), (2, 3, 34, 61), (1, 2, 5, 300)]:
        x = torch.from_numpy(np.abs(rng.standard_normal(shape)).astype(np.float32) + 0.1)
        for name, fn in fsnp_torch.NORMS.items():
            want = getattr(BaseModel, name)(x)
            got = fn(x)
            assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), name
    for shape, nb in [((2, 1, 257, 12), 15), ((2, 3, 40, 7), 1), ((1, 2, 9, 5), 0), ((2, 1, 33, 4), 8)]:
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        assert torch.equal(fsnp_torch.unfold(x, nb), BaseModel.unfold(x, nb))
