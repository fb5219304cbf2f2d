"""GPU parity tests of SURVEY.md 8(f-2): the original FullSubNet ``Model``
(speech_enhance/fullsubnet/model/fullsubnet.py:12-118) on the HIP kernels, through the C ABI, against the golden
vectors the REAL reference produced (tests/golden/fsn_*.npz, oracle/make_golden.py) and the torch-CPU oracle
(oracle/fsnp_torch.forward_fullsubnet) at the benchmark size.  Tolerance: 1e-3 rel (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest
import torch

from fullsubnet_plus_amd import FullSubNet
from oracle import fsnp_torch
from oracle.ref_loader import FULLSUBNET_MODEL_ARGS
from oracle.weights import make_inputs, make_state_dict_fullsubnet
from tests._util import Golden, golden_names, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}

torch.set_num_threads(min(16, os.cpu_count() or 1))


def _record(name, **kw):
    REPORT[name] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kw.items()}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report_fullsubnet.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _model(args, sd, mode="parity"):
    m = FullSubNet(**args)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    m.batch_mode = mode
    return m


def _cuda(t):
    g = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device="cuda")
    g.copy_(t)
    return g


@pytest.mark.parametrize("name", golden_names("fullsubnet"))
def test_fullsubnet_forward_vs_reference_golden(name):
    g = Golden(name)
    m = _model(g.args, g.state_dict(), "parity")
    mag = g.inputs()[0]
    B, T = mag.shape[0], mag.shape[-1]
    out = m(_cuda(mag)).cpu().numpy()
    m.check_errors()
    want = g.arrays["out"]
    assert out.shape == want.shape
    err, err64 = rel_err(out, want), rel_err(out, g.arrays["out64"])
    rec = dict(rel_vs_ref32=err, rel_vs_ref64=err64, ref32_vs_ref64=rel_err(want, g.arrays["out64"]))
    if "stage_fb_mag" in g.arrays:            # full-band LSTM + Linear + ReLU output, [B,F,T'] (forward hook on fb_model)
        fb = m.read_stage("fb_mag", B, T).permute(0, 2, 1).numpy()
        rec["fb_stage"] = rel_err(fb, g.arrays["stage_fb_mag"])
    _record(f"forward_{name}", **rec)
    assert err < TOL, rec
    if "fb_stage" in rec:
        assert rec["fb_stage"] < 2e-4, rec
    if "full" in g.arrays:
        m.batch_mode = "full"
        full = m(_cuda(mag)).cpu().numpy()
        errf = rel_err(full, g.arrays["full"])
        _record(f"forward_full_{name}", rel_vs_ref32=errf)
        assert errf < TOL, errf


@pytest.mark.parametrize("batch", [32, 40])
def test_fullsubnet_batch_vs_oracle(batch):
    """Benchmark-sized call (one / two full-band row tiles, all bins) vs the oracle, + bitwise repeatability."""
    sd = make_state_dict_fullsubnet(11, "harsh")
    mag = make_inputs(batch, 1.0 if batch > 32 else 2.0, 31)[0]
    m = _model(dict(FULLSUBNET_MODEL_ARGS), sd, "full")
    x = _cuda(mag)
    out = m(x)
    out2 = m(x)
    m.check_errors()
    assert torch.equal(out, out2)
    kw = {k: FULLSUBNET_MODEL_ARGS[k] for k in ("look_ahead", "sb_num_neighbors", "fb_num_neighbors", "norm_type",
                                                "num_groups_in_drop_band", "fb_output_activate_function",
                                                "sb_output_activate_function")}
    pick = [0, batch // 2, batch - 1]          # the oracle is slow: check three utterances of the batch
    want = fsnp_torch.forward_fullsubnet_full(sd, mag[pick], **kw).numpy()
    err = rel_err(out[pick].cpu().numpy(), want)
    _record(f"fullsubnet_b{batch}_full_vs_oracle", rel=err)
    assert err < TOL, err
    # parity mode rows are a sub-selection of the full rows (drop_band, feature.py:254-285)
    m.batch_mode = "parity"
    par = m(x).cpu().numpy()
    full = out.cpu().numpy()
    n0 = (batch + 1) // 2
    for r in (0, 1, n0 - 1, n0, batch - 1):
        s, p = (2 * r, 0) if r < n0 else (2 * (r - n0) + 1, 1)
        assert np.abs(par[r] - full[s][:, p:256:2, :]).max() <= 1e-6 * np.abs(full).max()


@pytest.mark.parametrize("batch,seconds", [(1, 2.0), (2, 0.7), (3, 1.0), (4, 0.5), (1, 30.0)])
def test_fullsubnet_small_batches_run_the_full_band_lstm_on_the_valu(batch, seconds):
    """csrc/lstm_fbv.hip (round 5): with 1 ... 4 utterances the full-band LSTM(257 -> 512 x 2) of fullsubnet.py:39-47 is a matrix-VECTOR
    product per step - 64 workgroups x 8 units, weights resident, plain FMAs, one hand-off per step - instead of 32-row MFMA tiles with
    one live row.  Against the oracle, against the K-split kernel it replaces (debug mode 2 keeps that one), bitwise repeatable, under
    drift injection; 1 ... 4 rows (NB = 1, 2, 4 instantiations, a padded row at B = 3) and a 30 s clip (1,877 steps)."""
    sd = make_state_dict_fullsubnet(14, "harsh")
    mag = make_inputs(batch, seconds, 60 + batch)[0]
    m = _model(dict(FULLSUBNET_MODEL_ARGS), sd, "full")
    x = _cuda(mag)
    out = m(x)
    m.check_errors()
    assert torch.equal(m(x), out)
    T = mag.shape[-1]
    fb_valu = m.read_stage("fb_mag", batch, T).numpy()
    for seed in (3, 99):
        m.debug_set_chaos(seed)
        assert torch.equal(m(x), out), seed
    m.debug_set_chaos(0)
    m.debug_set_lstm_coop(2)                       # the round-4 kernels: K-split full-band LSTM, serial schedules
    ref = m(x)
    m.check_errors()
    fb_mfma = m.read_stage("fb_mag", batch, T).numpy()
    m.debug_set_lstm_coop(1)
    assert rel_err(fb_valu, fb_mfma) < 1e-5 and not np.array_equal(fb_valu, fb_mfma)      # (another kernel really ran)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    if seconds <= 2.0:
        kw = {k: FULLSUBNET_MODEL_ARGS[k] for k in ("look_ahead", "sb_num_neighbors", "fb_num_neighbors", "norm_type",
                                                    "num_groups_in_drop_band", "fb_output_activate_function",
                                                    "sb_output_activate_function")}
        want = fsnp_torch.forward_fullsubnet_full(sd, mag, **kw).numpy()
        err = rel_err(out.cpu().numpy(), want)
        _record(f"fullsubnet_valu_b{batch}", rel=err, fb_valu_vs_mfma=rel_err(fb_valu, fb_mfma))
        assert err < TOL, err


@pytest.mark.parametrize("batch", [32, 16])
def test_fullsubnet_pipelined_loop_is_bit_identical(batch):
    """fsnp_set_pipeline on the original FullSubNet: the deferred remainder chunk of forward i (side stream) and the full-band LSTM of
    forward i + 1 (caller's stream) are both column-split launches.  Round 6 lets them run side by side (csrc/fsnp_abi.hip
    launch_coop_chained: launches of ONE handle whose workgroups fit the chip together are not chained - B = 32: 29.7 -> 28.5 ms per
    forward); the masks of a back-to-back loop over different inputs must be the plain call's, bit for bit, and no hand-off may time out."""
    sd = make_state_dict_fullsubnet(0, "default")
    m = _model(FULLSUBNET_MODEL_ARGS, sd, "full")
    m.error_check = "deferred"
    batches = [make_inputs(batch, 0.5, 700 + i)[0].cuda() for i in range(3)]
    plain = [m(b).clone() for b in batches]
    torch.cuda.synchronize()
    assert any(c["deferred_when_pipelined"] for c in m.describe_plan(batch)), m.describe_plan(batch)
    m.set_pipeline(True)
    piped = [m(b) for b in batches * 4]
    m.flush()
    torch.cuda.synchronize()
    m.poll_errors()
    for a, b in zip(piped, plain * 4):
        assert torch.equal(a, b)
    m.set_pipeline(False)
    assert torch.equal(m(batches[1]), plain[1])
    m.check_errors()


def test_fullsubnet_enhance_epilogue():
    sd = make_state_dict_fullsubnet(12, "default")
    mag, real, imag = make_inputs(2, 1.0, 32)
    X = torch.complex(real[:, 0], imag[:, 0])
    m = _model(dict(FULLSUBNET_MODEL_ARGS), sd, "parity")
    Xg = torch.empty_strided(X.shape, X.stride(), dtype=X.dtype, device="cuda")
    Xg.copy_(X)
    got = m.enhance(Xg).cpu()
    kw = {k: FULLSUBNET_MODEL_ARGS[k] for k in ("look_ahead", "sb_num_neighbors", "fb_num_neighbors", "norm_type",
                                                "num_groups_in_drop_band", "fb_output_activate_function",
                                                "sb_output_activate_function")}
    mask = fsnp_torch.forward_fullsubnet_full(sd, X.abs().unsqueeze(1), **kw)
    want = fsnp_torch.apply_cirm(mask, X)
    err = float((got - want).abs().max() / want.abs().max())
    _record("fullsubnet_enhance", rel=err)
    assert err < TOL


def test_fullsubnet_rejects_oversized_batch():
    m = _model(dict(FULLSUBNET_MODEL_ARGS), make_state_dict_fullsubnet(0), "full")
    x = torch.rand(513, 1, 257, 9, device="cuda")
    with pytest.raises(RuntimeError, match="split the batch"):
        m(x)


def test_fullsubnet_enhance_wave_vs_oracle():
    from oracle.weights import make_wave
    sd = make_state_dict_fullsubnet(13, "harsh")
    m = _model(dict(FULLSUBNET_MODEL_ARGS), sd)
    wav = torch.from_numpy(make_wave(2, 1.0, 501))
    got = m.enhance_wave(wav.cuda()).cpu()
    want = fsnp_torch.enhance_wave(sd, wav, fullsubnet=True)
    err = float((got - want).abs().max() / want.abs().max())
    _record("fullsubnet_enhance_wave", rel=err)
    assert err < TOL, err
