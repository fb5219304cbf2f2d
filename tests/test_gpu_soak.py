"""Long-recurrence and drift soaks of the column-split recurrent kernels (run LAST: tests/conftest.py orders the GPU suite).

Round 3 ended with one wrong answer on the driver's box: B = 3 x 126-second clips (T' = 7,878 steps - by accident: a
parametrisation written in frames was passed as seconds), default kernels, 0.35 rel on the 4th forward of a handle whose
first three forwards were right.  It never reproduced (round 4: 550 forwards of that exact shape on ten boxes, back to back
and behind 17 s of host-busy / GPU-idle time, all bit-identical - DESIGN.md section 5), so what this file pins is what CAN be
pinned: every column-split family at >= 8,000 steps, bit-repeatable and against the oracle, and every family with its
workgroups forced to drift apart by whole steps (fsnp_debug_set_chaos): the hand-off protocols may not depend on lockstep."""
import os

import numpy as np
import pytest
import torch

from fullsubnet_plus_amd import FullSubNet_Plus
from oracle import fsnp_torch
from oracle.ref_loader import DEFAULT_MODEL_ARGS
from oracle.weights import make_inputs, make_state_dict
from tests._util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3
torch.set_num_threads(min(16, os.cpu_count() or 1))   # the CPU oracle collapses when oversubscribed (256 hardware threads on the GPU boxes:
                                                      # run alone, without test_gpu_parity.py's identical line, this file took 395 s instead of 30)

# sequences -> launches of the default plan on 256 CUs (tests/test_host.py::_plan), round 5:
#   20 one row tile, layer-skewed K split @ 8 units | 131 five tiles @ 8 | 257 half-tile ping-pong | 514 wave-owned split @ 32 (17 tiles)
#   771 = 21 tiles wave-owned @ 32 + 4 @ 8 | 1028 wave-owned @ 64 (33 tiles) | 1285 wave-owned @ 64 (41)
#   2056 = 42 tiles wave-owned @ 64 + 21 @ 32 + 2 @ 8 | 2700 = 42 + 42 @ 64 + 1 @ 8
# and with the wave-owned split priced out (schedule "ksplit" / "serial": the round-4 plans, which GRU models and the other hidden sizes
# still run): 514 layer-skewed K split @ 32 | 771 = 640 @ 32 + 131 @ 8 (the B = 3 plan of the round-3 failure) | 1028 = @ 32 + half-tile
# ping-pong + @ 8 | 1285 layer-skewed @ 64 | 2056 three-way column split, one row tile per group | 2700 two row tiles per group
FAMILIES = [20, 131, 257, 514, 771, 1028, 1285, 2056, 2700]


def _round4_plans(m):
    """Price the wave-owned column split (csrc/lstm_coopw.hip) out of the handle's plans: the built-in table's first 21 values."""
    m.debug_set_costs(m.planner_costs_raw()[:21], 1)


def _model(sd, mode="full"):
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    m.batch_mode = mode
    return m


def _cuda(ts):
    out = []
    for t in ts:
        g = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device="cuda")
        g.copy_(t)
        out.append(g)
    return out


def _dense_input(n, steps, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn((n, 34, steps), generator=g, device="cuda", dtype=torch.float32)


def _sample_rows(n, count=20):
    return sorted(set(list(range(0, n, max(1, n // count))) + [n - 1]))


def _first_difference(a, b):
    """a, b: [n, 2, steps] -> (rows that differ, first / last differing step): which row tile broke, and when."""
    d = (a != b)
    rows = torch.nonzero(d.any(dim=2).any(dim=1)).flatten().tolist()
    steps = torch.nonzero(d.any(dim=0).any(dim=0)).flatten().tolist()
    return {"rows": rows[:8], "n_rows": len(rows), "tiles_of_32": sorted(set(r // 32 for r in rows))[:12],
            "first_step": steps[0] if steps else None, "last_step": steps[-1] if steps else None}


@pytest.mark.parametrize("n", FAMILIES)
@pytest.mark.parametrize("schedule", ["default", "ksplit", "serial"])
def test_column_split_kernels_under_drift(n, schedule):
    """Every column-split kernel family with pseudo-random per-workgroup delays at its phase boundaries (3 ... 24 us on one
    boundary in eight, ~200 us once in 1024: drifts of many whole steps; csrc/lstm_common.h chaos_delay): bit-identical to the
    undisturbed run for three seeds, and that run matches the oracle.  A buffer re-used while a slow peer still reads it, or
    a counter target off by one phase, fails here within a few hundred steps instead of once in a hundred long forwards."""
    if schedule == "serial" and n in (257, 2056, 2700):
        pytest.skip("the serial K-split schedule (debug mode 2) only differs for K-split launches")
    if schedule == "ksplit" and n in (20, 131, 257):
        pytest.skip("the default plan of these sizes has no wave-owned launch to price out")
    sd = make_state_dict(9, "harsh")
    m = _model(sd)
    m.lstm2_fc(_dense_input(1, 2, 1))
    m.debug_set_lstm_coop(2 if schedule == "serial" else 1)
    if schedule != "default":
        _round4_plans(m)
    steps = 160 if n >= 1285 else 320
    x = _dense_input(n, steps, 4000 + n)
    quiet = m.lstm2_fc(x)
    m.check_errors()
    sel = _sample_rows(n)
    want = fsnp_torch.lstm2_fc(x[sel].cpu(), sd).numpy()
    assert rel_err(quiet[sel].cpu().numpy(), want) < 2e-5
    for seed in (1, 7, 1234567):
        m.debug_set_chaos(seed)
        got = m.lstm2_fc(x)
        m.check_errors()
        assert torch.equal(got, quiet), (seed, _first_difference(got, quiet))
    m.debug_set_chaos(0)
    assert torch.equal(m.lstm2_fc(x), quiet)


@pytest.mark.parametrize("n,steps,plan", [(20, 8192, "default"), (131, 8192, "default"), (257, 8192, "default"), (514, 8192, "default"),
                                          (771, 8192, "default"), (1028, 8192, "default"), (1285, 8192, "default"), (2056, 8192, "default"),
                                          (2700, 8192, "default"), (514, 8192, "round4"), (771, 8192, "round4"), (1285, 8192, "round4"),
                                          (2056, 8192, "round4"), (2700, 8192, "round4")])
def test_long_recurrence_kernels(n, steps, plan):
    """>= 8k steps on every column-split family (the longest recurrence of the round-3 suite was 300 steps outside one
    accidental 126-second case): five runs bit-identical, a sample of rows against torch.lstm on all 8192 steps.  "round4": the plans
    without the wave-owned column split (layer-skewed K split at 32 / 64 units, three-way split)."""
    sd = make_state_dict(9, "default")
    m = _model(sd)
    if plan == "round4":
        m.lstm2_fc(_dense_input(1, 2, 1))
        _round4_plans(m)
    x = _dense_input(n, steps, 900 + n)
    first = m.lstm2_fc(x)
    m.check_errors()
    for rep in range(4):
        again = m.lstm2_fc(x)
        m.check_errors()
        assert torch.equal(again, first), (rep, _first_difference(again, first))
    sel = _sample_rows(n, 12)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                 # 8192 dependent steps of tiny matrix products: a thread pool only adds hand-over time
    try:
        want = fsnp_torch.lstm2_fc(x[sel].cpu(), sd).numpy()
    finally:
        torch.set_num_threads(threads)
    got = first[sel].cpu().numpy()
    per_step = np.abs(got - want).max(axis=(0, 1)) / np.abs(want).max()
    assert per_step.max() < 2e-5, (per_step.max(), int(per_step.argmax()))


def test_dense_input_beyond_2gib_is_not_read_as_zeros():
    """the half-tile ping-pong kernel (planned for 6..10 row tiles) addresses its input through ONE buffer descriptor with
    32-bit BYTE offsets (2 GiB); rows that start beyond that used to be read as zeros - out of the descriptor's range - with no
    error.  257 sequences x 62,000 steps x 34 features x 4 bytes = 2.17 GB: the planner now keeps that kernel away from such
    inputs (fsnp_abi.hip plan_sb: gather_bytes), and the LAST rows - the ones past 2 GiB - match the oracle."""
    n, steps = 257, 62000
    sd = make_state_dict(9, "default")
    m = _model(sd)
    x = _dense_input(n, steps, 77)
    assert x.numel() * 4 > 2 ** 31
    out = m.lstm2_fc(x)
    m.check_errors()
    sel = [0, 128, 250, 255, 256]
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        want = fsnp_torch.lstm2_fc(x[sel].cpu(), sd).numpy()
    finally:
        torch.set_num_threads(threads)
    got = out[sel].cpu().numpy()
    per_row = np.abs(got - want).max(axis=(1, 2)) / np.abs(want).max()
    assert per_row.max() < 2e-5, per_row
    small = m.lstm2_fc(x[:, :, :64].contiguous())          # the same handle still plans the half-tile ping-pong kernel for small inputs
    assert any("hp" in c["kernel"] for c in m.describe_plan(1))
    assert torch.equal(small[:, :, :8], out[:, :, :8]) or rel_err(small[:, :, :8].cpu().numpy(), out[:, :, :8].cpu().numpy()) < 2e-5


STAGES = ["att_mag", "att_real", "att_imag", "fb_mag", "fb_real", "fb_imag"]


@pytest.mark.parametrize("B,seconds", [(3, 126.0), (1, 200.0)])
def test_long_recurrence_forward(B, seconds):
    """The round-3 failure's shape on purpose: whole forward, B = 3 x 126 s (T' = 7,878) and B = 1 x 200 s (T' = 12,502) in
    full mode, default kernel plan (B = 1: the half-tile ping-pong kernel, 12,502 steps).  Six forwards on one handle - default GEMMs, general GEMM, default, 128-row DMA GEMM twice,
    default with drifting workgroups - bitwise equal where the kernels are the same, all against the oracle; on a mismatch the
    message says whether the attention / full-band stage buffers differ too and which rows broke from which frame on."""
    sd = make_state_dict(21, "default")
    m = _model(sd)
    mag, real, imag = make_inputs(B, seconds, 77)
    g = _cuda((mag, real, imag))
    T = mag.shape[-1]
    want = fsnp_torch.forward_full(sd, mag, real, imag).numpy()

    def run(mode, chaos=0):
        m.debug_set_gemm_dma(mode)
        m.debug_set_chaos(chaos)
        return m(*g)

    def read_stages():                        # the stage buffers of the LAST forward stay in the workspace until the next one
        return {s: m.read_stage(s, B, T) for s in STAGES}

    ref = run(1)
    ref_stages = read_stages()
    assert rel_err(ref.cpu().numpy(), want) < TOL
    for k, (mode, chaos) in enumerate([(0, 0), (1, 0), (2, 0), (2, 0), (1, 5)]):
        out = run(mode, chaos)
        if mode == 0:
            assert rel_err(out.cpu().numpy(), want) < TOL
            continue
        if not torch.equal(out, ref):
            # The recurrent kernels are deterministic; the full-band GroupNorm / TSSE statistics are fp64 sums fed by atomics in an
            # order the hardware chooses, so a last-bit flip of one converted statistic is legitimate (never seen in 550 forwards of
            # this shape, possible in principle).  Anything beyond rounding noise is the failure this test exists for.
            noise = rel_err(out.cpu().numpy(), ref.cpu().numpy())
            if noise < 1e-6:
                continue
            stages = read_stages()
            d = (out != ref)
            rows = torch.nonzero(d.any(dim=3).any(dim=1))            # [utterance, bin]
            frames = torch.nonzero(d.any(dim=2).any(dim=1).any(dim=0)).flatten()
            info = {"forward": k, "mode": mode, "chaos": chaos, "rel_vs_first_forward": noise, "rel_vs_oracle": rel_err(out.cpu().numpy(), want),
                    "rows": int(rows.shape[0]), "first_rows": rows[:6].tolist(), "first_frame": int(frames[0]), "last_frame": int(frames[-1]),
                    "stages_differing": [s for s in STAGES if not torch.equal(stages[s], ref_stages[s])]}
            raise AssertionError(info)
    m.debug_set_chaos(0)
    m.debug_set_gemm_dma(1)
