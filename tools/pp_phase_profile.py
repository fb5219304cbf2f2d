#!/usr/bin/env python3
"""Phase breakdown of the half-tile ping-pong kernel (csrc/lstm_hp.hip) from wall-clock stamps (fsnp_debug_pp_profile; until round 4 also of the
ping-pong K-split kernel lstm_pp.hip, since removed - its profile is profiles/r03_pp_phase_profile.txt).
usage: python tools/pp_phase_profile.py <sequences> <steps> <tiles per group>     (tiles per group 0 = the half-tile ping-pong kernel, csrc/lstm_hp.hip)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullsubnet_plus_amd import FullSubNet_Plus, _lib  # noqa: E402
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_state_dict  # noqa: E402


def main_hpw(s, n, steps):
    """csrc/lstm_hpw.hip (round 6, the default of planner kind 8; FSNP_HP_WAVE=0 selects lstm_hp.hip): wave 0 of workgroup 0 - which also
    sums the Linear partials of output row 0."""
    names = ["x k-groups (10 MFMAs) + s_waitcnt vmcnt(0) (previous stores, first operands) + arrival", "Linear of the previous step (summing waves only)",
             "-", "pass over h: 288 MFMAs, 36 operand loads, 12 x loads, counter poll", "counter check (+ wait) + prefill of the other half's operands",
             "cells + 2 h stores + Linear partial (2 x 2 cross-lane adds, 1 store)", "x normalise", "-"]
    d = np.diff(s[:, :9], axis=1) * 0.01
    res = {"kernel": "lstm2_coop_hpw_kernel", "sequences": n, "steps": steps, "us_per_half_phase": float((s[1:, 0] - s[:-1, 0]).mean() * 0.01),
           "counter_seen_complete_at_the_poll_fraction": float(s[:, 15].mean())}
    for i, nm in enumerate(names):
        if nm != "-":
            res[f"{i}: {nm}"] = round(float(d[:, i].mean()), 3)
    res["gap to the next phase"] = round(float((s[1:, 0] - s[:-1, 8]).mean() * 0.01), 3)
    print(json.dumps(res, indent=1))


def main_hp(s, n, steps):
    names = ["drain own DMA + barrier 1 (operands in LDS)", "pass: first quarter", "pass: deferred arrival (store drain + atomic)", "pass: rest (+ fetch of the other half)",
             "pre-activations -> LDS, barrier 2 (= waiting for the slowest wave)", "cell phase", "barrier 3", "publish (+ wait / late fetch)"]
    d = np.diff(s[:, :9], axis=1) * 0.01
    res = {"kernel": "lstm2_coop_hp_kernel", "sequences": n, "steps": steps, "us_per_half_phase": float((s[1:, 0] - s[:-1, 0]).mean() * 0.01),
           "early_fetch_fraction": float(s[:, 15].mean())}
    for i, nm in enumerate(names):
        res[f"{i}: {nm}"] = round(float(d[:, i].mean()), 3)
    res["gap to the next phase"] = round(float((s[1:, 0] - s[:-1, 8]).mean() * 0.01), 3)
    print(json.dumps(res, indent=1))


def main_coopw(s, n, steps, units):
    """csrc/lstm_coopw.hip: stamps 0..6 in A_t (layer 0), 8..14 in C_{t-1} (layer 1), wave 0 of workgroup 0."""
    a = ["A: h0 k-groups", "A: x k-groups", "A: x fetch, wait b1, prefill C", "A: staging + cells + h0 stores", "A: x normalise", "A: drain + arrive"]
    c = ["C: h1 k-groups", "C: h0 k-groups", "C: wait b0, prefill A", "C: staging + cells + h1 stores", "C: Linear partial", "C: drain + arrive"]
    res = {"kernel": "lstm2_coopw_kernel", "units_per_workgroup": units, "sequences": n, "steps": steps, "us_per_step": float((s[1:, 0] - s[:-1, 0]).mean() * 0.01)}
    da, dc = np.diff(s[:, 0:7], axis=1) * 0.01, np.diff(s[:, 8:15], axis=1) * 0.01
    for i, nm in enumerate(a):
        res[nm] = round(float(da[:, i].mean()), 3)
    res["gap A -> C (epilogue of step t - 2 on participant 0)"] = round(float((s[:, 8] - s[:, 6]).mean() * 0.01), 3)
    for i, nm in enumerate(c):
        res[nm] = round(float(dc[:, i].mean()), 3)
    res["gap C -> next A"] = round(float((s[1:, 0] - s[:-1, 14]).mean() * 0.01), 3)
    print(json.dumps(res, indent=1))


def main():
    n, steps, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m.load_state_dict(make_state_dict(0, "default"))
    m = m.cuda().eval()
    x = torch.randn(n, steps, 34, device="cuda")
    out = torch.empty(n, 2, steps, device="cuda")
    m.lstm2_fc(x.permute(0, 2, 1)[:8])          # creates the handle
    lib = _lib.load()
    if r in (32, 64):
        stamps = np.zeros(steps * 16, dtype=np.uint64)
        for _ in range(2):
            _lib.check(lib.fsnp_debug_pp_profile(m._handle, x.data_ptr(), out.data_ptr(), n, steps, r, stamps.ctypes.data, stamps.size), "profile")
        return main_coopw(stamps.reshape(steps, 16).astype(np.int64)[4:-1], n, steps, r)
    hp = r == 0
    rr = 2 if hp else r
    if hp and n <= 16:
        raise SystemExit("the half-tile profile assumes both halves of row tile 0 hold sequences (n > 16)")
    stamps = np.zeros(steps * rr * (16 if hp else 8), dtype=np.uint64)
    for _ in range(2):
        _lib.check(lib.fsnp_debug_pp_profile(m._handle, x.data_ptr(), out.data_ptr(), n, steps, r, stamps.ctypes.data, stamps.size), "profile")
    if hp:
        wave_owned = os.environ.get("FSNP_HP_WAVE", "1") != "0"
        return (main_hpw if wave_owned else main_hp)(stamps.reshape(steps * 2, 16).astype(np.int64)[8:], n, steps)
    s = stamps.reshape(steps * rr, 8).astype(np.int64)[4 * rr:]          # skip warm-up steps; 10 ns ticks
    names = ["MFMA pass", "barrier 1 (+flags)", "early fetch + partial tiles -> LDS, barrier 2", "cell phase", "barrier 3", "publish (+ wait / late fetch)"]
    d = np.diff(s[:, :7], axis=1) * 0.01
    res = {"kernel": "lstm2_coop_hp_kernel" if hp else "lstm2_coop_pp_kernel", "sequences": n, "steps": steps, "tiles_per_group": r, "us_per_tile_phase": float((s[1:, 0] - s[:-1, 0]).mean() * 0.01),
           "early_fetch_fraction": float(s[:, 7].mean())}
    for i, nm in enumerate(names):
        res[f"{i}: {nm}"] = round(float(d[:, i].mean()), 3)
    res["gap to the next pass"] = round(float((s[1:, 0] - s[:-1, 6]).mean() * 0.01), 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
