"""Soak / localisation harness for long recurrences (round-4 task 1: the driver-box failure of
test_dma_gemm_equals_general_gemm[default-3-126] - B = 3 x 126 s clips, T' = 7,878 steps, 0.35 rel on the 4th forward).

  python tools/repro_long.py forward --iters 12 [--batch 3 --seconds 126] [--tag NAME]
      the test's four-forward sequence (default GEMMs, general GEMM, default, 128-row DMA GEMM) on a FRESH handle per iteration;
      every output must be bit-identical to the first iteration's (same kernels, same inputs).  On a mismatch: which rows
      (utterance, bin) differ, from which frame on, and whether the attention / full-band stage buffers differ too.
  python tools/repro_long.py lstm --n 771 --steps 8000 --reps 10
      fsnp_lstm2_fc alone (dense input), bit-repeatable + vs torch on a sample of rows.
Writes gpurun_out/repro_<tag>.json."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fullsubnet_plus_amd import FullSubNet_Plus  # noqa: E402
from oracle import fsnp_torch  # noqa: E402
from oracle.ref_loader import DEFAULT_MODEL_ARGS  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402

STAGES = ["att_mag", "att_real", "att_imag", "fb_mag", "fb_real", "fb_imag"]


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


def describe_diff(got, ref):
    """got / ref: [B, 2, F, T]."""
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = max(float(np.abs(ref).max()), 1e-30)
    bad = d.max(axis=(1, 3)) > 0                  # [B, F]
    rows = np.argwhere(bad)
    info = {"rel": float(d.max() / scale), "rows_differing": int(bad.sum()), "rows_total": int(bad.size),
            "nan_count": int(np.isnan(got).sum())}
    if len(rows):
        per_b = {int(b): [int(rows[rows[:, 0] == b][:, 1].min()), int(rows[rows[:, 0] == b][:, 1].max()), int((rows[:, 0] == b).sum())]
                 for b in np.unique(rows[:, 0])}
        info["per_utterance_bin_min_max_count"] = per_b
        first_t = {}
        for b, f in rows[:: max(1, len(rows) // 12)]:
            t_bad = np.argwhere(d[b, :, f, :].max(axis=0) > 0)
            first_t[f"{int(b)}:{int(f)}"] = [int(t_bad.min()), int(t_bad.max()), int(len(t_bad)), float(d[b, :, f, :].max() / scale)]
        info["first_last_count_bad_frame_of_sample_rows"] = first_t
        # flat row index (utterance * F + bin) = the planner's sequence index in "full" mode
        flat = rows[:, 0] * got.shape[2] + rows[:, 1]
        info["flat_seq_min_max"] = [int(flat.min()), int(flat.max())]
    return info


def run_forward(args):
    sd = make_state_dict(21, "default")
    mag, real, imag = make_inputs(args.batch, args.seconds, 77)
    g = _cuda((mag, real, imag))
    T = mag.shape[-1]
    seq = [1, 0, 1, 2] if not args.same else [1, 1, 1, 1]
    report = {"batch": args.batch, "seconds": args.seconds, "frames": T, "iters": [], "env": {k: v for k, v in os.environ.items() if k.startswith("FSNP_")}}
    ref_out, ref_stage = {}, {}
    n_bad = 0
    for it in range(args.iters):
        m = _model(sd)
        rec = {"it": it, "fwd": []}
        for k, mode in enumerate(seq):
            m.debug_set_gemm_dma(mode)
            t0 = time.time()
            out = m(*g).cpu().numpy()
            dt = time.time() - t0
            key = 0 if mode == 0 else 1
            stages = {s: m.read_stage(s, args.batch, T).numpy() for s in STAGES} if args.stages else {}
            if key not in ref_out:
                ref_out[key] = out
                ref_stage[key] = stages
                rec["fwd"].append({"k": k, "mode": mode, "ms": dt * 1e3, "ref": True})
                continue
            same = np.array_equal(out, ref_out[key])
            e = {"k": k, "mode": mode, "ms": round(dt * 1e3, 1), "equal": bool(same)}
            if not same:
                n_bad += 1
                e["diff"] = describe_diff(out, ref_out[key])
                for s in stages:
                    a, b = stages[s], ref_stage[key][s]
                    if not np.array_equal(a, b):
                        dd = np.abs(a.astype(np.float64) - b)
                        where = np.argwhere(dd.max(axis=2) > 0)      # [utt, frame]
                        e.setdefault("stage_diff", {})[s] = {"rel": float(dd.max() / max(np.abs(b).max(), 1e-30)), "cells": int((dd > 0).sum()),
                                                             "utts": sorted(set(int(x) for x in where[:, 0])),
                                                             "frame_min_max": [int(where[:, 1].min()), int(where[:, 1].max())]}
                print("MISMATCH", json.dumps(e), flush=True)
                np.save(os.path.join(ROOT, "gpurun_out", f"repro_{args.tag}_bad_it{it}_k{k}.npy"), out[:, :, ::8, ::16])
            rec["fwd"].append(e)
        if it == 0:
            report["plan"] = [f"{c['kernel']} x{c['sequences']} tiles={c['tiles']} wg={c['workgroups']}" for c in m.describe_plan(args.batch)]
            print("plan:", report["plan"], flush=True)
        report["iters"].append(rec)
        print(f"iter {it}: " + " ".join(f"{f.get('mode')}:{'ref' if f.get('ref') else ('ok' if f['equal'] else 'BAD')}({f['ms']:.0f}ms)" for f in rec["fwd"]), flush=True)
        del m
    report["mismatches"] = n_bad
    if args.oracle:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        want = fsnp_torch.forward_full(sd, mag, real, imag).numpy()
        report["ref_vs_oracle"] = {str(k): float(np.abs(v - want).max() / np.abs(want).max()) for k, v in ref_out.items()}
        print("reference outputs vs oracle:", report["ref_vs_oracle"], flush=True)
    return report


def run_idle(args):
    """The failing test's shape of events: forwards, then the GPU idles while the host is busy (the CPU oracle, 16 threads),
    then ONE more forward - compared bitwise with the first."""
    sd = make_state_dict(21, "default")
    mag, real, imag = make_inputs(args.batch, args.seconds, 77)
    g = _cuda((mag, real, imag))
    T = mag.shape[-1]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    report = {"batch": args.batch, "seconds": args.seconds, "frames": T, "trials": [], "env": {k: v for k, v in os.environ.items() if k.startswith("FSNP_")}}
    ref = None
    ref_stage = None
    n_bad = 0
    for it in range(args.iters):
        m = _model(sd)
        out = m(*g).cpu().numpy()
        if ref is None:
            ref = out
            ref_stage = {s: m.read_stage(s, args.batch, T).numpy() for s in STAGES}
        first_ok = bool(np.array_equal(out, ref))
        t0 = time.time()
        if args.idle_kind == "oracle":
            want = fsnp_torch.forward_full(sd, mag, real, imag).numpy()
            if it == 0:
                report["ref_vs_oracle"] = float(np.abs(ref - want).max() / np.abs(want).max())
        else:
            time.sleep(args.idle)
        idle_s = time.time() - t0
        m.debug_set_gemm_dma(2)
        out = m(*g).cpu().numpy()
        same = bool(np.array_equal(out, ref))
        e = {"it": it, "idle_s": round(idle_s, 1), "first_equal": first_ok, "after_idle_equal": same}
        if not same:
            n_bad += 1
            e["diff"] = describe_diff(out, ref)
            stages = {s: m.read_stage(s, args.batch, T).numpy() for s in STAGES}
            for s in stages:
                a, b = stages[s], ref_stage[s]
                if not np.array_equal(a, b):
                    dd = np.abs(a.astype(np.float64) - b)
                    where = np.argwhere(dd.max(axis=2) > 0)
                    e.setdefault("stage_diff", {})[s] = {"rel": float(dd.max() / max(np.abs(b).max(), 1e-30)), "cells": int((dd > 0).sum()),
                                                         "utts": sorted(set(int(x) for x in where[:, 0])),
                                                         "frame_min_max": [int(where[:, 1].min()), int(where[:, 1].max())]}
            again = m(*g).cpu().numpy()
            e["immediately_again_equal"] = bool(np.array_equal(again, ref))
            np.save(os.path.join(ROOT, "gpurun_out", f"repro_{args.tag}_bad_it{it}.npy"), out[:, :, ::4, ::8])
        print("trial", json.dumps(e), flush=True)
        report["trials"].append(e)
        del m
    report["mismatches"] = n_bad
    return report


def run_lstm(args):
    sd = make_state_dict(9, "harsh" if args.harsh else "default")
    m = _model(sd)
    rng = np.random.Generator(np.random.PCG64(4321 + args.n))
    x = torch.from_numpy(rng.standard_normal((args.n, 34, args.steps)).astype(np.float32))
    xg = x.cuda()
    report = {"n": args.n, "steps": args.steps, "reps": [], "env": {k: v for k, v in os.environ.items() if k.startswith("FSNP_")}}
    first = None
    for r in range(args.reps):
        t0 = time.time()
        got = m.lstm2_fc(xg).cpu().numpy()
        m.check_errors()
        dt = time.time() - t0
        if first is None:
            first = got
            report["reps"].append({"r": r, "ms": dt * 1e3, "ref": True})
            continue
        same = np.array_equal(got, first)
        e = {"r": r, "ms": round(dt * 1e3, 1), "equal": bool(same)}
        if not same:
            d = np.abs(got.astype(np.float64) - first)
            rows = np.argwhere(d.max(axis=(1, 2)) > 0)[:, 0]
            e["rows_differing"] = int(len(rows))
            e["row_min_max"] = [int(rows.min()), int(rows.max())]
            e["tiles_32"] = sorted(set(int(q) // 32 for q in rows))[:40]
            tb = np.argwhere(d.max(axis=(0, 1)) > 0)[:, 0]
            e["step_min_max"] = [int(tb.min()), int(tb.max())]
            e["rel"] = float(d.max() / np.abs(first).max())
            print("MISMATCH", json.dumps(e), flush=True)
        report["reps"].append(e)
        print(f"rep {r}: {'ok' if same else 'BAD'} {dt * 1e3:.0f} ms", flush=True)
    report["mismatches"] = sum(1 for e in report["reps"] if e.get("equal") is False)
    # the oracle on a sample of rows (torch.lstm is per-row independent)
    sel = sorted(set(list(range(0, args.n, max(1, args.n // 24))) + [args.n - 1]))
    want = fsnp_torch.lstm2_fc(x[sel], sd).numpy()
    report["rel_vs_oracle_sample_rows"] = float(np.abs(first[sel] - want).max() / np.abs(want).max())
    print("vs oracle (sample rows):", report["rel_vs_oracle_sample_rows"], flush=True)
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["forward", "lstm", "idle"])
    ap.add_argument("--idle", type=float, default=30.0)
    ap.add_argument("--idle-kind", default="oracle", choices=["oracle", "sleep"])
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=126.0)
    ap.add_argument("--tag", default="default")
    ap.add_argument("--stages", type=int, default=1)
    ap.add_argument("--oracle", type=int, default=0)
    ap.add_argument("--same", type=int, default=0, help="four default-mode forwards instead of the test's 1,0,1,2 sequence")
    ap.add_argument("--n", type=int, default=771)
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--harsh", type=int, default=0)
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rep = run_forward(a) if a.what == "forward" else run_idle(a) if a.what == "idle" else run_lstm(a)
    with open(os.path.join(ROOT, "gpurun_out", f"repro_{a.tag}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print("mismatches:", rep["mismatches"])
