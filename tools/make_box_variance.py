#!/usr/bin/env python3
"""profiles/r06_box_variance.md from the bench lines profiles/r06_box_*.json (one per gpurun box: tools/gpu_r06_box.sh)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_box_*.json"))):
    r = json.load(open(p))
    b, rf = r["box"], r["roofline"]
    lc = rf.get("last_launch_clock") or {}
    d = b.get("during_timed_loops", {})
    rows.append((os.path.basename(p)[8:-5], (b.get("gpu_unique_id") or "-")[:8], r["ms_per_step"], r["ms_per_step_runs"], r["alt_ms_per_step"], r["dropin_ms_per_step"],
                 rf["avg_launch_ms"], rf["frac"], rf["frac_of_box_peak"], rf.get("frac_at_held_clock"), lc.get("wall_ms"), lc.get("slowest_workgroup_ms"),
                 lc.get("s_memtime_ticks"), lc.get("s_memtime_mhz"), b["probe_before"]["mfma_tflops"], b["probe_after"]["mfma_tflops"],
                 b["probe_after"]["clock_mhz_slowest_cu"], b["probe_after"]["clock_mhz_fastest_cu"],
                 (d.get("sclk_mhz") or {}).get("median"), (d.get("power_w") or {}).get("median"), d.get("power_cap_w")))
f = lambda v, n=3: "-" if v is None else f"{v:.{n}f}"
print("| box | GPU serial | ms/step (median of 3) | runs | back to back | drop-in | dominant launch, hipEvents (ms) | frac (spec peak) | frac of box peak | frac at held clock | "
      "workgroup 0 (ms) | slowest workgroup (ms) | shader cycles of workgroup 0 | clock held (MHz) | probe before / after (TFLOP/s) | probe: slowest / fastest CU (MHz) | "
      "sclk median (sysfs) | power median / cap (W) |")
print("|" + "---|" * 18)
for (tag, uid, ms, runs, alt, drop, lms, fr, fb, fh, w0, wmax, cyc, mhz, p0, p1, cmin, cmax, sclk, pw, cap) in rows:
    print(f"| {tag} | {uid} | **{ms:.3f}** | {' '.join(f'{x:.3f}' for x in runs)} | {f(alt)} | {f(drop)} | {lms:.3f} | {fr:.4f} | {fb:.4f} | {f(fh, 4)} | {f(w0)} | {f(wmax)} | "
          f"{'-' if cyc is None else f'{cyc / 1e6:.3f} M'} | {f(mhz, 0)} | {p0:.1f} / {p1:.1f} | {cmin:.0f} / {cmax:.0f} | {f(sclk, 0)} | {f(pw, 0)} / {f(cap, 0)} |")
