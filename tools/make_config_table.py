#!/usr/bin/env python3
"""gpurun_out/bench.log + gpurun_out/b_final.log (tools/gpu_rNN_final.sh) -> profiles/rNN_bench_configs.md; usage: make_config_table.py r03"""
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for line in open(os.path.join(ROOT, "gpurun_out", "b_final.log")):
    try:
        r = json.loads(line)
    except Exception:
        continue
    plan = " + ".join("%s ×%d" % (c["kernel"].split(" ")[0], c["sequences"]) for c in r["roofline"]["subband_plan"])
    rows.append((r["metric"].split("), ")[1], r["config"]["workload"], r["value"], r["ms_per_step"], r["alt_ms_per_step"], r["dtype"], plan))
h = json.load(open(os.path.join(ROOT, "gpurun_out", "bench.log")))
hp = " + ".join("%s ×%d" % (c["kernel"].split(" ")[0], c["sequences"]) for c in h["roofline"]["subband_plan"])
out = [f"# {TAG} - bench.py on one MI355X: every configuration measured at the end of round {int(TAG[1:])} (gpurun, `tools/gpu_{TAG}_final.sh`)", "",
       "fp32 unless stated; frames/s = B·T / wall time of the whole forward, inputs resident in HBM.  `ms/step` = the default loop of",
       "bench.py (pipelined serving loop, `fsnp_set_pipeline`: only differs where the plan has a remainder chunk behind a one-tile-per-CU",
       "chunk), `back to back` = `alt_ms_per_step` (forwards strictly serialised).  Plans are those of the built-in cost table.", "",
       "| configuration | frames/s | ms/step | back to back | plan of the sub-band model |", "|---|---|---|---|---|",
       "| **headline** `--gpus 1 --steps 20 --warmup 5`: batch 32 × 2 s, full mode | **%.0f** | **%.3f** | %.3f | %s; dominant kernel %.2f ms = %.3f of "
       "the fp32 MFMA peak; cpu_baseline (port, %d threads) %.0f frames/s; cIRM rel err vs oracle (%d utterances of the timed batch) %.1e |"
       % (h["value"], h["ms_per_step"], h["alt_ms_per_step"], hp, h["roofline"]["avg_launch_ms"], h["roofline"]["frac"],
          h["cpu_baseline"]["cores"], h["cpu_baseline"]["value"], len(h.get("cirm_checked_utterances", [])), h["cirm_rel_err"])]
for m, a, v, ms, alt, dt, plan in rows:
    a = a.replace(" clips per GPU", "").replace(", random-init weights (seed 0)", "").replace(", num_neighbors=15", "")
    out.append("| %s: %s%s | %.0f | %.3f | %s | %s |" % (m, a, "" if dt == "f32" else " **[" + dt + "]**", v, ms, "%.3f" % alt if alt else "-", plan))
out += ["", "Earlier rounds for comparison: `profiles/r01_bench_configs.md` (headline 29.03 ms), `profiles/r02_bench_configs.md` (27.62 ms; B = 1 2.22 / 2.65 ms;", "`profiles/r03_bench_configs.md` (27.59 ms; B = 1 1.80 / 1.97 ms; B = 8 9.79; B = 16 14.67; parity-mode B = 32 14.33; bf16-ih 21.20; 10 s clips 134.97 ms);",
        "B = 8 9.73 ms; B = 16 15.4 ms; parity-mode B = 32 14.8 ms; GRU B = 32 21.1 ms; bf16-ih 21.2 ms; 10 s clips 133 ms)."]
open(os.path.join(ROOT, "profiles", f"{TAG}_bench_configs.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
