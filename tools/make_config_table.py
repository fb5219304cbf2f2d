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
    plan = " + ".join("%s ×%d%s" % (c["kernel"].split(" ")[0], c["sequences"], "*" if c.get("deferred_when_pipelined") else "") for c in r["roofline"]["subband_plan"])
    rows.append((r["metric"].split("), ")[1], r["config"]["workload"], r["value"], r["ms_per_step"], r["alt_ms_per_step"], r["dtype"], plan, r.get("dropin_ms_per_step")))
h = json.load(open(os.path.join(ROOT, "gpurun_out", "bench.log")))
hp = " + ".join("%s ×%d" % (c["kernel"].split(" ")[0], c["sequences"]) for c in h["roofline"]["subband_plan"])
out = [f"# {TAG} - bench.py on one MI355X: every configuration measured at the end of round {int(TAG[1:])} (gpurun, `tools/gpu_{TAG}_final.sh`)", "",
       "fp32 unless stated; frames/s = B·T / wall time of the whole forward, inputs resident in HBM.  `ms/step` = the default loop of",
       "bench.py (pipelined serving loop, `fsnp_set_pipeline`: differs only where the planner defers launches - `*` behind a launch of the plan),",
       "`back to back` = `alt_ms_per_step` (forwards strictly serialised, `error_check=\"deferred\"`), `drop-in` = `dropin_ms_per_step` (the module's",
       "default `error_check=\"sync\"`: every forward waits for its launches - what editing the one TOML line gives).  Plans: built-in cost table.", "",
       "| configuration | frames/s | ms/step | back to back | drop-in | plan of the sub-band model |", "|---|---|---|---|---|---|",
       "| **headline** `--gpus 1 --steps 20 --warmup 5`: batch 32 × 2 s, full mode | **%.0f** | **%.3f** | %.3f | @DROPIN@ | %s; dominant kernel %.2f ms = %.3f of "
       "the fp32 MFMA peak; cpu_baseline (port, %d threads) %.0f frames/s; cIRM rel err vs oracle (%d utterances of the timed batch) %.1e |"
       % (h["value"], h["ms_per_step"], h["alt_ms_per_step"], hp, h["roofline"]["avg_launch_ms"], h["roofline"]["frac"],
          h["cpu_baseline"]["cores"], h["cpu_baseline"]["value"], len(h.get("cirm_checked_utterances", [])), h["cirm_rel_err"])]
out[-1] = out[-1].replace("@DROPIN@", "%.3f" % h["dropin_ms_per_step"] if h.get("dropin_ms_per_step") else "-")
for m, a, v, ms, alt, dt, plan, dropin in rows:
    a = a.replace(" clips per GPU", "").replace(", random-init weights (seed 0)", "").replace(", num_neighbors=15", "")
    out.append("| %s: %s%s | %.0f | %.3f | %s | %s | %s |" % (m, a, "" if dt == "f32" else " **[" + dt + "]**", v, ms, "%.3f" % alt if alt else "-", "%.3f" % dropin if dropin else "-", plan))
out += ["", "Earlier rounds for comparison: `profiles/r01_bench_configs.md` ... `profiles/r04_bench_configs.md` (round 4, ms/step / back to back: headline 27.51 / 28.32;",
        "B = 1 1.77 / 1.97; B = 2 3.10 / 3.35; B = 5 6.33 / 6.39; B = 8 9.77 / 9.87; B = 16 15.01 / 15.64; B = 21 20.08 / 20.08; B = 40 37.46 / 36.93; FullSubNet B = 1 3.34, B = 32 30.5)."]
open(os.path.join(ROOT, "profiles", f"{TAG}_bench_configs.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
