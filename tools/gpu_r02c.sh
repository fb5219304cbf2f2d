#!/bin/bash
# round 2, GPU call C: calibrated planner + two column-split workgroups per CU: GPU suite, headline, small-batch table.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
cat gpurun_out/planner_costs.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
: > gpurun_out/b_small.log
for args in "--batch 1" "--batch 2" "--batch 3" "--batch 5" "--batch 8" "--batch 12" "--batch 16" "--batch 21" "--batch 32 --mode parity" "--batch 40" "--batch 64" "--sequence-model GRU" "--batch 1 --sequence-model GRU"; do
  timeout 300 python bench.py $args --steps 6 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 2>&1 | tail -1 >> gpurun_out/b_small.log
  FSNP_COOP_OCC=1 FSNP_CALIBRATE=0 timeout 300 python bench.py $args --steps 6 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 2>&1 | tail -1 >> gpurun_out/b_small.log
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("HEADLINE ms/step %.3f alt %.3f value %.0f frac %.4f fullband %.3f" % (r["ms_per_step"], r["alt_ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["fullband_ms"]))
for i, l in enumerate(open("gpurun_out/b_small.log")):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    plan = " + ".join("%s x%d" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"]) for c in r["roofline"]["subband_plan"])
    print("%-9s| %-58s | %8.0f frames/s %8.3f ms  sub-band %7.3f | %s" % ("new" if i % 2 == 0 else "r01-table", r["config"]["workload"][:58], r["value"], r["ms_per_step"], r["roofline"]["subband_stage_ms"], plan))
PY
echo "== done"
