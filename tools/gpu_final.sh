#!/bin/bash
# Final evidence run: full GPU suite, smoke, headline bench (+cpu baseline), rocprof kernel stats, config table.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof -o trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
: > gpurun_out/b_final.log
for args in "--batch 31" "--batch 64" "--seconds 10" "--seconds 10 --norm cumulative_layer_norm" "--mode parity" "--precision bf16_ih" "--batch 1" "--batch 8" "--batch 40" "--wave" "--model fullsubnet" "--model fullsubnet --batch 1"; do
  timeout 400 python bench.py $args --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_final.log
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("HEADLINE %.0f frames/s %.3f ms frac %.4f stage %.3f ms cpu %.0f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["subband_stage_ms"], r["cpu_baseline"]["value"]))
for l in open("gpurun_out/b_final.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    print("%-22s | %-60s | %8.0f frames/s %8.3f ms  sub-band %7.3f  fullband %6.3f | %s" % (r["metric"][38:60], r["config"]["workload"][:60], r["value"], r["ms_per_step"],
          r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["dtype"][:8]))
PY
