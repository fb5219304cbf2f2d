#!/bin/bash
# round 2, GPU call B: GPU suite, headline bench (both loops), the unmodified reference CLI end to end (tools/cli_e2e.py,
# needs the reference staged under _refstage/ for this call only), rocprofv3 stats of the back-to-back loop.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("BENCH ms/step %.3f alt %.3f value %.0f frac %.4f fullband %.3f lstm %.3f cpu %.0f err %.2e" % (r["ms_per_step"], r["alt_ms_per_step"], r["value"],
      r["roofline"]["frac"], r["roofline"]["fullband_ms"], r["roofline"]["avg_launch_ms"], r["cpu_baseline"]["value"], r["cirm_rel_err"]))
PY
timeout 300 python bench.py --steps 10 --warmup 3 --pipeline 0 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_serial.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_serial.log").read())
print("SERIAL ms/step %.3f alt %.3f fullband %.3f subband %.3f" % (r["ms_per_step"], r["alt_ms_per_step"], r["roofline"]["fullband_ms"], r["roofline"]["subband_stage_ms"]))
PY
if [ -d _refstage/reference ]; then
  timeout 1500 python tools/cli_e2e.py _refstage/reference 2>&1 | tail -40 | tee gpurun_out/cli_e2e_stdout.log
fi
rm -rf gpurun_out/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --pipeline 0 --no-alt --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -12 "$f" | cut -c1-160
rm -rf gpurun_out/prof
echo "== done"
