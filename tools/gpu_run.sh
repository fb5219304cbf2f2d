#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT=${1:-all}
echo "== device" | tee gpurun_out/run.log
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" | tee -a gpurun_out/run.log
nproc | tee -a gpurun_out/run.log
if [[ "$WHAT" == "all" || "$WHAT" == *test* ]]; then
  echo "== pytest -m gpu" | tee -a gpurun_out/run.log
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == *smoke* ]]; then
  echo "== smoke" | tee -a gpurun_out/run.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == *bench* ]]; then
  echo "== bench" | tee -a gpurun_out/run.log
  timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench.log
fi
if [[ "$WHAT" == "all" || "$WHAT" == *prof* ]]; then
  echo "== rocprofv3 kernel trace" | tee -a gpurun_out/run.log
  rm -rf gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
  tail -2 gpurun_out/prof_bench.log
  find gpurun_out/prof -name "*stats*" | head
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" | tee gpurun_out/kernel_stats_head.csv
fi
echo "== done"
