#!/bin/bash
# round 2, GPU call A: full GPU suite, headline bench (pipelined + back-to-back), TCN GEMM prefetch-depth sweep, rocprofv3 stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" | tee gpurun_out/device.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
tail -c 1500 gpurun_out/bench.log
: > gpurun_out/pf_sweep.log
for pf in 1 2 3 4; do
  FSNP_GEMM_PF=$pf timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/pf_sweep.log
done
python - <<'PY'
import json
for i, l in enumerate(open("gpurun_out/pf_sweep.log")):
    try:
        r = json.loads(l)
        print("PF", i + 1, "ms/step %.3f alt %.3f fullband %.3f subband %.3f" % (r["ms_per_step"], r["alt_ms_per_step"], r["roofline"]["fullband_ms"], r["roofline"]["subband_stage_ms"]))
    except Exception as e:
        print("PF", i + 1, "??", l[:300])
PY
rm -rf gpurun_out/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -14 "$f" | cut -c1-200
echo "== done"
