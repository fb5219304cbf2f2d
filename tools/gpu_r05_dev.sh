#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "weight_update" 2>&1 | tail -3
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/prof_b1.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/kernel_stats_dev_b1.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kernel_stats_dev_b1.csv")))
for r in rows[:16]:
    print("%-70s calls %5s avg %8.1f ns total %10.0f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"])))
PY
rm -rf $R/gpurun_out/prof
cd $R
for args in "--batch 1" "--batch 2"; do
  timeout 400 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['config']['workload'][:24], 'ms %.3f b2b %.3f dropin %.3f subband %.3f fullband %.3f alt %.3f' % (r['ms_per_step'], r['alt_ms_per_step'], r['dropin_ms_per_step'], r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], r['roofline']['alt_fullband_ms']))"
done
} 2>&1 | tee gpurun_out/dev.log
