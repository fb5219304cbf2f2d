#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats (csv), then PMC passes (each in its own run).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH="python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rm -rf gpurun_out/prof gpurun_out/pmc*
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof -o trace -- $BENCH > gpurun_out/prof_bench.log 2>&1
tail -1 gpurun_out/prof_bench.log
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctr -f csv -d gpurun_out/pmc$i -o pmc -- $BENCH > gpurun_out/pmc$i.log 2>&1
  python - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc$i/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"pmc$i {k:60s} {c:32s} sum={v:.6g} n={n} per_launch={v/n:.6g}")
PY
done
echo "== done"
