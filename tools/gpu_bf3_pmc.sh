#!/bin/bash
# PMC passes for the optional bf16x3 kernel (each counter set in its own run, no trace domains)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctr -f csv -d $R/gpurun_out/bpmc$i -o pmc -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 --precision bf16x3 > $R/gpurun_out/bpmc$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/bf3_pmc_summary.txt
import csv, glob, collections
for i in (1, 2, 3):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/bpmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bf3" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"pmc{i} {k:60s} {c:30s} n={n} per_launch={v/n:.6g}")
PY
rm -rf gpurun_out/bpmc1 gpurun_out/bpmc2 gpurun_out/bpmc3
