#!/bin/bash
# round 4: LDS bank conflicts of the full-band GEMM kernels (the epilogue's transposition, the 64-row kernel's swizzled image)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmc_lds
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL -f csv -d $R/gpurun_out/pmc_lds -o pmc -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/pmc_lds.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/pmc_lds_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
kernels = sorted({k for k, _ in agg})
for k in kernels:
    c = {n: agg[(k, n)][0] / max(agg[(k, n)][1], 1) for kk, n in agg if kk == k}
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if act <= 0: continue
    print(f"{k:60s} LDS_IDX_ACTIVE {act:12.4g}  BANK_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT', 0):12.4g} ({100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / act:5.2f} %)  UNALIGNED_STALL {c.get('SQ_LDS_UNALIGNED_STALL', 0):10.4g}")
PY
rm -rf gpurun_out/pmc_lds
