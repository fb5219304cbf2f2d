#!/bin/bash
# scratch: matrix-pipe counters of the small-batch sub-band kernels (B = 1, 2, 5, 8)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/small_batch_pmc.txt
for B in 1 2 5 8; do
  rm -rf /tmp/pmc
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -f csv -d /tmp/pmc -o pmc -- python $R/bench.py --gpus 1 --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 > /dev/null 2>&1
  python - $B <<'PY' | tee -a $R/gpurun_out/small_batch_pmc.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lstm2_" not in k: continue
        a = agg[k.split("(")[0][:60]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, c in agg.items():
    if not {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(c): continue
    n = c["GRBM_GUI_ACTIVE"][1]
    busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n, c["GRBM_GUI_ACTIVE"][0] / n
    print("B=%s %-60s launches %d  MFMA_BUSY %.4g  GUI_ACTIVE %.4g  matrix-pipe busy over the whole chip %.3f" % (sys.argv[1], k, n, busy, act, busy * 8 / (1024 * act)))
PY
done
