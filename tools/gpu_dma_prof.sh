#!/bin/bash
# per-dispatch durations of the full-band stage (kernel trace of a short serial-loop bench) -> gpurun_out/dma_trace.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
FSNP_CALIBRATE=0 timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/prof_dma -o dma -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pipeline 0 --no-alt > /tmp/prof_dma.log 2>&1
cd $R; f=$(find /tmp/prof_dma -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P' > gpurun_out/dma_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last forward: from the last fe_repack_kernel on
start = max(i for i, n in enumerate(names) if "fe_repack" in n)
prev_end = None
for r in rows[start:start + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-60s dur %8.2f us  gap %6.2f us  grid %s" % (r["Kernel_Name"][:60], (e - s) / 1e3, gap, r.get("Grid_Size", "")))
    prev_end = e
P
cat gpurun_out/dma_trace.txt
