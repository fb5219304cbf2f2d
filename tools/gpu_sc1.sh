#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsubnet.py -k "under_load or cooperative or gru2 or composite or fullsubnet_forward or b1_coop or coopn or b32 or sharded" -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/t_sc1.log
: > gpurun_out/b_sc1.log
for args in "--batch 1" "--batch 2" "--batch 5" "--batch 8" "--batch 16" "--batch 32" "--batch 32 --mode parity" "--model fullsubnet --batch 1"; do
  timeout 300 python bench.py $args --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_sc1.log
done
python - <<'PY'
import json
for l in open("gpurun_out/b_sc1.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:300]); continue
    print(r["metric"][-22:], r["config"]["workload"][:10], r["config"]["workload"].split(",")[2][:8], "| %.0f frames/s  %.3f ms/fwd  sub-band %.3f (first %.3f)  fullband %.3f" % (
        r["value"], r["ms_per_step"], r["roofline"]["subband_stage_ms"], r["roofline"]["avg_launch_ms"], r["roofline"]["fullband_ms"]))
PY
