#!/bin/bash
# gpurun helper: column-split kernel tests, then medium-batch / parity-mode bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -k "cooperative or coopn or b32 or complex or enhance" -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/t_coopn.log
: > gpurun_out/b_mid.log
for args in "--batch 6" "--batch 8" "--batch 10" "--batch 16" "--batch 21" "--batch 32 --mode parity" "--batch 42 --mode parity"; do
  timeout 200 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_mid.log
done
python - <<'PY'
import json
for l in open("gpurun_out/b_mid.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:300]); continue
    print(r["config"]["workload"][:24], r["config"]["workload"].split(",")[2], "| %.0f frames/s  %.2f ms/fwd  lstm %.2f  fullband %.2f" % (
        r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["fullband_ms"]))
PY
