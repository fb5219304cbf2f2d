#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -k "composite or cooperative or gru2 or valu_rows or b32" -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/t_comp.log
: > gpurun_out/b_comp.log
for args in "--batch 32" "--batch 36" "--batch 40" "--batch 48" "--batch 64"; do
  for g in 0.97 0; do
    FSNP_COMPOSITE_GAIN=$g timeout 300 python bench.py $args --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_comp.log
  done
done
FSNP_COMPOSITE_GAIN=10 timeout 300 python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_comp.log
python - <<'PY'
import json
for i, l in enumerate(open("gpurun_out/b_comp.log")):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:300]); continue
    tag = "forced" if i == 10 else ("composite" if i % 2 == 0 else "single   ")
    print(tag, r["config"]["workload"][:10], "| %.0f frames/s  %.2f ms/fwd  lstm %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"]))
PY
