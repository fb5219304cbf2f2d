#!/bin/bash
# gpurun helper: FullSubNet + cooperative-kernel tests, then small-batch / FullSubNet bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullsubnet.py tests/test_gpu_parity.py -k "fullsubnet or cooperative or dense_vs_oracle" -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/t_fsn.log
: > gpurun_out/b_small.log
for b in 1 2 3 5; do
  timeout 200 python bench.py --batch $b --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_small.log
done
for b in 1 8 32; do
  timeout 300 python bench.py --model fullsubnet --batch $b --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_small.log
done
python - <<'PY'
import json
for l in open("gpurun_out/b_small.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:300]); continue
    print(r["metric"][-20:], r["config"]["workload"][:22], "| %.0f frames/s  %.2f ms/fwd  lstm %.2f  fullband %.2f" % (
        r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["fullband_ms"]))
PY
