#!/bin/bash
# round 4: bf16 ih-GEMM on the half-tile kernel - tests + bench (parity-mode B = 32, full-mode B = 16, fp32 vs bf16_ih)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or half_tile" 2>&1 | tail -12 | tee gpurun_out/bf16_pytest.log
: > gpurun_out/bf16_bench.log
for args in "--mode parity" "--mode parity --precision bf16_ih" "--batch 16" "--batch 16 --precision bf16_ih" "--precision bf16_ih"; do
  timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/bf16_bench.log
done
python - <<'PY'
import json
for l in open("gpurun_out/bf16_bench.log"):
    try:
        d = json.loads(l)
        print(d["config"]["workload"][:100], "|", d["dtype"], "| ms", round(d["ms_per_step"], 3), "| frames/s", round(d["value"]), "| alt", d.get("alt_ms_per_step"))
    except Exception as e:
        print("bad line", l[:200])
PY
