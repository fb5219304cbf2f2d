set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/bf16_error.py 2>/dev/null > gpurun_out/r06_bf16_error_hilo.md; cat gpurun_out/r06_bf16_error_hilo.md
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "bf16" 2>&1 | tail -6
for a in "--precision bf16_ih" "--mode parity --precision bf16_ih" "--batch 16 --precision bf16_ih"; do
timeout 300 python bench.py $a --steps 10 --warmup 3 --no-cpu-baseline --probe-ms 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$a', round(r['ms_per_step'],3), round(r['alt_ms_per_step'],3), round(r['value']))"
done
