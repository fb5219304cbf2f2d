set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "sampled_exchange or exchange_verification_detects or half_tile_kernel" 2>&1 | tail -6
for b in 1 4 5 8; do for vs in 16 0; do
timeout 300 python bench.py --batch $b --steps 64 --warmup 5 --no-cpu-baseline --probe-ms 0 --verify-sample $vs 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('B=$b verify_sample=$vs', 'serving', round(r['ms_per_step'],4), 'b2b', round(r['alt_ms_per_step'],4), 'dropin', round(r['dropin_ms_per_step'],4), [round(x,3) for x in r['dropin_ms_per_step_runs']])"
done; done | tee gpurun_out/r06_verify_sample_overhead.txt
