set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "wave_owned_column_split_kernel_vs_oracle or forward_b8_coopn" 2>&1 | tail -6
for u in 32 64; do for n in 514 672 1285 1344; do COOPW=$u timeout 120 python tools/time_lstm.py $n 128 7 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r06_coopw_times.txt
for b in 2 3 4 5 8; do
timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --probe-ms 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('B=$b', 'serving', round(r['ms_per_step'],4), 'b2b', round(r['alt_ms_per_step'],4), 'dropin', round(r['dropin_ms_per_step'],4))"
done | tee -a gpurun_out/r06_coopw_times.txt
