set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "wave_owned_column_split_kernel_vs_oracle or forward_b8_coopn or planner_cost_table" 2>&1 | tail -6
for n in 64 1799 2048; do COOPW=96 timeout 120 python tools/time_lstm.py $n 128 7 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_coopw96_times.txt
for b in 6 7 8 9 10 12 40; do
timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --probe-ms 0 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('B=$b', 'serving', round(r['ms_per_step'],4), 'b2b', round(r['alt_ms_per_step'],4), 'dropin', round(r['dropin_ms_per_step'],4), [c['kernel'].split(' ')[0]+' x'+str(c['sequences']) for c in r['roofline']['subband_plan']])"
done | tee -a gpurun_out/r06_coopw96_times.txt
timeout 300 python tools/dump_costs.py 2>&1 | grep -v amdgpu > gpurun_out/r06_dump_costs.log
