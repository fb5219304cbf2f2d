#!/bin/bash
# round 6: the wave-owned half-tile kernel (csrc/lstm_hpw.hip) against lstm_hp.hip - parity tests, kernel time, B = 1 forward
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -p no:cacheprovider -k "half_tile_ping_pong_kernel_vs_oracle or half_tile_ping_pong_under" 2>&1 | tail -25 | tee gpurun_out/r06_hpw_pytest.log
{
for v in 1 0; do
  for n in 257 32 288; do
    FSNP_HP_WAVE=$v HP=1 timeout 120 python tools/time_lstm.py $n 128 7 2>&1 | grep -v amdgpu.ids | sed "s/^/FSNP_HP_WAVE=$v /"
  done
done
} | tee gpurun_out/r06_hpw_times.txt
for v in 1 0; do
  FSNP_HP_WAVE=$v timeout 300 python bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_hpw_b1_v$v.json
  python -c "
import json; r=json.load(open('gpurun_out/r06_hpw_b1_v$v.json')); print('FSNP_HP_WAVE=$v B=1', r['ms_per_step'], r['alt_ms_per_step'], r['dropin_ms_per_step'], r['ms_per_step_runs'], r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'])" | tee -a gpurun_out/r06_hpw_times.txt
done
