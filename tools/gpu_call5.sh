#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cooperative or skewed or role_split or ping_pong or other_hidden or exchange_under_load or two_workgroups" 2>&1 | tail -6 | tee gpurun_out/pp_tests.txt
for cfg in "32 64 1" "64 64 2" "96 64 3" "257 64 2" "288 64 2" "257 64 3" "640 64 4"; do
  timeout 120 python tools/pp_phase_profile.py $cfg 2>&1 | tail -15
done 2>&1 | tee gpurun_out/pp_profile.txt
for n in 32 64 160 257 320 514 640 1285; do
  for r in 0 1 2 3 4; do
    PP_R=$r timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/pp_times.txt
for b in 1 2 4; do
  for pp in 0; do
    FSNP_COOP_PP=$pp timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench B=$b PP=$pp: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f plan %s' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], [(c['kernel'][:22], c['sequences']) for c in r['roofline']['subband_plan']]))"
  done
done 2>&1 | tee -a gpurun_out/pp_times.txt
