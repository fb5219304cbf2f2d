#!/bin/bash
# ping-pong K-split kernel (csrc/lstm_pp.hip): bit-identity with the serial schedule, then per-step times by tiles per group
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "ping_pong" 2>&1 | tail -4 | tee gpurun_out/pp_tests.txt
for n in 32 64 96 128 160 257 320 514 640; do
  for r in 0 1 2 3 4; do
    PP_R=$r timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/pp_times.txt
for b in 1 2; do
  for pp in 1 0; do
    FSNP_COOP_PP=$pp timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('bench B=$b PP=$pp: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f plan %s' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], [(c['kernel'][:22], c['sequences']) for c in r['roofline']['subband_plan']]))"
  done
done 2>&1 | tee -a gpurun_out/pp_times.txt
