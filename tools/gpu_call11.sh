#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cooperative or skewed or role_split or exchange_under_load or other_hidden or b1_cooperative or b32_full_vs_oracle or gru2_fc" 2>&1 | tail -5 | tee gpurun_out/fc_tests.txt
for n in 32 64 160 257 320 514; do
  for f in 1 0 1 0; do
    FSNP_FC_SPLIT=$f timeout 120 python tools/time_lstm.py $n 128 7 2>&1 | tail -1 | sed "s/^/FC_SPLIT=$f /"
  done
done 2>&1 | tee gpurun_out/fc_times.txt
for f in 1 0; do
  for b in 1 32; do
  FSNP_FC_SPLIT=$f timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('FC_SPLIT=$f B=$b: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms']))"
  done
done 2>&1 | tee -a gpurun_out/fc_times.txt
