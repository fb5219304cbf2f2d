#!/bin/bash
# one-off: padded arrival counters for every column-split kernel + the half-tile ping-pong kernel
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "half_tile_ping_pong or ping_pong_k_split or column or coop or lstm2_fc or skew or split or exchange" 2>&1 | tail -8 | tee gpurun_out/hp_tests.txt
for n in 32 64 160 257 320 514 1285 2056; do
  timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
done | tee gpurun_out/hp_times.txt
HP=1 timeout 120 python tools/time_lstm.py 257 128 5 2>&1 | tail -1 | tee -a gpurun_out/hp_times.txt
for b in 1 2 8; do timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench B=$b: %.3f ms/step (alt %.3f)' % (d['ms_per_step'], d.get('alt_ms_per_step') or -1), [c['kernel'][:24] for c in d['roofline']['subband_plan']])"; done | tee -a gpurun_out/hp_times.txt
