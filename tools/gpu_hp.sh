#!/bin/bash
# one-off: half-tile ping-pong kernel, x gather by buffer loads
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "half_tile_ping_pong" 2>&1 | tail -3 | tee gpurun_out/hp_tests.txt
for n in 32 257 320; do HP=1 timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1; done | tee gpurun_out/hp_times.txt
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench B=1: %.3f ms/step (alt %.3f)' % (d['ms_per_step'], d.get('alt_ms_per_step') or -1))" | tee -a gpurun_out/hp_times.txt
