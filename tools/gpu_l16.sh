#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "half_tile or b32_parity or golden or sharded_parity" 2>&1 | tail -5
for n in 3855 4096 4112; do
  for x in 1 0; do
    FSNP_LSTM16=$x python tools/time_lstm.py $n 128 5 2>&1 | tail -1 | sed "s/^/lstm16=$x /"
  done
done
for x in 1 0; do FSNP_LSTM16=$x python bench.py --mode parity --steps 6 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('parity B=32 lstm16=$x ms/step %.3f subband %.3f value %.0f' % (r['ms_per_step'], r['roofline']['subband_stage_ms'], r['value']))"; done
for x in 1 0; do FSNP_LSTM16=$x python bench.py --batch 16 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=16 lstm16=$x ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"; done
