#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "skewed or cooperative or gru2 or column_split or composite or b32_full or pipelined or golden" 2>&1 | tail -6
for n in 32 160 257 514 1285; do
  for x in 0 1; do
    PIN_R01=1 FSNP_COOP_SKEW=$x python tools/time_lstm.py $n 128 5 2>&1 | tail -1 | sed "s/^/skew=$x /"
  done
done
python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-alt --pipeline 0 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=1 ms/step %.3f subband %.3f fullband %.3f' % (r['ms_per_step'], r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms']))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 ms/step %.3f alt %.3f' % (r['ms_per_step'], r['alt_ms_per_step']))"
