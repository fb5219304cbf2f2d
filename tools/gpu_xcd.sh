#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cooperative or coopn or gru2 or column_split or composite or b32_full" 2>&1 | tail -4
for n in 257 514 1285 2056 2720 4112 5440; do
  for x in 0 1; do
    PIN_R01=1 FSNP_COOP_XCD=$x python tools/time_lstm.py $n 128 5 2>&1 | tail -1
  done
done
for p in 1 0; do FSNP_SIDE_PRIO=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('PRIO $p ms/step %.3f alt %.3f fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['fullband_ms']))"; done
