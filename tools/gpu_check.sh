#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined" 2>&1 | tail -4
for b in 5 2; do
  python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=$b ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
done
