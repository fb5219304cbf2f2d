#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "half_tile or b32_parity or lstm2_fc_dense" 2>&1 | tail -3
python tools/time_lstm.py 4096 128 5 2>&1 | tail -1
python bench.py --mode parity --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('parity B=32 ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
python bench.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=16 ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
