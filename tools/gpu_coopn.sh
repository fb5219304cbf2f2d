#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "column_split or gru2_fc_dense or two_workgroups or other_hidden or golden" 2>&1 | tail -3
python tools/time_lstm.py 2056 128 5 2>&1 | tail -1
python tools/time_lstm.py 2720 128 5 2>&1 | tail -1
python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=8 ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
