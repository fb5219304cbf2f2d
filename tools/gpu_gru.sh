#!/bin/bash
# GRU kernel with the packed cell update: parity + timing
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "gru" 2>&1 | tail -4
SEQ=GRU python tools/time_lstm.py 8192 128 5 2>&1 | tail -1
python bench.py --sequence-model GRU --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('GRU B=32 ms/step %.3f alt %.3f lstm %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['value']))"
