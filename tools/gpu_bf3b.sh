#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bf16x3" 2>&1 | tail -3
python bench.py --precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bf16x3 B=32 ms/step %.3f alt %.3f lstm %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['value']))"
