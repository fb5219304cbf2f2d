#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "role_split or b32 or skewed or pipelined or exchange_under_load or two_handles or sync_policy" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 ms/step %.3f alt %.3f lstm %.3f frac %.4f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['value']))"
FSNP_COOP_SPLIT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 split=0 ms/step %.3f alt %.3f lstm %.3f frac %.4f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['value']))"
python bench.py --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=64 ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
