#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bf16" 2>&1 | tail -5
python -c "
import json; d=json.load(open('gpurun_out/parity_report.json')); print({k:v for k,v in d.items() if 'bf16x3' in k})"
for p in fp32 bf16_ih bf16x3; do
python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$p ms/step %.3f alt %.3f value %.0f lstm %.3f fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value'], r['roofline']['avg_launch_ms'], r['roofline']['fullband_ms']))"
done
python bench.py --precision bf16x3 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('bf16x3 B=64 ms/step %.3f alt %.3f value %.0f lstm %.3f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value'], r['roofline']['avg_launch_ms']))"
