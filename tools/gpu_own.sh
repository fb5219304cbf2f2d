#!/bin/bash
# A/B: the deferred K-split chunk of the pipelined loop claims its CUs' whole LDS (FSNP_OWN_CU, fsnp_abi.hip launch_sb_lstm)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined or b32_parity or exchange_under_load" 2>&1 | tail -3
for rep in 1 2; do
for x in 1 0; do
  FSNP_OWN_CU=$x python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 own_cu=$x ms/step %.3f alt %.3f fullband %s value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline'].get('fullband_ms'), r['value']))"
done
done
for x in 1 0; do
  FSNP_OWN_CU=$x python bench.py --batch 64 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=64 own_cu=$x ms/step %.3f alt %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['value']))"
done
