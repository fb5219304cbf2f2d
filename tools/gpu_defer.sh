#!/bin/bash
# pipelined loop: plans that start with a column-split launch run on the side stream whole (FSNP_DEFER_SMALL) - A/B + tests
export TMPDIR=/tmp
for b in 1 8 16; do
for x in 1 0; do
  FSNP_DEFER_SMALL=$x python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=$b defer_small=$x ms/step %.3f alt %.3f subband %.3f fullband %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], r['value']))"
done
done
