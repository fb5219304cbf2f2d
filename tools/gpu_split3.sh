#!/bin/bash
# pipelined loop: remainder chunk on the role-split kernel (96 CUs, 0.95 ms) vs the one-set kernel (48 CUs, 1.1 ms)
export TMPDIR=/tmp
for rep in 1 2; do
for x in 1 3; do
  FSNP_COOP_SPLIT=$x python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('split=$x ms/step %.3f alt %.3f fullband %.3f alt_fullband %.3f plan %s' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline']['fullband_ms'], r['roofline']['alt_fullband_ms'], r['roofline']['subband_plan'][-1]['kernel'][:24]))"
done
done
