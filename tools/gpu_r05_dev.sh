#!/bin/bash
# scratch: quick checks between evidence runs
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ls -la _refstage 2>&1 | head -5
ls _refstage/reference 2>&1 | head -3
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "side_stream or weight_update or golden or bf16_ih_forward_b32 or attention or cbam or subband_num or stages or c_abi or plain_c" 2>&1 | tail -8 | tee gpurun_out/dev_pytest.log
for args in "--batch 1" "--batch 32" "--precision bf16_ih"; do
  timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['config']['workload'][:40], r['ms_per_step'], r['alt_ms_per_step'], r.get('dropin_ms_per_step'), r['roofline']['fullband_ms'], r['roofline'].get('alt_fullband_ms'), [c['kernel'].split(' ')[0]+' x%d'%c['sequences'] for c in r['roofline']['subband_plan']])" | tee -a gpurun_out/dev_bench.log
done
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --no-cpu-baseline --no-alt --batch 1 --steps 10 --warmup 2 --pipeline 0 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -22 "$f" | cut -c1-150 | tee $GRAFT_REPO_ROOT/gpurun_out/dev_stats_b1.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
