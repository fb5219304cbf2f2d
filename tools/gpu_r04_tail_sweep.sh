set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" "FSNP_COOP_SPLIT=3" "FSNP_OWN_CU=0" "FSNP_COOP_SPLIT=3 FSNP_OWN_CU=0" "FSNP_SIDE_PRIO=0" "FSNP_COOP_SPLIT=3 FSNP_SIDE_PRIO=0"; do
  for rep in 1 2; do
    env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$v'.ljust(40), 'ms_per_step %.3f' % d['ms_per_step'], 'fullband %.3f' % r['fullband_ms'], 'lstm_first %.3f' % r['avg_launch_ms'], 'stage %.3f' % r['subband_stage_ms'], [c['kernel'][:24] for c in r['subband_plan']])
"
  done
done | tee gpurun_out/tail_sweep.txt
