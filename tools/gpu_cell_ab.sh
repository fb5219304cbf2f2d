#!/bin/bash
# same-box A/B: libfsnp_hip.so (packed two-cell update) vs libfsnp_hip_oldcell.so (lstm.hip of the previous commit), swapped in place
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/fullsubnet_plus_amd
cp libfsnp_hip.so /tmp/new.so; cp libfsnp_hip_oldcell.so /tmp/old.so
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in new old; do
  cp /tmp/$v.so fullsubnet_plus_amd/libfsnp_hip.so
  python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v ms/step %.3f lstm %.3f frac %.4f' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
done
done
cp /tmp/new.so fullsubnet_plus_amd/libfsnp_hip.so
