#!/bin/bash
# scratch: prologue launch with and without the weight watch (B = 1 and B = 32)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/watch_cost.py <<'PY'
import sys, torch
from fullsubnet_plus_amd import FullSubNet_Plus
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
B, every = int(sys.argv[1]), int(sys.argv[2])
m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(make_state_dict(0, "default")); m = m.cuda().eval(); m.batch_mode = "full"; m.error_check = "deferred"
m.weight_watch_every = every
ins = [t.cuda() for t in make_inputs(B, 2.0, 5)]
for _ in range(12): m(*ins)
torch.cuda.synchronize(); m.check_errors()
PY
cd /tmp
for B in 1 32; do for every in 1 1000000; do
  rm -rf /tmp/prof
  PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof -o trace -- python /tmp/watch_cost.py $B $every > /dev/null 2>&1
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "B=$B weight_watch_every=$every: $(grep prologue_kernel $f | cut -d, -f2-4)" | tee -a $GRAFT_REPO_ROOT/gpurun_out/watch_cost.txt
done; done
