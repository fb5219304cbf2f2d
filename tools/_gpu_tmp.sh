#!/bin/bash
# scratch: fused TCN stack kernel - parity, phase profile, timings with / without
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "dma_gemm or stages_vs_reference or forward_vs_reference_golden" 2>&1 | tail -25 | tee gpurun_out/dev_pytest.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dev_tcn_profile.txt
import torch, time
from fullsubnet_plus_amd import FullSubNet_Plus
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
for B in (1, 2, 4):
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(make_state_dict(0, "default")); m = m.cuda().eval(); m.batch_mode = "full"
    ins = [t.cuda() for t in make_inputs(B, 2.0, 5)]
    m.tcn_profile(True)
    for _ in range(5): m(*ins)
    torch.cuda.synchronize()
    p = m.tcn_profile(True)
    print(f"B={B} stamps (shader cycles since block start; us at 2.1 GHz):", {k: (int(v), round(v / 2100, 2)) for k, v in p.items()})
PY
: > gpurun_out/dev_bench.log
for B in 1 2 4 8; do
  for F in 1 0; do
    FSNP_TCN_FUSED=$F timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --probe-ms 0 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('B=$B fused=$F', 'ms', r['ms_per_step'], 'b2b', r['alt_ms_per_step'], 'dropin', r.get('dropin_ms_per_step'), 'fullband', r['roofline']['fullband_ms'], 'alt_fullband', r['roofline'].get('alt_fullband_ms'), 'err', r.get('cirm_rel_err'))" | tee -a gpurun_out/dev_bench.log
  done
done
for F in 0 4000; do
  FSNP_TCN_FUSED=$F timeout 300 python bench.py --batch 1 --seconds 10 --steps 20 --warmup 5 --no-cpu-baseline --probe-ms 0 --pipeline 0 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('B=1 10s fused=$F', 'ms', r['ms_per_step'], 'b2b', r['alt_ms_per_step'], 'dropin', r.get('dropin_ms_per_step'), 'fullband', r['roofline']['fullband_ms'], 'alt_fullband', r['roofline'].get('alt_fullband_ms'), 'err', r.get('cirm_rel_err'))" | tee -a gpurun_out/dev_bench.log
done
cd /tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 --probe-ms 0 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 | tee $GRAFT_REPO_ROOT/gpurun_out/dev_kernel_stats_b1.csv
