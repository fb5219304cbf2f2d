#!/bin/bash
# the whole GPU suite, smoke, the headline as the driver runs it (+ the verification soak with "soak" as first argument); logs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
digest() { cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h | sha256sum | cut -c1-16; }
{
  echo "commit: ${FSNP_HEAD:-unknown}   csrc sha256[:16] at start: $(digest)   library stamp: $(cut -c1-16 fullsubnet_plus_amd/libfsnp_hip.so.stamp)   $(date -u +%FT%TZ)"
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 2>&1 | tail -20
  echo "csrc sha256[:16] at end: $(digest)"
} | tee gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_final.log
python -c "
import json; r=json.load(open('gpurun_out/bench_final.log')); print(r['value'], r['ms_per_step'], r['alt_ms_per_step'], r['dropin_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['value'], r['cirm_rel_err'])"
if [ "${1:-}" = "soak" ]; then timeout 600 python tools/verify_soak.py 12 2>&1 | grep -v amdgpu.ids | tail -12; fi
