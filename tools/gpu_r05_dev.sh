#!/bin/bash
# scratch: the extended verification test, a verification soak over the default plans, the whole suite once more (run C)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
digest() { cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h | sha256sum | cut -c1-16; }
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "exchange_verification" 2>&1 | tail -5 | tee gpurun_out/dev_pytest.log
timeout 900 python tools/verify_soak.py 25 2>&1 | grep -v amdgpu.ids | tail -20
{
  echo "commit: ${FSNP_HEAD:-unknown}   csrc sha256[:16] at start: $(digest)   library stamp: $(cut -c1-16 fullsubnet_plus_amd/libfsnp_hip.so.stamp)   $(date -u +%FT%TZ)"
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 2>&1 | tail -20
  echo "csrc sha256[:16] at end: $(digest)"
} | tee gpurun_out/pytest_gpu_runC.log
