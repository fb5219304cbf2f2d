#!/bin/bash
# round 4: the whole GPU suite in its new order (boundary / configuration-level tests first, soaks last) + smoke.
# Writes gpurun_out/pytest_gpu.log with the commit the sources were at and whether the kernel sources changed under the run.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
sum0=$(cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h | sha256sum | cut -c1-16)
{
  echo "commit: ${FSNP_HEAD:-unknown}   csrc sha256[:16] at start: $sum0   library stamp: $(cat fullsubnet_plus_amd/libfsnp_hip.so.stamp | cut -c1-16)"
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 2>&1 | tail -45
  sum1=$(cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h | sha256sum | cut -c1-16)
  echo "csrc sha256[:16] at end: $sum1"
} | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
