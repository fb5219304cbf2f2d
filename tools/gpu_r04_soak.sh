#!/bin/bash
# round 4: the drift / long-recurrence soaks on their own, then the whole GPU suite in its new order
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_soak.py -m gpu -q --tb=short -p no:cacheprovider --durations=12 2>&1 | tail -60 | tee gpurun_out/pytest_soak.log
