#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsubnet.py -k "gru or GRU" -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/t_gru.log
