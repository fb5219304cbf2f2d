#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python tools/cum_laplace_spread.py _refstage/reference > gpurun_out/cum_laplace.log 2>&1; tail -40 gpurun_out/cum_laplace.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
