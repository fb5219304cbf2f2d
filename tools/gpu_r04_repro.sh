#!/bin/bash
# round 4, task 1: soak of the round-3 failure's shape (B = 3 x 126 s) - the test's four-forward sequence on a fresh handle per
# iteration, every output and every stage buffer compared bitwise with the first iteration's; then forward - host busy / GPU idle - forward
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/repro_long.py forward --iters ${ITERS:-25} --oracle 1 --tag fwd_${TAG:-x} 2>&1 | tail -8 | tee gpurun_out/repro_fwd_${TAG:-x}.log
timeout 600 python tools/repro_long.py idle --iters ${IDLE_ITERS:-3} --idle-kind oracle --tag idle_${TAG:-x} 2>&1 | tail -6 | tee gpurun_out/repro_idle_${TAG:-x}.log
