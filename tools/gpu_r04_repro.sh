#!/bin/bash
# round 4, task 1: reproduce / localise the driver-box failure (B = 3 x 126 s, 4th forward 0.35 rel): forward - host busy / GPU idle - forward
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/repro_long.py idle --iters ${ITERS:-8} --idle-kind oracle --tag idle_oracle 2>&1 | tail -30 | tee gpurun_out/repro_idle_oracle.log
