#!/bin/bash
# scratch: verification soak over the default plans
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/verify_soak.py 22 2>&1 | grep -v amdgpu.ids | tail -20
