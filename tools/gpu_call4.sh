#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "32 64 1" "64 64 2" "96 64 3" "128 64 4" "257 64 2" "640 64 4"; do
  timeout 120 python tools/pp_phase_profile.py $cfg 2>&1 | tail -14
done 2>&1 | tee gpurun_out/pp_profile.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "variable_clip or host_time or pipelined_enhance or rccl or b32_10s_cumulative or bf16_ih_forward or weight_update or side_stream" 2>&1 | tail -25 | tee gpurun_out/new_tests.txt
