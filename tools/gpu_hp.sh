#!/bin/bash
# one-off: half-tile ping-pong kernel after the deferred publish
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "half_tile_ping_pong" 2>&1 | tail -6 | tee gpurun_out/hp_tests.txt
for n in 32 257 320; do
  HP=1 timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
done | tee gpurun_out/hp_times.txt
timeout 120 python tools/pp_phase_profile.py 257 64 0 2>&1 | tail -17 | tee gpurun_out/hp_phase_profile.txt
timeout 300 python tools/dump_costs.py 2>&1 | grep -v amdgpu > gpurun_out/planner_costs_padded.txt
