#!/bin/bash
# round 5 development run
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "wave_owned or weight_update or exchange_verification or device_side_failure or sync_error_policy or half_tile_ping_pong or coopn_equals" 2>&1 | tail -25
echo "== timing (us per step, 128 steps): early prefill"
for pair in "514 32" "1285 64" "2048 96" "32 32" "32 64" "32 96"; do set -- $pair; COOPW=$2 timeout 120 python tools/time_lstm.py $1 128 5 2>&1 | tail -1; done
echo "== late prefill"
for pair in "514 32" "1285 64" "2048 96" "32 32" "32 64" "32 96"; do set -- $pair; FSNP_W_LATE=1 COOPW=$2 timeout 120 python tools/time_lstm.py $1 128 5 2>&1 | tail -1; done
echo "== again early"
for pair in "514 32" "1285 64" "2048 96"; do set -- $pair; COOPW=$2 timeout 120 python tools/time_lstm.py $1 128 5 2>&1 | tail -1; done
echo "== phase profiles"
timeout 120 python tools/pp_phase_profile.py 514 64 32 2>&1 | grep -v amdgpu
timeout 120 python tools/pp_phase_profile.py 2048 64 96 2>&1 | grep -v amdgpu
} 2>&1 | tee gpurun_out/dev.log
