#!/bin/bash
# scratch: x gather behind the arrival (lstm_coopw.hip): per-step times and parity
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for pair in "514 32" "672 32" "1285 64" "1344 64"; do set -- $pair; COOPW=$2 timeout 120 python tools/time_lstm.py $1 128 5 2>&1 | tail -1; done
timeout 120 python tools/time_lstm.py 2056 128 5 2>&1 | tail -1
} | tee gpurun_out/dev_times.txt
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "wave_owned or exchange_verification or (drift and 514) or (long_recurrence_kernels and 1285)" 2>&1 | tail -5 | tee gpurun_out/dev_pytest.log
for args in "--batch 2" "--batch 5" "--batch 8"; do
  timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['config']['workload'][:40], r['ms_per_step'], r['alt_ms_per_step'], r.get('dropin_ms_per_step'))" | tee -a gpurun_out/dev_bench.log
done
