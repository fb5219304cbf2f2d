#!/bin/bash
# role-split K-split kernel (lstm2_coop_split_kernel): correctness under FSNP_COOP_SPLIT=2 (wherever it fits) + per-step times
export TMPDIR=/tmp
FSNP_COOP_SPLIT=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or lstm2_fc_dense or exchange_under_load or batch_independence" 2>&1 | tail -4
for n in 32 64 160 257 320 514 672; do
  for x in 2 0; do
    FSNP_COOP_SPLIT=$x timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1 | sed "s/^/split=$x /"
  done
done
