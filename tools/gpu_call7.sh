#!/bin/bash
export TMPDIR=/tmp
for st in 2 4; do
  echo "== STAGES=$st"
  FSNP_GEMM_STAGES=$st timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "dma_gemm or stages_vs_reference" 2>&1 | tail -8
done
