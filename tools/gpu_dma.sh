#!/bin/bash
# A/B of the DMA GEMM kernels of the full-band TCN stacks (tcn.hip tcn_gemm_dma_kernel, FSNP_GEMM_DMA)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or b32 or 10s or strided or tcn or batch_independ" 2>&1 | tail -4
for x in 1 0; do
  FSNP_GEMM_DMA=$x python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 dma=$x ms/step %.3f alt %.3f fullband %.3f value %.0f' % (r['ms_per_step'], r['alt_ms_per_step'], r['roofline'].get('fullband_ms'), r['value']))"
  FSNP_GEMM_DMA=$x python bench.py --steps 20 --warmup 4 --no-cpu-baseline --pipeline 0 --no-alt 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B=32 serial dma=$x ms/step %.3f fullband %.3f value %.0f' % (r['ms_per_step'], r['roofline'].get('fullband_ms'), r['value']))"
done
cd /tmp && FSNP_CALIBRATE=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_dma -o dma -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipeline 0 --no-alt > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_dma -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/dma_kernel_stats.csv; head -14 gpurun_out/dma_kernel_stats.csv | cut -c1-150
