#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsubnet.py -m gpu -q -x -p no:cacheprovider -k "dma_gemm or stages_vs_reference or golden or b32_full_vs_oracle or subband_tcn or stft or enhance" 2>&1 | tail -6 | tee gpurun_out/fb_tests.txt
for st in 4 2; do
  FSNP_GEMM_STAGES=$st timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('STAGES=$st B=32: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f alt_fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], r['roofline']['alt_fullband_ms'] or 0))"
  FSNP_GEMM_STAGES=$st timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('STAGES=$st B=1: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f alt_fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], r['roofline']['alt_fullband_ms'] or 0))"
done 2>&1 | tee gpurun_out/fb_times.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_serial.csv
rm -rf gpurun_out/prof
head -20 gpurun_out/kernel_stats_serial.csv | cut -c1-160
