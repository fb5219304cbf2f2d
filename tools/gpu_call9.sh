#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsubnet.py -m gpu -q -x -p no:cacheprovider -k "dma_gemm or stages_vs_reference or golden or b32_full_vs_oracle or graph_replay or pipelined_mode or two_handles or side_stream or variable_clip or fbh300" 2>&1 | tail -6 | tee gpurun_out/fb_tests.txt
for fs in 1 0; do
  for b in 32 1 8; do
  FSNP_FB_STREAMS=$fs timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('FB_STREAMS=$fs B=$b: %.3f ms/step (alt %.3f) sub-band %.3f fullband %.3f alt_fullband %.3f' % (r['ms_per_step'], r['alt_ms_per_step'] or 0, r['roofline']['subband_stage_ms'], r['roofline']['fullband_ms'], r['roofline']['alt_fullband_ms'] or 0))"
  done
done 2>&1 | tee gpurun_out/fb_times.txt
