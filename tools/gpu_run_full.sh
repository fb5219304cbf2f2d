mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
for b in 32 3 5 64; do timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench B=$b: %.3f ms/step (alt %.3f) %.0f frames/s' % (d['ms_per_step'], d.get('alt_ms_per_step') or -1, d['value']), [c['kernel'][:22]+' x%d' % c['sequences'] for c in d['roofline']['subband_plan']])"; done | tee gpurun_out/skew8_bench.txt
