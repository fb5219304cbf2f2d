#!/bin/bash
# round 4: the float4 epilogue of the DMA GEMM kernels - parity subset, bench A/B (graph replay on / off), rocprofv3 kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "dma_gemm or stages_vs_reference or forward_vs_reference_golden or b32_full_vs_oracle or b32_batch_independence or graph_replay" 2>&1 | tail -8 | tee gpurun_out/epi_pytest.log
for b in 32 1 8; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b$b.json
  FSNP_GRAPH=1 timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b${b}_graph.json
done
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/epi_prof.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/epi_kernel_stats_serial.csv
rm -rf $R/gpurun_out/prof
cd $R
python - <<'PY'
import json
for b in (32, 1, 8):
    for tag in ("", "_graph"):
        try:
            d = json.loads(open(f"gpurun_out/epi_bench_b{b}{tag}.json").read())
            r = d["roofline"]
            print(f"B={b}{tag}: ms_per_step {d['ms_per_step']:.3f} value {d['value']:.0f} fullband_ms {r.get('fullband_ms')} alt_fullband_ms {r.get('alt_fullband_ms')} alt_ms {r.get('alt_ms_per_step')}")
        except Exception as e:
            print(b, tag, "failed", e)
PY
head -30 gpurun_out/epi_kernel_stats_serial.csv
