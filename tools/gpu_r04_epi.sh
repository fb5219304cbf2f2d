#!/bin/bash
# round 4: full-band GEMM work - parity subset, bench A/B (FSNP_GEMM_BM64 on / off), rocprofv3 kernel stats
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -m gpu -q --tb=short -p no:cacheprovider -k "dma_gemm or stages_vs_reference or forward_vs_reference_golden or b32_full_vs_oracle or b32_batch_independence or b32_10s_full or long_recurrence_forward" 2>&1 | tail -12 | tee gpurun_out/epi_pytest.log
for b in 32 24 40; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b$b.json
  FSNP_GEMM_BM64=0 timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b${b}_bm128.json
done
timeout 300 python bench.py --seconds 10 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b32_10s.json
FSNP_GEMM_BM64=0 timeout 300 python bench.py --seconds 10 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/epi_bench_b32_10s_bm128.json
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/epi_prof.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/epi_kernel_stats_serial.csv
rm -rf $R/gpurun_out/prof
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/epi_bench_*.json")):
    try:
        d = json.loads(open(f).read())
        r = d["roofline"]
        print(f"{f[21:-5]:16s} ms_per_step {d['ms_per_step']:.3f} value {d['value']:.0f} fullband_ms {r.get('fullband_ms'):.4f} alt_fullband_ms {r.get('alt_fullband_ms'):.4f} alt_ms {d.get('alt_ms_per_step')}")
    except Exception as e:
        print(f, "failed", e)
PY
grep -E "tcn_|dwconv" gpurun_out/epi_kernel_stats_serial.csv | cut -c1-150
