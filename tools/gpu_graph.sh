#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "graph or side_stream or repacks or stages or b32" -q --tb=short -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/t_graph.log
: > gpurun_out/b_graph.log
for g in 1 0; do for b in 1 5 32; do
  FSNP_GRAPH=$g timeout 200 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_graph.log
done; done
python - <<'PY'
import json
for i, l in enumerate(open("gpurun_out/b_graph.log")):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:300]); continue
    print("graph" if i < 3 else "plain", r["config"]["workload"][:24], "| %.0f frames/s  %.3f ms/fwd  lstm %.3f  fullband %.3f" % (
        r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["fullband_ms"]))
PY
