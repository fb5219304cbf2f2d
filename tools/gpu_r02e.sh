#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
python tools/dump_costs.py 2>&1 | grep -E "per_cu|rowtile|workgroups|^LSTM|^GRU|x[0-9]" | paste -sd' ' | fold -w 400
: > gpurun_out/b_small.log
for args in "--batch 1" "--batch 2" "--batch 3" "--batch 5" "--batch 8" "--batch 16" "--batch 32 --mode parity" "--sequence-model GRU" "--batch 1 --sequence-model GRU"; do
  timeout 300 python bench.py $args --steps 6 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 2>&1 | tail -1 >> gpurun_out/b_small.log
done
python - <<'PY'
import json
for i, l in enumerate(open("gpurun_out/b_small.log")):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    plan = " + ".join("%s x%d" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"]) for c in r["roofline"]["subband_plan"])
    print("%-58s | %8.0f frames/s %8.3f ms  sub-band %7.3f | %s" % (r["config"]["workload"][:58], r["value"], r["ms_per_step"], r["roofline"]["subband_stage_ms"], plan))
PY
echo "== done"
