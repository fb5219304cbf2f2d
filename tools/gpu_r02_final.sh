#!/bin/bash
# round 2 evidence run: full GPU suite, smoke, the headline exactly as the driver runs it, rocprofv3 kernel stats of that
# command, PMC passes of the dominant kernel (each in its own run), the configuration table.  Everything -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
rm -rf gpurun_out/prof gpurun_out/pmc*
cd /tmp
# (profiling runs: FSNP_CALIBRATE=0 keeps the planner's one-off calibration launches out of the per-kernel statistics; the
#  built-in table yields the same B = 32 plan)
export FSNP_CALIBRATE=0   # (the default since the built-in table drives the planner; kept explicit)
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_bench.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/kernel_stats.csv
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctr -f csv -d $R/gpurun_out/pmc$i -o pmc -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/pmc$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_summary.txt
import csv, glob, collections
for i in (1, 2, 3, 4):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
        print(f"pmc{i} {k:70s} {c:30s} sum={v:.6g} n={n} per_launch={v/n:.6g}")
PY
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4
unset FSNP_CALIBRATE
: > gpurun_out/b_final.log
for args in "--batch 31" "--batch 64" "--seconds 10" "--seconds 10 --norm cumulative_layer_norm" "--mode parity" "--precision bf16_ih" "--precision bf16x3" "--precision bf16x3 --batch 64" "--batch 1" "--batch 2" "--batch 5" "--batch 8" "--batch 16" "--batch 21" "--batch 40" "--wave" "--model fullsubnet" "--model fullsubnet --batch 1" "--sequence-model GRU" "--sequence-model GRU --batch 1" "--sequence-model TCN"; do
  timeout 400 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_final.log
done
python tools/make_config_table.py > /dev/null
python tools/dump_costs.py > gpurun_out/dump_costs.log 2>&1
python - <<'PY' | tee gpurun_out/b_final.txt
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("HEADLINE %.0f frames/s %.3f ms (alt %.3f) frac %.4f lstm %.3f ms stage %.3f ms fullband %.3f cpu %.0f err %.2e" % (r["value"], r["ms_per_step"], r["alt_ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["cpu_baseline"]["value"], r["cirm_rel_err"]))
for l in open("gpurun_out/b_final.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    plan = " + ".join("%s x%d" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"]) for c in r["roofline"]["subband_plan"])
    alt = r["alt_ms_per_step"]
    print("%-22s | %-62s | %8.0f frames/s %8.3f ms (alt %s) sub-band %7.3f fullband %6.3f | %s | %s" % (r["metric"][38:60], r["config"]["workload"][:62], r["value"], r["ms_per_step"],
          "%.3f" % alt if alt else "-", r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["dtype"][:8], plan))
PY
head -14 gpurun_out/kernel_stats.csv | cut -c1-170
echo "== done"
