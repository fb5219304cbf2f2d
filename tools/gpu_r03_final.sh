#!/bin/bash
# round 3 evidence run: full GPU suite, smoke, the headline exactly as the driver runs it, rocprofv3 kernel stats of that command
# (serving loop) and of the back-to-back loop, PMC passes of the dominant kernel (each in its own run) -> profiles/lstm_pmc.json,
# the configuration table, the RCCL run at N = 1, the phase profile of the opt-in ping-pong kernel.  Everything -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
rm -rf gpurun_out/prof gpurun_out/pmc*
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_bench.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/kernel_stats.csv
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/prof_bench_serial.log 2>&1
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/kernel_stats_serial_loop.csv
rm -rf $R/gpurun_out/prof
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctr -f csv -d $R/gpurun_out/pmc$i -o pmc -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/pmc$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_summary.txt
import csv, glob, collections, json
vals = {}
for i in (1, 2, 3, 4):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
        print(f"pmc{i} {k:70s} {c:30s} sum={v:.6g} n={n} per_launch={v/n:.6g}")
        if "lstm2_fc_kernel<384, 40, 2, 0, false, 4, false>" in k:
            vals[c] = v / n
if {"FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(vals):
    out = {"kernel": "lstm2_fc_kernel<384,40,2,EX=0,NW=4> (the first chunk of the B=32 plan: 8192 of the 8224 sequences)",
           "workload": "B=32 x 2 s, full mode: 8192 sequences x 128 steps on the one-tile-per-CU kernel (+ 32 on a K-split kernel)",
           "source": "rocprofv3 --pmc, separate passes (tools/gpu_r03_final.sh, final run of round 3 at HEAD); profiles/r03_pmc_summary.txt",
           "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
           "TCC_HIT_per_launch": vals.get("TCC_HIT_sum"), "TCC_MISS_per_launch": vals.get("TCC_MISS_sum"),
           "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": vals["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE_per_launch": vals["GRBM_GUI_ACTIVE"],
           "mfma_busy_frac": vals["SQ_VALU_MFMA_BUSY_CYCLES"] * 8 / (1024 * vals["GRBM_GUI_ACTIVE"]),
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads are tallied at half); counts fabric requests, i.e. L2 misses served by the 256 MiB Infinity Cache are included - an upper bound on HBM bytes.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES * 8 / (1024 SIMDs * GRBM_GUI_ACTIVE)",
           "traffic_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
           "SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch": vals.get("SQ_INSTS_VALU_MFMA_MOPS_F32")}
    json.dump(out, open("gpurun_out/lstm_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4
: > gpurun_out/b_final.log
for args in "--batch 31" "--batch 64" "--seconds 10" "--seconds 10 --norm cumulative_layer_norm" "--mode parity" "--precision bf16_ih" "--precision bf16x3" "--batch 1" "--batch 2" "--batch 5" "--batch 8" "--batch 16" "--batch 21" "--batch 40" "--wave" "--model fullsubnet" "--model fullsubnet --batch 1" "--sequence-model GRU" "--sequence-model GRU --batch 1" "--sequence-model TCN"; do
  timeout 400 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/b_final.log
done
python tools/make_config_table.py r03 > /dev/null
python tools/dump_costs.py > gpurun_out/dump_costs.log 2>&1
# RCCL at world size 1: the driver's multi-GPU launch line with N = 1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_rccl_n1.log
# B = 1 without the half-tile ping-pong kernel (the round-2 kernels with padded counters)
FSNP_COOP_HP=0 timeout 300 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b1_pp.log
timeout 120 python tools/pp_phase_profile.py 64 64 2 > gpurun_out/pp_profile_64x2.txt 2>&1
python - <<'PY' | tee gpurun_out/b_final.txt
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("HEADLINE %.0f frames/s %.3f ms (alt %.3f) frac %.4f lstm %.3f ms stage %.3f ms fullband %.3f (alt %.3f) cpu %.0f err %.2e checked %d utterances" % (r["value"], r["ms_per_step"], r["alt_ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["roofline"]["alt_fullband_ms"], r["cpu_baseline"]["value"], r["cirm_rel_err"], len(r["cirm_checked_utterances"])))
for l in open("gpurun_out/b_final.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    plan = " + ".join("%s x%d" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"]) for c in r["roofline"]["subband_plan"])
    alt = r["alt_ms_per_step"]
    print("%-22s | %-62s | %8.0f frames/s %8.3f ms (alt %s) sub-band %7.3f fullband %6.3f | %s | %s" % (r["metric"][38:60], r["config"]["workload"][:62], r["value"], r["ms_per_step"],
          "%.3f" % alt if alt else "-", r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["dtype"][:8], plan))
for f in ("bench_rccl_n1.log", "bench_b1_pp.log"):
    try:
        r = json.loads(open("gpurun_out/" + f).read())
        print(f, "%.3f ms/step" % r["ms_per_step"], r.get("dist"), r.get("gather_ms"))
    except Exception as e:
        print(f, "??", e)
PY
# B = 1 (the reference CLI's batch): fabric reads of the sub-band kernel with / without the half-tile ping-pong kernel, its phase
# profile, isolated timings of the column-split kernels
cd /tmp
for hp in 1 0; do
  rm -rf $R/gpurun_out/pmcb1
  FSNP_COOP_HP=$hp timeout 300 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/gpurun_out/pmcb1 -o pmc -- python $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 > $R/gpurun_out/pmcb1_$hp.log 2>&1
  python - <<PY | tee -a $R/gpurun_out/b1_fetch_size.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$R/gpurun_out/pmcb1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "lstm2_coop" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][:60]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in agg.items():
    print("FSNP_COOP_HP=$hp  %-60s FETCH_SIZE per launch = %.4g KB (x2 per MI355X_MICROARCH.md = %.3f GB), %d launches" % (k, v / n, 2 * v / n * 1024 / 1e9, n))
PY
done
rm -rf $R/gpurun_out/pmcb1
cd $R
timeout 120 python tools/pp_phase_profile.py 257 64 0 2>&1 | grep -v amdgpu > gpurun_out/hp_phase_profile.txt
for n in 32 160 257 320; do
  FSNP_COOP_HP=0 timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
  HP=1 timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1
done > gpurun_out/hp_times_final.txt
head -14 gpurun_out/kernel_stats.csv | cut -c1-170
echo "== done"
