#!/bin/bash
# round 6 evidence run.  Usage (from the build container): gpurun -- "FSNP_HEAD=<commit> bash tools/gpu_r06_final.sh [ref]"
#   1. the whole GPU suite (run A), smoke, the headline exactly as the driver runs it
#   2. rocprofv3 kernel stats of that command (serving loop), of the back-to-back loop and of B = 1 / B = 2; PMC passes of the dominant kernel
#      (each counter set in its own run) -> profiles/lstm_pmc.json
#   3. the configuration table, RCCL at N = 1, phase profiles of the B = 1 and B = 2 sub-band kernels, the cost table, verification overhead
#   4. [ref] the unmodified reference CLI end to end (reference staged under the git-ignored _refstage/ for this one call)
#   5. the whole GPU suite again (run B): two green runs on one HEAD
# The pytest logs carry the commit and a digest of the kernel sources at start and end of each run.  Everything -> gpurun_out/.
set -u
REF=${1:-}          # (saved: the column-split timing loop below re-uses the positional parameters)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
digest() { cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h include/fsnp_debug.h | sha256sum | cut -c1-16; }
suite() {   # $1 = log name
  {
    echo "commit: ${FSNP_HEAD:-unknown}   csrc sha256[:16] at start: $(digest)   library stamp: $(cut -c1-16 fullsubnet_plus_amd/libfsnp_hip.so.stamp)   $(date -u +%FT%TZ)"
    timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=25 2>&1 | tail -50
    echo "csrc sha256[:16] at end: $(digest)"
  } | tee gpurun_out/$1
}
suite pytest_gpu_runA.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench.log
cd /tmp
prof() {   # $1 = output csv name, rest = bench.py arguments
  local name=$1; shift
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --gpus 1 --no-cpu-baseline --no-alt "$@" > $R/gpurun_out/prof_$name.log 2>&1
  f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/$name.csv
  rm -rf $R/gpurun_out/prof
}
prof kernel_stats --steps 5 --warmup 2
prof kernel_stats_serial_loop --steps 5 --warmup 2 --pipeline 0
prof kernel_stats_b1 --batch 1 --steps 10 --warmup 2 --pipeline 0
prof kernel_stats_b2 --batch 2 --steps 10 --warmup 2 --pipeline 0
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
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"pmc{i} {k:70s} {c:30s} sum={v:.6g} n={n} per_launch={v/n:.6g}")
        if "lstm2_fc_kernel<384, 40, 2, 0, false, 4, false>" in k:
            vals[c] = v / n
if {"FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(vals):
    out = {"kernel": "lstm2_fc_kernel<384,40,2,EX=0,NW=4> (the first chunk of the B=32 plan: 8192 of the 8224 sequences)",
           "workload": "B=32 x 2 s, full mode: 8192 sequences x 128 steps on the one-tile-per-CU kernel (+ 32 on a K-split kernel)",
           "source": "rocprofv3 --pmc, separate passes (tools/gpu_r06_final.sh, final run of round 6 at HEAD); profiles/r06_pmc_summary.txt",
           "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
           "TCC_HIT_per_launch": vals.get("TCC_HIT_sum"), "TCC_MISS_per_launch": vals.get("TCC_MISS_sum"),
           "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": vals["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE_per_launch": vals["GRBM_GUI_ACTIVE"],
           "mfma_busy_frac": vals["SQ_VALU_MFMA_BUSY_CYCLES"] * 8 / (1024 * vals["GRBM_GUI_ACTIVE"]),
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads are tallied at half); counts fabric requests, i.e. L2 misses served by the 256 MiB Infinity Cache are included - an upper bound on HBM bytes.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES * 8 / (1024 SIMDs * GRBM_GUI_ACTIVE)",
           "traffic_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
           "kernel_source_sha256": __import__("hashlib").sha256(b"".join(open("fullsubnet_plus_amd/csrc/" + n, "rb").read() for n in ("lstm.hip", "lstm_common.h", "fsnp_common.h"))).hexdigest(),
           "SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch": vals.get("SQ_INSTS_VALU_MFMA_MOPS_F32")}
    json.dump(out, open("gpurun_out/lstm_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4
# matrix-pipe counters of the small-batch sub-band kernels (busy on OCCUPIED CUs) -> profiles/r06_small_batch_pmc.txt
cd /tmp
: > $R/gpurun_out/small_batch_pmc.txt
for B in 1 2 4 8; do
  rm -rf /tmp/pmc
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -f csv -d /tmp/pmc -o pmc -- python $R/bench.py --gpus 1 --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-alt --pipeline 0 --probe-ms 0 > /dev/null 2>&1
  python - $B <<'PY' | tee -a $R/gpurun_out/small_batch_pmc.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
wgs = {}
for f in glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lstm2_" not in k: continue
        k = k.split("(")[0][:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        try:
            wgs[k] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        except Exception:
            pass
for k, c in agg.items():
    if not {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(c): continue
    n = c["GRBM_GUI_ACTIVE"][1]
    busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n, c["GRBM_GUI_ACTIVE"][0] / n
    chip = busy * 8 / (1024 * act)
    w = wgs.get(k)
    occ = ("  workgroups %d (one per CU)  busy on occupied CUs %.3f" % (w, chip * 256 / w)) if w and w <= 256 else ""
    print("B=%s %-60s launches %d  MFMA_BUSY %.4g  GUI_ACTIVE %.4g  matrix-pipe busy over the whole chip %.3f%s" % (sys.argv[1], k, n, busy, act, chip, occ))
PY
done
cd $R
: > gpurun_out/b_final.log
for args in "--batch 1" "--batch 2" "--batch 3" "--batch 4" "--batch 5" "--batch 6" "--batch 7" "--batch 8" "--batch 10" "--batch 12" "--batch 16" "--batch 21" "--batch 31" "--batch 40" "--batch 64" "--seconds 10" "--seconds 10 --norm cumulative_layer_norm" "--mode parity" "--mode parity --precision bf16_ih" "--precision bf16_ih" "--batch 16 --precision bf16_ih" "--wave" "--model fullsubnet" "--model fullsubnet --batch 1" "--model fullsubnet --batch 4" "--model fullsubnet --batch 8" "--model fullsubnet --batch 16" "--sequence-model GRU" "--sequence-model GRU --batch 1" "--sequence-model TCN"; do
  timeout 400 python bench.py $args --steps 10 --warmup 2 --no-cpu-baseline --probe-ms 0 2>&1 | tail -1 >> gpurun_out/b_final.log
done
python tools/make_config_table.py r06 > /dev/null
python tools/dump_costs.py > gpurun_out/dump_costs.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_rccl_n1.log
timeout 120 python tools/pp_phase_profile.py 257 64 0 2>&1 | grep -v amdgpu > gpurun_out/hp_phase_profile.txt
timeout 120 python tools/pp_phase_profile.py 514 64 32 2>&1 | grep -v amdgpu > gpurun_out/coopw_phase_profile.txt
timeout 120 python tools/pp_phase_profile.py 1285 64 64 2>&1 | grep -v amdgpu >> gpurun_out/coopw_phase_profile.txt
timeout 120 python tools/pp_phase_profile.py 2048 64 96 2>&1 | grep -v amdgpu >> gpurun_out/coopw_phase_profile.txt
{
  echo "== per-step times of the column-split kernels (tools/time_lstm.py, 128 steps)"
  for pair in "514 32" "672 32" "1285 64" "1344 64" "2048 96" "32 32" "32 64" "32 96"; do set -- $pair; COOPW=$2 timeout 120 python tools/time_lstm.py $1 128 5 2>&1 | tail -1; done
  for n in 514 1285 2056; do NOCOOPW=1 timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1; timeout 120 python tools/time_lstm.py $n 128 5 2>&1 | tail -1; done
  HP=1 timeout 100 python tools/time_lstm.py 257 128 5 2>&1 | tail -1
  FSNP_HP_WAVE=0 HP=1 timeout 100 python tools/time_lstm.py 257 128 5 2>&1 | tail -1 | sed "s/^/FSNP_HP_WAVE=0 (lstm_hp.hip) /"
} > gpurun_out/column_split_times.txt 2>&1
python - <<'PY' | tee gpurun_out/verify_overhead.txt
# exchange verification (fsnp_set_verify): per-forward overhead at N = 64 and the cost of one verified forward, B = 1 and B = 32
import time, torch
from fullsubnet_plus_amd import FullSubNet_Plus
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
for B in (1, 32):
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(make_state_dict(0, "default")); m = m.cuda().eval(); m.batch_mode = "full"; m.error_check = "deferred"
    ins = [t.cuda() for t in make_inputs(B, 2.0, 5)]
    res = {}
    for every in (0, 64, 1):
        m.verify_every = every
        for _ in range(3): m(*ins)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 128 if every != 1 else 8
        for _ in range(n): m(*ins)
        torch.cuda.synchronize(); res[every] = (time.perf_counter() - t0) / n * 1e3
    m.check_errors()
    print(f"B={B}: plain {res[0]:.3f} ms, verify_every=64 {res[64]:.3f} ms (+{res[64]-res[0]:.3f}), every forward verified {res[1]:.3f} ms; verification passes run: {m.verify_count()}")
PY
python - <<'PY' | tee gpurun_out/b_final.txt
import json
r = json.loads(open("gpurun_out/bench.log").read())
print("HEADLINE %.0f frames/s %.3f ms (b2b %.3f, drop-in %.3f) frac %.4f lstm %.3f ms stage %.3f ms fullband %.3f (alt %.3f) cpu %.0f err %.2e checked %d utterances" % (r["value"], r["ms_per_step"], r["alt_ms_per_step"], r["dropin_ms_per_step"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], r["roofline"]["alt_fullband_ms"], r["cpu_baseline"]["value"], r["cirm_rel_err"], len(r["cirm_checked_utterances"])))
for l in open("gpurun_out/b_final.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:200]); continue
    plan = " + ".join("%s x%d%s" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"], "*" if c.get("deferred_when_pipelined") else "") for c in r["roofline"]["subband_plan"])
    alt, di = r["alt_ms_per_step"], r.get("dropin_ms_per_step")
    print("%-62s | %8.0f frames/s %8.3f ms (b2b %s drop-in %s) sub-band %7.3f fullband %6.3f (alt %s) | %s | %s" % (r["config"]["workload"][:62], r["value"], r["ms_per_step"],
          "%.3f" % alt if alt else "-", "%.3f" % di if di else "-", r["roofline"]["subband_stage_ms"], r["roofline"]["fullband_ms"], "%.3f" % r["roofline"]["alt_fullband_ms"] if r["roofline"].get("alt_fullband_ms") else "-", r["dtype"][:8], plan))
try:
    r = json.loads(open("gpurun_out/bench_rccl_n1.log").read())
    print("bench_rccl_n1 %.3f ms/step" % r["ms_per_step"], r.get("dist"), r.get("gather_ms"))
except Exception as e:
    print("bench_rccl_n1 ??", e)
PY
if [ -n "$REF" ] && [ -d "$REF" ]; then
  timeout 900 python tools/cli_e2e.py "$REF" 2>&1 | tail -15 | tee gpurun_out/cli_e2e_stdout.log
fi
suite pytest_gpu_runB.log
head -16 gpurun_out/kernel_stats.csv | cut -c1-170
echo "== done"
