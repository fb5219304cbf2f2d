#!/bin/bash
# round 6, last check at the round's HEAD: the whole GPU suite twice, smoke, the headline as the driver runs it (-> one more row of
# profiles/r06_box_variance.md) and the small-batch configurations on THIS box (the bars of the round-5 review are "on two evidence boxes").
# Usage (from the build container): gpurun -- "FSNP_HEAD=<commit> bash tools/gpu_r06_head.sh <tag>"
set -u
TAG=${1:-head}
mkdir -p gpurun_out
export TMPDIR=/tmp
digest() { cat fullsubnet_plus_amd/csrc/*.hip fullsubnet_plus_amd/csrc/*.h fullsubnet_plus_amd/csrc/*.cpp include/fsnp.h include/fsnp_debug.h | sha256sum | cut -c1-16; }
suite() {
  {
    echo "commit: ${FSNP_HEAD:-unknown}   csrc sha256[:16] at start: $(digest)   library stamp: $(cut -c1-16 fullsubnet_plus_amd/libfsnp_hip.so.stamp)   $(date -u +%FT%TZ)"
    timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 2>&1 | tail -30
    echo "csrc sha256[:16] at end: $(digest)"
  } | tee gpurun_out/$1
}
suite r06_pytest_gpu_${TAG}_runA.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r06_smoke_${TAG}.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_box_${TAG}.err | tail -1 > gpurun_out/r06_box_${TAG}.json
: > gpurun_out/r06_small_batch_${TAG}.log
for B in 1 2 3 4 5 8 12; do
  timeout 400 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --probe-ms 0 2>&1 | tail -1 >> gpurun_out/r06_small_batch_${TAG}.log
done
python - "$TAG" <<'PY' | tee gpurun_out/r06_small_batch_${TAG}.txt
import json, sys
tag = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/r06_box_{tag}.json"))
    print(f"box {tag}: GPU serial {(r['box'].get('gpu_unique_id') or '-')[:8]}  headline {r['ms_per_step']:.3f} ms/step (back to back {r['alt_ms_per_step']:.3f}, drop-in {r['dropin_ms_per_step']:.3f})  "
          f"dominant launch {r['roofline']['avg_launch_ms']:.3f} ms at {r['roofline']['last_launch_clock']['s_memtime_mhz']:.0f} MHz  frac {r['roofline']['frac']:.4f}  at held clock {r['roofline']['frac_at_held_clock']:.4f}")
except Exception as e:
    print("headline unreadable:", repr(e))
print("| B | serving loop (ms) | back to back | drop-in | full-band stage | plan |")
print("|---|---|---|---|---|---|")
for l in open(f"gpurun_out/r06_small_batch_{tag}.log"):
    try:
        r = json.loads(l)
    except Exception:
        print("??", l[:120]); continue
    plan = " + ".join("%s x%d%s" % (c["kernel"].split(" ")[0].replace("lstm2_", ""), c["sequences"], "*" if c.get("deferred_when_pipelined") else "") for c in r["roofline"]["subband_plan"])
    print(f"| {r['config']['global_batch']} | {r['ms_per_step']:.3f} | {r['alt_ms_per_step']:.3f} | {r['dropin_ms_per_step']:.3f} | {r['roofline']['alt_fullband_ms'] or r['roofline']['fullband_ms']:.3f} | {plan} |")
PY
suite r06_pytest_gpu_${TAG}_runB.log
echo "== done"
