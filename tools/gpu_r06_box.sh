#!/bin/bash
# round 6: one line of box-variance evidence per GPU box (profiles/r06_box_variance.md is built from these).
# Usage (from the build container): gpurun -- "bash tools/gpu_r06_box.sh <tag> [suite]"
#   - the headline exactly as the driver runs it (bench.py --gpus 1 --steps 20 --warmup 5) -> gpurun_out/r06_box_<tag>.json
#     (its `box` object: fp32-MFMA probe before / after, sclk / power sampled during the timed loops, rocm-smi after)
#   - rocm-smi's own view of the box, and five more 50 ms probes 1 s apart (is the probe itself stable on this box?)
#   - [suite] the whole GPU suite first
set -u
TAG=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${2:-}" = "suite" ]; then
  {
    echo "commit: ${FSNP_HEAD:-unknown}   library stamp: $(cut -c1-16 fullsubnet_plus_amd/libfsnp_hip.so.stamp)   $(date -u +%FT%TZ)"
    timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 2>&1 | tail -45
  } | tee gpurun_out/r06_pytest_gpu_$TAG.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r06_smoke_$TAG.log
fi
{
  echo "== host: $(hostname)  $(date -u +%FT%TZ)"
  (rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showtemp 2>&1 || true) | grep -v "^$" | head -60
  ls /sys/class/drm/ 2>&1 | head -20
  for d in /sys/class/drm/card*/device; do echo "$d vendor=$(cat $d/vendor 2>/dev/null) perf=$(cat $d/power_dpm_force_performance_level 2>/dev/null)"; ls $d/hwmon/hwmon*/ 2>/dev/null | tr '\n' ' '; echo; done
} > gpurun_out/r06_box_${TAG}_smi.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_box_${TAG}.err | tail -1 > gpurun_out/r06_box_${TAG}.json
python - "$TAG" <<'PY' 2>&1 | tee gpurun_out/r06_box_${TAG}_probes.txt
import json, sys, time
import torch
from fullsubnet_plus_amd import box
tag = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/r06_box_{tag}.json"))
    b = r["box"]
    print(f"{tag}: {r['value']:.0f} frames/s  ms/step {r['ms_per_step']:.3f} runs {['%.3f' % x for x in r['ms_per_step_runs']]}  "
          f"back-to-back {r['alt_ms_per_step']:.3f}  drop-in {r['dropin_ms_per_step']:.3f}")
    print(f"   dominant launch {r['roofline']['avg_launch_ms']:.3f} ms runs {['%.3f' % x for x in r['roofline']['avg_launch_ms_runs']]}  frac {r['roofline']['frac']:.4f}  "
          f"frac_of_box_peak {r['roofline']['frac_of_box_peak']:.4f}  frac_at_held_clock {r['roofline']['frac_at_held_clock']:.4f}  clock {r['roofline']['last_launch_clock']}")
    print(f"   box peak {b['mfma_peak_tflops']:.2f} TFLOP/s (before {b['probe_before']['mfma_tflops']:.2f} @ {b['probe_before']['clock_mhz']:.0f} MHz, "
          f"after {b['probe_after']['mfma_tflops']:.2f} @ {b['probe_after']['clock_mhz']:.0f} MHz)  sampled {b['during_timed_loops']}")
except Exception as e:
    print("bench line unreadable:", repr(e))
torch.zeros(1, device="cuda")
for i in range(5):
    p = box.probe(50.0)
    print(f"probe {i}: {p['mfma_tflops']:.2f} TFLOP/s  in-kernel {p['mfma_tflops_in_kernel']:.2f}  clock {p['clock_mhz']:.0f} MHz "
          f"[{p['clock_mhz_slowest_cu']:.0f}, {p['clock_mhz_fastest_cu']:.0f}]  s_memtime {p['s_memtime_mhz']:.1f} MHz, {p['s_memtime_ticks_per_mfma']:.2f} ticks/MFMA")
    time.sleep(1.0)
for ms in (5.0, 200.0, 1000.0):
    p = box.probe(ms)
    print(f"probe {ms:.0f} ms: {p['mfma_tflops']:.2f} TFLOP/s  clock {p['clock_mhz']:.0f} MHz  launch {p['launch_ms']:.1f} ms")
PY
