"""What the box a measurement ran on can sustain - so that a slow box can be told from a slow kernel.

bench.py's line carries a `box` object built from this module:
  * `probe()`            fsnp_debug_box_probe (csrc/box_probe.hip): pure v_mfma_f32_32x32x2_f32 issue on every SIMD for ~50 ms - the
                         fp32 MFMA rate THIS box holds (data sheet: 157.3 TFLOP/s at 2.4 GHz) and the shader clock that implies;
  * `Sampler`            a thread that reads the GPU's sclk / socket power from sysfs (hwmon) while the timed loops run;
  * `smi_snapshot()`     rocm-smi's view (clocks, power, power cap, performance level), once, outside the timed region.
Everything except the probe is best effort: a container may hide sysfs or rocm-smi, and the line then says so instead of failing.
The reference has no counterpart (it is measured on whatever CPU it runs on)."""
import ctypes
import glob
import json
import os
import shutil
import subprocess
import threading
import time

from . import _lib

PEAK_FP32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)": 256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz


def probe(target_ms=50.0, stream=None):
    """-> dict: the fp32 MFMA rate and implied shader clock of the current device (see include/fsnp_debug.h)."""
    out = (ctypes.c_double * _lib.BOX_PROBE_VALUES)()
    _lib.check(_lib.load().fsnp_debug_box_probe(float(target_ms), ctypes.byref(out), ctypes.c_void_p(stream)), "fsnp_debug_box_probe")
    return {"mfma_tflops": out[0], "mfma_tflops_in_kernel": out[8], "clock_mhz": out[1], "clock_mhz_slowest_cu": out[2],
            "clock_mhz_fastest_cu": out[3], "s_memtime_mhz": out[4], "s_memtime_ticks_per_mfma": out[5], "launch_ms": out[6],
            "compute_units": int(out[7]), "frac_of_spec_peak": out[0] / PEAK_FP32_MFMA_TFLOPS}


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _pci_slot(dev_dir):
    ue = _read(os.path.join(dev_dir, "uevent")) or ""
    for line in ue.splitlines():
        if line.startswith("PCI_SLOT_NAME="):
            return line.split("=", 1)[1].lower()
    return None


def sysfs_device(pci_bus_id=None):
    """The /sys/class/drm/cardN/device directory of the AMD GPU with that PCI address ("0000:05:00.0"; None = the only / first one)."""
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if (_read(os.path.join(d, "vendor")) or "").lower() != "0x1002":
            continue
        cands.append(d)
        if pci_bus_id and _pci_slot(d) == pci_bus_id.lower():
            return d
    return cands[0] if cands and (pci_bus_id is None or len(cands) == 1) else None


def read_sysfs(dev_dir):
    """-> {"sclk_mhz", "power_w", "power_cap_w", "perf_level"} (entries the kernel does not expose are missing)."""
    out = {}
    if not dev_dir:
        return out
    for hw in glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*")):
        v = _read(os.path.join(hw, "freq1_input"))
        if v and v.isdigit():
            out["sclk_mhz"] = int(v) / 1e6
        for name in ("power1_average", "power1_input"):
            v = _read(os.path.join(hw, name))
            if v and v.isdigit():
                out["power_w"] = int(v) / 1e6
                break
        v = _read(os.path.join(hw, "power1_cap"))
        if v and v.isdigit():
            out["power_cap_w"] = int(v) / 1e6
    if "sclk_mhz" not in out:
        for line in (_read(os.path.join(dev_dir, "pp_dpm_sclk")) or "").splitlines():
            if line.rstrip().endswith("*"):
                try:
                    out["sclk_mhz"] = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except (IndexError, ValueError):
                    pass
    v = _read(os.path.join(dev_dir, "power_dpm_force_performance_level"))
    if v:
        out["perf_level"] = v
    return out


def gpu_unique_id(dev_dir):
    """The GPU's serial (sysfs unique_id): tells the boxes of a pool apart where every container's hostname is the same."""
    return _read(os.path.join(dev_dir, "unique_id")) if dev_dir else None


class Sampler(threading.Thread):
    """Samples sclk / socket power from sysfs every `period` seconds between start() and stop()."""

    def __init__(self, dev_dir, period=0.01):
        super().__init__(daemon=True)
        self.dev_dir, self.period = dev_dir, period
        self.samples = []
        self._stop_ev = threading.Event()

    def run(self):
        while not self._stop_ev.is_set():
            s = read_sysfs(self.dev_dir)
            if s:
                self.samples.append(s)
            self._stop_ev.wait(self.period)

    def stop(self):
        self._stop_ev.set()
        self.join(timeout=2.0)
        return self.summary()

    def summary(self):
        out = {"samples": len(self.samples)}
        for key in ("sclk_mhz", "power_w"):
            vals = sorted(s[key] for s in self.samples if key in s)
            if vals:
                out[key] = {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}
        for key in ("power_cap_w", "perf_level"):
            for s in self.samples:
                if key in s:
                    out[key] = s[key]
                    break
        return out


def smi_snapshot(timeout_s=15.0):
    """rocm-smi's clocks / power / power cap / performance level as a dict (or {"error": ...}); run OUTSIDE timed regions."""
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    try:
        res = subprocess.run([exe, "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=timeout_s)
        txt = res.stdout.strip()
        start = txt.find("{")
        data = json.loads(txt[start:]) if start >= 0 else {}
        keep = {}
        for card, vals in data.items():
            if not isinstance(vals, dict):
                continue
            keep[card] = {k: v for k, v in vals.items()
                          if any(t in k.lower() for t in ("sclk", "mclk", "power", "performance level", "fclk"))}
        return keep or {"error": (res.stderr or txt)[-300:]}
    except Exception as e:       # noqa: BLE001 - best effort by design
        return {"error": repr(e)[:300]}
