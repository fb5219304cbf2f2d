#!/usr/bin/env python3
"""End-to-end evidence for "drops into the existing pipeline" (VERDICT r01 #4): run the UNMODIFIED reference CLI

    python -m speech_enhance.tools.inference -C <toml> -M rand_ckpt.tar -I wavdir -O out

three times on the GPU box - (1) with config/inference_hip.toml (the reference's TOML with [model].path pointing at the HIP
model) on the MI355X, (2) with the reference's own config/inference.toml on the host CPU, (3) the same on the MI355X through
PyTorch-ROCm / MIOpen - and compare the enhanced wavs it wrote.  Also times the real reference FullSubNet_Plus forward on
the box's host cores (the number bench.py's `cpu_baseline` approximates with the oracle port).

The reference tree is NOT part of this repo: for the one gpurun call that runs this script it is staged (by hand, from
/root/reference) under _refstage/ (git-ignored, deleted afterwards).  Usage: python tools/cli_e2e.py _refstage/reference
Writes gpurun_out/cli_e2e.json and gpurun_out/cli_e2e_*.log.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
WORK = "/tmp/cli_e2e"


def run_cli(ref, toml, outdir, env_extra, log_name, pythonpath):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(pythonpath)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "speech_enhance.tools.inference", "-C", toml, "-M", os.path.join(WORK, "rand_ckpt.tar"),
           "-I", os.path.join(WORK, "wavs"), "-O", outdir]
    t0 = time.perf_counter()
    res = subprocess.run(cmd, cwd=ref, env=env, capture_output=True, text=True, timeout=1500)
    dt = time.perf_counter() - t0
    with open(os.path.join(OUT, log_name), "w") as f:
        f.write("$ " + " ".join(cmd) + "\n" + res.stdout[-20000:] + "\n--- stderr ---\n" + res.stderr[-8000:])
    rtf = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\S+), rtf: ([0-9.eE+-]+)", res.stdout + res.stderr)}
    return {"rc": res.returncode, "wall_s": dt, "rtf": rtf, "cmd": " ".join(cmd[1:])}


def read_wavs(outdir):
    from scipy.io import wavfile
    out = {}
    for root, _d, files in os.walk(outdir):
        for f in files:
            if f.endswith(".wav"):
                out[f] = wavfile.read(os.path.join(root, f))[1].astype(np.float64)
    return out


def main():
    ref = os.path.abspath(sys.argv[1])
    assert os.path.exists(os.path.join(ref, "speech_enhance", "tools", "inference.py")), ref
    import torch
    from scipy.io import wavfile
    from fullsubnet_plus_amd.synthetic import make_state_dict, make_wave
    os.makedirs(OUT, exist_ok=True)
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(os.path.join(WORK, "wavs"))
    secs = [2.0, 3.1, 5.0]
    for i, s in enumerate(secs):                                   # synthetic 16 kHz clips (noise, like bench.py's inputs)
        w = make_wave(1, s, 4000 + i)[0]
        wavfile.write(os.path.join(WORK, "wavs", f"clip{i}.wav"), 16000, np.int16(np.clip(w, -1, 1) * 32767))
    torch.save({"model": make_state_dict(0, "default"), "epoch": 0}, os.path.join(WORK, "rand_ckpt.tar"))

    shims = os.path.join(ROOT, "shims")
    runs = {}
    runs["hip_mi355x"] = run_cli(ref, os.path.join(ROOT, "config", "inference_hip.toml"), os.path.join(WORK, "out_hip"), {},
                                 "cli_e2e_hip.log", [ROOT, shims])
    runs["reference_cpu"] = run_cli(ref, os.path.join(ref, "config", "inference.toml"), os.path.join(WORK, "out_ref_cpu"),
                                    {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "OMP_NUM_THREADS": "16"},
                                    "cli_e2e_ref_cpu.log", [shims])
    runs["reference_rocm_miopen"] = run_cli(ref, os.path.join(ref, "config", "inference.toml"), os.path.join(WORK, "out_ref_gpu"),
                                            {}, "cli_e2e_ref_gpu.log", [shims])
    wavs = {k: read_wavs(os.path.join(WORK, d)) for k, d in
            (("hip_mi355x", "out_hip"), ("reference_cpu", "out_ref_cpu"), ("reference_rocm_miopen", "out_ref_gpu"))}
    cmp = {}
    for other in ("reference_cpu", "reference_rocm_miopen"):
        for name, a in wavs["hip_mi355x"].items():
            b = wavs[other].get(name)
            if b is None or b.shape != a.shape:
                cmp[f"{other}/{name}"] = "missing"
                continue
            peak = np.abs(b).max()
            cmp[f"{other}/{name}"] = {"max_abs_diff": float(np.abs(a - b).max()), "peak": float(peak),
                                      "rel_to_peak": float(np.abs(a - b).max() / peak), "samples": int(a.size)}

    # the real reference module on the host cores: B = 1 x 2 s clips (== "full" semantics), thread sweep
    sys.path[:0] = [os.path.join(ref, "speech_enhance"), ref, shims]
    from fullsubnet_plus.model.fullsubnet_plus import FullSubNet_Plus as RefModel
    from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs
    model = RefModel(**DEFAULT_MODEL_ARGS)
    model.load_state_dict(make_state_dict(0, "default"), strict=True)
    model.eval()
    mag, real, imag = make_inputs(4, 2.0, 1000)
    T = mag.shape[-1]
    sweep = {}
    with torch.no_grad():
        for th in (8, 16, 32, 64):
            torch.set_num_threads(th)
            model(mag[:1], real[:1], imag[:1])
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 6.0:
                model(mag[n % 4:n % 4 + 1], real[n % 4:n % 4 + 1], imag[n % 4:n % 4 + 1])
                n += 1
            sweep[th] = n * T / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    result = {"runs": runs, "wav_compare": cmp, "clips_s": secs,
              "reference_cpu_forward": {"frames_per_s_by_threads": sweep, "best_threads": best, "value": sweep[best],
                                        "host_threads": os.cpu_count(), "torch": torch.__version__,
                                        "what": "fullsubnet_plus.model.fullsubnet_plus.FullSubNet_Plus (the unmodified "
                                                "reference class), B=1 x 2 s clips, 6 s per thread count"}}
    with open(os.path.join(OUT, "cli_e2e.json"), "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps({"rc": {k: v["rc"] for k, v in runs.items()}, "cmp": cmp, "ref_cpu": result["reference_cpu_forward"]["value"],
                      "threads": best}, indent=1))


if __name__ == "__main__":
    main()
