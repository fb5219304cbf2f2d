#!/usr/bin/env python3
"""How far does the REFERENCE disagree with itself under norm_type="cumulative_laplace_norm" on 10 s clips of real STFT data?

VERDICT r02 "weak #1": on `b2_10s_cumulative_laplace_norm` (tests/test_gpu_parity.py::test_long_clip_cumulative_norms_vs_oracle)
the HIP forward is 1.5e-2 rel from the reference's fp32 CPU forward and 4e-5 from its fp64 forward.  This script runs the
reference CLASS ITSELF (fullsubnet_plus/model/fullsubnet_plus.py:16, audio_zen/model/base_model.py:227-258) on those very
inputs five ways - torch CPU with 1 / 8 / 16 threads, fp64 on the CPU, and on the MI355X through PyTorch-ROCm (fp32 and
fp64) - next to the HIP forward, and tabulates every pairwise rel error (max |a - b| / max |fp64|).

The reference tree is NOT part of this repo: it is staged under _refstage/ (git-ignored) for the one gpurun call that runs
this, exactly like tools/cli_e2e.py.  Usage: python tools/cum_laplace_spread.py _refstage/reference [--no-gpu]
Writes gpurun_out/cum_laplace.json and gpurun_out/cum_laplace.md.
"""
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def main():
    ref = os.path.abspath(sys.argv[1])
    use_gpu = "--no-gpu" not in sys.argv and torch.cuda.is_available()
    os.environ["FSNP_REFERENCE_ROOT"] = ref
    from oracle.ref_loader import load_reference
    from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
    Ref = load_reference()
    results = {}
    notes = {}
    for norm in ("cumulative_laplace_norm", "cumulative_layer_norm"):
        args = {**DEFAULT_MODEL_ARGS, "norm_type": norm}
        sd = make_state_dict(11, "default")
        mag, real, imag = make_inputs(2, 10.0, 200)              # the inputs of test_long_clip_cumulative_norms_vs_oracle
        model = Ref(**args)
        model.load_state_dict(sd, strict=True)
        model.eval()
        outs = {}

        def run(m, ins, dev=None):
            # B = 1 calls: the reference applies drop_band whenever batch_size > 1 (fullsubnet_plus.py:192-196); one utterance per
            # call keeps all 257 bins = the HIP path's "full" mode
            res = []
            with torch.no_grad():
                for b in range(ins[0].shape[0]):
                    one = [t[b:b + 1] for t in ins]
                    if dev is not None:
                        one = [t.to(dev) for t in one]
                    res.append(m(*one).cpu().double().numpy())
            return np.concatenate(res, 0)

        for th in (1, 8, 16):
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            outs[f"ref fp32 CPU {th} thr"] = run(model, (mag, real, imag))
            notes[f"{norm}/ref fp32 CPU {th} thr"] = f"{time.perf_counter() - t0:.1f} s"
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        m64 = Ref(**args).double()
        m64.load_state_dict({k: v.double() for k, v in sd.items()}, strict=True)
        m64.eval()
        outs["ref fp64 CPU"] = run(m64, [t.double() for t in (mag, real, imag)])
        if use_gpu:
            dev = torch.device("cuda", 0)
            try:
                mg = Ref(**args)
                mg.load_state_dict(sd, strict=True)
                mg = mg.to(dev).eval()
                outs["ref fp32 MI355X (PyTorch-ROCm)"] = run(mg, (mag, real, imag), dev)
            except Exception as e:  # noqa: BLE001
                notes[f"{norm}/ref fp32 MI355X"] = f"failed: {e!r}"[:300]
            try:
                mg64 = m64.to(dev)
                outs["ref fp64 MI355X (PyTorch-ROCm)"] = run(mg64, [t.double() for t in (mag, real, imag)], dev)
            except Exception as e:  # noqa: BLE001
                notes[f"{norm}/ref fp64 MI355X"] = f"failed: {e!r}"[:300]
            from fullsubnet_plus_amd import FullSubNet_Plus
            hip = FullSubNet_Plus(**args)
            hip.load_state_dict(sd, strict=True)
            hip = hip.to(dev).eval()
            hip.batch_mode = "full"
            gin = []
            for t in (mag, real, imag):
                g = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
                g.copy_(t)
                gin.append(g)
            with torch.no_grad():
                outs["HIP fp32 (this repo)"] = hip(*gin).cpu().double().numpy()
        scale = float(np.abs(outs["ref fp64 CPU"]).max())
        pair = {}
        for a, b in itertools.combinations(outs, 2):
            pair[f"{a} | {b}"] = float(np.abs(outs[a] - outs[b]).max() / scale)
        results[norm] = {"names": list(outs), "pairwise_rel": pair, "scale_max_abs_fp64": scale}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "cum_laplace.json"), "w") as f:
        json.dump({"results": results, "notes": notes, "torch": torch.__version__, "host_threads": os.cpu_count(),
                   "inputs": "make_inputs(2, 10.0, 200), make_state_dict(11, 'default'), one utterance per call"}, f, indent=1)
    lines = []
    for norm, r in results.items():
        names = r["names"]
        lines += [f"### {norm}: rel = max|a - b| / max|ref fp64| ({r['scale_max_abs_fp64']:.4g})", "",
                  "| | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
        for a in names:
            row = []
            for b in names:
                if a == b:
                    row.append("-")
                else:
                    v = r["pairwise_rel"].get(f"{a} | {b}", r["pairwise_rel"].get(f"{b} | {a}"))
                    row.append(f"{v:.2e}")
            lines.append(f"| {a} | " + " | ".join(row) + " |")
        lines.append("")
    with open(os.path.join(OUT, "cum_laplace.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print(json.dumps(notes, indent=1))


if __name__ == "__main__":
    main()
