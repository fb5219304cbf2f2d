#!/usr/bin/env python3
"""Phase breakdown of the fused LSTM kernel from s_memtime stamps (fsnp_debug_lstm_profile)."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullsubnet_plus_amd import FullSubNet_Plus, _lib
from oracle.ref_loader import DEFAULT_MODEL_ARGS
from oracle.weights import make_state_dict

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8224
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(make_state_dict(0)); m = m.cuda().eval()
    x = torch.randn(n, steps, 34, device="cuda")
    out = torch.empty(n, 2, steps, device="cuda")
    m.lstm2_fc(x.permute(0, 2, 1)[:64])          # creates the handle
    if len(sys.argv) > 3:
        m.debug_set_lstm_waves(int(sys.argv[3]))
    lib = _lib.load()
    stamps = np.zeros(steps * 8, dtype=np.uint64)
    for rep in range(2):
        _lib.check(lib.fsnp_debug_lstm_profile(m._handle, x.data_ptr(), out.data_ptr(), n, steps,
                                               stamps.ctypes.data, stamps.size), "profile")
    s = stamps.reshape(steps, 8).astype(np.int64)
    names = ["L0 mfma", "barrier", "cell0+x+fc", "barrier", "L1 mfma", "barrier", "cell1"]
    d = np.diff(s, axis=1)[4:]                     # skip warm-up steps
    step = (s[5:, 0] - s[4:-1, 0])
    res = {"rows": n, "steps": steps, "waves": int(sys.argv[3]) if len(sys.argv) > 3 else 12, "ticks_per_step": float(step.mean())}
    for i, nm in enumerate(names):
        res[f"{i}:{nm}"] = float(d[:, i].mean())
    res["gap_to_next_step"] = float((s[5:, 0] - s[4:-1, 7]).mean())
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
