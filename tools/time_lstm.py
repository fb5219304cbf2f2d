#!/usr/bin/env python3
"""Time the fused recurrent kernel alone on dense inputs: python tools/time_lstm.py n steps [reps] -> ms per call, us per step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fullsubnet_plus_amd import FullSubNet_Plus  # noqa: E402
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_state_dict  # noqa: E402

n, steps = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
seq = os.environ.get("SEQ", "LSTM")
hidden = int(os.environ.get("HIDDEN", "384"))          # e.g. 320: no tuned instantiation -> the runtime-sized kernel
m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "sequence_model": seq, "sb_model_hidden_size": hidden})
m.load_state_dict(make_state_dict(0, "default", sequence_model=seq, sb_hidden=hidden), strict=True)
m = m.to("cuda").eval()
torch.manual_seed(1234 + n)
x = torch.randn(n, 34, steps, device="cuda")
out = m.lstm2_fc(x)
if os.environ.get("PIN_R01"):
    m.debug_set_costs(None, 1)
if os.environ.get("COOPW"):                     # only the wave-owned column split (csrc/lstm_coopw.hip) at COOPW = 32 / 64 units per workgroup
    u = int(os.environ["COOPW"])
    wv = [5.0 if 32 * (i + 1) == u else 900.0 for i in range(2)]
    m.debug_set_costs([900.0] * 12 + [900.0, 0.11] + [900.0] * 4 + [900.0] + [900.0, 900.0] + wv + wv + ([5.0, 5.0] if u == 96 else [900.0, 900.0]), 1)
    assert all(c["kernel"].startswith("lstm2_coopw_kernel") for c in m.describe_plan(1)), m.describe_plan(1)
if os.environ.get("NOCOOPW"):                   # the round-4 plans: no wave-owned column split
    m.debug_set_costs(list(m.planner_costs_raw())[:21], 1)
if os.environ.get("HP"):                        # only the half-tile ping-pong kernel (csrc/lstm_hp.hip)
    m.debug_set_lstm_coop(4)
    m.debug_set_costs([900.0] * 12 + [900.0, 0.11] + [900.0] * 4 + [900.0] + [5.0, 5.0], 1)
    assert all(c["kernel"].startswith(("lstm2_coop_hp_kernel", "lstm2_coop_hpw_kernel")) for c in m.describe_plan(1)), m.describe_plan(1)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.lstm2_fc(x)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
m.check_errors()
print(f"H={hidden} HP={os.environ.get('HP')} COOPW={os.environ.get('COOPW')} NOCOOPW={os.environ.get('NOCOOPW')} n={n} steps={steps} env XCD={os.environ.get('FSNP_COOP_XCD', '0')}: {best * 1e3:.3f} ms, {best * 1e6 / steps:.2f} us/step, checksum {float(out.double().sum()):.6f}")
