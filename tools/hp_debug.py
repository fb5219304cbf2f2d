#!/usr/bin/env python3
"""one-off debug of csrc/lstm_hp.hip: which rows / steps differ from the oracle"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fullsubnet_plus_amd import FullSubNet_Plus  # noqa: E402
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_state_dict  # noqa: E402
from oracle import fsnp_torch  # noqa: E402  (debug script, not product)

n, steps = int(sys.argv[1]), int(sys.argv[2])
sd = make_state_dict(3, "harsh")
m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(sd); m = m.cuda().eval()
rng = np.random.Generator(np.random.PCG64(977 + n + steps))
x = torch.from_numpy(rng.standard_normal((n, 34, steps)).astype(np.float32)).cuda()
m.lstm2_fc(x[:1])
m.debug_set_lstm_coop(4)
m.debug_set_costs([900.0] * 12 + [900.0, 0.11] + [900.0] * 4 + [900.0, 0.0] + [900.0] * 4 + [5.0, 5.0], 1)
got = m.lstm2_fc(x).cpu().numpy()
try:
    m.check_errors()
except Exception as e:
    print("check_errors:", e)
want = fsnp_torch.lstm2_fc(x.cpu(), sd).numpy()
err = np.abs(got - want).max(axis=1)          # [n, steps]
bad = np.argwhere(err > 1e-4 * np.abs(want).max())
print(f"n={n} steps={steps} XCD={os.environ.get('FSNP_COOP_XCD')} rel_err={np.abs(got-want).max()/np.abs(want).max():.3e} bad rows={sorted(set(bad[:,0].tolist()))[:20]} bad steps={sorted(set(bad[:,1].tolist()))[:20]}")
if len(bad):
    r = bad[0, 0]
    print(" got ", got[r, :, :8]); print(" want", want[r, :, :8])
