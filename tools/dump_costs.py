#!/usr/bin/env python3
"""Print the planner's calibrated cost table (LSTM and GRU handles) and the plans it yields; gpurun_out/planner_costs_*.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fullsubnet_plus_amd import FullSubNet_Plus  # noqa: E402
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for seq in ("LSTM", "GRU"):
    m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "sequence_model": seq})
    m.load_state_dict(make_state_dict(0, "default", sequence_model=seq), strict=True)
    m = m.to("cuda").eval()
    m.batch_mode = "full"
    ins = [t.cuda() for t in make_inputs(1, 0.5, 3)]
    t0 = time.perf_counter()
    m(*ins)
    torch.cuda.synchronize()
    c = m.planner_costs()
    c["measured"] = m.measure_costs()
    c["first_forward_s"] = time.perf_counter() - t0
    c["plans"] = {b: [f'{k["kernel"].split(" ")[0]} x{k["sequences"]}' for k in m.describe_plan(b)] for b in (1, 2, 3, 4, 5, 8, 12, 16, 21, 32, 40)}
    c["plan_parity_b32"] = [f'{k["kernel"].split(" ")[0]} x{k["sequences"]}' for k in m.describe_plan(32, parity=True)]
    print(seq, json.dumps(c, indent=1))
    with open(os.path.join(ROOT, "gpurun_out", f"planner_costs_{seq}.json"), "w") as f:
        json.dump(c, f, indent=1)
