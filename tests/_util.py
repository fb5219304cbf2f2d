"""Shared helpers for the tests: golden-fixture loading and input regeneration."""
import json
import os

import numpy as np
import torch

from oracle.make_golden import make_spec
from oracle.weights import make_inputs, make_state_dict, make_state_dict_fullsubnet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(model="plus"):
    """Fixtures of FullSubNet+ ("plus") or of the original FullSubNet ("fullsubnet", files fsn_*)."""
    names = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))
    return [n for n in names if n.startswith("fsn_") == (model == "fullsubnet")]


class Golden:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.arrays = {k: z[k] for k in z.files if k != "meta"}
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.args = self.meta["args"]
        self.sub = self.meta.get("subsample_f", 1)

    @property
    def is_fullsubnet(self):
        return self.meta.get("model") == "fullsubnet"

    def state_dict(self):
        if self.is_fullsubnet:
            return make_state_dict_fullsubnet(self.meta["wseed"], self.meta["profile"],
                                              sequence_model=self.args.get("sequence_model", "LSTM"),
                                              fb_num_neighbors=self.args.get("fb_num_neighbors", 0),
                                              num_freqs=self.args.get("num_freqs", 257),
                                              sb_num_neighbors=self.args.get("sb_num_neighbors", 15),
                                              fb_hidden=self.args.get("fb_model_hidden_size", 512),
                                              sb_hidden=self.args.get("sb_model_hidden_size", 384))
        return make_state_dict(self.meta["wseed"], self.meta["profile"],
                               attention=self.args.get("channel_attention_model", "TSSE"),
                               sequence_model=self.args.get("sequence_model", "LSTM"),
                               fb_num_neighbors=self.args.get("fb_num_neighbors", 0),
                               num_freqs=self.args.get("num_freqs", 257),
                               sb_num_neighbors=self.args.get("sb_num_neighbors", 15),
                               kersize=tuple(self.args.get("kersize", (3, 5, 10))),
                               sb_hidden=self.args.get("sb_model_hidden_size", 384),
                               output_size=self.args.get("output_size", 2))

    def inputs(self):
        inp = self.meta["inp"]
        if "mag" in self.arrays:             # fsn_* stft fixtures keep the magnitude itself, [B,1,F,T]
            m = torch.from_numpy(self.arrays["mag"][:, 0].transpose(0, 2, 1).copy()).permute(0, 2, 1).unsqueeze(1)
            return m, None, None
        if "X" in self.arrays:
            X = torch.from_numpy(self.arrays["X"].transpose(0, 2, 1).copy()).permute(0, 2, 1)  # stft strides
            return X.abs().unsqueeze(1), X.real.unsqueeze(1), X.imag.unsqueeze(1)
        if inp["kind"] == "stft":
            return make_inputs(inp["B"], inp["t"], inp["seed"])
        return make_spec(inp["B"], inp["t"], inp["seed"], self.args.get("num_freqs", 257))

    def fwd_kwargs(self):
        a = self.args
        if self.is_fullsubnet:
            return {k: a[k] for k in ("look_ahead", "sb_num_neighbors", "fb_num_neighbors", "norm_type",
                                      "num_groups_in_drop_band", "fb_output_activate_function",
                                      "sb_output_activate_function")}
        return dict(look_ahead=a["look_ahead"], sb_num_neighbors=a["sb_num_neighbors"],
                    fb_num_neighbors=a["fb_num_neighbors"], norm_type=a["norm_type"],
                    num_groups_in_drop_band=a["num_groups_in_drop_band"],
                    channel_attention_model=a.get("channel_attention_model", "TSSE"),
                    subband_num=a.get("subband_num", 1),
                    fb_output_activate_function=a.get("fb_output_activate_function", "ReLU"),
                    sb_output_activate_function=a.get("sb_output_activate_function", False),
                    output_size=a.get("output_size", 2))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
