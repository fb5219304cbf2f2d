import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from fullsubnet_plus_amd.model import FullSubNet_Plus
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
sd = make_state_dict(0, "default")
m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(sd, strict=True); m = m.to("cuda").eval(); m.batch_mode = "full"
mag, real, imag = make_inputs(32, 2.0, 100)
ins = [t.cuda() for t in (mag, real, imag)]
ref = m(*ins).cpu().numpy()
for rep in range(3):
    m.set_precision("bf16x3")
    got = m(*ins).cpu().numpy()
    m.set_precision("fp32")
    again = m(*ins).cpu().numpy()
    d = np.abs(got - ref)
    per_utt = d.reshape(32, -1).max(1) / np.abs(ref).max()
    per_bin = d.transpose(2, 0, 1, 3).reshape(257, -1).max(1) / np.abs(ref).max()
    per_t = d.transpose(3, 0, 1, 2).reshape(d.shape[3], -1).max(1) / np.abs(ref).max()
    print("rep", rep, "max", d.max() / np.abs(ref).max(), "fp32 repeat equal", np.array_equal(again, ref))
    print(" per utt", np.array2string(per_utt, precision=1, max_line_width=250))
    print(" worst bins", np.argsort(per_bin)[-8:], np.sort(per_bin)[-8:])
    print(" per t (first 12)", np.array2string(per_t[:12], precision=1), "last", np.array2string(per_t[-4:], precision=1))
