import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fullsubnet_plus_amd import FullSubNet_Plus
from oracle.ref_loader import DEFAULT_MODEL_ARGS
from oracle.weights import make_inputs, make_state_dict
def cuda(ts):
    out = []
    for t in ts:
        g = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device="cuda"); g.copy_(t); out.append(g)
    return out
sd = make_state_dict(0, "default")
m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS); m.load_state_dict(sd); m = m.cuda().eval(); m.batch_mode = "full"
cpu_in = make_inputs(32, 2.0, 100)
x32 = cuda(cpu_in)
full = m(*x32).cpu().numpy()
def d(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for b in (0, 13):
    x1 = cuda([t[b:b + 1] for t in cpu_in])
    outs = [m(*x1).cpu().numpy() for _ in range(3)]
    m.debug_set_graph(0)
    plain = m(*x1).cpu().numpy()
    m.debug_set_graph(1)
    again = m(*x1).cpu().numpy()
    print("b", b, "vs full row:", [d(o, full[b:b + 1]) for o in outs], "plain", d(plain, full[b:b + 1]), "again", d(again, full[b:b + 1]))
    for nm in ("att_mag", "fb_mag", "fb_imag"):
        pass
# stage check: graph vs plain on B=1
x1 = cuda([t[5:6] for t in cpu_in])
m.debug_set_graph(0); p = m(*x1).cpu().numpy(); sp = {k: m.read_stage(k, 1, 126).numpy() for k in ("att_mag", "att_imag", "fb_mag", "fb_real", "fb_imag")}
m.debug_set_graph(1); g = m(*x1).cpu().numpy(); sg = {k: m.read_stage(k, 1, 126).numpy() for k in sp}
print("out graph vs plain", d(g, p), {k: d(sg[k], sp[k]) for k in sp})
