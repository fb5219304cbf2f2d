"""TEST INFRASTRUCTURE ONLY - generate tests/golden/*.npz from the REAL reference.

Run in the build container (needs /root/reference):

    python -m oracle.make_golden            # writes tests/golden/*.npz

Each fixture holds, for one (weights seed/profile, model args, input spec):
  out     - the reference's fp32 forward output (FullSubNet_Plus.forward,
            fullsubnet_plus/model/fullsubnet_plus.py:122-209), literal batched call
            (so drop_band is active when B > 1);
  out64   - the same module deep-copied to float64 (accuracy yard-stick), stored fp32;
  full    - (B > 1 only) the reference called per utterance at B=1 and stacked
            == "full" mode;
  stages  - per-stage intermediates captured with forward hooks (small cases);
  X       - the complex input spectrogram when it came from torch.stft.
Inputs that come from ``make_spec`` are regenerated from the seed by the tests.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

from . import ref_loader
from .weights import make_inputs, make_state_dict, make_state_dict_fullsubnet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_spec(batch, frames, seed, num_freqs=257):
    """Synthetic complex spectrogram with the memory layout torch.stft produces
    ([B][T][F] order, real/imag interleaved): returns (mag, real, imag) [B,1,F,T] where
    real/imag are strided views of one complex64 buffer (inferencer.py:142-147)."""
    rng = np.random.Generator(np.random.PCG64(20_000 + seed))
    env = (0.05 + rng.random(size=(batch, frames, 1))) * (0.2 + rng.random(size=(batch, 1, num_freqs)))
    re = (env * rng.standard_normal(size=(batch, frames, num_freqs))).astype(np.float32)
    im = (env * rng.standard_normal(size=(batch, frames, num_freqs))).astype(np.float32)
    X = torch.complex(torch.from_numpy(re), torch.from_numpy(im)).permute(0, 2, 1)  # [B,F,T], strides (T*F,1,F)
    return X.abs().unsqueeze(1), X.real.unsqueeze(1), X.imag.unsqueeze(1)


CASES = [
    # name, weights(seed, profile), args overrides, input(kind, B, T-or-seconds, seed), store stages?
    dict(name="b1_2s_default", wseed=0, profile="default", args={}, inp=("stft", 1, 2.0, 0), stages=False),
    dict(name="b1_t24_default_stages", wseed=1, profile="default", args={}, inp=("spec", 1, 24, 1), stages=True),
    dict(name="b1_t24_harsh_stages", wseed=2, profile="harsh", args={}, inp=("spec", 1, 24, 2), stages=True),
    dict(name="b1_t8_min", wseed=3, profile="default", args={}, inp=("spec", 1, 8, 3), stages=False),
    dict(name="b3_t20_harsh", wseed=4, profile="harsh", args={}, inp=("spec", 3, 20, 4), stages=False),
    dict(name="b4_t16_default", wseed=5, profile="default", args={}, inp=("spec", 4, 16, 5), stages=False),
    dict(name="b5_t16_default", wseed=6, profile="default", args={}, inp=("spec", 5, 16, 6), stages=False),
    dict(name="b1_t30_cum_laplace", wseed=7, profile="default", args={"norm_type": "cumulative_laplace_norm"},
         inp=("spec", 1, 30, 7), stages=True),
    dict(name="b1_t30_gaussian", wseed=8, profile="default", args={"norm_type": "offline_gaussian_norm"},
         inp=("spec", 1, 30, 8), stages=True),
    dict(name="b1_t30_cum_layer", wseed=9, profile="default", args={"norm_type": "cumulative_layer_norm"},
         inp=("spec", 1, 30, 9), stages=True),
    dict(name="b3_t18_cum_layer", wseed=10, profile="harsh", args={"norm_type": "cumulative_layer_norm"},
         inp=("spec", 3, 18, 10), stages=False),
    dict(name="b1_t20_att_SE", wseed=12, profile="default", args={"channel_attention_model": "SE"},
         inp=("spec", 1, 20, 12), stages=False),
    dict(name="b1_t20_att_ECA", wseed=13, profile="harsh", args={"channel_attention_model": "ECA"},
         inp=("spec", 1, 20, 13), stages=False),
    dict(name="b3_t20_att_CBAM", wseed=14, profile="default", args={"channel_attention_model": "CBAM"},
         inp=("spec", 3, 20, 14), stages=False),
    # sub-band GRU (sequence_model.py:39-46)
    dict(name="gru_b1_t24_harsh_stages", wseed=15, profile="harsh", args={"sequence_model": "GRU"},
         inp=("spec", 1, 24, 15), stages=True),
    dict(name="gru_b3_t20_default", wseed=16, profile="default", args={"sequence_model": "GRU"},
         inp=("spec", 3, 20, 16), stages=False),
    dict(name="gru_b1_t30_cum_layer", wseed=17, profile="default",
         args={"sequence_model": "GRU", "norm_type": "cumulative_layer_norm"}, inp=("spec", 1, 30, 17), stages=False),
    # sub-band TCN (sequence_model.py:47-58)
    dict(name="tcn_b1_t24_default", wseed=18, profile="default", args={"sequence_model": "TCN"},
         inp=("spec", 1, 24, 18), stages=False),
    dict(name="tcn_b3_t20_harsh", wseed=19, profile="harsh", args={"sequence_model": "TCN"},
         inp=("spec", 3, 20, 19), stages=False),
    dict(name="b5_t16_groups3", wseed=20, profile="harsh", args={"num_groups_in_drop_band": 3},
         inp=("spec", 5, 16, 20), stages=False),
    # fb_num_neighbors > 0: the full-band outputs are unfolded too (fullsubnet_plus.py:170-184); 31 + 3 x 3 = 40 features
    dict(name="b3_t16_fbn1", wseed=21, profile="harsh", args={"fb_num_neighbors": 1}, inp=("spec", 3, 16, 21), stages=False),
    dict(name="b1_t30_fbn1_cum_layer", wseed=22, profile="default", args={"fb_num_neighbors": 1, "norm_type": "cumulative_layer_norm"},
         inp=("spec", 1, 30, 22), stages=False),
    # constructor options the default config never moves: look_ahead, sb_num_neighbors, kersize, activations, num_freqs
    dict(name="b1_t20_la0", wseed=23, profile="harsh", args={"look_ahead": 0}, inp=("spec", 1, 20, 23), stages=False),
    dict(name="b3_t18_la4", wseed=24, profile="default", args={"look_ahead": 4}, inp=("spec", 3, 18, 24), stages=False),
    dict(name="b1_t20_nb10_k247", wseed=25, profile="harsh", args={"sb_num_neighbors": 10, "kersize": [2, 4, 7]},
         inp=("spec", 1, 20, 25), stages=False),
    dict(name="b1_t20_tanh_relu6", wseed=26, profile="default",
         args={"fb_output_activate_function": "Tanh", "sb_output_activate_function": "ReLU6"}, inp=("spec", 1, 20, 26), stages=False),
    dict(name="b3_t16_f161", wseed=27, profile="harsh", args={"num_freqs": 161}, inp=("spec", 3, 16, 27), stages=False),
    # subband_num > 1 (fullsubnet_plus.py:146-153): only ECA survives it in the reference
    dict(name="b1_t20_eca_sub2", wseed=28, profile="harsh", args={"channel_attention_model": "ECA", "subband_num": 2},
         inp=("spec", 1, 20, 28), stages=False),
    dict(name="b3_t18_eca_sub3", wseed=29, profile="default", args={"channel_attention_model": "ECA", "subband_num": 3},
         inp=("spec", 3, 18, 29), stages=False),
    dict(name="b1_t16_eca_sub7_f161", wseed=30, profile="harsh",
         args={"channel_attention_model": "ECA", "subband_num": 7, "num_freqs": 161}, inp=("spec", 1, 16, 30), stages=False),
    # sub-band inputs wider than 40 features (fb_num_neighbors >= 2): the K = 64 instantiations of the recurrent kernels
    dict(name="b3_t16_fbn2", wseed=31, profile="harsh", args={"fb_num_neighbors": 2}, inp=("spec", 3, 16, 31), stages=False),
    dict(name="gru_b1_t20_fbn3", wseed=32, profile="default", args={"fb_num_neighbors": 3, "sequence_model": "GRU"},
         inp=("spec", 1, 20, 32), stages=False),
    # sb_model_hidden_size off its default (the H = 256 / 512 instantiations of the recurrent kernels)
    dict(name="b3_t16_h256", wseed=33, profile="harsh", args={"sb_model_hidden_size": 256}, inp=("spec", 3, 16, 34), stages=False),
    dict(name="b1_t20_h512", wseed=34, profile="default", args={"sb_model_hidden_size": 512}, inp=("spec", 1, 20, 35), stages=False),
    dict(name="gru_b3_t16_h256", wseed=35, profile="default", args={"sb_model_hidden_size": 256, "sequence_model": "GRU"},
         inp=("spec", 3, 16, 36), stages=False),
    # sizes WITHOUT a tuned (MFMA) kernel instantiation: they run on the runtime-sized kernel (csrc/lstm_generic.hip)
    dict(name="b3_t16_h320", wseed=36, profile="harsh", args={"sb_model_hidden_size": 320}, inp=("spec", 3, 16, 37), stages=False),
    dict(name="b1_t20_fbn6", wseed=37, profile="default", args={"fb_num_neighbors": 6}, inp=("spec", 1, 20, 38), stages=False),
    dict(name="gru_b3_t16_h190", wseed=38, profile="default", args={"sb_model_hidden_size": 190, "sequence_model": "GRU"},
         inp=("spec", 3, 16, 39), stages=False),
    # CBAM with more than 512 bins: fc1's K is no longer sliced (F / 2 > 256) - the two partial-sum vectors of fe_gate_kernel
    # (mean and max squeeze) must not alias
    dict(name="b1_t16_cbam_f521", wseed=40, profile="harsh", args={"channel_attention_model": "CBAM", "num_freqs": 521},
         inp=("spec", 1, 16, 41), stages=False),
    # output_size off its default (fullsubnet_plus.py:30,104,206): the Linear(H, output_size) of the sub-band model and the final
    # reshape; B = 3 also in the reference's drop_band row order
    dict(name="b1_t20_out3", wseed=41, profile="harsh", args={"output_size": 3}, inp=("spec", 1, 20, 42), stages=False),
    dict(name="b3_t16_out1", wseed=42, profile="default", args={"output_size": 1}, inp=("spec", 3, 16, 43), stages=False),
    dict(name="gru_b3_t16_out5", wseed=43, profile="harsh", args={"output_size": 5, "sequence_model": "GRU"}, inp=("spec", 3, 16, 44), stages=False),
    dict(name="tcn_b1_t20_out3", wseed=44, profile="default", args={"output_size": 3, "sequence_model": "TCN"}, inp=("spec", 1, 20, 45), stages=False),
    dict(name="b1_10s_default", wseed=0, profile="default", args={}, inp=("stft", 1, 10.0, 11), stages=False,
         subsample_f=4),
]


# the original FullSubNet ``Model`` (fullsubnet/model/fullsubnet.py), fixtures named fsn_*
FSN_CASES = [
    dict(name="fsn_b1_t24_default_stages", wseed=1, profile="default", args={}, inp=("spec", 1, 24, 21), stages=True),
    dict(name="fsn_b1_t24_harsh_stages", wseed=2, profile="harsh", args={}, inp=("spec", 1, 24, 22), stages=True),
    dict(name="fsn_b3_t20_harsh", wseed=3, profile="harsh", args={}, inp=("spec", 3, 20, 23), stages=False),
    dict(name="fsn_b5_t16_default", wseed=4, profile="default", args={}, inp=("spec", 5, 16, 24), stages=False),
    dict(name="fsn_b1_t30_cum_layer", wseed=5, profile="default", args={"norm_type": "cumulative_layer_norm"},
         inp=("spec", 1, 30, 25), stages=True),
    dict(name="fsn_b3_t18_cum_laplace", wseed=6, profile="harsh", args={"norm_type": "cumulative_laplace_norm"},
         inp=("spec", 3, 18, 26), stages=False),
    dict(name="fsn_b1_t20_gaussian", wseed=7, profile="default", args={"norm_type": "offline_gaussian_norm"},
         inp=("spec", 1, 20, 27), stages=True),
    dict(name="fsn_b1_2s_default", wseed=8, profile="default", args={}, inp=("stft", 1, 2.0, 28), stages=False),
    dict(name="fsn_gru_b1_t24_harsh_stages", wseed=9, profile="harsh", args={"sequence_model": "GRU"},
         inp=("spec", 1, 24, 29), stages=True),
    dict(name="fsn_b3_t16_fbn2", wseed=11, profile="harsh", args={"fb_num_neighbors": 2}, inp=("spec", 3, 16, 31), stages=False),
    dict(name="fsn_b3_t18_la1_nb10_f161_tanh", wseed=12, profile="harsh",
         args={"look_ahead": 1, "sb_num_neighbors": 10, "num_freqs": 161, "sb_output_activate_function": "Tanh"},
         inp=("spec", 3, 18, 32), stages=False),
    dict(name="fsn_b3_t16_fbn8", wseed=13, profile="harsh", args={"fb_num_neighbors": 8}, inp=("spec", 3, 16, 33), stages=False),
    dict(name="fsn_gru_b3_t20_default", wseed=10, profile="default", args={"sequence_model": "GRU"},
         inp=("spec", 3, 20, 30), stages=False),
    # full-band / sub-band hidden sizes without a tuned kernel instantiation (csrc/lstm_generic.hip)
    dict(name="fsn_b3_t16_fbh256", wseed=14, profile="harsh", args={"fb_model_hidden_size": 256}, inp=("spec", 3, 16, 34), stages=False),
    dict(name="fsn_b1_t20_fbh300_h320", wseed=15, profile="default", args={"fb_model_hidden_size": 300, "sb_model_hidden_size": 320},
         inp=("spec", 1, 20, 35), stages=True),
]

SB_ROWS = [0, 1, 14, 15, 16, 128, 240, 241, 242, 255, 256]   # sub-bands kept from stage sb_input (B=1)


def build_inputs(kind, B, t, seed, num_freqs=257):
    if kind == "stft":
        assert num_freqs == 257
        return make_inputs(B, t, seed)
    return make_spec(B, t, seed, num_freqs)


def run_case(case, FullSubNet_Plus):
    args = dict(ref_loader.DEFAULT_MODEL_ARGS)
    args.update(case["args"])
    torch.manual_seed(0)
    model = FullSubNet_Plus(**args).eval()
    sd = make_state_dict(case["wseed"], case["profile"], attention=args["channel_attention_model"],
                         sequence_model=args["sequence_model"], fb_num_neighbors=args["fb_num_neighbors"],
                         num_freqs=args["num_freqs"], sb_num_neighbors=args["sb_num_neighbors"], kersize=tuple(args["kersize"]),
                         sb_hidden=args["sb_model_hidden_size"], output_size=args.get("output_size", 2))
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    kind, B, t, iseed = case["inp"]
    mag, real, imag = build_inputs(kind, B, t, iseed, args["num_freqs"])

    stages = {}
    hooks = []
    if case["stages"]:
        def grab(name):
            def fn(_m, _i, o):
                stages[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
            return fn
        for nm, mod in (("att_mag", model.channel_attention), ("att_real", model.channel_attention_real),
                        ("att_imag", model.channel_attention_imag), ("fb_mag", model.fb_model),
                        ("fb_real", model.fb_model_real), ("fb_imag", model.fb_model_imag),
                        ("tcn0_mag", model.fb_model.sequence_model[0]),
                        ("lstm_hidden", model.sb_model.sequence_model)):
            hooks.append(mod.register_forward_hook(grab(nm)))
        hooks.append(model.sb_model.register_forward_pre_hook(
            lambda _m, i: stages.__setitem__("sb_input", i[0].detach().numpy().copy())))

    with torch.no_grad():
        out = model(mag, real, imag).numpy()
        for h in hooks:
            h.remove()
        m64 = copy.deepcopy(model).double()
        out64 = m64(mag.double(), real.double(), imag.double()).numpy().astype(np.float32)
        payload = dict(out=out, out64=out64)
        if B > 1:
            full = torch.cat([model(mag[b:b + 1], real[b:b + 1], imag[b:b + 1]) for b in range(B)], 0).numpy()
            full64 = torch.cat([m64(mag[b:b + 1].double(), real[b:b + 1].double(), imag[b:b + 1].double())
                                for b in range(B)], 0).numpy().astype(np.float32)
            payload.update(full=full, full64=full64)
    sub = case.get("subsample_f")
    if sub:
        for k in ("out", "out64", "full", "full64"):
            if k in payload:
                payload[k] = np.ascontiguousarray(payload[k][:, :, ::sub, :])
    if kind == "stft" and mag.shape[-1] <= 200:
        X = torch.complex(real[:, 0], imag[:, 0])            # [B,F,T]
        payload["X"] = X.numpy()
    if "lstm_hidden" in stages:                               # [N,T',H] is big: keep the last frame only
        stages["lstm_hidden_last"] = stages.pop("lstm_hidden")[:, -1, :].copy()
    if "sb_input" in stages:                                  # [N=B*F,34,T']: keep edge + centre sub-bands
        stages["sb_input"] = stages["sb_input"][SB_ROWS].copy()
    for k, v in stages.items():
        payload["stage_" + k] = v.astype(np.float32)
    meta = dict(name=case["name"], wseed=case["wseed"], profile=case["profile"], args=args,
                inp=dict(kind=kind, B=B, t=t, seed=iseed), subsample_f=sub or 1,
                torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
                in_checksum=[float(mag.double().sum()), float(real.double().sum()), float(imag.double().sum())])
    payload["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return payload, (sd, args, mag, real, imag)


def run_case_fsn(case, Model):
    """Same as run_case for the original FullSubNet (one magnitude input)."""
    args = dict(ref_loader.FULLSUBNET_MODEL_ARGS)
    args.update(case["args"])
    torch.manual_seed(0)
    model = Model(**args).eval()
    sd = make_state_dict_fullsubnet(case["wseed"], case["profile"], sequence_model=args["sequence_model"],
                                    fb_num_neighbors=args["fb_num_neighbors"], num_freqs=args["num_freqs"],
                                    sb_num_neighbors=args["sb_num_neighbors"], fb_hidden=args["fb_model_hidden_size"],
                                    sb_hidden=args["sb_model_hidden_size"])
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    kind, B, t, iseed = case["inp"]
    mag, _, _ = build_inputs(kind, B, t, iseed, args["num_freqs"])
    stages = {}
    hooks = []
    if case["stages"]:
        hooks.append(model.fb_model.register_forward_hook(
            lambda _m, _i, o: stages.__setitem__("fb_mag", o.detach().numpy().copy())))
        hooks.append(model.sb_model.register_forward_pre_hook(
            lambda _m, i: stages.__setitem__("sb_input", i[0].detach().numpy()[SB_ROWS].copy())))
    with torch.no_grad():
        out = model(mag).numpy()
        for h in hooks:
            h.remove()
        m64 = copy.deepcopy(model).double()
        payload = dict(out=out, out64=m64(mag.double()).numpy().astype(np.float32))
        if B > 1:
            payload["full"] = torch.cat([model(mag[b:b + 1]) for b in range(B)], 0).numpy()
            payload["full64"] = torch.cat([m64(mag[b:b + 1].double()) for b in range(B)], 0).numpy().astype(np.float32)
    if kind == "stft":
        payload["mag"] = mag.numpy()
    for k, v in stages.items():
        payload["stage_" + k] = v.astype(np.float32)
    meta = dict(name=case["name"], model="fullsubnet", wseed=case["wseed"], profile=case["profile"], args=args,
                inp=dict(kind=kind, B=B, t=t, seed=iseed), subsample_f=1,
                torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
                in_checksum=[float(mag.double().sum())])
    payload["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return payload, (sd, args, mag)


def main_fsn(only):
    from . import fsnp_torch
    Model = ref_loader.load_reference_fullsubnet()
    for case in FSN_CASES:
        if only and case["name"] not in only:
            continue
        payload, (sd, args, mag) = run_case_fsn(case, Model)
        path = os.path.join(GOLDEN_DIR, case["name"] + ".npz")
        np.savez_compressed(path, **payload)
        kw = {k: args[k] for k in ("look_ahead", "sb_num_neighbors", "fb_num_neighbors", "norm_type",
                                   "num_groups_in_drop_band", "fb_output_activate_function",
                                   "sb_output_activate_function")}
        ot = fsnp_torch.forward_fullsubnet(sd, mag, **kw).numpy()
        scale = np.abs(payload["out"]).max()
        print(f"{case['name']:28s} out{payload['out'].shape} scale {scale:.3e} "
              f"ref32-vs-64 {np.abs(payload['out'] - payload['out64']).max() / scale:.2e} "
              f"torch-port {np.abs(ot - payload['out']).max() / scale:.2e} [{os.path.getsize(path) / 1024:.0f} KB]",
              flush=True)


def main():
    only = set(sys.argv[1:])
    if not only or any(n.startswith("fsn_") for n in only):
        main_fsn(only)
        if only and all(n.startswith("fsn_") for n in only):
            return
    FullSubNet_Plus = ref_loader.load_reference()
    assert ref_loader.reference_model_args() == ref_loader.DEFAULT_MODEL_ARGS, "inference.toml drifted"
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    from . import fsnp_numpy, fsnp_torch
    for case in CASES:
        if only and case["name"] not in only:
            continue
        payload, (sd, args, mag, real, imag) = run_case(case, FullSubNet_Plus)
        path = os.path.join(GOLDEN_DIR, case["name"] + ".npz")
        np.savez_compressed(path, **payload)
        # cross-check both restatements right away (report only; tests enforce)
        kw = dict(look_ahead=args["look_ahead"], sb_num_neighbors=args["sb_num_neighbors"],
                  fb_num_neighbors=args["fb_num_neighbors"], norm_type=args["norm_type"],
                  num_groups_in_drop_band=args["num_groups_in_drop_band"],
                  channel_attention_model=args["channel_attention_model"], subband_num=args.get("subband_num", 1),
                  fb_output_activate_function=args["fb_output_activate_function"],
                  sb_output_activate_function=args["sb_output_activate_function"], output_size=args.get("output_size", 2))
        sub = case.get("subsample_f") or 1
        ot = fsnp_torch.forward(sd, mag, real, imag, **kw).numpy()[:, :, ::sub, :]
        scale = np.abs(payload["out"]).max()
        msg = f"{case['name']:28s} out{payload['out'].shape} scale {scale:.3e} " \
              f"ref32-vs-64 {np.abs(payload['out'] - payload['out64']).max() / scale:.2e} " \
              f"torch-port {np.abs(ot - payload['out']).max() / scale:.2e}"
        if mag.shape[-1] <= 40 and args["channel_attention_model"] == "TSSE" and args["sequence_model"] == "LSTM":
            sdn = {k: v.numpy() for k, v in sd.items()}
            kwn = {k: v for k, v in kw.items() if k not in ("channel_attention_model", "subband_num")}
            on = fsnp_numpy.forward(sdn, mag.numpy(), real.numpy(), imag.numpy(), dtype=np.float64, **kwn)
            msg += f" numpy64-vs-ref64 {np.abs(on[:, :, ::sub, :] - payload['out64']).max() / scale:.2e}"
        print(msg, f"[{os.path.getsize(path) / 1024:.0f} KB]", flush=True)


if __name__ == "__main__":
    main()
