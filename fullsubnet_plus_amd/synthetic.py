"""Deterministic synthetic weights and inputs for benchmarks and tests (no checkpoint or dataset travels to the GPU box).

Produces state_dicts with exactly the reference's parameter names and shapes
(dumped from the live reference:
``channel_attention*`` attention_model.py:49-76, ``fb_model*`` sequence_model.py:48-57 +
causal_conv.py:68-94, ``sb_model`` sequence_model.py:31-38,78-79) from a numpy
PCG64 stream, so the build container (reference) and the GPU box (HIP path +
oracle restatements) hold bit-identical weights without shipping a 35 MB file.  This is data generation only - nothing
here computes the forward; `oracle/weights.py` re-exports it for the test infrastructure.

profiles
  "default": PyTorch-like U(-1/sqrt(fan_in), 1/sqrt(fan_in)) everywhere.
  "harsh"  : mimics ``weight_init=True`` (base_model.py:332-397): N(0,1) conv
             weights/biases, Xavier-normal linears with N(0,1) biases,
             orthogonal LSTM matrices with N(0,1) biases.
In both profiles the GroupNorm affine and PReLU slopes are randomised (a trained
checkpoint has non-trivial values; a kernel that ignored them must fail parity).
"""
import numpy as np


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _normal(rng, shape, std=1.0, mean=0.0):
    return (mean + std * rng.standard_normal(size=shape)).astype(np.float32)


def _orthogonal(rng, rows, cols):
    a = rng.standard_normal(size=(max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return np.ascontiguousarray(q[:rows, :cols]).astype(np.float32)


def make_state_dict(seed=0, profile="default", num_freqs=257, tcn_hidden=512, sb_hidden=384,
                    sb_num_neighbors=15, fb_num_neighbors=0, kersize=(3, 5, 10), output_size=2,
                    num_tcn_blocks=8, as_torch=True, attention="TSSE", sequence_model="LSTM"):
    assert profile in ("default", "harsh")
    harsh = profile == "harsh"
    rng = np.random.Generator(np.random.PCG64(seed))
    F = num_freqs
    Fr = F // 2
    sd = {}

    def conv(name, cout, cin_per_group, k):
        fan_in = cin_per_group * k
        if harsh:
            sd[name + ".weight"] = _normal(rng, (cout, cin_per_group, k))
            sd[name + ".bias"] = _normal(rng, (cout,))
        else:
            b = 1.0 / np.sqrt(fan_in)
            sd[name + ".weight"] = _uniform(rng, (cout, cin_per_group, k), b)
            sd[name + ".bias"] = _uniform(rng, (cout,), b)

    def linear(name, cout, cin):
        if harsh:
            std = np.sqrt(2.0 / (cin + cout))
            sd[name + ".weight"] = _normal(rng, (cout, cin), std)
            sd[name + ".bias"] = _normal(rng, (cout,))
        else:
            b = 1.0 / np.sqrt(cin)
            sd[name + ".weight"] = _uniform(rng, (cout, cin), b)
            sd[name + ".bias"] = _uniform(rng, (cout,), b)

    for att in ("channel_attention", "channel_attention_real", "channel_attention_imag"):
        if attention == "TSSE":
            for nm, k in zip(("smallConv1d", "middleConv1d", "largeConv1d"), kersize):
                conv(f"{att}.{nm}.0", F, 1, k)
            linear(f"{att}.feature_concate_fc", 1, 3)
        if attention in ("TSSE", "SE", "CBAM"):
            linear(f"{att}.fc1", Fr, F)
            linear(f"{att}.fc2", F, Fr)
        if attention == "ECA":                      # nn.Conv1d(1, 1, 3, padding=1, bias=False)
            sd[f"{att}.conv.weight"] = _normal(rng, (1, 1, 3)) if harsh else _uniform(rng, (1, 1, 3), 1.0 / np.sqrt(3))

    for fb in ("fb_model", "fb_model_real", "fb_model_imag"):
        for i in range(num_tcn_blocks):
            p = f"{fb}.sequence_model.{i}"
            conv(p + ".conv1x1", tcn_hidden, F, 1)
            sd[p + ".prelu1.weight"] = rng.uniform(0.1, 0.4, size=(1,)).astype(np.float32)
            sd[p + ".norm1.weight"] = _normal(rng, (tcn_hidden,), 0.1, 1.0)
            sd[p + ".norm1.bias"] = _normal(rng, (tcn_hidden,), 0.1)
            conv(p + ".depthwise_conv", tcn_hidden, 1, 3)
            sd[p + ".prelu2.weight"] = rng.uniform(0.1, 0.4, size=(1,)).astype(np.float32)
            sd[p + ".norm2.weight"] = _normal(rng, (tcn_hidden,), 0.1, 1.0)
            sd[p + ".norm2.bias"] = _normal(rng, (tcn_hidden,), 0.1)
            conv(p + ".sconv", F, tcn_hidden, 1)
        linear(f"{fb}.fc_output_layer", F, F)

    sb_in = (2 * sb_num_neighbors + 1) + 3 * (2 * fb_num_neighbors + 1)
    H = sb_hidden
    if sequence_model == "TCN":                        # sequence_model.py:47-58: 8 TCNBlocks(34 -> 512 -> 34) + Linear(34, 2)
        for i in range(num_tcn_blocks):
            p = f"sb_model.sequence_model.{i}"
            conv(p + ".conv1x1", tcn_hidden, sb_in, 1)
            sd[p + ".prelu1.weight"] = rng.uniform(0.1, 0.4, size=(1,)).astype(np.float32)
            sd[p + ".norm1.weight"] = _normal(rng, (tcn_hidden,), 0.1, 1.0)
            sd[p + ".norm1.bias"] = _normal(rng, (tcn_hidden,), 0.1)
            conv(p + ".depthwise_conv", tcn_hidden, 1, 3)
            sd[p + ".prelu2.weight"] = rng.uniform(0.1, 0.4, size=(1,)).astype(np.float32)
            sd[p + ".norm2.weight"] = _normal(rng, (tcn_hidden,), 0.1, 1.0)
            sd[p + ".norm2.bias"] = _normal(rng, (tcn_hidden,), 0.1)
            conv(p + ".sconv", sb_in, tcn_hidden, 1)
        linear("sb_model.fc_output_layer", output_size, sb_in)
        if as_torch:
            import torch
            return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
        return sd
    G = {"LSTM": 4, "GRU": 3}[sequence_model]          # gate blocks of nn.LSTM / nn.GRU (sequence_model.py:31-46)
    for layer, cin in ((0, sb_in), (1, H)):
        p = "sb_model.sequence_model."
        if harsh:
            sd[p + f"weight_ih_l{layer}"] = _orthogonal(rng, G * H, cin)
            sd[p + f"weight_hh_l{layer}"] = _orthogonal(rng, G * H, H)
            sd[p + f"bias_ih_l{layer}"] = _normal(rng, (G * H,))
            sd[p + f"bias_hh_l{layer}"] = _normal(rng, (G * H,))
        else:
            b = 1.0 / np.sqrt(H)
            sd[p + f"weight_ih_l{layer}"] = _uniform(rng, (G * H, cin), b)
            sd[p + f"weight_hh_l{layer}"] = _uniform(rng, (G * H, H), b)
            sd[p + f"bias_ih_l{layer}"] = _uniform(rng, (G * H,), b)
            sd[p + f"bias_hh_l{layer}"] = _uniform(rng, (G * H,), b)
    linear("sb_model.fc_output_layer", output_size, H)

    if as_torch:
        import torch
        return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    return sd


def make_state_dict_fullsubnet(seed=0, profile="default", num_freqs=257, fb_hidden=512, sb_hidden=384,
                                sb_num_neighbors=15, fb_num_neighbors=0, as_torch=True, sequence_model="LSTM"):
    """Original FullSubNet (speech_enhance/fullsubnet/model/fullsubnet.py:39-57): two SequenceModel(LSTM) stacks,
    keys ``fb_model.*`` (257 -> 512 x 2 -> 257) then ``sb_model.*`` (32 -> 384 x 2 -> 2)."""
    assert profile in ("default", "harsh")
    harsh = profile == "harsh"
    rng = np.random.Generator(np.random.PCG64(50_000 + seed))
    sd = {}

    G = {"LSTM": 4, "GRU": 3}[sequence_model]

    def lstm(prefix, cin, H):
        for layer, c in ((0, cin), (1, H)):
            p = prefix + ".sequence_model."
            if harsh:
                sd[p + f"weight_ih_l{layer}"] = _orthogonal(rng, G * H, c)
                sd[p + f"weight_hh_l{layer}"] = _orthogonal(rng, G * H, H)
                sd[p + f"bias_ih_l{layer}"] = _normal(rng, (G * H,))
                sd[p + f"bias_hh_l{layer}"] = _normal(rng, (G * H,))
            else:
                b = 1.0 / np.sqrt(H)
                sd[p + f"weight_ih_l{layer}"] = _uniform(rng, (G * H, c), b)
                sd[p + f"weight_hh_l{layer}"] = _uniform(rng, (G * H, H), b)
                sd[p + f"bias_ih_l{layer}"] = _uniform(rng, (G * H,), b)
                sd[p + f"bias_hh_l{layer}"] = _uniform(rng, (G * H,), b)

    def linear(name, cout, cin):
        if harsh:
            sd[name + ".weight"] = _normal(rng, (cout, cin), np.sqrt(2.0 / (cin + cout)))
            sd[name + ".bias"] = _normal(rng, (cout,))
        else:
            b = 1.0 / np.sqrt(cin)
            sd[name + ".weight"] = _uniform(rng, (cout, cin), b)
            sd[name + ".bias"] = _uniform(rng, (cout,), b)

    lstm("fb_model", num_freqs, fb_hidden)
    linear("fb_model.fc_output_layer", num_freqs, fb_hidden)
    lstm("sb_model", (2 * sb_num_neighbors + 1) + (2 * fb_num_neighbors + 1), sb_hidden)
    linear("sb_model.fc_output_layer", 2, sb_hidden)
    if as_torch:
        import torch
        return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    return sd


def make_wave(batch, seconds, seed, sr=16000, scale=0.1):
    """Seeded synthetic waveform, numpy PCG64 (SURVEY.md 8d uses torch.randn*0.1; we
    avoid the torch RNG so fixtures do not depend on the torch version)."""
    rng = np.random.Generator(np.random.PCG64(10_000 + seed))
    return (scale * rng.standard_normal(size=(batch, int(round(sr * seconds))))).astype(np.float32)


def make_inputs(batch, seconds, seed, n_fft=512, hop=256):
    """STFT exactly as the reference inferencer does (feature.py:10-31,
    inferencer.py:142-147): returns (mag, real, imag) each [B,1,F,T], the latter two
    being strided views of the complex buffer."""
    import torch
    wav = torch.from_numpy(make_wave(batch, seconds, seed))
    X = torch.stft(wav, n_fft, hop, n_fft, window=torch.hann_window(n_fft), return_complex=True)
    mag = X.abs().unsqueeze(1)
    return mag, X.real.unsqueeze(1), X.imag.unsqueeze(1)


# [model.args] of config/inference.toml:29-44, frozen so that the GPU box (which has no /root/reference) builds the same
# network; tests/test_oracle.py asserts it equals the reference's TOML whenever the reference is present.
DEFAULT_MODEL_ARGS = dict(
    sb_num_neighbors=15,
    fb_num_neighbors=0,
    num_freqs=257,
    look_ahead=2,
    sequence_model="LSTM",
    fb_output_activate_function="ReLU",
    sb_output_activate_function=False,
    channel_attention_model="TSSE",
    fb_model_hidden_size=512,
    sb_model_hidden_size=384,
    weight_init=False,
    norm_type="offline_laplace_norm",
    num_groups_in_drop_band=2,
    kersize=[3, 5, 10],
    subband_num=1,
)

# the same for the original FullSubNet (the commented alternative of config/inference.toml:11,28)
FULLSUBNET_MODEL_ARGS = dict(
    sb_num_neighbors=15,
    fb_num_neighbors=0,
    num_freqs=257,
    look_ahead=2,
    sequence_model="LSTM",
    fb_output_activate_function="ReLU",
    sb_output_activate_function=False,
    fb_model_hidden_size=512,
    sb_model_hidden_size=384,
    weight_init=False,
    norm_type="offline_laplace_norm",
    num_groups_in_drop_band=2,
)
