"""TEST INFRASTRUCTURE ONLY - torch-CPU functional restatement of the forward.

Same algorithm as oracle/fsnp_numpy.py but on the ATen / oneDNN / MKL CPU kernels the
reference itself runs on (torch.nn.functional.conv1d / group_norm / prelu / linear and
the fused ``torch.lstm``), so it is (a) fast enough for parity at BASELINE.json sizes
and (b) the stand-in for "the reference CPU PyTorch path" that bench.py times as
``cpu_baseline`` (kind "port") on the GPU box, where /root/reference does not exist.

Pinned against tests/golden/*.npz (real-reference outputs) by tests/test_oracle.py.
Citations relative to /root/reference/speech_enhance.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

EPSILON = float(np.finfo(np.float32).eps)  # audio_zen/constant.py:8
TCN_DILATIONS = (1, 2, 5, 9, 1, 2, 5, 9)   # audio_zen/model/module/sequence_model.py:48-57


def offline_laplace_norm(x):
    """base_model.py:210-225"""
    return x / (torch.mean(x, dim=(1, 2, 3), keepdim=True) + 1e-5)


def offline_gaussian_norm(x):
    """base_model.py:260-275"""
    mu = torch.mean(x, dim=(1, 2, 3), keepdim=True)
    std = torch.std(x, dim=(1, 2, 3), keepdim=True)
    return (x - mu) / (std + 1e-5)


def _cum_stats(x):
    B, C, F, T = x.shape
    y = x.reshape(B * C, F, T)
    count = torch.arange(F, F * T + 1, F, dtype=x.dtype).reshape(1, T)
    return y, count


def cumulative_laplace_norm(x):
    """base_model.py:227-258"""
    y, count = _cum_stats(x)
    mean = (torch.cumsum(y.sum(dim=1), dim=-1) / count).unsqueeze(1)
    return (y / (mean + EPSILON)).reshape(x.shape)


def cumulative_layer_norm(x):
    """base_model.py:277-316"""
    y, count = _cum_stats(x)
    cum = torch.cumsum(y.sum(dim=1), dim=-1)
    cum_pow = torch.cumsum(torch.square(y).sum(dim=1), dim=-1)
    mean = cum / count
    var = (cum_pow - 2 * mean * cum) / count + mean.pow(2)
    std = torch.sqrt(var + EPSILON)
    return ((y - mean.unsqueeze(1)) / std.unsqueeze(1)).reshape(x.shape)


NORMS = {
    "offline_laplace_norm": offline_laplace_norm,
    "cumulative_laplace_norm": cumulative_laplace_norm,
    "offline_gaussian_norm": offline_gaussian_norm,
    "cumulative_layer_norm": cumulative_layer_norm,
}


def tsse(x, p, prefix):
    """attention_model.py:78-98"""
    C = x.shape[1]
    feats = []
    for nm in ("smallConv1d", "middleConv1d", "largeConv1d"):
        y = Fn.conv1d(x, p[f"{prefix}.{nm}.0.weight"], p[f"{prefix}.{nm}.0.bias"], groups=C)
        feats.append(torch.relu(y.mean(dim=2, keepdim=True)))
    feature = torch.cat(feats, dim=2)
    squeeze = Fn.linear(feature, p[f"{prefix}.feature_concate_fc.weight"],
                        p[f"{prefix}.feature_concate_fc.bias"])[..., 0]
    h = torch.relu(Fn.linear(squeeze, p[f"{prefix}.fc1.weight"], p[f"{prefix}.fc1.bias"]))
    gate = torch.sigmoid(Fn.linear(h, p[f"{prefix}.fc2.weight"], p[f"{prefix}.fc2.bias"]))
    return x * gate.unsqueeze(2)


def attention(x, p, prefix, kind="TSSE"):
    """channel attention selected by `channel_attention_model` (fullsubnet_plus.py:51-70):
    SE attention_model.py:6-40, ECA :335-359, CBAM :296-332, TSSE :43-98."""
    if kind == "TSSE":
        return tsse(x, p, prefix)
    if kind == "SE":
        sq = x.mean(dim=2)
        h = torch.relu(Fn.linear(sq, p[f"{prefix}.fc1.weight"], p[f"{prefix}.fc1.bias"]))
        gate = torch.sigmoid(Fn.linear(h, p[f"{prefix}.fc2.weight"], p[f"{prefix}.fc2.bias"]))
        return x * gate.unsqueeze(2)
    if kind == "CBAM":
        h = torch.relu(Fn.linear(x.mean(dim=2), p[f"{prefix}.fc1.weight"], p[f"{prefix}.fc1.bias"])) + \
            torch.relu(Fn.linear(x.max(dim=2)[0], p[f"{prefix}.fc1.weight"], p[f"{prefix}.fc1.bias"]))
        gate = torch.sigmoid(Fn.linear(h, p[f"{prefix}.fc2.weight"], p[f"{prefix}.fc2.bias"]))
        return x * gate.unsqueeze(2)
    if kind == "ECA":
        y = x.mean(dim=2, keepdim=True)                                   # AdaptiveAvgPool1d(1): [B,C,1]
        y = Fn.conv1d(y.transpose(-1, -2), p[f"{prefix}.conv.weight"], padding=1).transpose(-1, -2)
        return x * torch.sigmoid(y)
    raise NotImplementedError(kind)


def decompress_cirm(mask, K=10.0, limit=9.9):
    """audio_zen/acoustics/mask.py:60-63"""
    mask = limit * (mask >= limit) - limit * (mask <= -limit) + mask * (torch.abs(mask) < limit)
    return -K * torch.log((K - mask) / (K + mask))


def apply_cirm(pred_crm, noisy_complex):
    """fullsubnet_plus/inferencer/inferencer.py:152-157: pred_crm [B,2,F,T], noisy_complex [B,F,T] complex
    -> enhanced complex [B,F,T]."""
    m = decompress_cirm(pred_crm.permute(0, 2, 3, 1))
    er = m[..., 0] * noisy_complex.real - m[..., 1] * noisy_complex.imag
    ei = m[..., 1] * noisy_complex.real + m[..., 0] * noisy_complex.imag
    return torch.complex(er, ei)


def tcn_block(x, p, prefix, dilation):
    """causal_conv.py:96-108"""
    Hc = p[prefix + ".conv1x1.weight"].shape[0]
    y = Fn.conv1d(x, p[prefix + ".conv1x1.weight"], p[prefix + ".conv1x1.bias"])
    y = Fn.group_norm(Fn.prelu(y, p[prefix + ".prelu1.weight"]), 1,
                      p[prefix + ".norm1.weight"], p[prefix + ".norm1.bias"], 1e-8)
    y = Fn.conv1d(y, p[prefix + ".depthwise_conv.weight"], p[prefix + ".depthwise_conv.bias"],
                  padding=dilation, dilation=dilation, groups=Hc)
    y = Fn.group_norm(Fn.prelu(y, p[prefix + ".prelu2.weight"]), 1,
                      p[prefix + ".norm2.weight"], p[prefix + ".norm2.bias"], 1e-8)
    return x + Fn.conv1d(y, p[prefix + ".sconv.weight"], p[prefix + ".sconv.bias"])


def _activation(x, name):
    if not name:
        return x
    return {"ReLU": torch.relu, "ReLU6": Fn.relu6, "Tanh": torch.tanh}[name](x)


def fb_sequence_model(x, p, prefix, activation="ReLU"):
    """sequence_model.py:106-112"""
    for i, d in enumerate(TCN_DILATIONS):
        x = tcn_block(x, p, f"{prefix}.sequence_model.{i}", d)
    x = torch.relu(x)
    o = Fn.linear(x.permute(0, 2, 1), p[prefix + ".fc_output_layer.weight"], p[prefix + ".fc_output_layer.bias"])
    return _activation(o, activation).permute(0, 2, 1)


def lstm2_fc(x, p, prefix="sb_model", activation=False):
    """sequence_model.py:113-123.  x [N,in,T] -> [N,out,T].  LSTM or GRU (sequence_model.py:31-46) is told apart by
    the gate-block count of weight_hh_l0 ([4H,H] vs [3H,H])."""
    N = x.shape[0]
    H = p[f"{prefix}.sequence_model.weight_hh_l0"].shape[1]
    gru = p[f"{prefix}.sequence_model.weight_hh_l0"].shape[0] == 3 * H
    flat = []
    for layer in (0, 1):
        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            flat.append(p[f"{prefix}.sequence_model.{nm}_l{layer}"])
    seq = x.permute(0, 2, 1).contiguous()
    h0 = torch.zeros(2, N, H, dtype=x.dtype)
    if gru:
        o, _ = torch.gru(seq, h0, flat, True, 2, 0.0, False, False, True)
    else:
        o, _, _ = torch.lstm(seq, (h0, h0.clone()), flat, True, 2, 0.0, False, False, True)
    o = Fn.linear(o, p[prefix + ".fc_output_layer.weight"], p[prefix + ".fc_output_layer.bias"])
    return _activation(o, activation).permute(0, 2, 1).contiguous()


def unfold(x, num_neighbor):
    """base_model.py:15-47"""
    B, C, F, T = x.shape
    if num_neighbor < 1:
        return x.permute(0, 2, 1, 3).reshape(B, F, C, 1, T)
    n = num_neighbor
    out = Fn.pad(x.reshape(B * C, 1, F, T), [0, 0, n, n], mode="reflect")
    out = Fn.unfold(out, (2 * n + 1, T))
    return out.reshape(B, C, 2 * n + 1, T, F).permute(0, 4, 1, 2, 3).contiguous()


def drop_band(x, num_groups=2):
    """audio_zen/acoustics/feature.py:254-285"""
    B, _, F, _ = x.shape
    assert B > num_groups, f"Batch size = {B}, num_groups = {num_groups}."
    if num_groups <= 1:
        return x
    if F % num_groups:
        x = x[..., :F - F % num_groups, :]
    return torch.cat([x[g::num_groups][:, :, g::num_groups, :] for g in range(num_groups)], dim=0)


@torch.no_grad()
def forward(p, noisy_mag, noisy_real, noisy_imag, *, look_ahead=2, sb_num_neighbors=15,
            fb_num_neighbors=0, norm_type="offline_laplace_norm", num_groups_in_drop_band=2,
            fb_output_activate_function="ReLU", sb_output_activate_function=False,
            output_size=2, apply_drop_band=None, stages=None, channel_attention_model="TSSE", subband_num=1):
    """fullsubnet_plus/model/fullsubnet_plus.py:122-209; see fsnp_numpy.forward."""
    assert noisy_mag.dim() == 4
    mag, real, imag = (Fn.pad(a, [0, look_ahead]) for a in (noisy_mag, noisy_real, noisy_imag))
    B, C, F, T = mag.shape
    assert C == 1
    if norm_type not in NORMS:
        raise NotImplementedError("You must set up a type of Norm.")
    norm = NORMS[norm_type]
    rec = (lambda k, v: stages.__setitem__(k, v)) if stages is not None else (lambda k, v: None)
    outs = []
    fb_in_mag = None
    for tag, x, att, fb in (("mag", mag, "channel_attention", "fb_model"),
                            ("real", real, "channel_attention_real", "fb_model_real"),
                            ("imag", imag, "channel_attention_imag", "fb_model_imag")):
        if tag == "mag" and subband_num != 1:                       # fullsubnet_plus.py:146-153
            pad_num = subband_num - F % subband_num
            xin = Fn.pad(norm(x), [0, 0, 0, pad_num], mode="reflect")
            xin = xin.reshape(B, (F + pad_num) // subband_num, T * subband_num)
            xin = attention(xin, p, att, channel_attention_model)
            xin = xin.reshape(B, F + pad_num, T)[:, :F, :]
        else:
            xin = attention(norm(x).reshape(B, F, T), p, att, channel_attention_model)
        rec(f"att_{tag}", xin)
        if tag == "mag":
            fb_in_mag = xin
        o = fb_sequence_model(xin, p, fb, fb_output_activate_function)
        rec(f"fb_{tag}", o)
        outs.append(o.reshape(B, 1, F, T))
    nfb, nsb = 2 * fb_num_neighbors + 1, 2 * sb_num_neighbors + 1
    parts = [unfold(fb_in_mag.reshape(B, 1, F, T), sb_num_neighbors).reshape(B, F, nsb, T)]
    parts += [unfold(o, fb_num_neighbors).reshape(B, F, nfb, T) for o in outs]
    sb_input = norm(torch.cat(parts, dim=2))
    rec("sb_input", sb_input)
    drop = (B > 1) if apply_drop_band is None else apply_drop_band
    Fo = F
    if drop:
        sb_input = drop_band(sb_input.permute(0, 2, 1, 3), num_groups_in_drop_band)
        Fo = sb_input.shape[2]
        sb_input = sb_input.permute(0, 2, 1, 3)
    sb_input = sb_input.reshape(B * Fo, nsb + 3 * nfb, T)
    if "sb_model.sequence_model.0.conv1x1.weight" in p:      # sequence_model="TCN" (sequence_model.py:47-58,106-112)
        sb_mask = fb_sequence_model(sb_input, p, "sb_model", sb_output_activate_function).contiguous()
    else:
        sb_mask = lstm2_fc(sb_input, p, "sb_model", sb_output_activate_function)
    sb_mask = sb_mask.reshape(B, Fo, output_size, T).permute(0, 2, 1, 3).contiguous()
    return sb_mask[:, :, :, look_ahead:]


def forward_full(p, mag, real, imag, **kw):
    """"full" mode: every utterance keeps all bins == the reference run per utterance at
    B=1 and stacked."""
    return forward(p, mag, real, imag, apply_drop_band=False, **kw)


@torch.no_grad()
def forward_fullsubnet(p, noisy_mag, *, look_ahead=2, sb_num_neighbors=15, fb_num_neighbors=0,
                       norm_type="offline_laplace_norm", num_groups_in_drop_band=2,
                       fb_output_activate_function="ReLU", sb_output_activate_function=False,
                       apply_drop_band=None, stages=None):
    """the original FullSubNet forward, speech_enhance/fullsubnet/model/fullsubnet.py:68-118."""
    assert noisy_mag.dim() == 4
    mag = Fn.pad(noisy_mag, [0, look_ahead])                                      # :82
    B, C, F, T = mag.shape
    assert C == 1
    if norm_type not in NORMS:
        raise NotImplementedError("You must set up a type of Norm.")
    norm = NORMS[norm_type]
    rec = (lambda k, v: stages.__setitem__(k, v)) if stages is not None else (lambda k, v: None)
    fb_input = norm(mag).reshape(B, F, T)                                          # :87
    fb_output = lstm2_fc(fb_input, p, "fb_model", fb_output_activate_function)     # :88, sequence_model.py:113-123
    rec("fb_mag", fb_output)
    nfb, nsb = 2 * fb_num_neighbors + 1, 2 * sb_num_neighbors + 1
    fb_unf = unfold(fb_output.reshape(B, 1, F, T), fb_num_neighbors).reshape(B, F, nfb, T)   # :91-92
    mag_unf = unfold(mag, sb_num_neighbors).reshape(B, F, nsb, T)                  # :95-96
    sb_input = norm(torch.cat([mag_unf, fb_unf], dim=2))                           # :99-100
    rec("sb_input", sb_input)
    drop = (B > 1) if apply_drop_band is None else apply_drop_band
    Fo = F
    if drop:                                                                       # :103-106
        sb_input = drop_band(sb_input.permute(0, 2, 1, 3), num_groups_in_drop_band)
        Fo = sb_input.shape[2]
        sb_input = sb_input.permute(0, 2, 1, 3)
    sb_input = sb_input.reshape(B * Fo, nsb + nfb, T)
    sb_mask = lstm2_fc(sb_input, p, "sb_model", sb_output_activate_function)       # :115
    sb_mask = sb_mask.reshape(B, Fo, 2, T).permute(0, 2, 1, 3).contiguous()
    return sb_mask[:, :, :, look_ahead:]                                           # :118


def forward_fullsubnet_full(p, mag, **kw):
    return forward_fullsubnet(p, mag, apply_drop_band=False, **kw)


def stft(y, n_fft=512, hop_length=256, win_length=512):
    """audio_zen/acoustics/feature.py:10-31"""
    return torch.stft(y, n_fft, hop_length, win_length, window=torch.hann_window(n_fft), return_complex=True)


def istft(spec, length, n_fft=512, hop_length=256, win_length=512):
    """audio_zen/acoustics/feature.py:34-56 (complex input)"""
    return torch.istft(spec, n_fft, hop_length, win_length, window=torch.hann_window(n_fft), length=length)


@torch.no_grad()
def enhance_wave(p, noisy, fullsubnet=False, **kw):
    """fullsubnet_plus/inferencer/inferencer.py:142-158 (`mag_complex_full_band_crm_mask`), per utterance (the
    reference inferencer runs batch 1: no drop_band); fullsubnet=True: `full_band_crm_mask` with the magnitude only."""
    X = stft(noisy)
    if fullsubnet:
        mask = forward_fullsubnet_full(p, X.abs().unsqueeze(1), **kw)
    else:
        mask = forward_full(p, X.abs().unsqueeze(1), X.real.unsqueeze(1), X.imag.unsqueeze(1), **kw)
    return istft(apply_cirm(mask, X), noisy.shape[-1])
