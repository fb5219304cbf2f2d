"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/fsnp.h
declares, the nn.Module mirrors the reference's plugin surface, and the LSTM weight packer agrees
with an emulation of the v_mfma_f32_32x32x2_f32 fragment layouts the kernel relies on."""
import copy
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from fullsubnet_plus_amd import FullSubNet, FullSubNet_Plus, Model, _lib
from oracle.ref_loader import DEFAULT_MODEL_ARGS, FULLSUBNET_MODEL_ARGS
from oracle.weights import make_state_dict, make_state_dict_fullsubnet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    core = open(os.path.join(ROOT, "include", "fsnp.h")).read()
    header = core + open(os.path.join(ROOT, "include", "fsnp_debug.h")).read()
    declared = set(re.findall(r"\b(fsnp_[a-z0-9_]+)\s*\(", header))
    declared -= {"fsnp_handle", "fsnp_config"}
    # fsnp.h is the surface a maintainer binds: no test / tuning hook, no planner internals (those live in fsnp_debug.h)
    core_syms = set(re.findall(r"^[a-z][a-z0-9_ \*]*\b(fsnp_[a-z0-9_]+)\s*\(", core, flags=re.M))
    assert not any(n.startswith("fsnp_debug_") for n in core_syms) and len(core_syms) <= 34, sorted(core_syms)
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.fsnp_version()


def test_create_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)._config()
    hp = ctypes.c_void_p()
    rc = lib.fsnp_create(ctypes.byref(cfg), ctypes.byref(hp))
    assert rc != 0 and not hp.value
    assert b"no CPU fallback" in lib.fsnp_last_error() or b"HIP" in lib.fsnp_last_error()


@pytest.mark.parametrize("field,value,msg", [("num_freqs", 1, b"num_freqs"), ("look_ahead", -1, b"look_ahead"),
                                             ("sb_num_neighbors", -1, b"neighbors"), ("fb_num_neighbors", 300, b"exceed"),
                                             ("subband_num", 2, b"ECA"), ("subband_num", -1, b"subband_num"),
                                             ("num_groups_in_drop_band", 0, b"num_groups_in_drop_band"),
                                             ("output_size", 0, b"output_size"), ("sb_hidden", 0, b"sb_model_hidden_size"),
                                             ("sb_hidden", 6000, b"too large"),
                                             ("norm_type", 7, b"norm_type"), ("attention", 9, b"attention"),
                                             ("model", 5, b"model"), ("sequence_model", 3, b"sequence_model")])
def test_create_validates_the_config_before_touching_the_device(field, value, msg):
    """Bad configurations are rejected with rc 2 and a message naming the field - with or without a GPU."""
    lib = _lib.load()
    cfg = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)._config()
    setattr(cfg, field, value)
    hp = ctypes.c_void_p()
    assert lib.fsnp_create(ctypes.byref(cfg), ctypes.byref(hp)) == 2 and not hp.value
    assert msg in lib.fsnp_last_error(), lib.fsnp_last_error()


def test_module_has_reference_parameter_tree_and_strict_load():
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    sd = make_state_dict(0)
    assert set(m.state_dict()) == set(sd)
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 8_675_102
    assert Model is FullSubNet_Plus
    for attr in ("num_groups_in_drop_band", "look_ahead", "sb_num_neighbors", "fb_num_neighbors", "output_size",
                 "subband_num"):
        assert hasattr(m, attr)
    m2 = copy.deepcopy(m)
    assert m2._hip is not m._hip


def test_weights_key_sees_every_way_of_replacing_a_parameter():
    """The handle re-packs its device weights when _weights_key changes.  The key walks a CACHED parameter list, so every way of
    swapping Parameter objects must drop that cache - not only .to() / _apply: load_state_dict(assign=True), assigning a new
    nn.Parameter to a submodule attribute, replacing a submodule (the forward would otherwise silently keep the
    previous checkpoint's weights), in-place updates, and - every 256th call - a deleted parameter."""
    import torch.nn as nn
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    keys = [m._weights_key()]
    assert m._weights_key() == keys[0]

    def changed():
        k = m._weights_key()
        assert k not in keys
        keys.append(k)
        assert m._weights_key() == k                     # stable again

    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)
    changed()
    m.sb_model.fc_output_layer.weight = nn.Parameter(torch.zeros_like(m.sb_model.fc_output_layer.weight))
    changed()
    m.sb_model.fc_output_layer = nn.Linear(384, 2)
    changed()
    with torch.no_grad():
        m.fb_model.fc_output_layer.bias.add_(1.0)          # in place: same object, version bumped
    changed()
    m.double()
    changed()
    del m.sb_model.fc_output_layer.bias                   # no registration hook fires for a deletion ...
    n_before = len(m._weights_key()[0])
    for _ in range(256):
        k = m._weights_key()                              # ... the periodic full walk catches it
    assert len(k[0]) == n_before - 1


def test_edits_through_dot_data_are_not_silently_ignored():
    """`p.data.add_()` / `p.data.copy_()` - the idiom of the reference's own BaseModel.weight_init (base_model.py:339-355:
    `init.normal_(m.weight.data)`), of EMA / weight averaging - change neither a storage pointer nor a version counter, so the
    (pointer, version) key alone would keep the handle on the previous weights for ever.  Nets (model.py: _weights_key): the key carries
    a content fingerprint refreshed by the periodic walk (here, on CPU parameters; on a GPU the handle watches the storage itself:
    tests/test_gpu_parity.py::test_weight_update_repacks_device_weights), refresh_weights() forces the re-pack at once, and
    `model.apply(model.weight_init)` - the method the reference exposes - goes through it."""
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m._hip.packed_key = k0 = m._weights_key()

    def noticed_within_a_walk():
        for _ in range(256):
            if m._weights_key() != m._hip.packed_key:
                return True
        return False

    assert not noticed_within_a_walk()                          # nothing changed: 256 calls, one walk, same key
    m.sb_model.fc_output_layer.bias.data.add_(1.0)
    assert m._weights_key()[0] == k0[0]                         # pointer and version did not move: the hole
    assert noticed_within_a_walk()                              # ... the fingerprint of the next walk does
    m._hip.packed_key = m._weights_key()
    m.fb_model.sequence_model[3].sconv.weight.data.copy_(torch.randn_like(m.fb_model.sequence_model[3].sconv.weight))
    assert noticed_within_a_walk()
    m._hip.packed_key = m._weights_key()
    m.refresh_weights()                                         # the explicit form: the next forward re-packs whatever the key says
    assert m._hip.packed_key is None
    # the reference's entry point: model.apply(model.weight_init) (fullsubnet_plus.py:119-120) re-initialises through .data
    m._hip.packed_key = m._weights_key()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(11)
    m.apply(m.weight_init)
    assert m._hip.packed_key is None
    after = m.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert any("sequence_model.weight_hh_l0" in k for k in moved) and any("conv1x1.weight" in k for k in moved) and any("fc1.bias" in k for k in moved)
    w = after["sb_model.sequence_model.weight_hh_l1"]            # orthogonal_ on the [1536, 384] matrix: orthonormal columns
    assert torch.allclose(w.T @ w, torch.eye(384), atol=1e-4)
    untouched = [k for k in before if k not in moved]            # PReLU / GroupNorm parameters: not in weight_init's list
    assert untouched and all(("prelu" in k or "norm" in k) for k in untouched), untouched[:5]
    from fullsubnet_plus_amd import FullSubNet
    f = FullSubNet(**FULLSUBNET_MODEL_ARGS)
    f._hip.packed_key = f._weights_key()
    f.apply(f.weight_init)
    assert f._hip.packed_key is None


def test_subband_num_follows_the_reference():
    """fullsubnet_plus.py:47-50,146-163: subband_num > 1 only runs with ECA in the reference (its other attention layers are
    sized num_freqs // subband_num + 1 but the real / imag branches feed them num_freqs); the HIP model accepts exactly that."""
    with pytest.raises(NotImplementedError):
        FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "subband_num": 2})
    m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "subband_num": 2, "channel_attention_model": "ECA"})
    assert m.subband_num == 2 and m._config().subband_num == 2


def test_wide_subband_inputs_pick_the_k64_kernels():
    """fb_num_neighbors = 2..5 (41..64 sub-band features) run on the K = 64 instantiations of the MFMA kernels; wider inputs (and any
    sb_model_hidden_size besides 256 / 384 / 512) are accepted too - they run on the runtime-sized kernel (csrc/lstm_generic.hip),
    as the reference accepts them (fullsubnet_plus.py:102-110)."""
    for fbn in (2, 5, 6):
        m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "fb_num_neighbors": fbn})
        assert m.sb_model.sequence_model.input_size == 31 + 3 * (2 * fbn + 1)
    m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "sb_model_hidden_size": 320})
    assert m.sb_model.sequence_model.hidden_size == 320 and m._config().sb_hidden == 320


@pytest.mark.parametrize("att", ["SE", "ECA", "CBAM"])
def test_attention_variants_have_reference_parameter_tree(att):
    m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "channel_attention_model": att})
    sd = make_state_dict(0, attention=att)
    assert set(m.state_dict()) == set(sd)
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd, strict=True)


def test_error_behaviour_matches_reference():
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    x = torch.zeros(1, 1, 257, 12)
    with pytest.raises(AssertionError):
        m(x[0], x[0], x[0])                       # fullsubnet_plus.py:136
    with pytest.raises(AssertionError):
        m(x.expand(1, 2, 257, 12), x, x)          # fullsubnet_plus.py:141
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x, x)
    with pytest.raises(NotImplementedError):      # base_model.py:328
        FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "norm_type": "forgetting_norm"})
    with pytest.raises(AssertionError):           # fullsubnet_plus.py:45
        FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "sequence_model": "RNN"})
    with pytest.raises(NotImplementedError):      # fullsubnet_plus.py:70
        FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "channel_attention_model": "XYZ"})
    m_tcn = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "sequence_model": "TCN"})      # sequence_model.py:47-58
    sd_tcn = make_state_dict(0, sequence_model="TCN")
    assert list(m_tcn.state_dict().keys()) == list(sd_tcn.keys())
    m_tcn.load_state_dict(sd_tcn, strict=True)


def test_weight_init_true_reinitialises():
    torch.manual_seed(0)
    m = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "weight_init": True})
    w = m.sb_model.sequence_model.weight_hh_l0
    # orthogonal init (base_model.py:380-385): columns orthonormal
    g = (w.T @ w).detach()
    assert torch.allclose(g, torch.eye(384), atol=1e-4)


def _emulate_mfma_gates(pack, A_lds, H, KX, layer, kgroups, NW=4):
    """Replays lstm.hip's inner loop in numpy: for every wave/tile/lane, accumulate
    D[row][col] += A[row][k] * B[k][col] with the documented 32x32x2 f32 fragment maps
    (A: lane l -> A[l&31][l>>5], B: lane l -> B[l>>5][l&31], C: col=l&31,
    row=(r&3)+8*(r>>2)+4*(l>>5)).  Returns gates [32 rows][4H] in reference row order."""
    UW, ST = H // NW, H // NW // 32
    NT = 4 * ST
    KGX, KGH = KX // 8, H // 8
    KG0, KGT = KGX + KGH, KGX + 3 * KGH
    packv = pack.reshape(NW, KGT, NT, 64, 4)
    gates = np.zeros((32, 4 * H), dtype=np.float64)
    g0 = 0 if layer == 0 else KG0
    lanes = np.arange(64)
    for wave in range(NW):
        for n in range(NT):
            D = np.zeros((32, 32))
            for gi in range(kgroups):
                a4 = A_lds[gi]                      # [64 lanes][4]  (what ds_read_b128 returns per lane)
                b4 = packv[wave, g0 + gi, n]        # [64 lanes][4]
                for p in range(4):
                    for kh in range(2):
                        sel = lanes[lanes >> 5 == kh]
                        arow = a4[sel, p]           # A[row = l&31][k]
                        bcol = b4[sel, p]           # B[k][col = l&31]
                        D += np.outer(arow, bcol)
            gate, s = n // ST, n % ST
            cols = gate * H + wave * UW + s * 32 + np.arange(32)
            gates[:, cols] = D
    return gates


@pytest.mark.parametrize("NW", [4, 12])
def test_lstm_pack_matches_mfma_fragment_emulation(NW):
    lib = _lib.load()
    H, NIN, KX = 384, 34, 40
    rng = np.random.default_rng(0)
    wih0 = rng.standard_normal((4 * H, NIN)).astype(np.float32)
    whh0 = rng.standard_normal((4 * H, H)).astype(np.float32)
    wih1 = rng.standard_normal((4 * H, H)).astype(np.float32)
    whh1 = rng.standard_normal((4 * H, H)).astype(np.float32)
    n = (KX // 8 + 3 * H // 8) * (H // 32) * 4 * 64 * 4
    pack = np.zeros(n, dtype=np.float32)
    rc = lib.fsnp_debug_lstm_pack(H, NIN, KX, NW, wih0.ctypes.data, whh0.ctypes.data, wih1.ctypes.data,
                                  whh1.ctypes.data, pack.ctypes.data, n)
    assert rc == 0, lib.fsnp_last_error()

    def a_frag(mat):  # mat [32 rows][K] -> LDS image [K/8][64 lanes][4] per lstm.hip:a_frag_index
        K = mat.shape[1]
        img = np.zeros((K // 8) * 64 * 4)
        for row in range(32):
            for k in range(K):
                img[(((k >> 3) * 64) + ((k & 1) * 32) + row) * 4 + ((k >> 1) & 3)] = mat[row, k]
        return img.reshape(K // 8, 64, 4)

    x = rng.standard_normal((32, NIN))
    h0 = rng.standard_normal((32, H))
    h1 = rng.standard_normal((32, H))
    xp = np.zeros((32, KX)); xp[:, :NIN] = x
    # layer 0: K order [x | h0]
    A0 = np.concatenate([a_frag(xp), a_frag(h0)], axis=0)
    got0 = _emulate_mfma_gates(pack, A0, H, KX, 0, KX // 8 + H // 8, NW)
    want0 = x @ wih0.T.astype(np.float64) + h0 @ whh0.T.astype(np.float64)
    assert np.abs(got0 - want0).max() < 1e-9
    # layer 1: K order [h1 | h0]
    A1 = np.concatenate([a_frag(h1), a_frag(h0)], axis=0)
    got1 = _emulate_mfma_gates(pack, A1, H, KX, 1, 2 * (H // 8), NW)
    want1 = h1 @ whh1.T.astype(np.float64) + h0 @ wih1.T.astype(np.float64)
    assert np.abs(got1 - want1).max() < 1e-9


def test_lstm_hot_loops_have_no_scratch_or_drain():
    """Static check of hipcc's gfx950 assembly (tools/check_lstm_asm.py): the k-group loops of every
    lstm2_fc_kernel instantiation keep the refill-in-place weight pipeline (no scratch traffic; no
    vmcnt(0) drain in the production EX=0 kernels)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse()
    assert len(res) >= 16
    for key, loops in res.items():
        # layer 0, layer 1 (h1 part), layer 1 (h0 part: fp32 groups, or bf16 k-steps which hipcc may fully unroll)
        assert len(loops) == 3 or (key.endswith("_BF1") and len(loops) == 2), (key, loops)
        for i, l in enumerate(loops):
            assert l["scratch"] == 0, (key, l)
            if key.endswith("_BF1") and i == 2:       # bf16 segment: one refill per tile and k-step and TWO MFMAs (hi and lo image of h0, round 6);
                hilo = "_EX0_" in key                 # tiles with VALU rows keep the hi image only (their E images take the LDS the lo image needs)
                assert (2 if hilo else 1) * l["gload"] == l["mfma"], (key, l)
                continue
            assert l["mfma"] % 16 == 0 and l["gload"] * 4 == l["mfma"], (key, l)
            if "_NW4_" in key and ("_EX0_" in key or "_EX1_" in key):   # the production fp32 kernels
                assert l["drain"] == 0, (key, l)


def test_groupnorm_fold_algebra_of_the_dma_sconv_gemm():
    """The identity csrc/tcn.hip's tcn_gemm_dma_kernel<EPI_RESIDUAL> relies on (weights and constants packed by fsnp_create in
    fsnp_abi.hip pack_tcn):  sum_k ((a_k - m) r g_k + b_k) W[n][k] + bias[n] = r * sum_k a_k (g_k W[n][k]) + c1[n] - r m c2[n]
    with c1 = bias + sum_k b_k W, c2 = sum_k g_k W.  In fp32 with fp64-summed constants the two sides agree to < 1e-6 of the
    output scale for a plane whose mean is a fraction of its standard deviation (what PReLU outputs look like) and the
    cancellation in r (acc - m c2) costs ~3e-7 per unit of |mean| / sigma: still < 1e-5 at ten standard deviations."""
    rng = np.random.Generator(np.random.PCG64(11))
    K, N, T = 512, 257, 128
    for mean, bar in ((0.3, 1e-6), (10.0, 1e-5)):
        a = (rng.standard_normal((T, K)) + mean).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        g = (1.0 + 0.3 * rng.standard_normal(K)).astype(np.float32)
        b = (0.2 * rng.standard_normal(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        m = np.float32(a.astype(np.float64).mean())
        r = np.float32(1.0 / np.sqrt(a.astype(np.float64).var() + 1e-8))
        direct = (((a.astype(np.float64) - m) * r * g + b) @ W.astype(np.float64).T + bias)          # the reference's order, fp64
        Wg = (W.astype(np.float64) * g).astype(np.float32)                                           # packed at create time
        c1 = (bias + W.astype(np.float64) @ b).astype(np.float32)
        c2 = (W.astype(np.float64) @ g).astype(np.float32)
        acc = a @ Wg.T                                                                               # the GEMM, fp32
        folded = r * acc + (c1 - (m * r) * c2)
        assert np.abs(folded - direct).max() / np.abs(direct).max() < bar, mean


def test_dma_gemm_lds_image_is_a_conflict_free_permutation():
    """csrc/tcn.hip tcn_gemm_dma_kernel, index arithmetic restated: the DMA writes lane-linear (piece base + 16 bytes x lane), so
    lane l of a piece FETCHES row s >> 2, k-quad (s & 3) ^ swz(row) for slot s; readers take slot row * 4 + (kq ^ swz(row)).
    Checked here: every (row, k-quad) of the 128 x 16 A tile and the 64 x 16 B tile lands in exactly one slot, readers find
    the k-quad they ask for, the 16 lanes of each quarter of a ds_read_b128 touch 64 distinct LDS banks, and the k
    permutation (MFMA j of k-group kg multiplies k = 8 kg + j and 8 kg + 4 + j) is shared by A and B and covers the tile."""
    swz = lambda row: (row >> 2) & 3
    for rows, pieces_per_wave in ((128, 2), (64, 1)):                 # A tile, B tile
        image = {}
        for wave in range(4):
            for piece in range(pieces_per_wave):
                for lane in range(64):
                    s = (wave * pieces_per_wave + piece) * 64 + lane  # lane-linear destination slot
                    row, kq = s >> 2, (s & 3) ^ swz(s >> 2)           # what that lane fetches
                    assert s not in image
                    image[s] = (row, kq)
        assert sorted(image.values()) == [(r, q) for r in range(rows) for q in range(4)]
        for base in range(0, rows, 32):                               # a wave's 32 rows (A) / a 32-column accumulator (B)
            for kg in range(2):
                for quarter in range(4):                              # ds_read_b128: 16 lanes per pass
                    banks = []
                    for lane in range(quarter * 16, quarter * 16 + 16):
                        r, kh = lane & 31, lane >> 5
                        row, kq = base + r, kg * 2 + kh
                        slot = row * 4 + (kq ^ swz(row))
                        assert image[slot] == (row, kq)
                        banks += [(slot * 4 + w) % 64 for w in range(4)]
                    assert len(set(banks)) == 64, (rows, base, kg, quarter)
    ks = sorted(8 * kg + 4 * kh + j for kg in range(2) for kh in range(2) for j in range(4))
    assert ks == list(range(16))


def test_dma_gemm_k_loop_is_stripped_to_the_matrix_pipe():
    """tcn_gemm_dma_kernel (csrc/tcn.hip): between its first and last MFMA the k-loop holds no LDS stores, no scratch, no
    accumulator shuttling and only the offset selects as VALU work - the general kernel issues ~200 VALU/SALU per 16 MFMAs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_dma_gemm()
    assert len(res) == 3                    # conv1x1 (PReLU + statistics), sconv (GroupNorm folded + residual), final Linear
    for key, l in res.items():
        assert l["mfma"] >= 32 and l["dma"] >= 3 and l["ds_read"] >= 6, (key, l)
        assert l["scratch"] == 0 and l["acc_moves"] == 0 and l["ds_write"] == 0 and l["valu"] <= 12, (key, l)
        # round 4: the epilogue moves 16-byte rows only (the tile is transposed through LDS) and drains vmcnt at most twice - the
        # sconv epilogue used to be 32 dependent 4-byte load -> add -> store round trips per lane
        assert l["epi_store16"] == 8 and l["epi_store4"] == 0 and l["epi_load4"] == 0 and l["epi_drains"] <= 2 and l["epi_scratch"] == 0, (key, l)
        assert l["epi_load16"] == (8 if "ILi1E" in key else 0), (key, l)             # EPI_RESIDUAL: the eight residual rows


def test_float4_epilogue_covers_every_element_once():
    """csrc/tcn.hip EpiF4 / the staging loops of the three DMA GEMM kernels, restated: the accumulator layout of
    v_mfma_f32_32x32x2_f32 (lane -> column lane & 31, register q -> row (q & 3) + 8 (q >> 2) + 4 (lane >> 5)) written into a row-major
    [rows][64] LDS slice covers every (row, column) exactly once with conflict-free ds_write_b32 (the 32 lanes of a half wave hit 32
    consecutive banks), and the float4 phase (lane -> column quad lane & 15, rows (lane >> 4) + 4 i) reads every float4 exactly once
    with each 16-lane quarter of a ds_read_b128 covering one whole 256-byte row."""
    # 128-row kernel: a wave's 32 x 64 slice from two accumulators; split-K kernel: 8 x 64 from sum[2][4]; 64-row kernel: 2 x 2 waves on a 64 x 64 tile
    def staged_128(lane, j, q):
        return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), j * 32 + (lane & 31)
    cells = [staged_128(lane, j, q) for lane in range(64) for j in range(2) for q in range(16)]
    assert sorted(cells) == [(r, c) for r in range(32) for c in range(64)]
    for j in range(2):
        for q in range(16):
            for half in range(2):
                banks = [(r * 64 + c) % 32 for r, c in (staged_128(lane, j, q) for lane in range(32 * half, 32 * half + 32))]
                assert len(set(banks)) == 32
    cells = [(qq + 4 * (lane >> 5), j * 32 + (lane & 31)) for lane in range(64) for j in range(2) for qq in range(4)]
    assert sorted(cells) == [(r, c) for r in range(8) for c in range(64)]
    cells = [(wr * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), wc * 32 + (lane & 31)) for wr in range(2) for wc in range(2) for lane in range(64) for q in range(16)]
    assert sorted(cells) == [(r, c) for r in range(64) for c in range(64)]
    for rows in (32, 8, 16):                                  # float4 phase: EpiF4<EPI, ROWS>
        quads = [((lane >> 4) + 4 * i, lane & 15) for lane in range(64) for i in range(rows // 4)]
        assert sorted(quads) == [(r, c) for r in range(rows) for c in range(16)]
        for i in range(rows // 4):
            for quarter in range(4):
                lanes = range(16 * quarter, 16 * quarter + 16)
                assert len({(lane >> 4) + 4 * i for lane in lanes}) == 1 and sorted(lane & 15 for lane in lanes) == list(range(16))


def test_dma64_gemm_image_column_ownership_and_asm():
    """csrc/tcn.hip tcn_gemm_dma64_kernel (round 4: the sconv GEMM on 64 x 64 tiles, column 256 on the VALU), restated:
    * the LDS image [row][8 k-quads] XOR-swizzled by (row >> 1) & 7: the lane-linear DMA pieces (4 per wave and k-tile) cover every
      (row, k-quad) of the 64 x 32 A tile and the 64 x 32 B tile exactly once, readers find the k-quad they ask for, the 16 lanes of
      each quarter of a ds_read_b128 touch 64 distinct banks, and the k permutation is shared by A and B and covers the 32 k of a tile;
    * the VALU column: thread -> slots tid and tid + 256 = rows tid >> 3 and 32 + (tid >> 3); the eight threads of a row hold its eight
      k-quads (each once), so three __shfl_xor steps over lanes 8 i ... 8 i + 7 complete the dot product;
    * the launcher's round count: the kernel runs exactly where it needs no more rounds of workgroups than the 128-row kernel;
    * static: 32 MFMAs and 12 DMA pieces in the unrolled pair of k-tiles (+ the first tile's 4), no scratch, no accumulator moves or LDS
      stores in the loop, 16-byte global traffic only behind it."""
    swz = lambda row: (row >> 1) & 7
    for tile in ("A", "B"):
        image = {}
        for wave in range(4):
            for piece in range(2):
                for lane in range(64):
                    sl = (wave * 2 + piece) * 64 + lane
                    row, kq = sl >> 3, (sl & 7) ^ swz(sl >> 3)
                    assert sl not in image
                    image[sl] = (row, kq)
        assert sorted(image.values()) == [(r, q) for r in range(64) for q in range(8)]
        for base in (0, 32):                                  # wave row half (A: wr) / column half (B: wc)
            for kg in range(4):
                for quarter in range(4):
                    banks = []
                    for lane in range(quarter * 16, quarter * 16 + 16):
                        r, kh = lane & 31, lane >> 5
                        row, kq = base + r, kg * 2 + kh
                        slot = row * 8 + (kq ^ swz(row))
                        assert image[slot] == (row, kq)
                        banks += [(slot * 4 + w) % 64 for w in range(4)]
                    assert len(set(banks)) == 64, (tile, base, kg, quarter)
    assert sorted(8 * kg + 4 * kh + j for kg in range(4) for kh in range(2) for j in range(4)) == list(range(32))
    # the VALU column's ownership
    per_row = {}
    for tid in range(256):
        xrow = tid >> 3
        for row, slot in ((xrow, tid), (xrow + 32, tid + 256)):
            assert slot >> 3 == row
            per_row.setdefault(row, []).append((slot & 7) ^ swz(row))
        assert (tid >> 3) == ((tid ^ 1) >> 3) == ((tid ^ 2) >> 3) == ((tid ^ 4) >> 3)      # the shuffle partners share the row
    assert sorted(per_row) == list(range(64)) and all(sorted(q) == list(range(8)) for q in per_row.values())
    # round counts (launch_gemm_dma64): B = 32 x 2 s takes it (768 workgroups = 3 rounds of half-size tiles vs 2 x 2), B = 24 ties (taken)
    for B, Tp, want in ((32, 128, True), (24, 128, True), (40, 128, True), (32, 628, True)):
        wg64, wg128 = 4 * (-(-Tp // 64)) * B * 3, 5 * (-(-Tp // 128)) * B * 3
        assert (-(-wg64 // 256) <= 2 * (-(-wg128 // 256))) == want, (B, Tp)
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_dma64_gemm()
    assert len(res) == 1                                      # EPI_RESIDUAL only
    for key, l in res.items():
        assert l["mfma"] == 32 and l["dma"] == 12 and l["scratch"] == 0 and l["acc_moves_in_loop"] == 0 and l["ds_write_in_loop"] == 0, (key, l)
        assert l["epi_store4"] == 0 and l["epi_store16"] == 6 and l["epi_load16"] == 4 and l["epi_drains"] <= 3, (key, l)


def _run_exchange_model(program, S, steps, seed, slow=None):
    """Event-level model of one row tile's S workgroups running `program(w)` (a generator of events) under a random scheduler.
    Events: ("wait", counter, target) blocks until the counter reaches the target; ("arrive", counter); ("read_begin", image, version)
    / ("read_end", image): between them the image must hold `version` in EVERY slice and nobody may write it; ("write", image, version):
    this workgroup's slice.  Returns None, or a description of the first hazard."""
    import random
    rng = random.Random(seed)
    gens = [program(w) for w in range(S)]
    pending = [next(g) for g in gens]
    counters, images, readers = {}, {}, {}
    done = [False] * S
    while not all(done):
        runnable = [w for w in range(S) if not done[w] and not (pending[w][0] == "wait" and counters.get(pending[w][1], 0) < pending[w][2])]
        if not runnable:
            return "deadlock"
        if slow is not None and slow in runnable and len(runnable) > 1 and rng.random() < 0.9:
            runnable.remove(slow)                                   # one workgroup that almost never gets a turn
        w = rng.choice(runnable)
        ev = pending[w]
        if ev[0] == "arrive":
            counters[ev[1]] = counters.get(ev[1], 0) + 1
        elif ev[0] == "read_begin":
            img = images.setdefault(ev[1], [-1] * S)                # -1 = the zeroed initial state
            if any(v != ev[2] for v in img):
                return f"workgroup {w} reads {ev[1]} expecting version {ev[2]}, slices hold {img}"
            readers.setdefault(ev[1], set()).add(w)
        elif ev[0] == "read_end":
            readers[ev[1]].discard(w)
        elif ev[0] == "write":
            if readers.get(ev[1]):
                return f"workgroup {w} writes {ev[1]} (version {ev[2]}) while {sorted(readers[ev[1]])} still read it"
            images.setdefault(ev[1], [-1] * S)[w] = ev[2]
        try:
            pending[w] = next(gens[w])
        except StopIteration:
            done[w] = True
    return None


def test_exchange_schedules_are_race_free_under_any_interleaving():
    """The inter-workgroup hand-off SCHEDULES of csrc/lstm_coop.hip restated as event programs - which image / counter every phase
    waits for, reads and overwrites, in the kernels' program order - and run under random and adversarial (one starved workgroup)
    interleavings: no image is read before all its slices carry the expected step, none is overwritten while a peer still reads it,
    nobody deadlocks.  (The device-side counterpart is the drift injection of tests/test_gpu_soak.py.)  Negative controls: the
    layer-skewed schedule with TWO h0 images and the serial schedule without its barrier both fail here at once."""
    def serial(S, T, barrier=True):                                  # lstm2_coop_kernel / lstm2_coopn_kernel: ONE barrier per step
        def prog(w):
            for t in range(T):
                cur, prv = t & 1, (t & 1) ^ 1
                yield ("read_begin", f"h0[{prv}]", t - 1); yield ("read_end", f"h0[{prv}]")            # layer 0 over h0_{t-1}
                yield ("write", f"h0[{cur}]", t)
                yield ("arrive", "b")
                if barrier:
                    yield ("wait", "b", S * (t + 1))
                if w == 0 and t > 0:                                                                    # Linear partials of step t - 1
                    yield ("read_begin", f"fc[{prv}]", t - 1); yield ("read_end", f"fc[{prv}]")
                yield ("read_begin", f"h1[{prv}]", t - 1); yield ("read_begin", f"h0[{cur}]", t)       # layer 1 over [h1_{t-1} | h0_t]
                yield ("read_end", f"h1[{prv}]"); yield ("read_end", f"h0[{cur}]")
                yield ("write", f"h1[{cur}]", t); yield ("write", f"fc[{cur}]", t)
            yield ("arrive", "b"); yield ("wait", "b", S * (T + 1))
            if w == 0:
                yield ("read_begin", f"fc[{(T - 1) & 1}]", T - 1); yield ("read_end", f"fc[{(T - 1) & 1}]")
        return prog

    def skewed(S, T, images=3):                                      # lstm2_coop_skew_kernel: A_t = layer 0 of step t, C_t = layer 1
        def phase_a(t, with_c):
            yield ("read_begin", f"h0[{(t - 1) % images}]", t - 1); yield ("read_end", f"h0[{(t - 1) % images}]")
            yield ("write", f"h0[{t % images}]", t)
            if with_c:
                yield ("wait", "b1", S * (t - 1))                    # h1_{t-2}, Linear partials of step t - 2
            yield ("arrive", "b0")
        def phase_c(t):
            cur, prv = t & 1, (t & 1) ^ 1
            yield ("read_begin", f"h1[{prv}]", t - 1); yield ("read_begin", f"h0[{t % images}]", t)
            yield ("read_end", f"h1[{prv}]"); yield ("read_end", f"h0[{t % images}]")
            yield ("write", f"h1[{cur}]", t); yield ("write", f"fc[{cur}]", t)
            yield ("arrive", "b1")
        def fc(w, t):
            if w == 0:
                yield ("read_begin", f"fc[{t & 1}]", t); yield ("read_end", f"fc[{t & 1}]")
        def prog(w):
            yield from phase_a(0, False)
            for t in range(1, T):
                yield ("wait", "b0", S * t)
                yield from phase_a(t, True)
                if t >= 2:
                    yield from fc(w, t - 2)
                yield from phase_c(t - 1)
            yield ("wait", "b0", S * T); yield ("wait", "b1", S * (T - 1))
            if T >= 2:
                yield from fc(w, T - 2)
            yield from phase_c(T - 1)
            yield ("wait", "b1", S * T)
            yield from fc(w, T - 1)
        return prog

    def half_tile_ping_pong(S, T, h0_images=2):                      # lstm2_coop_hp_kernel, one half tile (the halves are independent:
        def prog(w):                                                 # own images, own counter); fused phase t = [layer 1 of t, layer 0 of t + 1]
            yield ("write", "h0[0]", 0); yield ("arrive", "c")       # phase -1: h0_0
            for t in range(T):
                yield ("wait", "c", S * (t + 1))
                yield ("read_begin", f"h1[{(t - 1) & 1}]", t - 1); yield ("read_begin", f"h0[{t % h0_images}]", t)
                if w == 0 and t >= 1:
                    yield ("read_begin", f"fc[{(t - 1) & 1}]", t - 1); yield ("read_end", f"fc[{(t - 1) & 1}]")
                yield ("read_end", f"h1[{(t - 1) & 1}]"); yield ("read_end", f"h0[{t % h0_images}]")
                yield ("write", f"h1[{t & 1}]", t); yield ("write", f"fc[{t & 1}]", t); yield ("write", f"h0[{(t + 1) % h0_images}]", t + 1)
                yield ("arrive", "c")
            yield ("wait", "c", S * (T + 1))
            if w == 0:
                yield ("read_begin", f"fc[{(T - 1) & 1}]", T - 1); yield ("read_end", f"fc[{(T - 1) & 1}]")
        return prog

    for S, T in ((3, 9), (6, 14)):
        for seed in range(40):
            for slow in (None, 0, S - 1):
                assert _run_exchange_model(serial(S, T), S, T, seed, slow) is None
                assert _run_exchange_model(skewed(S, T), S, T, seed, slow) is None
                assert _run_exchange_model(half_tile_ping_pong(S, T), S, T, seed, slow) is None
    assert any(_run_exchange_model(half_tile_ping_pong(4, 12, h0_images=1), 4, 12, seed, slow=1) for seed in range(20))
    # negative controls: the model does catch what it is there to catch
    assert any(_run_exchange_model(skewed(4, 12, images=2), 4, 12, seed, slow=1) for seed in range(20))
    assert any(_run_exchange_model(serial(4, 12, barrier=False), 4, 12, seed, slow=1) for seed in range(20))


def test_half_tile_ping_pong_index_arithmetic():
    """csrc/lstm_hp.hip, index arithmetic restated: (1) hp_a16 puts element (row, k) of a half-tile image where lane (k & 3) * 16 + row
    of k-group k >> 4 reads component (k >> 2) & 3 - the A operand of v_mfma_f32_16x16x4_f32 number j = (k >> 2) & 3 of that group;
    (2) a wave's accumulator (lane l, register i = row 4 (l >> 4) + i, column l & 15), stored as one float4 per lane, is read back by
    the cell thread (wave = row quad, lane = (row & 3) * 16 + unit) at float (wave * 16 + unit) * 4 + (lane >> 4): all 256 threads
    distinct, 64 distinct banks per wave; (3) the staged h slice lands in hp_a16 order of the workgroup's own k-group; (4) the Linear
    partials are found where the summing workgroup looks for them."""
    a16 = lambda row, k: ((((k >> 4) * 4) + (k & 3)) * 16 + row) * 4 + ((k >> 2) & 3)
    seen = set()
    for k in range(48):
        for row in range(16):
            idx = a16(row, k)
            g, lane, j = idx // 256, (idx // 4) % 64, idx % 4
            assert g == k >> 4 and lane == (k & 3) * 16 + row and j == (k >> 2) & 3
            assert k == 16 * g + 4 * j + (lane >> 4)                 # = the k of the B fragment lstm_hp_pack_weights puts in (f = g, lane, j)
            seen.add(idx)
    assert seen == set(range(48 * 16))
    read_idx = {}
    for wave in range(4):
        banks = set()
        for lane in range(64):
            cu, crow = lane & 15, 4 * wave + (lane >> 4)
            gidx = (wave * 16 + cu) * 4 + (lane >> 4)
            producer_lane, reg = (crow >> 2) * 16 + cu, crow & 3      # MFMA D layout: row 4 (l >> 4) + i, column l & 15
            assert gidx == producer_lane * 4 + reg
            read_idx[(crow, cu)] = gidx
            banks.add(gidx % 64)
            sdst = ((cu & 3) * 16 + crow) * 4 + (cu >> 2)
            assert sdst == a16(crow, cu)                              # k-group 0 of the staging image = the workgroup's 16 units
        assert len(banks) == 64
    assert sorted(read_idx.values()) == list(range(256))
    S = 24
    for cs in range(16):                                              # summing workgroup cs (row cs): lane tid < 2 S reads partial (cs', o)
        for tid in range(2 * S):
            csp, o = tid % S, tid // S
            voff = csp * 128 + (o * 16 + cs) * 4
            assert voff == csp * 128 + 4 * (o * 16 + cs) and voff // 128 == csp and (voff % 128) // 4 == o * 16 + cs   # stage[512 + o * 16 + row]


def test_wave_owned_half_tile_kernel_data_flow_on_the_cpu():
    """csrc/lstm_hpw.hip restated with numpy on the weights its packer really produces (fsnp_debug_lstm_hpw_pack): the TRANSPOSED product
    (weights = A operand of v_mfma_f32_16x16x4_f32, h / x = B operand) leaves gate i of cell (sequence lane & 15, unit 16 cs + w + 4 (lane >> 4))
    in accumulator register i of that lane - for every participant, both layers, K = [x | h0] and [h1 | h0]; the lane's result lands at
    hp_a16(sequence, unit) of the exchange image = participant * 256 + sequence * 16 + (lane >> 4) * 4 bytes (one contiguous 256-byte row per
    wave); the Linear partials are found where the summing wave looks for them."""
    lib = _lib.load()
    H, NIN, KX = 64, 34, 40                       # (the index arithmetic does not depend on H; 64 keeps the emulation small)
    GX, GH = (KX + 15) // 16, H // 16
    NF = GX + 3 * GH
    rng = np.random.default_rng(5)
    wih0 = rng.standard_normal((4 * H, NIN)).astype(np.float32)
    whh0, wih1, whh1 = (rng.standard_normal((4 * H, H)).astype(np.float32) for _ in range(3))
    n = (H // 16) * 4 * NF * 256
    out = np.zeros(n, np.float32)
    assert lib.fsnp_debug_lstm_hpw_pack(H, NIN, KX, wih0.ctypes.data, whh0.ctypes.data, wih1.ctypes.data, whh1.ctypes.data, out.ctypes.data, n) == 0
    pack = out.reshape(H // 16 * 4, NF, 64, 4)                        # [participant][fragment][lane][j]
    x = rng.standard_normal((16, KX)).astype(np.float32); x[:, NIN:] = 7.0     # features >= NIN: garbage that must meet zero weights
    h0 = rng.standard_normal((16, H)).astype(np.float32)
    h1 = rng.standard_normal((16, H)).astype(np.float32)
    a16 = lambda row, k: ((((k >> 4) * 4) + (k & 3)) * 16 + row) * 4 + ((k >> 2) & 3)
    img0, img1 = np.zeros(16 * H, np.float32), np.zeros(16 * H, np.float32)
    for r in range(16):
        for k in range(H):
            img0[a16(r, k)] = h0[r, k]; img1[a16(r, k)] = h1[r, k]
    lanes = np.arange(64)

    def mfma(acc, a_lane, b_lane):               # D[m][n] += sum_kk A[m][kk] B[kk][n]; A from lane kk * 16 + m, B from lane kk * 16 + n
        A = a_lane.reshape(4, 16).T              # [m][kk]
        B = b_lane.reshape(4, 16)                # [kk][n]
        return acc + A.astype(np.float64) @ B.astype(np.float64)

    want0 = np.concatenate([x[:, :NIN], h0], 1).astype(np.float64) @ np.concatenate([wih0, whh0], 1).astype(np.float64).T      # [seq][4H]
    want1 = np.concatenate([h1, h0], 1).astype(np.float64) @ np.concatenate([whh1, wih1], 1).astype(np.float64).T
    for part in range(H // 16 * 4):
        cs, wv = part >> 2, part & 3
        d0, d1 = np.zeros((16, 16)), np.zeros((16, 16))
        for i in range(KX // 4):                 # x k-groups: the lane's fragment component i & 3 of group i >> 2 = x[seq][4 i + (lane >> 4)]
            xb = x[lanes & 15, 4 * i + (lanes >> 4)]
            d0 = mfma(d0, pack[part, i >> 2, :, i & 3], xb)
        for g in range(GH):
            p = img1.reshape(GH, 64, 4)[g]       # one 16-byte load per lane: float4 index g * 64 + lane
            q = img0.reshape(GH, 64, 4)[g]
            for j in range(4):
                d1 = mfma(d1, pack[part, GX + GH + g, :, j], p[:, j])
                d0 = mfma(d0, pack[part, GX + g, :, j], q[:, j])
                d1 = mfma(d1, pack[part, GX + 2 * GH + g, :, j], q[:, j])
        for lane in range(64):
            seq, jj = lane & 15, lane >> 4
            unit = 16 * cs + wv + 4 * jj
            for gate in range(4):                # accumulator register i of lane L = D[m = 4 (L >> 4) + i][n = L & 15]
                assert abs(d0[4 * jj + gate, seq] - want0[seq, gate * H + unit]) < 1e-3
                assert abs(d1[4 * jj + gate, seq] - want1[seq, gate * H + unit]) < 1e-3
            assert a16(seq, unit) * 4 == part * 256 + seq * 16 + jj * 4
    P = 4 * 24
    for cs in range(16):                         # summing wave of row cs: lane + 64 j reads the float2 of participant p at p * 128 + cs * 8
        for p_ in range(P):
            assert (p_ * 32 + cs * 2) * 4 == p_ * 128 + cs * 8
    for part in range(P):                        # a wave's partial: lanes < 16 (sequence = lane) store float2 at (part * 32 + seq * 2) * 4
        assert {(part * 32 + seq * 2) * 4 for seq in range(16)} == {part * 128 + 8 * r for r in range(16)}


def test_models_carry_the_reference_base_model_helpers():
    """SURVEY.md 8(b) "module protocol": the reference model objects expose `norm` (fullsubnet_plus.py:115, fullsubnet.py:61), `norm_wrapper`
    (base_model.py:318-330) and `unfold` (base_model.py:15-47) next to forward().  The HIP classes carry them too (fsnp_norm / fsnp_unfold
    kernels); like forward() they refuse CPU tensors - there is no CPU fallback - and unknown norm names raise the reference's error."""
    for m in (FullSubNet_Plus(**DEFAULT_MODEL_ARGS), FullSubNet(**FULLSUBNET_MODEL_ARGS)):
        assert callable(m.norm) and m.norm.__name__ == m.norm_type
        assert callable(m.norm_wrapper("cumulative_layer_norm")) and callable(m.unfold) and callable(type(m).unfold)
        with pytest.raises(NotImplementedError, match="You must set up a type of Norm"):
            m.norm_wrapper("forgetting_norm")
        x = torch.zeros(2, 1, 9, 5)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.norm(x)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.unfold(x, 2)
        with pytest.raises(AssertionError, match="four dim"):
            m.unfold(x[0], 2)
        # the sub-band model is callable as a submodule, like the reference's `self.sb_model(sb_input)` (fullsubnet_plus.py:203,
        # fullsubnet.py:114): it runs its OWNER's kernels, also after a deepcopy / a whole-module pickle round trip
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.sb_model(torch.zeros(3, m.sb_model.sequence_model.input_size, 4))
        with pytest.raises(AssertionError, match="shape of input"):
            m.sb_model(torch.zeros(3, 4))
        twin = copy.deepcopy(m)
        assert twin.sb_model.__dict__["_fsnp_owner"]() is twin and m.sb_model.__dict__["_fsnp_owner"]() is m
        if isinstance(m, FullSubNet_Plus):      # ... and so are the attention layers and full-band stacks (fullsubnet_plus.py:160-165, 171-173)
            for branch, tag in enumerate(("", "_real", "_imag")):
                for name in ("channel_attention" + tag, "fb_model" + tag):
                    holder = getattr(m, name)
                    assert holder.__dict__["_fsnp_branch"] == branch and getattr(twin, name).__dict__["_fsnp_owner"]() is twin
                    with pytest.raises(RuntimeError, match="no CPU fallback"):
                        holder(torch.zeros(2, 257, 6))
        else:                                   # (the original FullSubNet's full-band model is recurrent: no stage entry point, said so)
            with pytest.raises(RuntimeError, match="not a callable stage"):
                m.fb_model(torch.zeros(2, 257, 6))
        import io
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)
        assert back.sb_model.__dict__["_fsnp_owner"]() is back and list(back.state_dict()) == list(m.state_dict())
    lib = _lib.load()
    st = (ctypes.c_int64 * 4)(1, 1, 1, 1)
    assert lib.fsnp_norm(9, 1, ctypes.byref(st), 1, 1, 1, 1, 1, None) == 2 and b"norm_type" in lib.fsnp_last_error()
    assert lib.fsnp_unfold(1, ctypes.byref(st), 1, 1, 1, 4, 1, 4, None) == 2 and b"reflect" in lib.fsnp_last_error()
    assert lib.fsnp_norm(0, None, ctypes.byref(st), 1, 1, 1, 1, 1, None) == 1


def test_splitk_gemm_has_no_barrier_in_its_k_loop():
    """tcn_gemm_sk_kernel (csrc/tcn.hip, small batches): a wave multiplies ITS k-tiles out of a private double buffer - 16 MFMAs per
    k-tile, 6 + 6 DMA pieces, no scratch, no workgroup barrier between the first and the last MFMA."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_splitk_gemm()
    assert len(res) == 3, sorted(res)
    for key, l in res.items():
        assert l["mfma"] == 16 and l["dma"] == 12 and l["scratch"] == 0 and l["barriers_between_mfmas"] == 0, (key, l)
    # after the partial tiles are added, wave w finishes accumulator registers q in [4 w, 4 w + 4): with the 32x32 C/D layout
    # (row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)) that is rows t0 + qq + 8 w + 4 (lane >> 5) - every row of the tile exactly once
    rows = sorted((q & 3) + 8 * (q >> 2) + 4 * half for w in range(4) for q in range(4 * w, 4 * w + 4) for half in (0, 1))
    assert rows == list(range(32))
    assert all((q & 3) + 8 * (q >> 2) == (q - 4 * w) + 8 * w for w in range(4) for q in range(4 * w, 4 * w + 4))
    # k-tiles w, w + 4, ... of the four waves partition [0, ktiles) for the two GEMM shapes (K = 257 -> 17 tiles, K = 512 -> 32)
    for ktiles in (17, 32, 1, 3):
        assert sorted(kt for w in range(4) for kt in range(w, ktiles, 4)) == list(range(ktiles))


def test_half_tile_hot_loops_keep_their_accumulators_in_agprs():
    """lstm2_fc16_kernel (csrc/lstm16.hip): every k-group loop is 96 MFMAs + 24 weight loads, no scratch, no drain and no
    AGPR<->VGPR shuttling of the 24 accumulator tiles (the asm pins in the kernel exist for exactly that)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_half_tile()
    assert len(res) >= 1
    for key, loops in res.items():
        assert len(loops) >= 3, (key, loops)
        bf = "ELi2ELb1ELb" in key                    # the bf16 ih-GEMM instantiation <384, 40, 2, BF = true, OWN> (round 4): its last loop is 2 x 24 bf16 MFMAs per k-step of 32
        for l in loops:                              # (round 6: every weight fragment multiplies the hi AND the lo bf16 image of h0)
            assert l["gload"] == 24 and (l["mfma"] == 96 or (bf and l["mfma"] == 48)), (key, l)
            assert l["scratch"] == 0 and l["drain"] == 0 and l["acc_moves"] == 0, (key, l)
        assert sum(1 for l in loops if l["mfma"] == 48) == (1 if bf else 0), (key, loops)


# ---------------------------------------------------------------- cooperative kernel weight stream (csrc/lstm_coop.hip)
@pytest.mark.parametrize("H,NIN,KX,units", [(384, 34, 40, 8), (384, 34, 40, 16), (384, 32, 40, 32), (384, 34, 40, 64),
                                            (512, 257, 264, 8), (512, 257, 264, 32)])
def test_lstm_coop_pack_matches_mfma_fragment_emulation(H, NIN, KX, units):
    """Emulates what lstm2_coop_kernel does with the packed stream: workgroup cs, wave w walks its local k-groups
    (global k-group 4 i + w), multiplies the A fragments of [x | h0] / [h1 | h0] with the packed B fragments, the four
    waves' partial tiles are summed, column j = gate * units + unit."""
    lib = _lib.load()
    rng = np.random.default_rng(units + H)
    wih0 = rng.standard_normal((4 * H, NIN)).astype(np.float32)
    whh0 = rng.standard_normal((4 * H, H)).astype(np.float32)
    wih1 = rng.standard_normal((4 * H, H)).astype(np.float32)
    whh1 = rng.standard_normal((4 * H, H)).astype(np.float32)
    S, NT = H // units, units // 8
    KGXP, KGH = (KX // 8 + 3) // 4 * 4, H // 8
    G0W, G1W = (KGXP + KGH) // 4, KGH // 2
    GW = G0W + G1W
    n = S * 4 * GW * NT * 64 * 4
    pack = np.zeros(n, dtype=np.float32)
    rc = lib.fsnp_debug_lstm_coop_pack(H, NIN, KX, units, wih0.ctypes.data, whh0.ctypes.data, wih1.ctypes.data,
                                       whh1.ctypes.data, pack.ctypes.data, n)
    assert rc == 0, lib.fsnp_last_error()
    pack = pack.reshape(S, 4, GW, NT, 64, 4).astype(np.float64)

    def a_img(mat, groups):  # [32][K] -> [groups][64 lanes][4], lane = row + 32 (k & 1), component (k >> 1) & 3
        img = np.zeros((groups, 64, 4))
        for k in range(mat.shape[1]):
            img[k >> 3, (k & 1) * 32 + np.arange(32), (k >> 1) & 3] = mat[:, k]
        return img

    x = rng.standard_normal((32, NIN)); h0 = rng.standard_normal((32, H)); h1 = rng.standard_normal((32, H))
    ximg, h0img, h1img = a_img(x, KGXP), a_img(h0, KGH), a_img(h1, KGH)
    got = np.zeros((2, 32, 4 * H))
    for cs in range(S):
        for layer, (G, base, first, nfirst, second) in enumerate(((G0W, 0, ximg, KGXP, h0img), (G1W, G0W, h1img, KGH, h0img))):
            acc = np.zeros((NT, 32, 32))
            for wave in range(4):
                for i in range(G):
                    g = 4 * i + wave
                    A = first[g] if g < nfirst else second[g - nfirst]          # [64][4]
                    Bf = pack[cs, wave, base + i]                               # [NT][64][4]
                    for p in range(4):
                        Am = np.stack([A[:32, p], A[32:, p]], axis=1)            # [row][k half]
                        for t in range(NT):
                            Bm = np.stack([Bf[t, :32, p], Bf[t, 32:, p]], axis=0)  # [k half][col]
                            acc[t] += Am @ Bm
            for t in range(NT):
                for col in range(32):
                    j = t * 32 + col
                    got[layer, :, (j // units) * H + cs * units + j % units] = acc[t, :, col]
    want0 = x @ wih0.T.astype(np.float64) + h0 @ whh0.T.astype(np.float64)
    want1 = h1 @ whh1.T.astype(np.float64) + h0 @ wih1.T.astype(np.float64)
    assert np.abs(got[0] - want0).max() < 1e-9
    assert np.abs(got[1] - want1).max() < 1e-9


# ---------------------------------------------------------------- the original FullSubNet
def test_fullsubnet_module_has_reference_parameter_tree_and_strict_load():
    m = FullSubNet(**FULLSUBNET_MODEL_ARGS)
    sd = make_state_dict_fullsubnet(0)
    own = m.state_dict()
    assert list(own.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    bad = dict(sd); bad.pop("fb_model.fc_output_layer.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    from fullsubnet_plus_amd.fullsubnet import Model as FsnModel
    assert FsnModel is FullSubNet
    for attr in ("num_groups_in_drop_band", "look_ahead", "sb_num_neighbors", "fb_num_neighbors"):
        assert getattr(m, attr) == FULLSUBNET_MODEL_ARGS[attr]


def test_fullsubnet_error_behaviour_matches_reference():
    args = dict(FULLSUBNET_MODEL_ARGS)
    with pytest.raises(AssertionError):                       # fullsubnet.py:37
        FullSubNet(**{**args, "sequence_model": "TCN"})
    with pytest.raises(NotImplementedError):                  # base_model.py:328
        FullSubNet(**{**args, "norm_type": "nope"})
    m = FullSubNet(**args)
    with pytest.raises(AssertionError):                       # fullsubnet.py:81
        m(torch.zeros(1, 257, 10))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 257, 10))


@pytest.mark.parametrize("cls,args,maker", [(FullSubNet_Plus, DEFAULT_MODEL_ARGS, make_state_dict),
                                            (FullSubNet, FULLSUBNET_MODEL_ARGS, make_state_dict_fullsubnet)])
def test_gru_variant_has_reference_parameter_tree(cls, args, maker):
    """sequence_model="GRU" (sequence_model.py:39-46) - nn.GRU key names and [3H, .] shapes."""
    m = cls(**{**args, "sequence_model": "GRU"})
    sd = maker(0, sequence_model="GRU")
    own = m.state_dict()
    assert list(own.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    assert own["sb_model.sequence_model.weight_hh_l0"].shape == (3 * 384, 384)
    m.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):             # an LSTM checkpoint must not load into the GRU model
        m.load_state_dict(maker(0), strict=True)


def test_column_split_kernels_keep_their_asm_invariants():
    """Static check (tools/check_lstm_asm.py analyse_column_split) of every lstm2_coop / lstm2_coopn instantiation:
    MFMAs present, no scratch inside the time loop, no buffer_wbl2 / buffer_inv (the write-through hand-off
    of csrc/lstm_common.h needs no cache maintenance) and the exchange images really are read / written with sc1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_column_split()
    assert len(res) >= 18, sorted(res)
    for name, r in res.items():
        assert r["mfma"] >= 48, (name, r)
        assert r["scratch_in_loop"] == 0 and r["cache_maint"] == 0, (name, r)
        assert r["sc1_loads"] > 0 and r["sc1_stores"] > 0, (name, r)


def test_full_band_matrix_vector_kernel_weight_layout():
    """csrc/lstm_fbv.hip (original FullSubNet, <= 4 utterances), restated on the CPU: thread (c, ks) of column slice cs holds gate c & 3
    of unit 8 cs + (c >> 2) over k slice ks - 100 weights of layer 0 over [x (288, zero padded) | h0] and 128 of layer 1 over
    [h0 | h1].  Emulating the kernel's sums on the packer's output (eight k slices per column, then the cell's four gates) reproduces
    W_ih x + W_hh h for both layers, for num_freqs 257 and 161; the vector layout [x | h0 | h1] makes layer 1's k range contiguous."""
    import ctypes as ct
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(11))
    H, XP, K0, K1 = 512, 288, 100, 128
    assert XP + H == 8 * K0 and 2 * H == 8 * K1
    vp = lambda a: a.ctypes.data_as(ct.c_void_p)
    for NIN in (257, 161):
        wih0, whh0 = rng.standard_normal((4 * H, NIN)).astype(np.float32), rng.standard_normal((4 * H, H)).astype(np.float32)
        wih1, whh1 = rng.standard_normal((4 * H, H)).astype(np.float32), rng.standard_normal((4 * H, H)).astype(np.float32)
        out = np.zeros(H // 8 * (K0 + K1) * 256, dtype=np.float32)
        assert lib.fsnp_debug_lstm_fbv_pack(H, NIN, vp(wih0), vp(whh0), vp(wih1), vp(whh1), vp(out), out.size) == 0, lib.fsnp_last_error()
        assert lib.fsnp_debug_lstm_fbv_pack(H, NIN, vp(wih0), vp(whh0), vp(wih1), vp(whh1), vp(out), out.size - 4) == 2
        pack = out.reshape(H // 8, (K0 + K1) // 4, 256, 4).astype(np.float64)
        x, h0, h1 = (rng.standard_normal(n).astype(np.float32).astype(np.float64) for n in (NIN, H, H))
        V = np.concatenate([x, np.zeros(XP - NIN), h0, h1])                    # the kernel's LDS row: [x | h0 | h1]
        want0 = wih0.astype(np.float64) @ x + whh0.astype(np.float64) @ h0
        want1 = wih1.astype(np.float64) @ h0 + whh1.astype(np.float64) @ h1
        for cs in (0, 13, H // 8 - 1):
            w = pack[cs].transpose(1, 0, 2).reshape(256, K0 + K1)              # [tid][j]
            for c in range(32):
                a0 = sum(w[ks * 32 + c, :K0] @ V[K0 * ks:K0 * ks + K0] for ks in range(8))
                a1 = sum(w[ks * 32 + c, K0:] @ V[XP + K1 * ks:XP + K1 * ks + K1] for ks in range(8))
                row = (c & 3) * H + 8 * cs + (c >> 2)
                assert abs(a0 - want0[row]) < 1e-9 * (1 + abs(a0)) and abs(a1 - want1[row]) < 1e-9 * (1 + abs(a1))
    assert lib.fsnp_debug_lstm_fbv_pack(384, 257, vp(wih0), vp(whh0), vp(wih1), vp(whh1), vp(out), out.size) == 2


def test_wave_owned_column_split_index_maps_and_asm():
    """csrc/lstm_coopw.hip restated on the CPU.  (1) The packed stream: participant `part` of an NT-tile split multiplies 8-unit
    blocks part * NT + n of every k-group; column c of a block = gate c & 3 of unit c >> 2; the B operand of MFMA p of k-group g in
    lane l is k = 8 g' + 2 p + (l >> 5) - emulating the kernel's indexing on the packer's output reproduces W x for both layers, for
    both split widths, K = 40 and 64.  (2) Lane / slot identities: lane (row, hi)'s features hi + 2 i are the components of ITS A
    fragment of x k-group i >> 2, and its cells (units hi + 2 j of block ub) are the components of ITS float4 of k-group ub of the h
    image (a_frag_index).  (3) Round 6, transposed product: accumulator register q of lane (sequence, hi) is gate q & 3 of unit hi + 2 (q >> 2)
    - the cells are lane-local straight out of the accumulators (the round-5 staging tile through LDS is gone); the bias table's index math.
    (4) The layer-skewed schedule with per-wave participants is the one test_exchange_schedules_are_race_free_under_any_interleaving
    model-checks (three h0 images, two h1 images).  (5) Static asm: clean k-group loops, no scratch, no workgroup barrier in the time
    loop, no cache maintenance, 16-byte sc1 stores."""
    import ctypes as ct
    import importlib.util
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(5))
    H = 384
    for NIN, KX in ((34, 40), (52, 64)):
        wih0, whh0 = rng.standard_normal((4 * H, NIN)).astype(np.float32), rng.standard_normal((4 * H, H)).astype(np.float32)
        wih1, whh1 = rng.standard_normal((4 * H, H)).astype(np.float32), rng.standard_normal((4 * H, H)).astype(np.float32)
        KGX, KGH = KX // 8, H // 8
        KG0, NUB = KGX + KGH, H // 8
        out = np.zeros((KG0 + 2 * KGH) * NUB * 256, dtype=np.float32)
        vp = lambda a: a.ctypes.data_as(ct.c_void_p)
        assert lib.fsnp_debug_lstm_coopw_pack(H, NIN, KX, vp(wih0), vp(whh0), vp(wih1), vp(whh1), vp(out), out.size) == 0, lib.fsnp_last_error()
        pack = out.reshape(KG0 + 2 * KGH, NUB, 64, 4)
        x, h0, h1 = rng.standard_normal(NIN).astype(np.float32), rng.standard_normal(H).astype(np.float32), rng.standard_normal(H).astype(np.float32)
        k0 = np.concatenate([x, np.zeros(KX - NIN, np.float32), h0]).astype(np.float64)       # layer 0: [x (padded) | h0]
        k1 = np.concatenate([h1, h0]).astype(np.float64)                                      # layer 1: [h1 | h0]
        want0 = wih0.astype(np.float64) @ x + whh0.astype(np.float64) @ h0
        want1 = whh1.astype(np.float64) @ h1 + wih1.astype(np.float64) @ h0
        lane = np.arange(64)
        for NT in (1, 2):
            for part in (0, 5, H // (8 * NT) - 1):
                for n in range(NT):
                    ub = part * NT + n
                    for c in (0, 1, 2, 3, 17, 31):
                        lanes = lane[(lane & 31) == c]                                        # the two k halves of column c
                        a0 = sum(float(pack[g, ub, l, p]) * k0[8 * g + 2 * p + (l >> 5)] for g in range(KG0) for l in lanes for p in range(4))
                        a1 = sum(float(pack[KG0 + g, ub, l, p]) * k1[8 * g + 2 * p + (l >> 5)] for g in range(2 * KGH) for l in lanes for p in range(4))
                        gate, unit = c & 3, 8 * ub + (c >> 2)
                        assert abs(a0 - want0[gate * H + unit]) < 1e-9 * (1 + abs(a0)) and abs(a1 - want1[gate * H + unit]) < 1e-9 * (1 + abs(a1))
    a_frag = lambda row, k: (((k >> 3) * 64) + ((k & 1) * 32) + row) * 4 + ((k >> 1) & 3)      # lstm_common.h: a_frag_index
    for l in range(64):
        row, hi = l & 31, l >> 5
        for i in range(32):                                                                   # x: feature hi + 2 i <-> component i & 3 of float4 (i >> 2) * 64 + l
            assert a_frag(row, hi + 2 * i) == ((i >> 2) * 64 + l) * 4 + (i & 3)
        for ub in (0, 7, 47):
            for j in range(4):                                                                # h: unit 8 ub + hi + 2 j <-> component j of float4 ub * 64 + l
                assert a_frag(row, 8 * ub + hi + 2 * j) == (ub * 64 + l) * 4 + j
    # (3) round 6: TRANSPOSED product.  D = W-fragment (A operand, M = the tile's 32 gate-interleaved columns) x h (B operand, N = 32 sequences):
    # accumulator register q of lane l is D[m = (q & 3) + 8 (q >> 2) + 4 (l >> 5)][n = l & 31] (the 32x32 C/D layout); with column m = gate
    # (m & 3) of unit (m >> 2) that is gate q & 3 of unit hi + 2 (q >> 2) for sequence l & 31: the four gates of each of the lane's four
    # cells sit in consecutive registers, every (sequence, unit, gate) exactly once per wave - no staging tile.  The bias table
    # [layer][n][hi][16] read as float4 u holds gates 0..3 of unit hi + 2 u.
    for NT in (1, 2):
        seen = set()
        for l in range(64):
            seq, hi = l & 31, l >> 5
            for q in range(16):
                m = (q & 3) + 8 * (q >> 2) + 4 * hi
                gate, unit = m & 3, m >> 2
                assert gate == q & 3 and unit == hi + 2 * (q >> 2)
                seen.add((seq, unit, gate))
        assert len(seen) == 32 * 8 * 4
        for i in range(2 * NT * 2 * 16):                                                      # the fill loop's decode of float index i
            q, bh, n, layer = i & 15, (i >> 4) & 1, (i >> 5) % NT, i // (32 * NT)
            assert i == ((layer * NT + n) * 2 + bh) * 16 + q
            u, g = q >> 2, q & 3                                                              # float4 u of lane hi = bh, component g
            assert (((layer * NT + n) * 2 + bh) * 4 + u) * 4 + g == i
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.analyse_wave_owned()
    assert len(res) == 6, sorted(res)                                                         # NT in {1, 2, 3} x K in {40, 64}
    for name, r in res.items():
        nt = int(re.search(r"ILi384ELi\d+ELi(\d)E", name).group(1))
        depth = 8 if nt == 1 else 4
        assert r["scratch"] == 0 and r["cache_maint"] == 0 and r["barriers_after_first_mfma"] == 0, (name, r)
        assert r["sc1_stores16"] >= 2 * nt and r["sc1_loads"] > 0 and len(r["loops"]) >= 3, (name, r)
        for lp in r["loops"]:
            assert lp["mfma"] == depth * nt * 4 and lp["loads"] == depth * (nt + 1), (name, lp)
            assert lp["scratch"] == 0 and lp["drain"] == 0 and lp["acc_moves"] == 0, (name, lp)


def test_round3_kernels_keep_their_asm_invariants():
    """Static checks of the round-3 kernels (tools/check_lstm_asm.py).  lstm2_coop_hp_kernel: exactly the
    16x16x4 MFMAs of two unrolled half-phases, layer-1 weights from AGPRs, operands by LDS DMA, 16-byte sc1 stores, DPP row sums, no
    scratch.  lstm2_generic_kernel: plain FMAs, no MFMA, no scratch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lstm_asm", os.path.join(ROOT, "tools", "check_lstm_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hp = mod.analyse_half_tile_ping_pong()
    assert len(hp) == 4, sorted(hp)                           # H 256 / 384 x K 40 / 64
    for name, r in hp.items():
        h, kx = (int(v) for v in re.search(r"ILi(\d+)ELi(\d+)EEEv", name).groups())
        per_pass = 4 * ((kx + 15) // 16) + 12 * (h // 16)      # MFMAs of one half-phase; the time loop is unrolled for the two halves
        assert r["scratch"] == 0 and r["cache_maint"] == 0 and r["mfma_other"] == 0 and r["bpermute"] == 0, (name, r)
        assert r["mfma"] == 2 * per_pass + 2 * 4 * ((kx + 15) // 16), (name, r)
        assert r["mfma_b_in_agpr"] >= 0.55 * r["mfma"], (name, r)      # layer 1 = two thirds of the weights, pinned to AGPRs
        assert r["lds_dma"] >= 2 and r["sc1_stores16"] >= 7 and r["dpp_ror"] >= 8, (name, r)
    gen = mod.analyse_generic()
    assert len(gen) == 8, sorted(gen)                         # 1 / 2 / 4 / 8 sequences per workgroup x {sub-band, full-band}
    for name, r in gen.items():
        assert r["mfma"] == 0 and r["scratch"] == 0 and r["fma"] > 0 and r["ds_read128"] > 0, (name, r)


# ---------------------------------------------------------------- sub-band planner (fsnp_abi.hip plan_sb), host only
def _plan(rows, cus=256, gru=0, coop=1, gain=0.97):
    import ctypes as ct
    lib = _lib.load()
    buf = (ct.c_int32 * (8 * 64))()
    n = lib.fsnp_debug_plan_rows(rows, cus, 384, gru, coop, gain, buf, 64)
    assert n > 0, lib.fsnp_last_error()
    keys = ("kind", "row0", "rows", "tiles", "ex", "par", "rpg", "slot0")
    return [dict(zip(keys, buf[8 * i:8 * i + 8])) for i in range(n)]


@pytest.mark.parametrize("gru", [0, 1])
@pytest.mark.parametrize("rows", [1, 31, 257, 514, 1285, 1344, 1345, 1542, 2056, 4112, 5440, 5441, 5654, 8192, 8224, 8481,
                                  9252, 10280, 12336, 16448, 24672, 65792])
def test_subband_plans_cover_every_sequence_and_fit_the_chip(rows, gru):
    """Every plan: chunks are consecutive, cover [0, rows), their slots do not overlap, every tile holds <= 32 + ex
    sequences, and the column-split chunks are co-resident by construction (workgroups <= CUs)."""
    chunks = _plan(rows, gru=gru)
    nxt, slot = 0, 0
    for c in chunks:
        assert c["row0"] == nxt and c["slot0"] == slot and c["rows"] > 0
        nxt += c["rows"]
        per_tile = 16 if c["kind"] == 4 else 32 + c["ex"]             # kind 4 = half tiles (csrc/lstm16.hip)
        slot += c["tiles"] * per_tile
        assert c["tiles"] * per_tile >= c["rows"]
        if c["kind"] == 4:
            assert not gru and c["ex"] == 0
        elif c["kind"] == 1:
            assert c["par"] in (8, 16, 32, 64) and c["tiles"] * (384 // c["par"]) <= 256
        elif c["kind"] == 2:
            assert c["rpg"] in (1, 2) and c["par"] * 3 <= 256 and c["par"] * c["rpg"] >= c["tiles"]
        elif c["kind"] == 9:
            assert not gru and c["par"] in (32, 64, 96) and c["tiles"] * (384 // c["par"]) <= 256   # wave-owned column split (lstm_coopw.hip)
        else:
            assert not gru, "there is no row-tile GRU kernel"
    assert nxt == rows


def test_subband_plan_choices_match_the_design():
    """The cuts DESIGN.md 4.1 / 4.1b describe, for the batch sizes of BASELINE.json and of the tables in profiles/ (built-in
    cost table = the round-1 measurements, one workgroup per CU; the shortest-path planner may move tiles between chunks of
    equal total cost, so multi-chunk cuts are checked by kernel sequence, coverage and capacity)."""
    kinds = lambda rows, **kw: [(c["kind"], c["rows"]) for c in _plan(rows, **kw)]
    seq = lambda rows, **kw: [c["kind"] for c in _plan(rows, **kw)]
    assert kinds(257) == [(8, 257)] and _plan(257)[0]["tiles"] == 9                 # B = 1: 9 row tiles on the half-tile ping-pong kernel (lstm_hp.hip)
    assert kinds(257, gru=1) == [(1, 257)] and _plan(257, gru=1)[0]["par"] == 16    # (GRU: K split, 16 units)
    assert kinds(160) == [(1, 160)] and _plan(160)[0]["par"] == 8 and kinds(192) == [(8, 192)]   # up to 5 row tiles: K split at 8 units; 6 ... 10: lstm_hp
    assert kinds(514) == [(9, 514)] and _plan(514)[0]["par"] == 32 and _plan(514)[0]["tiles"] == 17     # B = 2: wave-owned split, 12 workgroups per tile
    assert kinds(1285) == [(9, 1285)] and _plan(1285)[0]["par"] == 64               # B = 5: 41 tiles, 6 workgroups per tile
    assert kinds(1285, gru=1) == [(1, 1285)] and _plan(1285, gru=1)[0]["par"] == 64   # (GRU: the K split, as in round 4)
    p = _plan(2056)                                                                 # B = 8 (round 6): 64 of the 65 tiles in ONE launch at 96 units per
    assert [(c["kind"], c["par"], c["tiles"]) for c in p] == [(9, 96, 64), (1, 8, 1)] and sum(c["rows"] for c in p) == 2056   # workgroup + 1 on the K split (round 5: 42 + 21 + 2)
    assert kinds(2056, gru=1) == [(2, 2056)] and _plan(2056, gru=1)[0]["rpg"] == 1  # (GRU: three-way split)
    assert kinds(4096) == [(4, 4096)] and _plan(4096)[0]["tiles"] == 256            # parity-mode B = 32: one round of 256 half tiles
    assert kinds(4112) == [(4, 4096), (1, 16)]                                      # B = 16: a half-tile round + 16 sequences K split
    p = _plan(4096, gru=1)                                                          # GRU has no half-tile kernel: 128 tiles = one per
    assert [c["kind"] for c in p] == [2, 1, 1] and p[0]["rpg"] == 1 and p[1]["par"] == 64 and p[2]["par"] == 8   # group + 42 + 1
    assert sum(c["rows"] for c in p) == 4096 and p[0]["tiles"] <= 85 and p[1]["tiles"] == 42 and p[2]["tiles"] <= 5
    assert seq(3300) == [9, 9] and kinds(3500) == [(4, 3500)] and kinds(3855) == [(4, 3855)]   # the half tiles pay from ~106 row tiles up (104 = 62 + 42 wave-owned: 95 us against 103)
    assert seq(2800) == [9, 9, 1] and kinds(2800, gru=1)[0][0] == 2
    assert seq(4256) == [4, 1]                                                      # 133 tiles: half-tile round + 5 tiles K split
    assert kinds(5397) == [(4, 4096), (9, 1301)] and _plan(5397)[1]["par"] == 64    # B = 21, 169 tiles: half-tile round + 41 tiles wave-owned split
    assert kinds(5397, gru=1) == [(2, 5397)] and _plan(5397, gru=1)[0]["rpg"] == 2  # (108 + 48.5 us against 157 for two per group); GRU: two per group
    assert kinds(8224) == [(0, 8192), (1, 32)] and _plan(8224)[1]["par"] == 8       # B = 32: full round + leftover tile
    assert _plan(8224)[1]["rpg"] == 0 and _plan(16448)[1]["rpg"] == 0 and _plan(16448)[1]["tiles"] == 2   # (the role-split schedule was removed in round 4)
    assert seq(10280) == [0, 9, 1] and kinds(10280)[0] == (0, 8192) and _plan(10280)[1]["par"] == 96   # B = 40: a full round + 64 + 2 tiles
    assert kinds(16448) == [(0, 16384), (1, 64)]                                    # B = 64: two rounds + 2 tiles
    assert kinds(7967) == [(0, 7967)] and _plan(7967)[0]["ex"] == 0                 # B = 31: 249 tiles, one launch
    assert kinds(8224, coop=0) == [(0, 8224)] and _plan(8224, coop=0)[0]["ex"] == 1  # column-split kernels off: VALU rows
    assert kinds(8224, gain=0.0) == [(0, 8224)]                                     # composite plans off
    g = _plan(65792, gru=1)                                                         # GRU, B = 256: 170-tile launches (two per group)
    assert [c["kind"] for c in g[:12]] == [2] * 12 and all(c["rpg"] == 2 and c["tiles"] <= 170 for c in g[:12])
    assert all(c["tiles"] == 170 for c in g[:11])
    assert sum(c["rows"] for c in g) == 65792 and all(c["kind"] == 1 for c in g[12:])   # the short rest K split
    g = _plan(8224, gru=1)                                                          # GRU, B = 32: 257 tiles = 170 + 85 + 2
    assert [c["kind"] for c in g] == [2, 2, 1] and g[0]["rpg"] == 2 and g[1]["rpg"] == 1 and sum(c["rows"] for c in g) == 8224
    # B = 12: 97 tiles = 55 at 96 units per workgroup + 42 at 64 on the wave-owned split (round 6: 12.4 ms per forward; round 5: three
    # launches, 13.5); half tiles from ~103 tiles
    assert seq(3084) == [9, 9] and [c["par"] for c in _plan(3084)] == [96, 64] and kinds(3400) == [(4, 3400)]
    p = _plan(3084, gru=1)                                                          # (GRU: 97 tiles = one per group + the rest K split)
    assert p[0]["kind"] == 2 and p[0]["rpg"] == 1 and p[0]["tiles"] == 85 and all(c["kind"] == 1 for c in p[1:]) and sum(c["rows"] for c in p) == 3084
    assert seq(4112, gru=1) == [2, 1, 1]                                            # GRU B = 16: 129 tiles = 85 + 42 + 2
    p = _plan(1376)                                                                 # 43 tiles: a full wave-owned launch + one tile on the K split
    assert [(c["kind"], c["par"], c["tiles"]) for c in p] == [(9, 64, 42), (1, 8, 1)] and sum(c["rows"] for c in p) == 1376
    p = _plan(1376, gru=1)                                                          # (GRU: a full K-split launch + a tiny one)
    assert [c["kind"] for c in p] == [1, 1] and p[0]["par"] == 64 and p[1]["par"] == 8 and sum(c["rows"] for c in p) == 1376


def test_planner_follows_the_cost_table_and_two_workgroups_per_cu():
    """fsnp_debug_plan_rows2: the planner minimises whatever cost table it is given (on the device: the calibrated one).  With
    two column-split workgroups allowed per CU and a table in which that pays (each CU then overlaps the hand-off stalls of
    two independent row tiles), B = 1 runs at 8 units per workgroup (432 workgroups), B = 16 at one row tile per group on
    129 groups; with the built-in table (two per CU priced prohibitively) nothing changes; capacity is never exceeded."""
    import ctypes as ct
    lib = _lib.load()

    def plan(rows, occ, costs=None, gru=0):
        buf = (ct.c_int32 * (8 * 64))()
        nc = _lib.NUM_COSTS
        arr = (ct.c_double * nc)(*(list(costs) + [1e9] * (nc - len(costs)))) if costs else None      # (launch shapes the table does not name are priced out)
        n = lib.fsnp_debug_plan_rows2(rows, 256, 384, gru, 1, 0.97, occ, arr, buf, 64)
        assert n > 0, lib.fsnp_last_error()
        keys = ("kind", "row0", "rows", "tiles", "ex", "par", "rpg", "slot0")
        return [dict(zip(keys, buf[8 * i:8 * i + 8])) for i in range(n)]

    cheap2 = [9, 12, 19, 25, 29, 38, 55, 70, 76, 95, 151, 190, 208, 0.11, 9, 19, 29, 55, 1000]
    p = plan(257, 2, cheap2)
    assert len(p) == 1 and p[0]["kind"] == 1 and p[0]["par"] == 8 and 9 * 48 <= 512
    p = plan(4112, 2, cheap2)                    # 129 tiles
    assert len(p) == 1 and p[0]["kind"] == 2 and p[0]["rpg"] == 1 and p[0]["par"] == 129
    assert [(c["kind"], c["rows"]) for c in plan(257, 2)] == [(c["kind"], c["rows"]) for c in plan(257, 1)]
    assert [(c["kind"], c["rows"]) for c in plan(4112, 2)] == [(c["kind"], c["rows"]) for c in plan(4112, 1)]
    for rows in (257, 514, 1285, 2056, 4096, 4112, 5440, 8224, 10280, 65792):
        for gru in (0, 1):
            for c in plan(rows, 2, cheap2, gru):
                wgs = c["tiles"] * (384 // c["par"]) if c["kind"] in (1, 9) else c["par"] * 3 if c["kind"] == 2 else 0
                assert wgs <= 512 and (c["kind"] != 2 or c["par"] * c["rpg"] >= c["tiles"])
    slow_k = [100, 100, 100, 100, 100, 100, 100, 100, 76, 95, 151, 190, 208, 0.11, 100, 100, 100, 100, 1000]      # K split suddenly slow: 9 tiles move
    p = plan(257, 1, slow_k)
    assert p[0]["kind"] == 2
    # the half-tile ping-pong kernel (kernel 8: 24 workgroups per row tile, at most 10 row tiles per launch)
    cheap_hp = [100] * 8 + [760, 950, 1510, 1900, 208, 0.11, 100, 100, 100, 100, 1000, 9, 11]
    p = plan(257, 1, cheap_hp)
    assert len(p) == 1 and p[0]["kind"] == 8 and p[0]["tiles"] == 9 and p[0]["rows"] == 257
    p = plan(32, 1, cheap_hp)
    assert len(p) == 1 and p[0]["kind"] == 8 and p[0]["tiles"] == 1
    p = plan(352, 1, cheap_hp)                   # 11 tiles: two launches, each within 10 tiles
    assert [c["kind"] for c in p] == [8, 8] and sorted(c["tiles"] for c in p) == [1, 10] and sum(c["rows"] for c in p) == 352
    p = plan(8224, 1, cheap_hp)                  # B = 32: the leftover tile
    assert [(c["kind"], c["rows"]) for c in p] == [(0, 8192), (8, 32)]
    assert [c["kind"] for c in plan(257, 1)] == [8] and all(c["kind"] != 8 for rows in (32, 160, 640, 8224) for c in plan(rows, 1))   # built-in table: B = 1
    # the wave-owned column split (kernel 9: 12 / 6 workgroups per row tile at 32 / 64 units, at most 21 / 42 row tiles per launch)
    cheap_w = [100] * 8 + [760, 950, 1510, 1900, 208, 0.11, 100, 100, 100, 100, 1000, 1e9, 1e9, 11, 19, 10, 18]
    p = plan(514, 1, cheap_w)
    assert len(p) == 1 and p[0]["kind"] == 9 and p[0]["par"] == 32 and p[0]["tiles"] == 17
    p = plan(1344, 1, cheap_w)                   # 42 tiles: one launch at 64 units (19) beats two at 32 (22)
    assert [(c["kind"], c["par"], c["tiles"]) for c in p] == [(9, 64, 42)]
    p = plan(2056, 1, cheap_w)                   # 65 tiles: never more than 21 / 42 tiles per launch
    assert all(c["kind"] == 9 and c["tiles"] <= (21 if c["par"] == 32 else 42) for c in p) and sum(c["rows"] for c in p) == 2056
    assert all(c["kind"] != 9 for rows in (514, 1285, 2056) for c in plan(rows, 1, cheap_w, gru=1))          # LSTM only


def test_oracle_is_only_reachable_from_the_allowed_places():
    """The oracle is test infrastructure: the product package never imports it, bench.py only inside its cpu_baseline
    leg, __graft_entry__ only inside build() (import check of the checker) and smoke()."""
    import ast

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        found = []
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            for node in ast.iter_child_nodes(fn) if isinstance(fn, ast.Module) else ast.walk(fn):
                mod = node.module if isinstance(node, ast.ImportFrom) else None
                names = [a.name for a in node.names] if isinstance(node, ast.Import) else []
                if (mod and mod.split(".")[0] == "oracle") or any(n.split(".")[0] == "oracle" for n in names):
                    found.append(fn.name if isinstance(fn, ast.FunctionDef) else "<module>")
        return set(found)

    pkg = os.path.join(ROOT, "fullsubnet_plus_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert oracle_imports(os.path.join(pkg, f)) == set(), f
    assert oracle_imports(os.path.join(ROOT, "bench.py")) <= {"cpu_baseline"}
    # build() may BUILD the checker (here: import-check the pure-Python restatements), smoke() uses it
    assert oracle_imports(os.path.join(ROOT, "__graft_entry__.py")) <= {"build", "smoke"}


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """include/fsnp.h compiles as C99 with -Wall -Werror -pedantic, libfsnp_hip.so links from a plain C program, and the
    error paths reachable without a GPU return codes + messages (tests/c_abi/abi_check.c)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    _lib.load()
    from fullsubnet_plus_amd import _build
    exe = tmp_path / "abi_check"
    libdir = os.path.dirname(_build.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", str(exe), "-L", libdir, "-lfsnp_hip",
           "-Wl,-rpath," + libdir]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "gfx950" in run.stdout and "bad config rc=2" in run.stdout and "plan chunks=2" in run.stdout


def test_synthetic_generators_are_frozen():
    """The seeded weight / waveform generators (fullsubnet_plus_amd/synthetic.py, numpy PCG64) are what the golden fixtures
    were generated with: known-answer checksums, so a change of generator or numpy stream shows up here and not as a
    mysterious parity failure."""
    from fullsubnet_plus_amd.synthetic import make_wave
    f64 = lambda a: float(np.asarray(a, dtype=np.float64).sum())
    sd = make_state_dict(0, "default", as_torch=False)
    assert abs(f64(sd["sb_model.sequence_model.weight_hh_l0"]) - 14.657252892301898) < 1e-6
    assert abs(f64(sd["fb_model.sequence_model.3.sconv.weight"]) - 1.1025237809649013) < 1e-6
    assert abs(f64(make_state_dict(4, "harsh", as_torch=False)["sb_model.sequence_model.weight_ih_l1"]) - 19.92727242918871) < 1e-5
    assert abs(f64(make_state_dict_fullsubnet(3, "harsh", as_torch=False)["fb_model.sequence_model.weight_hh_l1"]) + 15.92294563049542) < 1e-5
    assert abs(f64(make_wave(2, 0.5, 7)) + 3.7271236432115984) < 1e-6


def test_every_environment_switch_is_documented_and_reported():
    """The FSNP_* variables the library reads (getenv in csrc/) are exactly the rows of INTEGRATION.md's table, and fsnp_dump_config names each
    one with the value in force - a switch cannot be added, or removed with what it selected, without its documentation."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fullsubnet_plus_amd", "csrc")
    read = set()
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, name)) as f:
                read |= set(re.findall(r'getenv\("(FSNP_[A-Z0-9_]+)"\)', f.read()))
    with open(os.path.join(root, "INTEGRATION.md")) as f:
        text = f.read()
    table = set(re.findall(r"^\| `(FSNP_[A-Z0-9_]+)=", text, flags=re.M))
    assert read == table, (sorted(read - table), sorted(table - read))
    assert {3: "Three", 10: "Ten", 11: "Eleven", 12: "Twelve"}.get(len(read), str(len(read))) + " switches" in text
    with open(os.path.join(csrc, "fsnp_abi.hip")) as f:
        abi = f.read()
    dump = abi[abi.index("int64_t fsnp_dump_config("):]
    for var in read:
        assert f'{var}=%s' in dump, var
