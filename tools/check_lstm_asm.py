#!/usr/bin/env python3
"""Static check of the fused LSTM kernel's hot loops in the gfx950 assembly hipcc emits.

For every instantiation of lstm2_fc_kernel: each depth-2 loop (the k-group loops) must contain MFMAs,
no scratch access and no full `s_waitcnt vmcnt(0)` drain, i.e. the refill-in-place weight pipeline survived
the compiler.  Used by tests/test_host.py (CPU, hipcc cross-compiles) and by hand while tuning."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "lstm.hip")


def analyse(flags=("-fno-slp-vectorize",)):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "lstm.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", SRC, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp15lstm2_fc_kernelI\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        tag = re.search(r"Li(\d+)ELi(\d+)ELi2ELi(\d)ELb(\d)ELi(\d+)ELb(\d)E", name)
        key = (f"H{tag.group(1)}_" if tag.group(1) != "384" else "") + \
            f"KX{tag.group(2)}_EX{tag.group(3)}_PROF{tag.group(4)}_NW{tag.group(5)}_BF{tag.group(6)}"
        loops = []
        for i, l in enumerate(body):
            if "Inner Loop Header: Depth=2" not in l:
                continue
            lab = None
            for k in range(i, max(i - 4, 0), -1):
                mm = re.match(r"^(\.LBB\d+_\d+):", body[k])
                if mm:
                    lab = mm.group(1)
                    break
            end = next(k for k in range(i, len(body)) if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", body[k]))
            seg = body[i:end]
            cnt = lambda pat: sum(1 for x in seg if re.search(pat, x))
            loops.append(dict(mfma=cnt(r"v_mfma"), scratch=cnt(r"scratch_"), drain=cnt(r"vmcnt\(0\)"),
                              gload=cnt(r"global_load_dwordx4|buffer_load_dwordx4"), valu=cnt(r"\bv_fma|\bv_fmac"), lines=len(seg)))
        res[key] = [l for l in loops if l["mfma"] > 0]
    return res


def analyse_column_split(flags=("-fno-slp-vectorize",)):
    """Whole-kernel invariants of the column-split kernels (lstm_coop.hip, lstm_coopn.hip), per instantiation:
    scratch accesses from the first MFMA on (i.e. inside the time loop), flat accesses (reported only), cache-maintenance
    instructions (the write-through hand-off needs none), sc1 loads / stores of the exchange images, MFMAs."""
    res = {}
    for fn, sym in (("lstm_coop.hip", "_ZN4fsnp17lstm2_coop_kernelI"), ("lstm_coopn.hip", "_ZN4fsnp18lstm2_coopn_kernelI")):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", fn)
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                           check=True, capture_output=True)
            text = open(out).read()
        for m in re.finditer(r"^(" + sym + r"\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
            name, body = m.group(1), m.group(2).split("\n")
            first_mfma = next(i for i, l in enumerate(body) if "v_mfma" in l)
            inloop = body[first_mfma:]                                                            # the time loop and after
            cnt = lambda seg, pat: sum(1 for x in seg if re.search(pat, x))
            res[name] = dict(mfma=cnt(body, r"v_mfma"), scratch_in_loop=cnt(inloop, r"scratch_"), flat=cnt(body, r"flat_load|flat_store"),
                             cache_maint=cnt(body, r"buffer_wbl2|buffer_inv"), sc1_loads=cnt(body, r"buffer_load_dwordx4.*sc1"),
                             sc1_stores=cnt(body, r"global_store_dword\b.*sc1|buffer_store_dword\b.*sc1"))
    return res


def analyse_half_tile_ping_pong(flags=("-fno-slp-vectorize",)):
    """Whole-kernel invariants of lstm2_coop_hp_kernel (lstm_hp.hip), per instantiation: 16x16x4 MFMAs (and how many take their B
    operand - a resident weight - from an AGPR), scratch, cache maintenance, LDS-DMA operand loads (sc1), 16-byte sc1 stores of the
    exchange images, DPP row rotations (the Linear partial sums), ds_bpermute (must be none)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "lstm_hp.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp20lstm2_coop_hp_kernelI\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        cnt = lambda pat: sum(1 for x in body if re.search(pat, x))
        res[name] = dict(mfma=cnt(r"v_mfma_f32_16x16x4_f32"), mfma_other=cnt(r"v_mfma") - cnt(r"v_mfma_f32_16x16x4_f32"),
                         mfma_b_in_agpr=cnt(r"v_mfma_f32_16x16x4_f32 [av]\[\d+:\d+\], v\d+, a\d+, "), scratch=cnt(r"scratch_"),
                         cache_maint=cnt(r"buffer_wbl2|buffer_inv"), lds_dma=cnt(r"buffer_load_dwordx4.*sc1.*lds|buffer_load_dwordx4.*lds.*sc1"),
                         sc1_stores16=cnt(r"buffer_store_dwordx4.*sc1"), dpp_ror=cnt(r"row_ror"), bpermute=cnt(r"ds_bpermute"))
    return res


def analyse_wave_owned(flags=("-fno-slp-vectorize",)):
    """lstm2_coopw_kernel (lstm_coopw.hip), per instantiation: the k-group loops (depth-2 loops with MFMAs: NT x 4 MFMAs and NT + 1
    16-byte loads per k-group, D groups per iteration, counted waits only - no drain, no scratch, no accumulator moves), and
    whole-kernel facts: no scratch anywhere, no cache maintenance, no workgroup barrier behind the first MFMA, 16-byte sc1 stores."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "lstm_coopw.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp18lstm2_coopw_kernelI\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        cnt = lambda seg, pat: sum(1 for x in seg if re.search(pat, x))
        labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = body[labels[mm.group(1)]:i]
                if 0 < cnt(seg, r"v_mfma") <= 64 and len(seg) < 200:
                    loops.append(dict(mfma=cnt(seg, r"v_mfma"), loads=cnt(seg, r"buffer_load_dwordx4"), scratch=cnt(seg, r"scratch_"),
                                      drain=cnt(seg, r"vmcnt\(0\)"), acc_moves=cnt(seg, r"v_accvgpr"), lines=len(seg)))
        first = next(i for i, l in enumerate(body) if "v_mfma" in l)
        res[name] = dict(loops=loops, scratch=cnt(body, r"scratch_"), cache_maint=cnt(body, r"buffer_wbl2|buffer_inv"),
                         barriers_after_first_mfma=cnt(body[first:], r"s_barrier"), sc1_stores16=cnt(body, r"buffer_store_dwordx4.*sc1"),
                         sc1_loads=cnt(body, r"buffer_load_dwordx4.*sc1"), mfma=cnt(body, r"v_mfma_f32_32x32x2_f32"))
    return res


def analyse_splitk_gemm(flags=("-fno-slp-vectorize",)):
    """tcn_gemm_sk_kernel (tcn.hip), per instantiation: one k-tile body per wave (16 MFMAs, 6 fragment reads, 6 DMA pieces + the 6 of the
    first tile), no scratch, and NO workgroup barrier between the first and the last MFMA (a wave waits for its own DMA only)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "tcn.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp18tcn_gemm_sk_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), [l for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        cnt = lambda seg, pat: sum(1 for x in seg if re.search(pat, x))
        idx = [i for i, l in enumerate(body) if "v_mfma" in l]
        res[name] = dict(mfma=cnt(body, r"v_mfma"), dma=cnt(body, r"buffer_load_dwordx4 .* lds"), scratch=cnt(body, r"scratch_"),
                         barriers_between_mfmas=cnt(body[idx[0]:idx[-1] + 1], r"s_barrier"))
    return res


def analyse_generic(flags=("-fno-slp-vectorize",)):
    """lstm2_generic_kernel (lstm_generic.hip): fp32 FMAs only - no MFMA - and no scratch in any instantiation."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "lstm_generic.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp20lstm2_generic_kernelI\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        cnt = lambda pat: sum(1 for x in body if re.search(pat, x))
        res[name] = dict(mfma=cnt(r"v_mfma"), fma=cnt(r"\bv_fma_f32|\bv_fmac_f32|\bv_pk_fma_f32"), scratch=cnt(r"scratch_"), ds_read128=cnt(r"ds_read_b128"))
    return res


def analyse_half_tile(flags=("-fno-slp-vectorize",)):
    """k-group loops of lstm2_fc16_kernel (lstm16.hip): MFMAs, weight loads, scratch, drains and AGPR<->VGPR moves per loop
    (hipcc once shuttled the 24 accumulator tiles through VGPRs inside one of the loops: 124 moves per k-group)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "lstm16.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp17lstm2_fc16_kernelI\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        loops = []
        for i, l in enumerate(body):
            if "Inner Loop Header: Depth=2" not in l:
                continue
            lab = None
            for k in range(i, max(i - 4, 0), -1):
                mm = re.match(r"^(\.LBB\d+_\d+):", body[k])
                if mm:
                    lab = mm.group(1)
                    break
            end = next(k for k in range(i, len(body)) if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", body[k]))
            seg = body[i:end]
            cnt = lambda pat: sum(1 for x in seg if re.search(pat, x))
            loops.append(dict(mfma=cnt(r"v_mfma"), scratch=cnt(r"scratch_"), drain=cnt(r"vmcnt\(0\)"),
                              gload=cnt(r"buffer_load_dwordx4"), acc_moves=cnt(r"v_accvgpr"), lines=len(seg)))
        res[name] = [l for l in loops if l["mfma"] > 0]
    return res


def analyse_dma_gemm(flags=("-fno-slp-vectorize",)):
    """k-loop of tcn_gemm_dma_kernel (tcn.hip), per instantiation: per k-tile 16 MFMAs, 6 ds_read_b128, 3 LDS-DMA loads and two
    offset selects - no ds_write, no scratch, no AGPR<->VGPR moves (counts over the span between the first and the last MFMA)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "tcn.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp19tcn_gemm_dma_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), [l for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        idx = [i for i, l in enumerate(body) if "v_mfma" in l]            # the k-loop = everything between the first and the last MFMA
        seg = body[idx[0] - 12:idx[-1] + 1]                               # (+ the fragment reads and DMA issue in front of the first)
        cnt = lambda pat: sum(1 for x in seg if re.search(pat, x))
        res[name] = dict(mfma=cnt(r"v_mfma"), ds_read=cnt(r"ds_read_b128"), dma=cnt(r"buffer_load_dwordx4 .* lds"), scratch=cnt(r"scratch_"),
                         acc_moves=cnt(r"v_accvgpr"), valu=cnt(r"^\s+v_(?!mfma)"), ds_write=cnt(r"ds_write"))
        # the epilogue = everything behind the last MFMA: 16-byte global traffic only, at most two full vmcnt drains (round 4: the
        # accumulator tile is transposed through LDS so that a lane owns float4s along a row; before, the sconv epilogue was 32
        # dependent 4-byte load -> add -> store round trips per lane)
        epi = body[idx[-1] + 1:]
        ecnt = lambda pat: sum(1 for x in epi if re.search(pat, x))
        res[name].update(epi_drains=ecnt(r"vmcnt\(0\)"), epi_load16=ecnt(r"global_load_dwordx4"), epi_load4=ecnt(r"global_load_dword\s"),
                         epi_store16=ecnt(r"global_store_dwordx4"), epi_store4=ecnt(r"global_store_dword\s"), epi_scratch=ecnt(r"scratch_"))
    return res


def analyse_dma64_gemm(flags=("-fno-slp-vectorize",)):
    """tcn_gemm_dma64_kernel (tcn.hip, round 4): whole-kernel counts.  The k-loop body exists twice (two k-tiles per trip), each with 16
    MFMAs, 4 LDS-DMA pieces and 8 fragment reads (+ 4 reads of the VALU column's operands); no scratch, no AGPR<->VGPR moves; the epilogue
    moves 16-byte rows (4 residual loads + 4 stores per lane for the 64 x 64 tile, one float4 store per row of the VALU column)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        src = os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", "tcn.hip")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN4fsnp21tcn_gemm_dma64_kernel\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), [l for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        idx = [i for i, l in enumerate(body) if "v_mfma" in l]
        loop, epi = body[idx[0]:idx[-1] + 1], body[idx[-1] + 1:]
        cnt = lambda seg, pat: sum(1 for x in seg if re.search(pat, x))
        res[name] = dict(mfma=cnt(body, r"v_mfma_f32_32x32x2"), dma=cnt(body, r"buffer_load_dwordx4 .* lds"), scratch=cnt(body, r"scratch_"),
                         acc_moves_in_loop=cnt(loop, r"v_accvgpr"), ds_write_in_loop=cnt(loop, r"ds_write"),
                         epi_store16=cnt(epi, r"global_store_dwordx4"), epi_store4=cnt(epi, r"global_store_dword\s"),
                         epi_load16=cnt(epi, r"global_load_dwordx4"), epi_drains=cnt(epi, r"vmcnt\(0\)"))
    return res


if __name__ == "__main__":
    for k, v in analyse_dma_gemm().items():
        print(k, v)
    for k, v in analyse_dma64_gemm().items():
        print(k, v)
    for k, loops in analyse_half_tile().items():
        for l in loops:
            print(k, l)
    r = analyse()
    bad = 0
    for k in sorted(r):
        for l in r[k]:
            ok = l["scratch"] == 0 and l["drain"] == 0
            bad += not ok
            print(k, l, "" if ok else "  <-- BAD")
    sys.exit(1 if bad else 0)
