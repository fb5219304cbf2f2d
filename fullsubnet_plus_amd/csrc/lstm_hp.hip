// lstm_hp.hip - "half-tile ping-pong" column-split two-layer LSTM + Linear for SMALL row counts on gfx950 (round 3).
//
// Same arithmetic as lstm_coop.hip / lstm_pp.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123) and again a group of S workgroups per 32-row tile that
// exchange h through global images every step - rebuilt around what the round-3 measurements of lstm_pp.hip said a K-split
// tile-phase spends outside its MFMAs (profiles/r03_column_split.md: LDS reduction of four K-split partial tiles 0.3 us,
// 2 cells per thread behind 32 LDS reads 1.0 us, 24 KB of operand fragments issued through VGPRs 1.1 us, three workgroup
// barriers 0.5 us, and - with 48 workgroups per group on two XCDs - fabric contention):
//
//   * 16 hidden units per workgroup: S = H / 16 = 24 workgroups share a row tile and sit on ONE XCD (lstm_common.h
//     xcd_local_decode) - the exchange goes through that XCD's L2;
//   * the four waves split the GATES, not K: wave g owns gate g of the 16 units = one 16-column block of
//     v_mfma_f32_16x16x4_f32 over the whole K, so there are no partial tiles to add up: a wave hands 4 registers per layer
//     to the cell phase (one ds_write_b128) instead of 32, a cell reads its 8 pre-activations with conflict-free
//     ds_read_b32s, and both layers' cells of a thread go through the packed two-cell update (lstm_common.h);
//   * the wave's weights (16 columns x 1200 k = 300 registers: layer 1 in AGPRs, layer 0 in VGPRs) stay in registers for
//     the whole sequence - no weight stream at all;
//   * a row tile is worked on as TWO HALF TILES of 16 sequences in turn (M = 16 is the MFMA's own height, so nothing is
//     wasted): while the h of one half is on its way through the fabric the workgroup computes the other half - the
//     ping-pong of lstm_pp.hip without needing a second row tile, i.e. with every CU busy at B = 1 (9 row tiles x 24);
//   * operands come global -> LDS by DMA (buffer_load_dwordx4 ... lds, waves 1-3: 16 KB each): no VGPRs, no issue slots of
//     the matrix pipe's wave, and the fetch of the next half-phase is issued from INSIDE the current MFMA pass as soon as a
//     poll (ONE wave polls; the others learn the outcome through an LDS token) shows the other half's counter complete;
//     wave 0 owns the publishing side: three 16-byte write-through stores from an LDS staging buffer, issued behind the NEXT
//     half-phase's first barrier (which thereby doubles as the "staging complete" barrier), the arrival a quarter pass later
//     behind an honest vmcnt(0) - wave 0's queue holds nothing else;
//   * fused phase as in lstm_pp.hip: [layer 1 of step t, layer 0 of step t + 1] is ONE pass over h0_t.
// (Round 5 measured the alternative of fetching the next half-phase's operands BEHIND the pass, under the cell phase - an LDS-DMA issue
//  costs 100-185 matrix-pipe cycles inside an MFMA stream: the pass shrank by 0.5 us, but the DMA then lands 0.75 us into the next
//  half-phase and the cell phase grew by 0.2: 13.8 -> 14.2 us per step, profiles/r05_column_split.md.  The in-pass fetch stays.)
// Exchange region and abort protocol: those of lstm_coop.hip; arrival counters: one per half tile, each in a 128-byte line of its own.
// 16x16x4 MFMAs sum K in another order than the 32x32x2 kernels, so results are bit-identical to no sibling kernel; same
// oracle tolerance (tests/test_gpu_parity.py::test_half_tile_ping_pong_kernel_vs_oracle), bitwise repeatable.
//
// Phases of one half tile (images double buffered by parity; its counter counts arrivals, S per phase):
//   phase -1     : acc0 = W_ih0 x_0 -> cell 0 -> h0_0 -> h0img[0]                                                  arrive
//   phase t >= 0 : wait counter >= S (t + 1);  DMA h1img[(t-1) & 1], h0img[t & 1] -> LDS
//                  acc1 = W_hh1 h1_{t-1} + W_ih1 h0_t,  acc0 = W_ih0 x_{t+1} + W_hh0 h0_t
//                  cells -> h1_t -> h1img[t & 1], Linear partials -> fcp[t & 1], h0_{t+1} -> h0img[(t+1) & 1]     arrive
//                  (workgroup q < 16: out[row q][t - 1] = bias + the S partials of step t - 1)
//   final        : wait counter >= S (Tp + 1); out[.][Tp - 1]
#include <type_traits>
#include <utility>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using lds_ptr = __attribute__((address_space(3))) void*;

template <typename F, int... I>
__device__ __forceinline__ void hp_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void hp_static_for(F&& f) { hp_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// float index of A element (row < 16, k) inside a half-tile image: [k-group of 16][k & 3][row][(k >> 2) & 3] - lane l of a
// k-group reads the float4 at (l >> 4 = k & 3, l & 15 = row); component j feeds the group's MFMA j (k = 16 g + 4 j + (l >> 4))
__host__ __device__ __forceinline__ int hp_a16(int row, int k) { return ((((k >> 4) * 4) + (k & 3)) * 16 + row) * 4 + ((k >> 2) & 3); }

constexpr int hp_gx(int KX) { return (KX + 15) / 16; }
constexpr size_t hp_smem_bytes(int HID, int KX) {
    return (size_t)(2 * hp_gx(KX) * 64 + 4 * (HID / 16) * 64 + 8 * 64 + 2 * 64 + 8) * 16 + 2 * (HID / 16) * 4 + 32 * sizeof(RowDesc) + 64;
}

// sum over the 16 lanes of a DPP row, in every lane (row_ror 8, 4, 2, 1: no LDS round trips)
template <int N>
__device__ __forceinline__ float hp_ror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false)); }
__device__ __forceinline__ float row_sum16(float v) { v += hp_ror<8>(v); v += hp_ror<4>(v); v += hp_ror<2>(v); v += hp_ror<1>(v); return v; }

#define HP_MF(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0)

}  // namespace

template <int HID, int KX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lstm2_coop_hp_kernel(LstmWeights w, LstmArgs a) {
    constexpr int S = HID / 16;                      // workgroups per row tile
    constexpr int GX = hp_gx(KX), GH = HID / 16;     // k-groups of 16: x (zero padded), one h image
    constexpr int NW0 = GX + GH, NW1 = 2 * GH;       // the wave's weight fragments (float4): layer 0 [x | h0], layer 1 [h1 | h0]
    constexpr int HALF_B = GH * 1024;                // bytes of one half image (16 rows x HID)
    constexpr int IMG_B = (HID / 8) * 1024;          // bytes of one 32-row image slot of the exchange region (lstm_common.h)
    constexpr int TILE_BYTES = coop_tile_f4(HID) * 16;
    constexpr int H0OFF = 0, H1OFF = 2 * IMG_B, FCOFF = 4 * IMG_B;
    // events inside the MFMA pass, in k-groups of the h part (a group = 12 MFMAs = 384 matrix-pipe cycles)
    // (measured: arrival after 3 ... 6 of 24 groups, poll at 8 ... 12, check at 14 ... 16 make no difference - profiles/r03_column_split.md)
    constexpr int ARRIVE_G = GH / 4, POLL1_G = GH / 2, CHECK1_G = (GH * 2) / 3, PEER_G = CHECK1_G + GH / 8, POLL2_G = PEER_G + 1;
    static_assert(KX <= 64 && GX <= 4, "gathered sub-band input");
    static_assert(S >= 16 && 2 * S <= 64, "one workgroup per output row of a half tile; the Linear partials are fetched by wave 0");
    static_assert((NW0 + NW1) * 4 <= 320 && NW1 * 4 <= 192, "the wave's weights must fit the register file (layer 1: AGPRs)");
    static_assert(4 * S * 128 <= 2 * (HID / 8) * 256, "Linear partials fit their slot of the exchange region");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);                  // [2 halves][GX][64]  A images of x
    float4* H1s = Xs + 2 * GX * 64;                                     // [2][GH][64]         h1_{t-1}
    float4* H0s = H1s + 2 * GH * 64;                                    // [2][GH][64]         h0_t
    float4* gat = H0s + 2 * GH * 64;                                    // [2 layers][4 gates][64] pre-activations
    float4* stage = gat + 8 * 64;                                       // [h0 | h1][64] + Linear partials [8]
    float* fc_red = reinterpret_cast<float*>(stage + 2 * 64 + 8);       // [2][S]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(fc_red + 2 * S);       // [32]
    int* flags = reinterpret_cast<int*>(rows_s + 32);                   // [0] abort, [1] token of the half-phase whose successor's operands may be fetched

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rt = blockIdx.x / S, cs = blockIdx.x % S;
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, a.num_tiles, a.coop_xcd, rt, cs)) return;
    const int Tp = a.Tp;

    if (tid == 0) { flags[0] = 0; flags[1] = 0; }
    for (int i = tid; i < 2 * GX * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) rows_s[tid] = a.rows[rt * 32 + tid];
    __syncthreads();
    // half tiles of this row tile that hold sequences (a launch's sequences are spread evenly over its tiles, from slot 0 up: build_rows_kernel)
    const int nh = rows_s[16].valid ? 2 : 1;

    // ---- input plan: thread owns row tid & 15 of each half, features (tid >> 4) + 16 i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    const int xrow = tid & 15, j0 = tid >> 4;
    int goff[2][GX];
    NormMD md[2];
    const NormMD* md_t[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const RowDesc rd = rows_s[hf * 16 + xrow];
        md[hf] = NormMD{0.0f, 1.0f};
        md_t[hf] = nullptr;
#pragma unroll
        for (int i = 0; i < GX; ++i) {
            const int j = j0 + 16 * i;
            int off = -1;
            if (rd.valid && j < w.NIN) {
                if (dense) off = rd.b * Tp * gstep + j;
                else off = sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            goff[hf][i] = off;
        }
        if (rd.valid) {
            if (a.md_seq != nullptr) md_t[hf] = a.md_seq + (size_t)rd.b * Tp;
            else if (!dense && a.md_row != nullptr) md_t[hf] = a.md_row + (size_t)(rt * 32 + hf * 16 + xrow) * Tp;
            else if (!dense) md[hf] = a.md_utt[rd.b];
        }
    }
    const int xdst0 = hp_a16(xrow, j0);               // (+ 256 floats per further k-group)
    // (one buffer load per element and step: the per-lane byte offset is loop invariant, the step goes into the scalar offset;
    //  elements without a source - goff < 0 - load offset 0 and are never written to the x image)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbase), 0, 0x7FFFFFFC, 0x00020000);
    int xvo[2][GX];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < GX; ++i) xvo[hf][i] = goff[hf][i] >= 0 ? goff[hf][i] * 4 : 0;
    auto x_load = [&](int hf, int i, int t) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xvo[hf][i], t * gstep * 4, 0));
    };

    // ---- the wave's weights, resident: [cs][gate = wave][fragment][lane][4]
    float4 w0[NW0], w1[NW1];
    {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(w.wpack_hp) + (size_t)(cs * 4 + wave) * (NW0 + NW1) * 256, 0, (NW0 + NW1) * 1024, 0x00020000);
#pragma unroll
        for (int i = 0; i < NW0; ++i) w0[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, i * 1024, 0));
#pragma unroll
        for (int i = 0; i < NW1; ++i) w1[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, (NW0 + i) * 1024, 0));
#pragma unroll
        for (int i = 0; i < NW1; ++i) asm volatile("" : "+a"(w1[i].x), "+a"(w1[i].y), "+a"(w1[i].z), "+a"(w1[i].w));
    }
    const float bias0 = w.bias[wave * HID + cs * 16 + (lane & 15)], bias1 = w.bias[4 * HID + wave * HID + cs * 16 + (lane & 15)];

    // ---- exchange region of this row tile
    const __amdgpu_buffer_rsrc_t hr =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.coop_hx) + (size_t)rt * (TILE_BYTES / 4), 0, TILE_BYTES, 0x00020000);
    unsigned* const bars[2] = {FSNP_COOP_BAR(a, rt, 0), FSNP_COOP_BAR(a, rt, 1)};     // one 128-byte line each
    auto store16 = [&](const float4& v, int voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), hr, voff, soff, kSc1);
    };

    // ---- cell ownership: thread = (unit u = lane & 15, row 4 wave + (lane >> 4)): its pre-activations are float gidx of every gate block
    // (64 lanes -> 64 distinct banks), and the units of a row are the 16 lanes of a DPP row (the Linear partial sums)
    const int cu = lane & 15, crow = 4 * wave + (lane >> 4);
    const float wfc0 = w.wfc[cs * 16 + cu], wfc1 = w.wfc[HID + cs * 16 + cu];
    const int sdst = ((cu & 3) * 16 + crow) * 4 + (cu >> 2);             // float index inside a staged k-group (hp_a16)
    const int gidx = (wave * 16 + cu) * 4 + (lane >> 4);                 // float index inside one gate block [64 lanes][4 rows]
    f32x2 cst[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};                    // {c1, c0} of each half
    const float* gf = reinterpret_cast<const float*>(gat);
    float* sf = reinterpret_cast<float*>(stage);
    float* Xf = reinterpret_cast<float*>(Xs);

    // Linear(H, 2): workgroup q < 16 owns output row q of both halves; lanes [0, 2 S) of wave 0 fetch the S partials of both outputs
    const bool fc_wg = cs < 16;
    const int fc_voff = (tid % S) * 128 + ((tid / S) * 16 + cs) * 4;
    auto fc_finish = [&](int hf, int t_done) {         // wave 2, lanes 0..31: output o = lane >> 4, a fixed tree over the S partials
        if (fc_wg && wave == 2 && lane < 32) {
            const int o = lane >> 4, q = lane & 15;
            const RowDesc rd = rows_s[hf * 16 + cs];
            float sum = fc_red[o * S + q] + (q + 16 < S ? fc_red[o * S + q + 16] : 0.0f);
            sum = row_sum16(sum) + w.bfc[o];
            if (q == 0 && rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };

    // ---- x_0 of both halves
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const NormMD m0 = md_t[hf] ? md_t[hf][0] : md[hf];
#pragma unroll
        for (int i = 0; i < GX; ++i)
            if (goff[hf][i] >= 0) Xf[hf * GX * 256 + xdst0 + i * 256] = (x_load(hf, i, 0) - m0.m) / m0.d;
    }
    __syncthreads();

    // ================= phase -1 of every half: h0_0 =================
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (hf >= nh) break;
        float xr[GX];
        NormMD mdn = md[hf];
        if (Tp > 1) {
            if (md_t[hf]) mdn = md_t[hf][1];
#pragma unroll
            for (int i = 0; i < GX; ++i) xr[i] = x_load(hf, i, 1);
        }
        f32x4 a0a = {bias0, bias0, bias0, bias0}, a0b = {0.f, 0.f, 0.f, 0.f};
        const float4* Xh = Xs + hf * GX * 64 + lane;
#pragma unroll
        for (int i = 0; i < GX; ++i) {
            const float4 ax = Xh[i * 64];
            HP_MF(a0a, ax.x, w0[i].x); HP_MF(a0b, ax.y, w0[i].y); HP_MF(a0a, ax.z, w0[i].z); HP_MF(a0b, ax.w, w0[i].w);
        }
        a0a += a0b;
        gat[wave * 64 + lane] = make_float4(a0a[0], a0a[1], a0a[2], a0a[3]);
        __syncthreads();
        {
            f32x2 c = {0.f, cst[hf].y};
            const f32x2 hh = lstm_cell_pair(f32x2{0.f, gf[0 * 256 + gidx]}, f32x2{0.f, gf[1 * 256 + gidx]}, f32x2{0.f, gf[2 * 256 + gidx]},
                                            f32x2{0.f, gf[3 * 256 + gidx]}, c);
            cst[hf].y = c.y;
            sf[sdst] = hh.y;
        }
        if (Tp > 1) {
#pragma unroll
            for (int i = 0; i < GX; ++i)
                if (goff[hf][i] >= 0) Xf[hf * GX * 256 + xdst0 + i * 256] = (xr[i] - mdn.m) / mdn.d;
        }
        __syncthreads();
        if (wave == 0) {
            store16(stage[lane], (cs * 64 + lane) * 16, H0OFF + hf * HALF_B);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(bars[hf], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ================= phases 0 .. Tp - 1, the two halves in turn =================
    // fetch(half, t): operands of half-phase (half, t).  Waves 1-3: 2 GH chunks of 1 KB, global -> LDS; wave 0 (workgroups that
    // own an output row): the S Linear partials of step t - 1.  Called once the half's counter shows S (t + 1) arrivals.
    float fcv = 0.0f;
    // (LDS byte addresses of the two image arrays as integers: a generic -> LDS pointer conversion per chunk carries a null check)
    const unsigned lds_h1 = (unsigned)(size_t)(lds_ptr)H1s, lds_h0 = (unsigned)(size_t)(lds_ptr)H0s;
    auto fetch = [&](auto HN, int tn) {
        constexpr int hn = decltype(HN)::value;
        if (wave != 0) {
            const int h1src = H1OFF + ((tn & 1) ^ 1) * IMG_B + hn * HALF_B, h0src = H0OFF + (tn & 1) * IMG_B + hn * HALF_B;
            for (int c = wave - 1; c < 2 * GH; c += 3) {
                if (c < GH) __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr)(size_t)(lds_h1 + (hn * GH + c) * 1024), 16, lane * 16, h1src + c * 1024, 0, kSc1);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr)(size_t)(lds_h0 + (hn * GH + c - GH) * 1024), 16, lane * 16, h0src + (c - GH) * 1024, 0, kSc1);
            }
        } else if (fc_wg && tn >= 1 && tid < 2 * S) {
            fcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr, fc_voff, FCOFF + ((((tn & 1) ^ 1) * 2 + hn) * S) * 128, kSc1));
        }
    };
    // a wave's blocking wait for a counter (lane 0 polls; the whole wave learns the outcome)
    auto wave_wait = [&](unsigned* bar, unsigned target) -> bool {
        int ok = 1;
        if (lane == 0) ok = xchg_wait(bar, target, a.coop_abort, a.coop_err) ? 1 : 0;
        ok = __builtin_amdgcn_readfirstlane(ok);
        if (!ok && lane == 0) flags[0] = 1;
        return ok != 0;
    };
    // optional phase profile (fsnp_debug_pp_profile): thread 0 of workgroup 0 stamps the 100 MHz wall clock: prof[(t * 2 + hf) * 16 + k]
    unsigned long long* prof = (a.prof != nullptr && blockIdx.x == 0 && tid == 0) ? a.prof : nullptr;
#define FSNP_HP_STAMP(k) do { if (prof) prof[(t * 2 + hf) * 16 + (k)] = (unsigned long long)wall_clock64(); } while (0)

    bool pending = false;                 // wave 0: the arrival of the previous half-phase has not been issued yet
    unsigned* pending_bar = nullptr;
    // wave 0 publishes the staged h0 / h1 / Linear partials of half-phase (hp, tp): three 16-byte write-through stores per lane
    auto publish = [&](int hp, int tp) {
        store16(stage[lane], (cs * 64 + lane) * 16, H0OFF + ((tp + 1) & 1) * IMG_B + hp * HALF_B);
        store16(stage[64 + lane], (cs * 64 + lane) * 16, H1OFF + (tp & 1) * IMG_B + hp * HALF_B);
        if (lane < 8) store16(stage[128 + lane], lane * 16, FCOFF + (((tp & 1) * 2 + hp) * S + cs) * 128);
    };
    bool pub_pending = false;             // (two halves) the previous half-phase's cells are staged but not published yet: that happens behind
    int pub_t = 0;                        // the NEXT half-phase's first barrier, which saves a workgroup barrier per half-phase
    bool dead = false;
    // first half-phase: nothing to overlap with
    if (wave != 0 || fc_wg) { if (wave_wait(bars[0], (unsigned)S)) fetch(std::integral_constant<int, 0>{}, 0); }

    for (int t = 0; t < Tp && !dead; ++t) {
        hp_static_for<2>([&](auto HC) {
            constexpr int hf = decltype(HC)::value;
            constexpr int ho = hf ^ 1;
            if (hf >= nh || dead) return;
            const bool two = nh == 2;
            const int nt = (two && hf == 0) ? t : t + 1;              // the next half-phase is (two ? ho : 0, nt)
            const bool has_next = nt < Tp;
            unsigned* nbar = bars[two ? ho : 0];
            const unsigned ntarget = (unsigned)S * (unsigned)(nt + 1);
            const bool fc_now = fc_wg && t >= 1;
            const bool polls = two && has_next && (wave != 0 || fc_wg);     // (one half only: the next phase needs THIS phase's arrivals)
            const int token = 2 * t + hf + 1;                               // wave 1 polls the counter; the others learn the outcome through flags[1]
            auto fetch_next = [&]() {
                if (two) { if constexpr (hf == 0) fetch(std::integral_constant<int, 1>{}, nt); else fetch(std::integral_constant<int, 0>{}, nt); }
                else fetch(std::integral_constant<int, 0>{}, nt);
            };
            FSNP_HP_STAMP(0);
            chaos_delay(a.coop_chaos, t, hf);
            // ---- operands of this half-phase are in LDS once every DMA wave has drained its queue
            if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (flags[0]) { dead = true; return; }
            FSNP_HP_STAMP(1);
            if (wave == 0 && pub_pending) {      // the other half's previous half-phase: staged before this barrier
                publish(ho, pub_t);
                pending = true; pending_bar = bars[ho]; pub_pending = false;
            }

            // ---- one pass: acc0 = W_ih0 x_{t+1} + W_hh0 h0_t, acc1 = W_hh1 h1_{t-1} + W_ih1 h0_t  (wave = gate)
            f32x4 a0a = {bias0, bias0, bias0, bias0}, a0b = {0.f, 0.f, 0.f, 0.f};
            f32x4 a1a = {bias1, bias1, bias1, bias1}, a1b = {0.f, 0.f, 0.f, 0.f};
            {
                const float4* Xh = Xs + hf * GX * 64 + lane;
#pragma unroll
                for (int i = 0; i < GX; ++i) {
                    const float4 ax = Xh[i * 64];
                    HP_MF(a0a, ax.x, w0[i].x); HP_MF(a0b, ax.y, w0[i].y); HP_MF(a0a, ax.z, w0[i].z); HP_MF(a0b, ax.w, w0[i].w);
                }
            }
            const float4* P1 = H1s + hf * GH * 64 + lane;
            const float4* P0 = H0s + hf * GH * 64 + lane;
            float xr[GX];
            NormMD mdn = md[hf];
            const bool have_x = t + 2 < Tp;
            unsigned seen = 0;
            bool fetched = false;
            float4 p = P1[0], q = P0[0];
#pragma unroll
            for (int g = 0; g < GH; ++g) {
                // (the next group's fragments are read first: an LDS round trip per group would otherwise sit in front of its MFMAs)
                const float4 pn = P1[(g + 1 < GH ? g + 1 : g) * 64], qn = P0[(g + 1 < GH ? g + 1 : g) * 64];
                __builtin_amdgcn_sched_barrier(0);
                HP_MF(a1a, p.x, w1[g].x); HP_MF(a0a, q.x, w0[GX + g].x); HP_MF(a1b, q.x, w1[GH + g].x);
                HP_MF(a1a, p.y, w1[g].y); HP_MF(a0a, q.y, w0[GX + g].y); HP_MF(a1b, q.y, w1[GH + g].y);
                HP_MF(a1a, p.z, w1[g].z); HP_MF(a0a, q.z, w0[GX + g].z); HP_MF(a1b, q.z, w1[GH + g].z);
                HP_MF(a1a, p.w, w1[g].w); HP_MF(a0a, q.w, w0[GX + g].w); HP_MF(a1b, q.w, w1[GH + g].w);
                if (g == ARRIVE_G) {
                    FSNP_HP_STAMP(2);
                    if (wave == 0 && pending) {      // the previous half-phase's stores are a quarter pass old; nothing else is in wave 0's queue
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add(pending_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pending = false;
                    }
                    FSNP_HP_STAMP(3);
                    if (fc_now && tid < 2 * S) fc_red[tid] = fcv;
                    if (have_x) {
                        if (md_t[hf]) mdn = md_t[hf][t + 2];
#pragma unroll
                        for (int i = 0; i < GX; ++i) xr[i] = x_load(hf, i, t + 2);
                    }
                }
                // ONE wave polls the other half's counter (the load is issued here, looked at four groups later: no stall unless the
                // fabric is slow); the other waves learn the outcome through LDS three groups after that, and again behind barrier 2
                if ((g == POLL1_G || g == POLL2_G) && wave == 1 && polls && !fetched) seen = __hip_atomic_load(nbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g == CHECK1_G && wave == 1 && polls && __builtin_amdgcn_readfirstlane(seen) >= ntarget) {
                    if (lane == 0) __hip_atomic_store(&flags[1], token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    fetch_next(); fetched = true;
                }
                if (g == PEER_G && wave != 1 && polls && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == token) { fetch_next(); fetched = true; }
                __builtin_amdgcn_sched_barrier(0);
                p = pn; q = qn;
            }
            if (wave == 1 && polls && !fetched && __builtin_amdgcn_readfirstlane(seen) >= ntarget) {
                if (lane == 0) __hip_atomic_store(&flags[1], token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                fetch_next(); fetched = true;
            }
            FSNP_HP_STAMP(4);
            a0a += a0b; a1a += a1b;
            gat[wave * 64 + lane] = make_float4(a0a[0], a0a[1], a0a[2], a0a[3]);
            gat[(4 + wave) * 64 + lane] = make_float4(a1a[0], a1a[1], a1a[2], a1a[3]);
            __syncthreads();
            FSNP_HP_STAMP(5);
            if (polls && !fetched && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == token) { fetch_next(); fetched = true; }
            // ---- cells: {layer 1 (h1_t), layer 0 (h0_{t+1})} of (row crow, unit cu) as ONE packed two-cell update
            {
                const f32x2 hh = lstm_cell_pair(f32x2{gf[4 * 256 + gidx], gf[0 * 256 + gidx]}, f32x2{gf[5 * 256 + gidx], gf[1 * 256 + gidx]},
                                                f32x2{gf[6 * 256 + gidx], gf[2 * 256 + gidx]}, f32x2{gf[7 * 256 + gidx], gf[3 * 256 + gidx]}, cst[hf]);
                // (test hook, LstmArgs::coop_corrupt: h0 of step t + 1, row 0, unit 0 of row tile 0 is published with 1.0 added)
                sf[sdst] = (a.coop_corrupt != 0 && rt == 0 && cs == 0 && hf == 0 && cu == 0 && crow == 0 && t + 2 == a.coop_corrupt) ? hh.y + 1.0f : hh.y;
                sf[256 + sdst] = hh.x;
                float p0 = hh.x * wfc0, p1 = hh.x * wfc1;               // partial Linear over this workgroup's 16 units
                p0 = row_sum16(p0); p1 = row_sum16(p1);
                if (cu == 0) { sf[512 + crow] = p0; sf[512 + 16 + crow] = p1; }
            }
            if (have_x) {
#pragma unroll
                for (int i = 0; i < GX; ++i)
                    if (goff[hf][i] >= 0) Xf[hf * GX * 256 + xdst0 + i * 256] = (xr[i] - mdn.m) / mdn.d;
            }
            if (fc_now) fc_finish(hf, t - 1);
            chaos_delay(a.coop_chaos, t, 2 + hf);
            FSNP_HP_STAMP(6);
            if (!two) {                           // one half only: the next wait is for this very half - publish and arrive now
                __syncthreads();
                if (wave == 0) {
                    publish(hf, t);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(bars[hf], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                pub_pending = true; pub_t = t;    // (published behind the next half-phase's first barrier)
            }
            FSNP_HP_STAMP(7);
            if (prof) prof[(t * 2 + hf) * 16 + 15] = fetched ? 1ull : 0ull;        // (wave 0's view: the Linear partials)
            if (has_next && !fetched && (wave != 0 || fc_wg)) {      // not seen complete during the pass: wait for it for real
                if (wave_wait(nbar, ntarget)) fetch_next();
            }
            FSNP_HP_STAMP(8);
        });
    }
    __syncthreads();                      // the last half-phase's cells are staged
    if (wave == 0 && !dead) {
        if (pending) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(pending_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pub_pending) {
            publish(nh - 1, pub_t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(bars[nh - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (flags[0] || dead) return;
    // ================= the Linear of the last step =================
    if (!fc_wg) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (hf >= nh) break;
        __syncthreads();                             // fc_red of the previous half has been read
        if (tid == 0 && !xchg_wait(bars[hf], (unsigned)S * (unsigned)(Tp + 1), a.coop_abort, a.coop_err)) flags[0] = 1;
        __syncthreads();
        if (flags[0]) return;
        if (tid < 2 * S)
            fc_red[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr, fc_voff, FCOFF + ((((Tp - 1) & 1) * 2 + hf) * S) * 128, kSc1));
        __syncthreads();
        fc_finish(hf, Tp - 1);
    }
}
#undef FSNP_HP_STAMP

// ------------------------------------------------------------------------------------------------
size_t lstm_hp_pack_floats(int H, int KX) { return (size_t)(H / 16) * 4 * (hp_gx(KX) + 3 * (H / 16)) * 256; }

// [column slice cs][gate][fragment: x k-groups | W_hh0 | W_hh1 | W_ih1][lane][4]: the B operand of MFMA j of a k-group is
// W[gate * H + cs * 16 + (lane & 15)][k = 16 g + 4 j + (lane >> 4)]
void lstm_hp_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out) {
    const int S = H / 16, GX = hp_gx(KX), GH = H / 16, NF = GX + 3 * GH;
    for (int cs = 0; cs < S; ++cs)
        for (int gate = 0; gate < 4; ++gate)
            for (int f = 0; f < NF; ++f)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const size_t wrow = (size_t)gate * H + cs * 16 + (lane & 15);
                        float v = 0.0f;
                        if (f < GX) { const int k = 16 * f + 4 * j + (lane >> 4); if (k < NIN) v = wih0[wrow * NIN + k]; }
                        else if (f < GX + GH) v = whh0[wrow * H + 16 * (f - GX) + 4 * j + (lane >> 4)];
                        else if (f < GX + 2 * GH) v = whh1[wrow * H + 16 * (f - GX - GH) + 4 * j + (lane >> 4)];
                        else v = wih1[wrow * H + 16 * (f - GX - 2 * GH) + 4 * j + (lane >> 4)];
                        out[((((size_t)cs * 4 + gate) * NF + f) * 64 + lane) * 4 + j] = v;
                    }
}

template <int HID, int KX>
static void launch_hp_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    constexpr int S = HID / 16;
    const size_t smem_need = hp_smem_bytes(HID, KX);
    auto kern = lstm2_coop_hp_kernel<HID, KX>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    if (occ) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(kern), 256, smem_need) != hipSuccess) *occ = 0;
        return;
    }
    const size_t smem = a.coop_own_cu > 0 && (size_t)a.coop_own_cu > smem_need ? (size_t)a.coop_own_cu : smem_need;
    const int grid = a.coop_xcd ? 8 * xcd_local_blocks_per_xcd(S, a.num_tiles, a.coop_xcd) : a.num_tiles * S;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, w, a);
}

bool lstm_hp_available(const LstmWeights& w) { return !w.gru && (w.H == 384 || w.H == 256) && (w.KX == 40 || w.KX == 64); }

// a.num_tiles row tiles x H / 16 workgroups, all co-resident
void launch_lstm_hp(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.H == 256) { if (w.KX == 64) launch_hp_inst<256, 64>(w, a, s, nullptr); else launch_hp_inst<256, 40>(w, a, s, nullptr); return; }
    if (w.KX == 64) launch_hp_inst<384, 64>(w, a, s, nullptr); else launch_hp_inst<384, 40>(w, a, s, nullptr);
}

}  // namespace fsnp
