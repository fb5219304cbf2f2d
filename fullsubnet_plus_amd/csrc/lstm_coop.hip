// lstm_coop.hip - column-split ("cooperative") two-layer LSTM for SMALL row counts on gfx950.
//
// Same arithmetic as lstm.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123), different decomposition.  The row-tile
// kernel gives one CU 32 sequences and ALL 4H gate columns, so a step costs one CU ~200 us however few tiles exist:
// the reference CLI's batch of ONE utterance (257 sequences = 9 tiles) kept 247 CUs idle for 26 ms, and the
// full-band LSTM of the original FullSubNet (speech_enhance/fullsubnet/model/fullsubnet.py:39-47: 257 -> 512 x 2,
// only B sequences) is a single tile.  Here a 32-row tile is shared by S = H / UNITS workgroups:
//   * workgroup (rt, cs) owns hidden units [cs * UNITS, (cs+1) * UNITS) of both layers = 4 UNITS gate columns =
//     NT = UNITS / 8 accumulator tiles (column j of the workgroup = gate * UNITS + unit, tile j / 32);
//   * its 4 waves split K (each wave reduces a quarter of the k-groups into its own copy of the tiles); the partial
//     tiles are summed through LDS by the cell update, which is spread over all 256 threads: thread -> UNITS / 8
//     (row, unit) pairs, c stays in registers;
//   * every step each workgroup publishes its 32 x UNITS slice of h0_t / h1_t into a per-tile, double-buffered
//     exchange image in global memory that is ALREADY in MFMA A-fragment order, so consumers read their A operands
//     straight from L2 with one coalesced 16-byte load per lane per k-group (no LDS copy);
//   * ONE inter-workgroup barrier per step (after h0_t is published) orders everything: h1_{t-1} was published
//     before its writer arrived.  The hand-off is the write-through recipe (lstm_common.h; cdna_hip_programming.md G16 R1):
//     sc1 stores, every storing wave drains vmcnt, __syncthreads, ONE lane: asm vmcnt(0) + relaxed atomic arrive, relaxed
//     polling with s_sleep, __syncthreads, sc1 loads - no release / acquire fence.  Spins are bounded; all
//     workgroups of a launch must be co-resident (the host only launches RT * S <= number of CUs).
//   * SEQ = false (sub-band model): the Linear(H, 2) epilogue is a per-workgroup partial dot over its own units,
//     published with h1 and summed in a fixed order by workgroup cs == 0 one step later (deterministic, no float
//     atomics).  SEQ = true (full-band model): h1_t is also written row-major to seq_out[seq][t][H]; the wide
//     Linear(512, 257) + activation is one GEMM afterwards (tcn.hip).
#include <cstdlib>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

struct CoopStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};

template <int NT>
__device__ __forceinline__ float4 coop_wload(const CoopStream& ws, int group, int n) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, (group * NT + n) * 1024, 0);
    return __builtin_bit_cast(float4, v);
}

// One layer's share of this wave: G local k-groups (global k-group 4 i + wave - the waves interleave so that every
// trip count is a compile-time constant and the loop unrolls into straight-line code with immediate offsets), weights
// at stream groups [WBASE, WBASE + G).  The first XG local groups take their A operand from srcx(i), the others from
// srch(i - XG).  Register pipeline D groups deep: a k-group lasts NT * 256 cycles, D = 16 / NT keeps ~4096 cycles
// (> the L2 round trip) of loads in flight.
template <int NT, int G>
struct CoopDepth { static constexpr int value = (16 / NT < G) ? 16 / NT : G; };
template <int NT, int G>
using CoopA = float4[CoopDepth<NT, G>::value];            // A fragments of the D groups in flight
template <int NT, int G>
using CoopB = float4[CoopDepth<NT, G>::value][NT];        // their weight fragments

// the first D groups' loads of a layer (coop_layer = coop_layer_prefill + coop_layer_run; the layer-skewed kernel issues the
// prefill of its layer-1 phase BEFORE the drain / arrival / wait that end its layer-0 phase: those operands were published
// a whole phase earlier, so their round trip overlaps the hand-off instead of following it)
template <int NT, int G, int XG, int WBASE, typename SrcX, typename SrcH>
__device__ __forceinline__ void coop_layer_prefill(CoopA<NT, G>& a, CoopB<NT, G>& b,
                                                   const CoopStream& ws, SrcX srcx, SrcH srch) {
    constexpr int D = CoopDepth<NT, G>::value;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        a[k] = k < XG ? srcx(k) : srch(k - XG);
#pragma unroll
        for (int n = 0; n < NT; ++n) b[k][n] = coop_wload<NT>(ws, WBASE + k, n);
    }
}

template <int NT, int G, int XG, int WBASE, typename SrcX, typename SrcH>
__device__ __forceinline__ void coop_layer_run(f32x16 (&acc)[NT], CoopA<NT, G>& a, CoopB<NT, G>& b,
                                               const CoopStream& ws, SrcX srcx, SrcH srch) {
    constexpr int D = CoopDepth<NT, G>::value;
    auto fill = [&](int slot, int i) {
        a[slot] = i < XG ? srcx(i) : srch(i - XG);
#pragma unroll
        for (int n = 0; n < NT; ++n) b[slot][n] = coop_wload<NT>(ws, WBASE + i, n);
    };
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int slot = i % D;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].x, b[slot][n].x, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].y, b[slot][n].y, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].z, b[slot][n].z, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].w, b[slot][n].w, acc[n], 0, 0, 0);
        }
        if (i + D < G) fill(slot, i + D);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT, int G, int XG, int WBASE, typename SrcX, typename SrcH>
__device__ __forceinline__ void coop_layer(f32x16 (&acc)[NT], const CoopStream& ws, SrcX srcx, SrcH srch) {
    constexpr int D = CoopDepth<NT, G>::value;
    float4 a[D];
    float4 b[D][NT];
    coop_layer_prefill<NT, G, XG, WBASE>(a, b, ws, srcx, srch);
    coop_layer_run<NT, G, XG, WBASE>(acc, a, b, ws, srcx, srch);
}

// The same with the wave's weights RESIDENT in registers (bw[i][n], loaded once before the time loop): with <= 16 hidden
// units per workgroup a wave's whole share of both layers is 152..228 registers, so a step issues no weight loads at
// all - only the A operands travel (D-deep register pipeline as above).
template <int NT, int G, int XG, typename SrcX, typename SrcH>
__device__ __forceinline__ void coop_layer_resident(f32x16 (&acc)[NT], const float4 (&bw)[G][NT], SrcX srcx, SrcH srch) {
    constexpr int D = 8 < G ? 8 : G;
    float4 a[D];
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = k < XG ? srcx(k) : srch(k - XG);
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int slot = i % D;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].x, bw[i][n].x, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].y, bw[i][n].y, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].z, bw[i][n].z, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot].w, bw[i][n].w, acc[n], 0, 0, 0);
        }
        if (i + D < G) a[slot] = (i + D) < XG ? srcx(i + D) : srch(i + D - XG);
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int coop_units_index(int units) { return units == 8 ? 0 : units == 16 ? 1 : units == 32 ? 2 : 3; }

}  // namespace

// GRU = true: nn.GRU instead of nn.LSTM (sequence_model.py:39-46).  The four column slots of a hidden unit are then
// (r, z, n_x, n_h) - the host packs W_in only into the input part of K and W_hn only into the hidden part (zero blocks
// elsewhere), so the MFMA loops are unchanged - and the per-unit state kept in registers is h itself:
//   r = s(a_r), z = s(a_z), n = tanh(a_nx + r * a_nh), h' = (1 - z) n + z h          (torch.nn.GRU)
template <int HID, int KX, int UNITS, bool SEQ, bool GRU>
__global__ __launch_bounds__(256) void lstm2_coop_kernel(LstmWeights w, LstmArgs a) {
    constexpr int NT = UNITS / 8;                  // 32-column accumulator tiles per workgroup
    constexpr int NP = UNITS / 8;                  // (row, unit) pairs per thread in the cell update
    constexpr int KGX = KX / 8, KGH = HID / 8;
    constexpr int KGXP = (KGX + 3) / 4 * 4;        // x k-groups padded (zero weights, zero A) so that 4 waves split evenly
    constexpr int G0W = (KGXP + KGH) / 4;          // local k-groups per wave, layer 0: [x | h0_{t-1}]
    constexpr int G1W = KGH / 2;                   //                          layer 1: [h1_{t-1} | h0_t]
    constexpr int S = HID / UNITS;
    constexpr int HIMG = KGH * 64;                 // float4 per exchange image (32 rows x HID)
    constexpr bool GATHER = KX <= 64;              // sub-band input built from att_mag / fb; else dense rows only
    constexpr bool BIAS_REGS = UNITS <= 32;
    // weights resident in VGPRs (see coop_layer_resident): 152 (H = 384) / 228 (H = 512) registers at 8 units per workgroup.
    // Not at 16 units (304): half of them would live in AGPRs and be copied back before every MFMA - measured 16 % SLOWER.
    constexpr bool WREG = NT * (G0W + G1W) * 4 <= 232;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);                     // [KGXP][64] A image of x_t
    float* red = reinterpret_cast<float*>(Xs + KGXP * 64);                // [4 waves][NT][16][64]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(red + 4 * NT * 16 * 64); // [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup id -> (row tile, column slice).  Flat: consecutive ids = the S slices of one tile (spread over the 8 XCDs by
    // the dispatcher's id % 8 placement).  XCD-local (a.coop_xcd = CUs per XCD): see lstm_common.h xcd_local_decode.
    int rt = blockIdx.x / S, cs = blockIdx.x % S;
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, a.num_tiles, a.coop_xcd, rt, cs)) return;
    const int slot0 = rt * 32;
    const int Tp = a.Tp;

    // exchange region of this row tile: [h0 p0][h0 p1][h1 p0][h1 p1] images + FC partials [2][H/8][64]
    float4* hx = reinterpret_cast<float4*>(a.coop_hx) + (size_t)rt * coop_tile_f4(HID);
    float4* h0img[2] = {hx, hx + HIMG};
    float4* h1img[2] = {hx + 2 * HIMG, hx + 3 * HIMG};
    float* fcp = reinterpret_cast<float*>(hx + 4 * HIMG);                  // [2][S][64]
    unsigned* bar = FSNP_COOP_BAR(a, rt, 0);

    __shared__ int abort_s;                        // set by thread 0 when an inter-workgroup wait gives up
    if (tid == 0) abort_s = 0;
    for (int i = tid; i < KGXP * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) rows_s[tid] = a.rows[slot0 + tid];
    __syncthreads();

    // ---- input plan: thread owns row = tid & 31, features j = (tid >> 5) + 8 i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    constexpr int NG = KGX;
    const int grow = tid & 31;
    const int jrow = tid >> 5;
    int goff[GATHER ? NG : 1];
    int dbase = -1;                                   // dense-only variant: offset of (row, t = 0, j = 0)
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_t = nullptr;                     // per-step table of this row (stride 1 in t)
    {
        const RowDesc rd = rows_s[grow];
        if constexpr (GATHER) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int j = jrow + 8 * i;
                int off = -1;
                if (rd.valid && j < w.NIN) {
                    if (dense) off = rd.b * Tp * gstep + j;
                    else off = sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
                }
                goff[i] = off;
            }
        } else {
            goff[0] = 0;
            if (rd.valid) dbase = rd.b * Tp * gstep;
        }
        if (rd.valid) {
            if (a.md_seq != nullptr) md_t = a.md_seq + (size_t)rd.b * Tp;                            // [sequence][t]
            else if (!dense && a.md_row != nullptr) md_t = a.md_row + (size_t)(slot0 + grow) * Tp;   // [slot][t]
            else if (!dense) md = a.md_utt[rd.b];
        }
    }
    auto x_valid = [&](int i) -> bool {
        if constexpr (GATHER) return goff[i] >= 0;
        else return dbase >= 0 && jrow + 8 * i < w.NIN;
    };
    auto x_load = [&](int i, int t) -> float {
        if constexpr (GATHER) return goff[i] >= 0 ? gbase[goff[i] + t * gstep] : 0.0f;
        else return x_valid(i) ? gbase[dbase + t * gstep + jrow + 8 * i] : 0.0f;
    };
    const int xdst0 = a_frag_index(grow, jrow);
    float* Xf = reinterpret_cast<float*>(Xs);
    {
        const NormMD m0 = md_t ? md_t[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = x_valid(i) ? (x_load(i, 0) - m0.m) / m0.d : 0.0f;
    }

    // ---- this wave's private weight stream: [cs][wave][local k-group][tile][lane][4]
    CoopStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(w.wpack) + (size_t)(cs * 4 + wave) * (G0W + G1W) * NT * 256, 0, (G0W + G1W) * NT * 1024, 0x00020000);
    ws.voff = lane * 16;
    float4 bw0[WREG ? G0W : 1][NT], bw1[WREG ? G1W : 1][NT];
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < G0W; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) bw0[i][n] = coop_wload<NT>(ws, i, n);
#pragma unroll
        for (int i = 0; i < G1W; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) bw1[i][n] = coop_wload<NT>(ws, G0W + i, n);
    }
    const float4* Xw = Xs + wave * 64 + lane;           // local group i of this wave = global k-group 4 i + wave
    // A operands from the exchange images: buffer loads too (a flat load would make hipcc drain vmcnt AND lgkmcnt)
    CoopStream hs;
    hs.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(hx), 0, 4 * HIMG * 16, 0x00020000);
    hs.voff = (wave * 64 + lane) * 16;
    auto hload = [&](int image, int i) -> float4 {      // image: 0/1 = h0 parity 0/1, 2/3 = h1 parity 0/1
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hs.rsrc, hs.voff, image * (HIMG * 16) + i * 4096, kSc1));
    };

    // ---- cell update ownership: pair p = tid + 256 i  ->  unit u = p % UNITS (fastest: conflict-free LDS reads),
    //      row = p / UNITS.  In the accumulator layout (row, column j) sits in tile j / 32, register
    //      (row & 3) + 4 (row >> 3), lane (j & 31) + 32 ((row >> 2) & 1).
    int prow[NP], pk[NP], pred[NP][4];
    float c0[NP], c1[NP];
    float bias0[BIAS_REGS ? NP : 1][4], bias1[BIAS_REGS ? NP : 1][4];
    float wfc0[SEQ ? 1 : NP], wfc1[SEQ ? 1 : NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = tid + 256 * i;
        const int u = p % UNITS, row = p / UNITS;
        prow[i] = row;
        pk[i] = cs * UNITS + u;
        c0[i] = 0.f; c1[i] = 0.f;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const int j = gate * UNITS + u;
            pred[i][gate] = (((j >> 5) * 16) + (row & 3) + 4 * (row >> 3)) * 64 + (j & 31) + 32 * ((row >> 2) & 1);
            if constexpr (BIAS_REGS) {
                bias0[i][gate] = w.bias[gate * HID + pk[i]];
                bias1[i][gate] = w.bias[4 * HID + gate * HID + pk[i]];
            }
        }
        if constexpr (!SEQ) {
            wfc0[i] = w.wfc[pk[i]];
            wfc1[i] = w.wfc[HID + pk[i]];
        }
    }
    if constexpr (!BIAS_REGS) { bias0[0][0] = 0.f; bias1[0][0] = 0.f; }
    if constexpr (SEQ) { wfc0[0] = 0.f; wfc1[0] = 0.f; }

    // partial tiles of the 4 waves -> LDS; afterwards red_sum(idx) is the full pre-activation
    auto publish_tiles = [&](f32x16 (&acc)[NT]) {
        __syncthreads();                       // previous readers of `red` are done
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NT + n) * 16 + r) * 64 + lane] = acc[n][r];
        __syncthreads();
    };
    auto red_sum = [&](int idx) -> float {
        return red[idx] + red[idx + NT * 1024] + red[idx + 2 * NT * 1024] + red[idx + 3 * NT * 1024];
    };

    // returns false (to every thread) once the launch is aborted: a peer never arrived (lstm_common.h: xchg_wait)
    auto inter_wg_barrier = [&](unsigned target) -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!xchg_wait(bar, target, a.coop_abort, a.coop_err)) abort_s = 1;
        }
        __syncthreads();
        return abort_s == 0;
    };
    auto fc_epilogue = [&](int t_done) {     // workgroup cs == 0 sums the S partials of step t_done in a fixed order
        if (cs == 0 && tid < 64) {
            const int row = tid & 31, o = tid >> 5;
            const RowDesc rd = rows_s[row];
            const float* part = fcp + (size_t)(t_done & 1) * S * 64;
            float sum = w.bfc[o];
            for (int p = 0; p < S; ++p) sum += xchg_load(part + p * 64 + o * 32 + row);
            if (rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };

    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        chaos_delay(a.coop_chaos, t, 0);
        // prefetch x(t+1)
        float xr[NG];
        NormMD mdn = md;
        const bool have_next = t + 1 < Tp;
        if (have_next) {
            if (md_t) mdn = md_t[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = x_load(i, t + 1);
        }

        f32x16 acc[NT];
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        if constexpr (WREG)
            coop_layer_resident<NT, G0W, KGXP / 4>(acc, bw0, [&](int i) -> float4 { return Xw[i * 256]; },
                                                   [&](int i) -> float4 { return hload(prv, i); });
        else
            coop_layer<NT, G0W, KGXP / 4, 0>(acc, ws, [&](int i) -> float4 { return Xw[i * 256]; },
                                             [&](int i) -> float4 { return hload(prv, i); });
        publish_tiles(acc);
        {
            float* img = reinterpret_cast<float*>(h0img[cur]);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float bi, bf, bg, bo;
                if constexpr (BIAS_REGS) { bi = bias0[i][0]; bf = bias0[i][1]; bg = bias0[i][2]; bo = bias0[i][3]; }
                else { bi = w.bias[pk[i]]; bf = w.bias[HID + pk[i]]; bg = w.bias[2 * HID + pk[i]]; bo = w.bias[3 * HID + pk[i]]; }
                float hval;
                if constexpr (GRU) {
                    const float rg = fast_sigmoid(red_sum(pred[i][0]) + bi);
                    const float zg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                    const float ng = fast_tanh(red_sum(pred[i][2]) + bg + rg * (red_sum(pred[i][3]) + bo));
                    hval = ng + zg * (c0[i] - ng);
                    c0[i] = hval;
                } else {
                    const float ig = fast_sigmoid(red_sum(pred[i][0]) + bi);
                    const float fg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                    const float gg = fast_tanh(red_sum(pred[i][2]) + bg);
                    const float og = fast_sigmoid(red_sum(pred[i][3]) + bo);
                    const float cn = fg * c0[i] + ig * gg;
                    c0[i] = cn;
                    hval = og * fast_tanh(cn);
                }
                if (a.coop_corrupt != 0 && rt == 0 && pk[i] == 0 && prow[i] == 0 && t + 1 == a.coop_corrupt) hval += 1.0f;     // test hook: published value only
                xchg_store(img + a_frag_index(prow[i], pk[i]), hval);
            }
        }
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = x_valid(i) ? (xr[i] - mdn.m) / mdn.d : 0.0f;
        }
        chaos_delay(a.coop_chaos, t, 1);
        if (!inter_wg_barrier((unsigned)S * (unsigned)(t + 1))) return;   // h0_t, h1_{t-1} and the FC partials of step t-1 are now visible
        chaos_delay(a.coop_chaos, t, 2);

        if constexpr (!SEQ) { if (t > 0) fc_epilogue(t - 1); }

        // ---------------- layer 1: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        if constexpr (WREG)
            coop_layer_resident<NT, G1W, KGH / 4>(acc, bw1, [&](int i) -> float4 { return hload(2 + prv, i); },
                                                  [&](int i) -> float4 { return hload(cur, i); });
        else
            coop_layer<NT, G1W, KGH / 4, G0W>(acc, ws, [&](int i) -> float4 { return hload(2 + prv, i); },
                                              [&](int i) -> float4 { return hload(cur, i); });
        publish_tiles(acc);
        {
            float* img = reinterpret_cast<float*>(h1img[cur]);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                float bi, bf, bg, bo;
                if constexpr (BIAS_REGS) { bi = bias1[i][0]; bf = bias1[i][1]; bg = bias1[i][2]; bo = bias1[i][3]; }
                else {
                    bi = w.bias[4 * HID + pk[i]]; bf = w.bias[5 * HID + pk[i]];
                    bg = w.bias[6 * HID + pk[i]]; bo = w.bias[7 * HID + pk[i]];
                }
                float h;
                if constexpr (GRU) {
                    const float rg = fast_sigmoid(red_sum(pred[i][0]) + bi);
                    const float zg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                    const float ng = fast_tanh(red_sum(pred[i][2]) + bg + rg * (red_sum(pred[i][3]) + bo));
                    h = ng + zg * (c1[i] - ng);
                    c1[i] = h;
                } else {
                    const float ig = fast_sigmoid(red_sum(pred[i][0]) + bi);
                    const float fg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                    const float gg = fast_tanh(red_sum(pred[i][2]) + bg);
                    const float og = fast_sigmoid(red_sum(pred[i][3]) + bo);
                    const float cn = fg * c1[i] + ig * gg;
                    c1[i] = cn;
                    h = og * fast_tanh(cn);
                }
                xchg_store(img + a_frag_index(prow[i], pk[i]), h);
                if constexpr (SEQ) {
                    const RowDesc rd = rows_s[prow[i]];
                    if (rd.valid) a.seq_out[((size_t)rd.b * Tp + t) * HID + pk[i]] = h;
                } else {
                    // partial Linear over this workgroup's units: reduce over the UNITS lanes that share the row
                    float p0 = h * wfc0[i], p1 = h * wfc1[i];
#pragma unroll
                    for (int m = UNITS / 2; m > 0; m >>= 1) { p0 += __shfl_xor(p0, m); p1 += __shfl_xor(p1, m); }
                    if ((tid & (UNITS - 1)) == 0) {
                        float* part = fcp + ((size_t)cur * S + cs) * 64;
                        xchg_store(part + prow[i], p0);
                        xchg_store(part + 32 + prow[i], p1);
                    }
                }
            }
        }
    }
    if constexpr (!SEQ) {
        // last step's Linear: one more barrier so that every partial of step Tp-1 is visible
        if (!inter_wg_barrier((unsigned)S * (unsigned)(Tp + 1))) return;
        fc_epilogue(Tp - 1);
    }
}

// ------------------------------------------------------------------------------------------------
// Layer-skewed schedule of the same kernel (sub-band model only).  In lstm2_coop_kernel a step is a serial chain
//     L0 MFMAs -> cell 0 -> publish h0_t -> [inter-workgroup barrier] -> L1 MFMAs -> cell 1 -> publish h1_t
// and the barrier's round trips (drain, arrive, poll) sit in the middle of it with nothing to overlap.  Layer 0 of step t+1
// and layer 1 of step t only need what was published up to h0_t / h1_{t-1}, so here every workgroup runs
//     A_0,  [A_1, C_0],  [A_2, C_1],  ...,  [A_{T-1}, C_{T-2}],  C_{T-1}          A_t = layer 0 of step t, C_t = layer 1
// with TWO arrival counters per row tile: b0 counts finished A phases, b1 finished C phases.  A_t waits for b0 >= S t (all of
// h0_{t-1}), C_t for b0 >= S (t+1) (h0_t: the very wait A_{t+1} just passed) and b1 >= S t (h1_{t-1}).  Every wait is for an
// arrival that happened one whole phase earlier - the peers arrived on b0 before their C phase, on b1 before their next A
// phase - so in lockstep the poll finds the counter complete on its first read.  Buffers: h0_t is read by A_{t+1} AND C_t,
// and A_{t+2} (which overwrites the image of parity t in a double buffer) may start in a fast workgroup while a slow one is
// still in C_t, so h0 cycles through THREE images (A_{t+3} starts only after every workgroup finished A_{t+2}, which follows
// C_t); h1_t is read by C_{t+1} only: two images; the Linear partials of step t are summed by slice 0 after it passed the
// b1 wait of C_{t+1}: two buffers.  Same arithmetic and summation order as lstm2_coop_kernel: results are bit-identical.
template <int HID, int KX, int UNITS, bool GRU>
__global__ __launch_bounds__(256) void lstm2_coop_skew_kernel(LstmWeights w, LstmArgs a) {
    constexpr int NT = UNITS / 8, NP = UNITS / 8;
    constexpr int KGX = KX / 8, KGH = HID / 8;
    constexpr int KGXP = (KGX + 3) / 4 * 4;
    constexpr int G0W = (KGXP + KGH) / 4, G1W = KGH / 2;
    constexpr int S = HID / UNITS;
    constexpr int HIMG = KGH * 64;
    constexpr int FCP4 = 2 * (HID / 8) * 16;
    constexpr bool BIAS_REGS = UNITS <= 32;
    constexpr bool WREG = NT * (G0W + G1W) * 4 <= 232;
    static_assert(KX <= 64, "gathered sub-band input");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);                     // [KGXP][64] A image of x_t
    float* red = reinterpret_cast<float*>(Xs + KGXP * 64);                // [4 waves][NT][16][64]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(red + 4 * NT * 16 * 64); // [32]
    __shared__ int abort_s;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rt = blockIdx.x / S, cs = blockIdx.x % S;
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, a.num_tiles, a.coop_xcd, rt, cs)) return;
    const int slot0 = rt * 32;
    const int Tp = a.Tp;

    float4* hx = reinterpret_cast<float4*>(a.coop_hx) + (size_t)rt * coop_tile_f4(HID);
    // float4 offsets of the images inside the tile's region (lstm_common.h: coop_tile_f4)
    auto h0off = [](int m3) -> int { return m3 < 2 ? m3 * HIMG : 4 * HIMG + FCP4; };
    auto h1off = [](int par) -> int { return (2 + par) * HIMG; };
    float* fcp = reinterpret_cast<float*>(hx + 4 * HIMG);                  // [2][S][64]
    unsigned* bar0 = FSNP_COOP_BAR(a, rt, 0);
    unsigned* bar1 = FSNP_COOP_BAR(a, rt, 1);

    if (tid == 0) abort_s = 0;
    for (int i = tid; i < KGXP * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) rows_s[tid] = a.rows[slot0 + tid];
    __syncthreads();

    // ---- input plan: thread owns row = tid & 31, features j = (tid >> 5) + 8 i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    constexpr int NG = KGX;
    const int grow = tid & 31, jrow = tid >> 5;
    int goff[NG];
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_t = nullptr;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = jrow + 8 * i;
            int off = -1;
            if (rd.valid && j < w.NIN) {
                if (dense) off = rd.b * Tp * gstep + j;
                else off = sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            goff[i] = off;
        }
        if (rd.valid) {
            if (a.md_seq != nullptr) md_t = a.md_seq + (size_t)rd.b * Tp;
            else if (!dense && a.md_row != nullptr) md_t = a.md_row + (size_t)(slot0 + grow) * Tp;
            else if (!dense) md = a.md_utt[rd.b];
        }
    }
    auto x_load = [&](int i, int t) -> float { return goff[i] >= 0 ? gbase[goff[i] + t * gstep] : 0.0f; };
    const int xdst0 = a_frag_index(grow, jrow);
    float* Xf = reinterpret_cast<float*>(Xs);
    {
        const NormMD m0 = md_t ? md_t[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[i] >= 0 ? (x_load(i, 0) - m0.m) / m0.d : 0.0f;
    }

    CoopStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(w.wpack) + (size_t)(cs * 4 + wave) * (G0W + G1W) * NT * 256, 0, (G0W + G1W) * NT * 1024, 0x00020000);
    ws.voff = lane * 16;
    float4 bw0[WREG ? G0W : 1][NT], bw1[WREG ? G1W : 1][NT];
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < G0W; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) bw0[i][n] = coop_wload<NT>(ws, i, n);
#pragma unroll
        for (int i = 0; i < G1W; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n) bw1[i][n] = coop_wload<NT>(ws, G0W + i, n);
    }
    const float4* Xw = Xs + wave * 64 + lane;
    CoopStream hs;
    hs.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(hx), 0, coop_tile_f4(HID) * 16, 0x00020000);
    hs.voff = (wave * 64 + lane) * 16;
    auto hload = [&](int off_f4, int i) -> float4 {      // local k-group i (global 4 i + wave) of the image at float4 offset off_f4
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hs.rsrc, hs.voff, off_f4 * 16 + i * 4096, kSc1));
    };

    int prow[NP], pk[NP], pred[NP][4];
    float c0[NP], c1[NP];
    float bias0[BIAS_REGS ? NP : 1][4], bias1[BIAS_REGS ? NP : 1][4];
    float wfc0[NP], wfc1[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = tid + 256 * i;
        const int u = p % UNITS, row = p / UNITS;
        prow[i] = row;
        pk[i] = cs * UNITS + u;
        c0[i] = 0.f; c1[i] = 0.f;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const int j = gate * UNITS + u;
            pred[i][gate] = (((j >> 5) * 16) + (row & 3) + 4 * (row >> 3)) * 64 + (j & 31) + 32 * ((row >> 2) & 1);
            if constexpr (BIAS_REGS) {
                bias0[i][gate] = w.bias[gate * HID + pk[i]];
                bias1[i][gate] = w.bias[4 * HID + gate * HID + pk[i]];
            }
        }
        wfc0[i] = w.wfc[pk[i]];
        wfc1[i] = w.wfc[HID + pk[i]];
    }
    if constexpr (!BIAS_REGS) { bias0[0][0] = 0.f; bias1[0][0] = 0.f; }

    auto publish_tiles = [&](f32x16 (&acc)[NT]) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NT + n) * 16 + r) * 64 + lane] = acc[n][r];
        __syncthreads();
    };
    auto red_sum = [&](int idx) -> float { return red[idx] + red[idx + NT * 1024] + red[idx + 2 * NT * 1024] + red[idx + 3 * NT * 1024]; };
    // one layer's cell update of this thread's NP (row, unit) pairs from the summed tiles; returns h through `emit`
    auto cell = [&](float (&c)[NP], const float (&bias)[BIAS_REGS ? NP : 1][4], int layer, auto emit) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float bi, bf, bg, bo;
            if constexpr (BIAS_REGS) { bi = bias[i][0]; bf = bias[i][1]; bg = bias[i][2]; bo = bias[i][3]; }
            else {
                const float* b = w.bias + layer * 4 * HID + pk[i];
                bi = b[0]; bf = b[HID]; bg = b[2 * HID]; bo = b[3 * HID];
            }
            float hval;
            if constexpr (GRU) {
                const float rg = fast_sigmoid(red_sum(pred[i][0]) + bi);
                const float zg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                const float ng = fast_tanh(red_sum(pred[i][2]) + bg + rg * (red_sum(pred[i][3]) + bo));
                hval = ng + zg * (c[i] - ng);
                c[i] = hval;
            } else {
                const float ig = fast_sigmoid(red_sum(pred[i][0]) + bi);
                const float fg = fast_sigmoid(red_sum(pred[i][1]) + bf);
                const float gg = fast_tanh(red_sum(pred[i][2]) + bg);
                const float og = fast_sigmoid(red_sum(pred[i][3]) + bo);
                const float cn = fg * c[i] + ig * gg;
                c[i] = cn;
                hval = og * fast_tanh(cn);
            }
            emit(i, hval);
        }
    };
    // split-phase barrier: arrive now, wait a phase later
    auto arrive = [&](unsigned* bar) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // `early` = the counter as thread 0 read it at the end of the previous MFMA phase (poll_early): in lockstep it is already
    // complete, and the wait costs one __syncthreads instead of a fabric round trip
    auto poll_early = [&](unsigned* bar) -> unsigned {
        return tid == 0 ? __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    };
    auto wait_for = [&](unsigned* bar, unsigned target, unsigned early) -> bool {
        if (tid == 0 && early < target && !xchg_wait(bar, target, a.coop_abort, a.coop_err)) abort_s = 1;
        __syncthreads();
        return abort_s == 0;
    };
    auto fc_epilogue = [&](int t_done) {     // slice 0 sums the S partials of step t_done in a fixed order
        if (cs == 0 && tid < 64) {
            const int row = tid & 31, o = tid >> 5;
            const RowDesc rd = rows_s[row];
            const float* part = fcp + (size_t)(t_done & 1) * S * 64;
            float sum = w.bfc[o];
            for (int p = 0; p < S; ++p) sum += xchg_load(part + p * 64 + o * 32 + row);
            if (rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };

    // A_t: layer 0 of step t.  m3 = t % 3, pm3 = (t - 1) % 3
    unsigned early0 = 0, early1 = 0;                    // counters as read ahead of the next waits
    // operands of the first groups of the NEXT layer-1 phase, issued before the hand-off that ends a layer-0 phase
    CoopA<NT, G1W> ca;
    CoopB<NT, G1W> cb;
    // A_t; with_c: C_{t-1} follows - its b1 wait and the prefill of its first operand groups happen in here, between the
    // stores of h0_t and the drain + arrival, so that their round trips overlap.  Returns false once the launch is aborted.
    auto phase_a = [&](int t, int m3, int pm3, bool with_c) -> bool {
        chaos_delay(a.coop_chaos, t, 0);
        float xr[NG];
        NormMD mdn = md;
        const bool have_next = t + 1 < Tp;
        if (have_next) {
            if (md_t) mdn = md_t[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = x_load(i, t + 1);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        const int hprev = h0off(pm3);
        if constexpr (WREG)
            coop_layer_resident<NT, G0W, KGXP / 4>(acc, bw0, [&](int i) -> float4 { return Xw[i * 256]; },
                                                   [&](int i) -> float4 { return hload(hprev, i); });
        else
            coop_layer<NT, G0W, KGXP / 4, 0>(acc, ws, [&](int i) -> float4 { return Xw[i * 256]; },
                                             [&](int i) -> float4 { return hload(hprev, i); });
        if (with_c) early1 = poll_early(bar1);           // for the wait in front of the C phase that follows
        publish_tiles(acc);
        float* img = reinterpret_cast<float*>(hx + h0off(m3));
        cell(c0, bias0, 0, [&](int i, float hval) {
            if (a.coop_corrupt != 0 && rt == 0 && pk[i] == 0 && prow[i] == 0 && t + 1 == a.coop_corrupt) hval += 1.0f;     // test hook: published value only
            xchg_store(img + a_frag_index(prow[i], pk[i]), hval);
        });
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
        }
        if (with_c) {
            if (!wait_for(bar1, (unsigned)S * (unsigned)(t - 1), early1)) return false;   // h1_{t-2}, Linear partials of step t-2
            if constexpr (!WREG) {
                const int h1p = h1off(t & 1), h0c = h0off(pm3);      // C_{t-1} reads h1_{t-2} (parity t & 1) and h0_{t-1}
                coop_layer_prefill<NT, G1W, KGH / 4, G0W>(ca, cb, ws, [&](int i) -> float4 { return hload(h1p, i); },
                                                          [&](int i) -> float4 { return hload(h0c, i); });
            }
        }
        chaos_delay(a.coop_chaos, t, 1);
        arrive(bar0);
        return true;
    };
    // C_t: layer 1 of step t over [h1_{t-1} | h0_t]
    auto phase_c = [&](int t, int m3, bool prefilled) {
        const int cur = t & 1, prv = cur ^ 1;
        chaos_delay(a.coop_chaos, t, 2);
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        const int h1p = h1off(prv), h0c = h0off(m3);
        if constexpr (WREG)
            coop_layer_resident<NT, G1W, KGH / 4>(acc, bw1, [&](int i) -> float4 { return hload(h1p, i); },
                                                  [&](int i) -> float4 { return hload(h0c, i); });
        else {
            if (!prefilled)
                coop_layer_prefill<NT, G1W, KGH / 4, G0W>(ca, cb, ws, [&](int i) -> float4 { return hload(h1p, i); },
                                                          [&](int i) -> float4 { return hload(h0c, i); });
            coop_layer_run<NT, G1W, KGH / 4, G0W>(acc, ca, cb, ws, [&](int i) -> float4 { return hload(h1p, i); },
                                                  [&](int i) -> float4 { return hload(h0c, i); });
        }
        early0 = poll_early(bar0);                       // for the wait in front of the next A phase
        publish_tiles(acc);
        float* img = reinterpret_cast<float*>(hx + h1off(cur));
        cell(c1, bias1, 1, [&](int i, float h) {
            xchg_store(img + a_frag_index(prow[i], pk[i]), h);
            float p0 = h * wfc0[i], p1 = h * wfc1[i];                 // partial Linear over this workgroup's units
#pragma unroll
            for (int m = UNITS / 2; m > 0; m >>= 1) { p0 += __shfl_xor(p0, m); p1 += __shfl_xor(p1, m); }
            if ((tid & (UNITS - 1)) == 0) {
                float* part = fcp + ((size_t)cur * S + cs) * 64;
                xchg_store(part + prow[i], p0);
                xchg_store(part + 32 + prow[i], p1);
            }
        });
        chaos_delay(a.coop_chaos, t, 3);
        arrive(bar1);
    };

    __syncthreads();
    (void)phase_a(0, 0, 2, false);                      // h0_{-1} = the (zeroed) third image
    int m3 = 1, pm3 = 0;                                // t % 3, (t - 1) % 3 for t = 1
    for (int t = 1; t < Tp; ++t) {
        if (!wait_for(bar0, (unsigned)S * (unsigned)t, early0)) return;            // h0_{t-1} published by every slice
        if (!phase_a(t, m3, pm3, true)) return;                                    // (waits for b1 >= S (t-1) inside)
        if (t >= 2) fc_epilogue(t - 2);
        phase_c(t - 1, pm3, !WREG);
        pm3 = m3;
        m3 = m3 == 2 ? 0 : m3 + 1;
    }
    if (!wait_for(bar0, (unsigned)S * (unsigned)Tp, early0)) return;
    if (!wait_for(bar1, (unsigned)S * (unsigned)(Tp - 1), Tp >= 2 ? poll_early(bar1) : 0u)) return;
    if (Tp >= 2) fc_epilogue(Tp - 2);
    phase_c(Tp - 1, pm3, false);
    if (!wait_for(bar1, (unsigned)S * (unsigned)Tp, 0u)) return;
    fc_epilogue(Tp - 1);
}

// ------------------------------------------------------------------------------------------------
static int coop_kgxp(int KX) { return (KX / 8 + 3) / 4 * 4; }

size_t lstm_coop_pack_floats(int H, int KX, int units) {
    const int S = H / units, NT = units / 8;
    const int GW = (coop_kgxp(KX) + H / 8) / 4 + H / 16;     // local k-groups per wave, both layers
    return (size_t)S * 4 * GW * NT * 64 * 4;
}

// [cs][wave][local k-group i (layer 0: G0W groups, then layer 1: G1W groups)][tile n][lane][k-pair]; local group i of
// wave w is global k-group 4 i + w of its layer: layer 0 = [x (KGXP groups, zero padded) | h0], layer 1 = [h1 | h0].
// Column j = n * 32 + (lane & 31) of the workgroup is gate j / units, hidden unit cs * units + j % units.
void lstm_coop_pack_weights(int H, int NIN, int KX, int units, const float* wih0, const float* whh0, const float* wih1,
                            const float* whh1, float* wpack) {
    const int S = H / units, NT = units / 8;
    const int KGXP = coop_kgxp(KX), KGH = H / 8;
    const int G0W = (KGXP + KGH) / 4, G1W = KGH / 2, GW = G0W + G1W;
    for (int cs = 0; cs < S; ++cs)
        for (int wave = 0; wave < 4; ++wave)
            for (int i = 0; i < GW; ++i)
                for (int n = 0; n < NT; ++n)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int p = 0; p < 4; ++p) {
                            const int j = n * 32 + (lane & 31);
                            const int gate = j / units, u = j % units;
                            const int wrow = gate * H + cs * units + u;
                            float v = 0.0f;
                            if (i < G0W) {
                                const int g = 4 * i + wave;
                                if (g < KGXP) {
                                    const int k = 8 * g + 2 * p + (lane >> 5);
                                    if (k < NIN) v = wih0[(size_t)wrow * NIN + k];
                                } else {
                                    const int k = 8 * (g - KGXP) + 2 * p + (lane >> 5);
                                    v = whh0[(size_t)wrow * H + k];
                                }
                            } else {
                                const int g = 4 * (i - G0W) + wave;
                                const int k = 8 * g + 2 * p + (lane >> 5);
                                if (k < H) v = whh1[(size_t)wrow * H + k];
                                else v = wih1[(size_t)wrow * H + (k - H)];
                            }
                            wpack[(((((size_t)cs * 4 + wave) * GW + i) * NT + n) * 64 + lane) * 4 + p] = v;
                        }
}

// per row tile: 4 h images + Linear partials sized for the finest split (units = 8)
size_t lstm_coop_exchange_bytes(int H, int row_tiles) {
    return (size_t)row_tiles * (size_t)coop_tile_f4(H) * 16;
}

// dynamic LDS a launch may claim to keep a CU to itself (LstmArgs::coop_own_cu): 160 KiB per CU minus the static __shared__ words
constexpr size_t kOwnCuLds = 160 * 1024 - 256;

// occ != nullptr: do not launch; report how many workgroups of this instantiation fit one CU at once
template <int HID, int KX, int UNITS, bool SEQ, bool GRU>
static void launch_coop_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    constexpr int S = HID / UNITS, NT = UNITS / 8;
    const size_t smem_need = (size_t)coop_kgxp(KX) * 64 * 16 + (size_t)4 * NT * 16 * 64 * 4 + 32 * sizeof(RowDesc);
    auto kern = lstm2_coop_kernel<HID, KX, UNITS, SEQ, GRU>;
    static PerDeviceOnce attr_once;            // the attribute is per device: one process may drive several GPUs
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOwnCuLds); });
    if (occ) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(kern), 256, smem_need) != hipSuccess) *occ = 0;
        return;
    }
    const size_t smem = a.coop_own_cu > 0 && (size_t)a.coop_own_cu > smem_need ? (size_t)a.coop_own_cu : smem_need;
    LstmWeights wv = w;
    wv.wpack = w.wpack_coop[coop_units_index(UNITS)];
    const int grid = a.coop_xcd ? 8 * xcd_local_blocks_per_xcd(S, a.num_tiles, a.coop_xcd) : a.num_tiles * S;
    if constexpr (!SEQ) {
        // layer-skewed schedule (lstm2_coop_skew_kernel), same launch shape.  Measured (profiles/r02_column_split.md): 16 units per
        // workgroup and up: 9 tiles 19.8 -> 18.0 us per step, 17 tiles 28.1 -> 25.8, 41 tiles 53.0 -> 48.4.  At 8 units it did not
        // pay in round 2 (1 tile 11.0 -> 10.8, 5 tiles 16 -> 28) - its two arrival counters per tile sat in the one cache line all
        // counters shared; with a line per counter (round 3, profiles/r03_column_split.md section 8) it does: 1 tile 9.7 -> 8.8,
        // 5 tiles 10.4 -> 9.1.
        if (a.coop_skew) {
            auto skew = lstm2_coop_skew_kernel<HID, KX, UNITS, GRU>;
            static PerDeviceOnce skew_once;
            skew_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skew), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOwnCuLds); });
            hipLaunchKernelGGL(skew, dim3(grid), dim3(256), smem, s, wv, a);
            return;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, wv, a);
}

// Finest column split (fewest units per workgroup, >= min_units) whose row_tiles * H / units workgroups are all
// resident at once (one per CU); 0 = none.
int lstm_coop_pick_units(int H, int row_tiles, int num_cus, int min_units) {
    for (int u = 8; u <= 64; u *= 2)
        if (u >= min_units && H % u == 0 && row_tiles * (H / u) <= num_cus) return u;
    return 0;
}

template <int HID, int KX, bool SEQ, bool GRU>
static void launch_coop_units(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ = nullptr) {
    switch (a.coop_units) {
        case 8: launch_coop_inst<HID, KX, 8, SEQ, GRU>(w, a, s, occ); break;
        case 16: launch_coop_inst<HID, KX, 16, SEQ, GRU>(w, a, s, occ); break;
        case 32: launch_coop_inst<HID, KX, 32, SEQ, GRU>(w, a, s, occ); break;
        default:
            if constexpr (!SEQ) launch_coop_inst<HID, KX, 64, SEQ, GRU>(w, a, s, occ);
            else launch_coop_inst<HID, KX, 32, SEQ, GRU>(w, a, s, occ);
            break;
    }
}

// sub-band model: H = 384, x gathered (or dense [seq][t][NIN]), fused Linear(384, 2); w.gru selects nn.GRU
template <int HID>
static void dispatch_lstm_coop_h(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {   // other hidden sizes: K = 40 / 64
    if (w.KX == 64) {
        if (w.gru) launch_coop_units<HID, 64, false, true>(w, a, s, occ);
        else launch_coop_units<HID, 64, false, false>(w, a, s, occ);
        return;
    }
    if (w.gru) launch_coop_units<HID, 40, false, true>(w, a, s, occ);
    else launch_coop_units<HID, 40, false, false>(w, a, s, occ);
}
static void dispatch_lstm_coop(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    if (w.H == 256) { dispatch_lstm_coop_h<256>(w, a, s, occ); return; }      // sb_model_hidden_size 256 / 512
    if (w.H == 512) { dispatch_lstm_coop_h<512>(w, a, s, occ); return; }
    if (w.KX == 64) {                  // sub-band inputs of 41..64 features
        if (w.gru) launch_coop_units<384, 64, false, true>(w, a, s, occ);
        else launch_coop_units<384, 64, false, false>(w, a, s, occ);
        return;
    }
    if (w.gru) launch_coop_units<384, 40, false, true>(w, a, s, occ);
    else launch_coop_units<384, 40, false, false>(w, a, s, occ);
}
void launch_lstm_coop(const LstmWeights& w, const LstmArgs& a, hipStream_t s) { dispatch_lstm_coop(w, a, s, nullptr); }
// workgroups of the sub-band K-split kernel at `units` hidden units per workgroup that fit one CU at once (0 = unknown)
int lstm_coop_occupancy(const LstmWeights& w, int units) {
    LstmArgs a{};
    a.coop_units = units;
    int occ = 0;
    dispatch_lstm_coop(w, a, nullptr, &occ);
    return occ;
}

// full-band model of the original FullSubNet: H = 512, dense input rows of <= 264 features, h1 sequence out
void launch_lstm_coop_seq(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.gru) launch_coop_units<512, 264, true, true>(w, a, s);
    else launch_coop_units<512, 264, true, false>(w, a, s);
}

}  // namespace fsnp
