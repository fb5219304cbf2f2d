// lstm_pp.hip - "ping-pong" K-split two-layer LSTM + Linear for SMALL row counts on gfx950 (round 3).
//
// Same arithmetic as lstm_coop.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123) and the same K-split decomposition at its finest width
// (8 hidden units = one 32-column gate-interleaved accumulator tile per workgroup, S = H / 8 workgroups share a 32-row
// tile, the 4 waves split K, the wave's weights stay in registers for the whole sequence).  What changes is the SCHEDULE:
//
//   * fused phase.  lstm2_coop_kernel runs layer 0 and layer 1 of a step as two MFMA passes with two LDS reductions, and
//     reads h0_t twice (layer 1 of step t, layer 0 of step t + 1).  Between two inter-workgroup barriers a workgroup has
//     exactly [layer 1 of step t, layer 0 of step t + 1] to do, and both multiply h0_t: here that is ONE pass - every
//     k-group of h0_t is loaded once and feeds both accumulators (W_ih1 and W_hh0), one LDS reduction and one cell phase
//     for both layers.  Per step and workgroup: 96 KB of operands instead of 144, 3 workgroup barriers instead of 6.
//   * R independent row tiles per group ("ping-pong").  A K-split step is latency-bound: write-through stores -> drain ->
//     arrive -> poll -> first operand fetch is ~5-6 us against 4 us of MFMA time (profiles/r02_column_split.md).  A group
//     owns R row tiles and works on them in turn, so the hand-off of tile A is in flight while tile B computes; the
//     arrival of A is issued by wave 0 once B's first operand loads have been issued (vmcnt is in order: the stores are
//     older), and the counter of the next tile is polled before the cell phase of the current one.
//   * 16-byte write-through stores.  The cell phase stages the workgroup's 32 x 8 slice of h0 / h1 in LDS in A-fragment
//     order (it is exactly one k-group = 1 KB) and wave 0 publishes it with ONE 16-byte sc1 store per lane instead of
//     512 four-byte ones (MI355X_MICROARCH.md: a dword sc1 store costs ~6x a dwordx4 one per byte).  Only wave 0 stores,
//     so only wave 0 drains and arrives: no workgroup barrier between the drain and the arrival.
//   * the Linear(H, 2) partial sums of row q are added up (in the same fixed order as lstm2_coop_kernel) by workgroup q of
//     the group, with the loads issued a phase ahead - no workgroup is slower than the others.
// Same exchange images, arrival counters, abort protocol and weight pack (units = 8) as lstm_coop.hip; every accumulator
// sees the same k order and every sum the same operand order: results are BIT-IDENTICAL to lstm2_coop_kernel
// (tests/test_gpu_parity.py::test_ping_pong_k_split_equals_serial_schedule).
//
// Phases of one row tile (h0img / h1img / fcp are double buffered by parity; bar counts arrivals, S per phase):
//   phase -1     : acc0 = W_ih0 x_0                          -> cell 0 -> h0_0 -> h0img[0]                          arrive
//   phase t >= 0 : wait bar >= S (t + 1)
//                  acc1 = W_hh1 h1_{t-1} + W_ih1 h0_t,  acc0 = W_ih0 x_{t+1} + W_hh0 h0_t
//                  -> cell 1 -> h1_t -> h1img[t & 1], Linear partials -> fcp[t & 1];  cell 0 -> h0_{t+1} -> h0img[(t+1) & 1]   arrive
//                  (workgroup q < 32: out[row q][t - 1] = bias + sum of the S partials of step t - 1)
//   final        : wait bar >= S (Tp + 1); out[.][Tp - 1]
// A phase writes only buffers whose last readers finished a phase earlier (they arrived), so two images per layer suffice.
#include <type_traits>
#include <utility>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <typename F, int... I>
__device__ __forceinline__ void pp_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void pp_static_for(F&& f) { pp_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct PpStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};

__device__ __forceinline__ void pp_mfma4(f32x16& acc, const float4& a, const float4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

constexpr int pp_kgxp(int KX) { return (KX / 8 + 3) / 4 * 4; }
// dynamic LDS of one workgroup (bytes): R x-images, two reduction buffers, staging, Linear sums, row descriptors, flags
constexpr size_t pp_smem_bytes(int HID, int KX, int R) {
    return (size_t)R * pp_kgxp(KX) * 64 * 16 + 2 * 4 * 16 * 64 * 4 + (2 * 64 + 16) * 16 + 2 * (HID / 8) * 4 + (size_t)R * 32 * sizeof(RowDesc) + 64;
}

}  // namespace

template <int HID, int KX, int R>
__global__ __launch_bounds__(256) void lstm2_coop_pp_kernel(LstmWeights w, LstmArgs a) {
    constexpr int KGX = KX / 8, KGH = HID / 8;
    constexpr int KGXP = pp_kgxp(KX);
    constexpr int XW = KGXP / 4;                   // x k-groups per wave
    constexpr int HW = KGH / 4;                    // k-groups of one h image per wave
    constexpr int G0W = XW + HW, G1W = 2 * HW;     // the wave's weight groups: layer 0 [x | h0], layer 1 [h1 | h0]  (lstm_coop_pack_weights)
    constexpr int S = HID / 8;
    constexpr int HIMG = KGH * 64;                 // float4 per exchange image
    constexpr int NG = KGX;
    constexpr int D = 2 * HW;                      // A fragments in flight per wave: the whole tile-phase (the weights live in AGPRs)
    static_assert(KX <= 64, "gathered sub-band input");
    static_assert(S >= 32 && 2 * S <= 128, "one workgroup per output row, Linear loads by the first two waves");
    static_assert((G0W + G1W) * 4 <= 232, "the wave's weights must fit the register file");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);                        // [R][KGXP][64] A images of x
    float* red = reinterpret_cast<float*>(Xs + R * KGXP * 64);               // [2 layers][4 waves][16][64]
    float4* stage = reinterpret_cast<float4*>(red + 2 * 4096);               // [h0 | h1][64] + Linear partials [16]
    float* fc_red = reinterpret_cast<float*>(stage + 2 * 64 + 16);           // [2][S]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(fc_red + 2 * S);            // [R][32]
    int* flags = reinterpret_cast<int*>(rows_s + R * 32);                    // [0] abort, [1] next tile ready

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x / S, cs = blockIdx.x % S;
    const int tile0 = grp * R;
    const int nt = a.num_tiles - tile0 < R ? a.num_tiles - tile0 : R;         // row tiles of this group (>= 1)
    const int Tp = a.Tp;
    constexpr int TILE_BYTES = coop_tile_f4(HID) * 16;

    if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; }
    for (int i = tid; i < R * KGXP * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tid < 32) rows_s[r * 32 + tid] = r < nt ? a.rows[(tile0 + r) * 32 + tid] : RowDesc{0, 0, 0, 0};
    __syncthreads();

    // ---- input plan (as lstm2_coop_kernel): thread owns row tid & 31, features (tid >> 5) + 8 i of every tile
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    const int grow = tid & 31, jrow = tid >> 5;
    int goff[R][NG];
    NormMD md[R];
    const NormMD* md_t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const RowDesc rd = rows_s[r * 32 + grow];
        md[r] = NormMD{0.0f, 1.0f};
        md_t[r] = nullptr;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = jrow + 8 * i;
            int off = -1;
            if (rd.valid && j < w.NIN) {
                if (dense) off = rd.b * Tp * gstep + j;
                else off = sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            goff[r][i] = off;
        }
        if (rd.valid) {
            if (a.md_seq != nullptr) md_t[r] = a.md_seq + (size_t)rd.b * Tp;
            else if (!dense && a.md_row != nullptr) md_t[r] = a.md_row + (size_t)((tile0 + r) * 32 + grow) * Tp;
            else if (!dense) md[r] = a.md_utt[rd.b];
        }
    }
    const int xdst0 = a_frag_index(grow, jrow);

    // ---- the wave's weights, resident: [cs][wave][local k-group][lane][4]
    float4 bw0[G0W], bw1[G1W];
    {
        PpStream ws;
        ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(w.wpack) + (size_t)(cs * 4 + wave) * (G0W + G1W) * 256, 0, (G0W + G1W) * 1024, 0x00020000);
        ws.voff = lane * 16;
#pragma unroll
        for (int i = 0; i < G0W; ++i) bw0[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, i * 1024, 0));
#pragma unroll
        for (int i = 0; i < G1W; ++i) bw1[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, (G0W + i) * 1024, 0));
        // pinned to AGPRs (the MFMA takes its B operand from there directly): the VGPRs hold a whole tile-phase of A fragments
#pragma unroll
        for (int i = 0; i < G0W; ++i) asm volatile("" : "+a"(bw0[i].x), "+a"(bw0[i].y), "+a"(bw0[i].z), "+a"(bw0[i].w));
#pragma unroll
        for (int i = 0; i < G1W; ++i) asm volatile("" : "+a"(bw1[i].x), "+a"(bw1[i].y), "+a"(bw1[i].z), "+a"(bw1[i].w));
    }
    // ---- exchange region of this group's tiles: one descriptor, the tile / image / k-group offset is scalar
    PpStream hs;
    hs.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.coop_hx) + (size_t)tile0 * (TILE_BYTES / 4), 0, nt * TILE_BYTES, 0x00020000);
    hs.voff = (wave * 64 + lane) * 16;
    auto hload = [&](int soff) -> float4 {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hs.rsrc, hs.voff, soff, kSc1));
    };
    constexpr int H0OFF = 0, H1OFF = 2 * HIMG * 16, FCOFF = 4 * HIMG * 16;      // byte offsets inside a tile's region (coop_tile_f4)

    // ---- cell ownership (as lstm2_coop_kernel at 8 units): unit u = tid % 8, row = tid / 8
    const int cu = tid & 7, crow = tid >> 3;
    const int pk = cs * 8 + cu;
    int pred[4];
    float bias0[4], bias1[4];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
        const int j = gate * 8 + cu;
        pred[gate] = ((crow & 3) + 4 * (crow >> 3)) * 64 + j + 32 * ((crow >> 2) & 1);
        bias0[gate] = w.bias[gate * HID + pk];
        bias1[gate] = w.bias[4 * HID + gate * HID + pk];
    }
    const float wfc0 = w.wfc[pk], wfc1 = w.wfc[HID + pk];
    const int sdst = ((cu & 1) * 32 + crow) * 4 + (cu >> 1);             // float index inside a staged k-group (a_frag_index)
    float c0[R], c1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { c0[r] = 0.f; c1[r] = 0.f; }

    auto red_sum = [&](const float* rb, int idx) -> float { return rb[idx] + rb[idx + 1024] + rb[idx + 2048] + rb[idx + 3072]; };
    auto cell = [&](const float* rb, const float (&bias)[4], float& c) -> float {
        const float ig = fast_sigmoid(red_sum(rb, pred[0]) + bias[0]);
        const float fg = fast_sigmoid(red_sum(rb, pred[1]) + bias[1]);
        const float gg = fast_tanh(red_sum(rb, pred[2]) + bias[2]);
        const float og = fast_sigmoid(red_sum(rb, pred[3]) + bias[3]);
        const float cn = fg * c + ig * gg;
        c = cn;
        return og * fast_tanh(cn);
    };
    auto x_load = [&](int r, int i, int t) -> float { return goff[r][i] >= 0 ? gbase[goff[r][i] + t * gstep] : 0.0f; };
    auto store16 = [&](const float4& v, int voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), hs.rsrc, voff, soff, kSc1);
    };
    // Linear(H, 2): workgroup q < 32 owns output row q of every tile; lanes [0, 2 S) fetch the S partials of both outputs
    const bool fc_wg = cs < 32;
    const int fc_voff = ((tid % S) * 64 + (tid / S) * 32 + cs) * 4;
    auto fc_finish = [&](int r, int t_done) {          // threads 128, 129: sum in the fixed order of lstm2_coop_kernel, write out
        if (fc_wg && (tid == 128 || tid == 129)) {
            const int o = tid - 128;
            const RowDesc rd = rows_s[r * 32 + cs];
            float sum = w.bfc[o];
            float4 pv[S / 4];
#pragma unroll
            for (int p = 0; p < S / 4; ++p) pv[p] = reinterpret_cast<const float4*>(fc_red + o * S)[p];
#pragma unroll
            for (int p = 0; p < S / 4; ++p) { sum += pv[p].x; sum += pv[p].y; sum += pv[p].z; sum += pv[p].w; }
            if (rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };

    // ---- x_0 of every tile
    float* Xf = reinterpret_cast<float*>(Xs);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const NormMD m0 = md_t[r] ? md_t[r][0] : md[r];
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (goff[r][i] >= 0) Xf[r * KGXP * 256 + xdst0 + i * 256] = (x_load(r, i, 0) - m0.m) / m0.d;
    }
    __syncthreads();

    // ================= phase -1 of every tile: h0_0 =================
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r >= nt) break;
        float xr[NG];
        NormMD mdn = md[r];
        if (Tp > 1) {
            if (md_t[r]) mdn = md_t[r][1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = x_load(r, i, 1);
        }
        f32x16 acc0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc0[q] = 0.0f;
        const float4* Xw = Xs + r * KGXP * 64 + wave * 64 + lane;
#pragma unroll
        for (int i = 0; i < XW; ++i) pp_mfma4(acc0, Xw[i * 256], bw0[i]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc0[q];
        __syncthreads();
        reinterpret_cast<float*>(stage)[sdst] = cell(red, bias0, c0[r]);
        if (Tp > 1) {
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if (goff[r][i] >= 0) Xf[r * KGXP * 256 + xdst0 + i * 256] = (xr[i] - mdn.m) / mdn.d;
        }
        __syncthreads();
        if (wave == 0) {
            store16(stage[lane], (cs * 64 + lane) * 16, r * TILE_BYTES + H0OFF);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(FSNP_COOP_BAR(a, tile0 + r, 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ================= phases 0 .. Tp - 1, the group's tiles in turn =================
    // Software pipeline over tile-phases: the operands of the NEXT tile-phase (A fragments of [h1 | h0], x two steps ahead, the
    // Linear partials) are issued right after the MFMA pass of the current one - before its LDS reduction, cell phase and
    // stores - whenever the poll issued at 3/4 of the pass found the next tile's counter complete ("ready"); otherwise after
    // the stores and a real wait.  ab / xr / fcv always hold what was issued for the upcoming tile-phase.
    // Only the first EARLY fragments are fetched ahead (issuing 24 KB per wave costs the CU's 64 B/clk vector-memory path 0.3 us
    // per 8 fragments - profiles/r03_column_split.md); the last LATE = D - EARLY ones (k-groups of h0_t, needed in the second half
    // of the pass) are issued INSIDE the pass, one per k-group, behind the MFMAs.
    constexpr int LATE = D / 3, EARLY = D - LATE;
    constexpr int ARRIVE_AT = LATE, POLL_AT = D * 3 / 4;
    static_assert(EARLY >= HW && ARRIVE_AT < EARLY, "in-pass fetches are h0 k-groups; the arrival follows them");
    float4 ab[D];
    float xr[NG], fcv = 0.0f;
    NormMD mdn = NormMD{0.0f, 1.0f};
    bool have_x = false;
    auto issue = [&](auto RN, int tt) {
        constexpr int rn = decltype(RN)::value;
        have_x = tt + 2 < Tp;
        mdn = md[rn];
        if (have_x) {
            if (md_t[rn]) mdn = md_t[rn][tt + 2];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = x_load(rn, i, tt + 2);
        }
        if (fc_wg && tt >= 1 && tid < 2 * S)
            fcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hs.rsrc, fc_voff, rn * TILE_BYTES + FCOFF + ((tt & 1) ^ 1) * S * 256, kSc1));
        const int h1base = rn * TILE_BYTES + H1OFF + ((tt & 1) ^ 1) * (HIMG * 16), h0base = rn * TILE_BYTES + H0OFF + (tt & 1) * (HIMG * 16);
#pragma unroll
        for (int k = 0; k < EARLY; ++k) ab[k] = k < HW ? hload(h1base + k * 4096) : hload(h0base + (k - HW) * 4096);
    };
    // optional phase profile (fsnp_debug_pp_profile): thread 0 of workgroup 0 stamps the 100 MHz wall clock at 7 points of every
    // tile-phase + whether the early fetch happened: prof[(t * R + r) * 8 + k]
    unsigned long long* prof = (a.prof != nullptr && blockIdx.x == 0 && tid == 0) ? a.prof : nullptr;
#define FSNP_PP_STAMP(k) do { if (prof) prof[(t * R + r) * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
    bool pending = false, pending_fast = false;       // wave 0: the arrival of the previous tile-phase has not been issued yet
    unsigned* pending_bar = nullptr;
    bool dead = false;                                // the launch was aborted (a peer never arrived)
    if (tid == 0 && !xchg_wait(FSNP_COOP_BAR(a, tile0, 0), (unsigned)S, a.coop_abort, a.coop_err)) flags[0] = 1;
    __syncthreads();
    if (flags[0]) return;
    issue(std::integral_constant<int, 0>{}, 0);
    for (int t = 0; t < Tp && !dead; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        pp_static_for<R>([&](auto RC) {
            constexpr int r = decltype(RC)::value;
            constexpr int rn = r + 1 < R ? r + 1 : 0;
            if (r >= nt || dead) return;
            unsigned* bar = FSNP_COOP_BAR(a, tile0 + r, 0);
            const bool last_tile = r + 1 >= nt;
            const bool has_next = !(last_tile && t + 1 >= Tp);
            const unsigned next_target = (unsigned)S * (unsigned)((last_tile ? t + 1 : t) + 1);
            unsigned* next_bar = FSNP_COOP_BAR(a, tile0 + (last_tile ? 0 : r + 1), 0);
            const bool fc_now = fc_wg && t >= 1;
            // ---- one pass: acc0 = W_ih0 x_{t+1} + W_hh0 h0_t, acc1 = W_hh1 h1_{t-1} + W_ih1 h0_t
            FSNP_PP_STAMP(0);
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
            const float4* Xw = Xs + r * KGXP * 64 + wave * 64 + lane;
#pragma unroll
            for (int i = 0; i < XW; ++i) pp_mfma4(acc0, Xw[i * 256], bw0[i]);
            unsigned early = 0;
            const int h0late = r * TILE_BYTES + H0OFF + cur * (HIMG * 16) + (EARLY - HW) * 4096;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                if (i < HW) {
                    pp_mfma4(acc1, ab[i], bw1[i]);
                } else {
                    pp_mfma4(acc0, ab[i], bw0[XW + i - HW]);
                    pp_mfma4(acc1, ab[i], bw1[i]);
                }
                if (i < LATE) ab[EARLY + i] = hload(h0late + i * 4096);
                if (i == ARRIVE_AT && wave == 0 && pending) {
                    // the previous tile-phase's write-through stores have drained once at most the LATE in-pass fetches (all newer) are
                    // outstanding - or, when this tile-phase's early fragments were issued AFTER the stores, those behind fragment
                    // ARRIVE_AT as well (vmcnt retires in order)
                    if (pending_fast) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LATE) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LATE + EARLY - 1 - ARRIVE_AT) : "memory");
                    if (lane == 0) __hip_atomic_fetch_add(pending_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pending = false;
                }
                if (i == POLL_AT && nt > 1 && has_next && tid == 0)
                    early = __hip_atomic_load(next_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_sched_barrier(0);
            }
            FSNP_PP_STAMP(1);
            if (tid == 0) flags[1] = (nt > 1 && has_next && early >= next_target) ? 1 : 0;
            if (fc_now && tid < 2 * S) fc_red[tid] = fcv;
            __syncthreads();
            const bool ready = flags[1] != 0;
            unsigned early2 = 0;             // not ready: ask again now, look at the answer after the cell phase
            if (!ready && nt > 1 && has_next && tid == 0) early2 = __hip_atomic_load(next_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FSNP_PP_STAMP(2);
            if (prof) prof[(t * R + r) * 8 + 7] = ready ? 1ull : 0ull;       // (+ 2 below if the fetch happened after the cell phase without a wait)
            // what the cell phase still needs of this tile-phase's prefetch
            float xc[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) xc[i] = xr[i];
            const NormMD mdc = mdn;
            const bool have_xc = have_x;
            if (ready) {
                if (last_tile) issue(std::integral_constant<int, 0>{}, t + 1);
                else issue(std::integral_constant<int, rn>{}, t);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                red[(wave * 16 + q) * 64 + lane] = acc0[q];
                red[4096 + (wave * 16 + q) * 64 + lane] = acc1[q];
            }
            __syncthreads();
            FSNP_PP_STAMP(3);
            {
                const float h1 = cell(red + 4096, bias1, c1[r]);
                const float h0n = cell(red, bias0, c0[r]);
                float* sf = reinterpret_cast<float*>(stage);
                sf[sdst] = h0n;
                sf[256 + sdst] = h1;
                float p0 = h1 * wfc0, p1 = h1 * wfc1;                  // partial Linear over this workgroup's 8 units
#pragma unroll
                for (int m = 4; m > 0; m >>= 1) { p0 += __shfl_xor(p0, m); p1 += __shfl_xor(p1, m); }
                if (cu == 0) { sf[512 + crow] = p0; sf[512 + 32 + crow] = p1; }
            }
            if (have_xc) {
#pragma unroll
                for (int i = 0; i < NG; ++i)
                    if (goff[r][i] >= 0) Xf[r * KGXP * 256 + xdst0 + i * 256] = (xc[i] - mdc.m) / mdc.d;
            }
            if (fc_now) fc_finish(r, t - 1);
            if (tid == 0) flags[2] = (!ready && early2 >= next_target) ? 1 : 0;
            FSNP_PP_STAMP(4);
            __syncthreads();
            FSNP_PP_STAMP(5);
            if (prof && !ready && flags[2]) prof[(t * R + r) * 8 + 7] = 2ull;
            // ---- publish: wave 0, 16 bytes per lane, write-through
            if (wave == 0) {
                if (t + 1 < Tp) store16(stage[lane], (cs * 64 + lane) * 16, r * TILE_BYTES + H0OFF + prv * (HIMG * 16));
                store16(stage[64 + lane], (cs * 64 + lane) * 16, r * TILE_BYTES + H1OFF + cur * (HIMG * 16));
                if (lane < 16) store16(stage[128 + lane], lane * 16, r * TILE_BYTES + FCOFF + (cur * S + cs) * 256);
                if (nt == 1) {                   // the next wait is for this very tile: arrive now
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    pending = true;
                    pending_fast = ready;
                    pending_bar = bar;
                }
            }
            if (!ready && has_next) {            // the next tile was not seen complete at the end of the pass ...
                if (flags[2] == 0) {             // ... nor after the cell phase: wait for it for real
                    if (tid == 0 && !xchg_wait(next_bar, next_target, a.coop_abort, a.coop_err)) flags[0] = 1;
                    __syncthreads();
                    if (flags[0]) { dead = true; return; }
                }
                if (last_tile) issue(std::integral_constant<int, 0>{}, t + 1);
                else issue(std::integral_constant<int, rn>{}, t);
            }
            FSNP_PP_STAMP(6);
        });
    }
    if (dead) return;
    if (wave == 0 && pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(pending_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ================= the Linear of the last step =================
    if (!fc_wg) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r >= nt) break;
        __syncthreads();                             // fc_red of the previous tile has been read
        if (tid == 0 && !xchg_wait(FSNP_COOP_BAR(a, tile0 + r, 0), (unsigned)S * (unsigned)(Tp + 1), a.coop_abort, a.coop_err)) flags[0] = 1;
        __syncthreads();
        if (flags[0]) return;
        if (tid < 2 * S)
            fc_red[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hs.rsrc, fc_voff, r * TILE_BYTES + FCOFF + ((Tp - 1) & 1) * S * 256, kSc1));
        __syncthreads();
        fc_finish(r, Tp - 1);
    }
}

// ------------------------------------------------------------------------------------------------
template <int HID, int KX, int R>
static void launch_pp_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    constexpr int S = HID / 8;
    const size_t smem_need = pp_smem_bytes(HID, KX, R);
    auto kern = lstm2_coop_pp_kernel<HID, KX, R>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    if (occ) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(kern), 256, smem_need) != hipSuccess) *occ = 0;
        return;
    }
    const size_t smem = a.coop_own_cu > 0 && (size_t)a.coop_own_cu > smem_need ? (size_t)a.coop_own_cu : smem_need;
    LstmWeights wv = w;
    wv.wpack = w.wpack_coop[0];
    const int groups = cdiv(a.num_tiles, R);
    hipLaunchKernelGGL(kern, dim3(groups * S), dim3(256), smem, s, wv, a);
}

template <int HID, int KX>
static void launch_pp_r(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    switch (a.coop_rows_per_group) {
        case 1: launch_pp_inst<HID, KX, 1>(w, a, s, occ); break;
        case 2: launch_pp_inst<HID, KX, 2>(w, a, s, occ); break;
        case 3: launch_pp_inst<HID, KX, 3>(w, a, s, occ); break;
        default: launch_pp_inst<HID, KX, 4>(w, a, s, occ); break;
    }
}

bool lstm_pp_available(const LstmWeights& w) { return !w.gru && (w.H == 384 || w.H == 256) && (w.KX == 40 || w.KX == 64); }

// a.num_tiles row tiles, a.coop_rows_per_group (1..4) per group of H / 8 workgroups; all cdiv(tiles, R) * H / 8 workgroups co-resident
void launch_lstm_pp(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.H == 256) { if (w.KX == 64) launch_pp_r<256, 64>(w, a, s, nullptr); else launch_pp_r<256, 40>(w, a, s, nullptr); return; }
    if (w.KX == 64) launch_pp_r<384, 64>(w, a, s, nullptr); else launch_pp_r<384, 40>(w, a, s, nullptr);
}

}  // namespace fsnp
