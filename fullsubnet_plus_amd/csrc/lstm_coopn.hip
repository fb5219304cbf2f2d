// lstm_coopn.hip - three-way column-split two-layer LSTM + Linear for MEDIUM row counts on gfx950.
//
// Same arithmetic as lstm.hip / lstm_coop.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123).  It fills the gap between the two:
//   lstm_coop.hip  (K split over 4 waves, 8..64 hidden units per workgroup)  row tiles * 6 <= CUs : <= 42 tiles (B <= 5)
//   lstm.hip       (one 32-row tile per CU, all 1536 gate columns)           needs >= 256 tiles to fill the chip
// With 43..170 row tiles (batches of 6..21 utterances, or the reference's literal drop_band call at B = 32: 128 tiles)
// the row-tile kernel leaves 1/3..5/6 of the CUs idle for ~208 us per step.  Here S = 3 workgroups share the row
// tiles of a GROUP: workgroup (g, cs) owns hidden units [128 cs, 128 cs + 128) of both layers for the R <= 2 row tiles
// g, g + G of its group; wave w owns 32 of those units as 4 gate tiles over the FULL K (no cross-wave reduction: i/f/g/o of
// one (row, unit) share lane and register index, the cell update is lane-local, c stays in registers - the row-tile
// kernel's 12-wave layout spread over three CUs).  h0_t / h1_t travel through the per-tile, double-buffered global
// exchange images of lstm_coop.hip (already in MFMA A-fragment order; consumers read A operands straight from L2),
// with ONE inter-workgroup barrier per step per group (write-through hand-off of lstm_common.h: sc1 stores, drained
// arrive, sc1 loads, no fences; bounded spins).
// The Linear(H, 2) epilogue is a per-wave partial dot, summed in a fixed order by workgroup cs == 0 one step later.
#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

struct NStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};

__device__ __forceinline__ float4 nload(const NStream& s, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff, soff, 0));
}
__device__ __forceinline__ float4 nload_sc1(const NStream& s, int soff) {    // exchange images: bypass L1 (lstm_common.h)
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff, soff, kSc1));
}

__device__ __forceinline__ void mfma16(f32x16 (&acc)[4], const float4& a, const float4 (&b)[4]) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[n].x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[n].y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[n].z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[n].w, acc[n], 0, 0, 0);
    }
}

// G (<= 8) k-groups, fully unrolled, every load issued up front.  aload(g) -> A fragment of group g,
// weights at stream groups [wbase, wbase + G).
template <int G, typename ALoad>
__device__ __forceinline__ void coopn_unrolled(f32x16 (&acc)[4], const NStream& ws, int wbase, ALoad aload) {
    float4 a[G];
    float4 b[G][4];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        a[k] = aload(k);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[k][n] = nload(ws, (wbase + k) * 4096 + n * 1024);
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
        mfma16(acc, a[k], b[k]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// G (multiple of 4, >= 8) k-groups: register pipeline 4 groups (= 4096 MFMA cycles) deep, refill in place.
template <int G, typename ALoad>
__device__ __forceinline__ void coopn_loop(f32x16 (&acc)[4], const NStream& ws, int wbase, ALoad aload) {
    static_assert(G % 4 == 0 && G >= 8, "pipeline shape");
    float4 a[4];
    float4 b[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = aload(k);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[k][n] = nload(ws, (wbase + k) * 4096 + n * 1024);
    }
    for (int g0 = 0; g0 < G - 4; g0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mfma16(acc, a[k], b[k]);
            a[k] = aload(g0 + k + 4);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[k][n] = nload(ws, (wbase + g0 + k + 4) * 4096 + n * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mfma16(acc, a[k], b[k]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// GRU = true: nn.GRU (see lstm_coop.hip): column slots (r, z, n_x, n_h), zero weight blocks packed by the host, the
// register state is h itself.
template <int HID, int KX, int R, bool GRU>
__global__ __launch_bounds__(256) void lstm2_coopn_kernel(LstmWeights w, LstmArgs a) {
    constexpr int UNITS = 128;                     // hidden units per workgroup, 32 per wave
    constexpr int S = HID / UNITS;
    constexpr int KGX = KX / 8, KGH = HID / 8, KG0 = KGX + KGH, KG1 = 2 * KGH;
    constexpr int HIMG = KGH * 64;                 // float4 per exchange image (32 rows x HID)
    constexpr int HXT = coop_tile_f4(HID);         // float4 per row tile of the exchange region (lstm_common.h)
    constexpr int NG = KGX;

    __shared__ __attribute__((aligned(16))) float4 Xs[2][R][KGX * 64];  // A images of x_t, double buffered by step parity
    __shared__ RowDesc rows_s[R][32];
    __shared__ int abort_s;                        // set by thread 0 when an inter-workgroup wait gives up

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.coop_groups;
    int g = blockIdx.x / S, cs = blockIdx.x % S;       // group, 128-unit column slice; XCD-local: lstm_common.h
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, G, a.coop_xcd, g, cs)) return;
    const int Tp = a.Tp;
    const int ub = cs * 4 + wave;                  // 32-unit block of this wave
    const int unit = ub * 32 + (lane & 31);        // hidden unit of this lane

    int rt[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rt[r] = g + G * r;
        live[r] = rt[r] < a.num_tiles;
        if (!live[r]) rt[r] = g;                   // harmless duplicate addresses; nothing is computed for it
    }
    unsigned* bar = FSNP_COOP_BAR(a, g, 0);
    if (tid == 0) abort_s = 0;

#pragma unroll
    for (int r = 0; r < R; ++r) {
        for (int i = tid; i < KGX * 64; i += 256) {
            Xs[0][r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            Xs[1][r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 32) {
            RowDesc rd{0, 0, 0, 0};
            if (live[r]) rd = a.rows[rt[r] * 32 + tid];
            rows_s[r][tid] = rd;
        }
    }
    __syncthreads();

    // ---- input plan (as lstm_coop.hip): thread owns row = tid & 31, features j = (tid >> 5) + 8 i of every row tile
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    const int grow = tid & 31, jrow = tid >> 5;
    int goff[R][NG];
    NormMD md[R];
    const NormMD* md_t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const RowDesc rd = rows_s[r][grow];
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
        if (rd.valid && !dense) {
            if (a.md_row != nullptr) md_t[r] = a.md_row + (size_t)(rt[r] * 32 + grow) * Tp;
            else md[r] = a.md_utt[rd.b];
        }
    }
    const int xdst0 = a_frag_index(grow, jrow);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float* Xf = reinterpret_cast<float*>(Xs[0][r]);
        const NormMD m0 = md_t[r] ? md_t[r][0] : md[r];
#pragma unroll
        for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[r][i] >= 0 ? (gbase[goff[r][i]] - m0.m) / m0.d : 0.0f;
    }

    // ---- this wave's weight stream [unit block][k-group: x | h0, then h1 | h0][gate][lane][4]
    NStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)ub * (KG0 + KG1) * 1024, 0,
                                                (KG0 + KG1) * 4096, 0x00020000);
    ws.voff = lane * 16;
    // A operands from the exchange images of each row tile
    NStream hs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        hs[r].rsrc = __builtin_amdgcn_make_buffer_rsrc(a.coop_hx + (size_t)rt[r] * HXT * 4, 0, 4 * HIMG * 16, 0x00020000);
        hs[r].voff = lane * 16;
    }

    // ---- lane-local cell state: register q of the accumulators <-> row (q&3) + 8 (q>>2) + 4 (lane>>5), unit = lane&31
    float c0[R][16], c1[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) { c0[r][q] = 0.f; c1[r][q] = 0.f; }
    float b0[4], b1[4];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
        b0[gate] = w.bias[gate * HID + unit];
        b1[gate] = w.bias[4 * HID + gate * HID + unit];
    }
    const float wfc0 = w.wfc[unit], wfc1 = w.wfc[HID + unit];
    const int rowbase = 4 * (lane >> 5);
    // cells q, q + 1 of this lane (rows row, row + 1 of its unit): gate pre-activations = accumulator + bias; state in `c`
    auto cell_pair = [&](const f32x16 (&acc)[4], int q, const float (&b)[4], float (&c)[16]) -> f32x2 {
        const f32x2 x0 = f32x2{acc[0][q], acc[0][q + 1]} + b[0], x1 = f32x2{acc[1][q], acc[1][q + 1]} + b[1];
        const f32x2 x2 = f32x2{acc[2][q], acc[2][q + 1]} + b[2], x3 = f32x2{acc[3][q], acc[3][q + 1]} + b[3];
        f32x2 st{c[q], c[q + 1]};
        f32x2 h;
        if constexpr (GRU) { h = gru_cell_pair(x0, x1, x2, x3, st); st = h; }
        else h = lstm_cell_pair(x0, x1, x2, x3, st);
        c[q] = st.x; c[q + 1] = st.y;
        return h;
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
    auto fc_epilogue = [&](int r, int t_done) {   // workgroup cs == 0 sums the 4 S per-wave partials in a fixed order
        if (cs == 0 && tid < 64 && live[r]) {
            const int row = tid & 31, o = tid >> 5;
            const RowDesc rd = rows_s[r][row];
            const float* part = a.coop_hx + ((size_t)rt[r] * HXT + 4 * HIMG) * 4 + (size_t)(t_done & 1) * (4 * S) * 64;
            float sum = w.bfc[o];
            for (int p = 0; p < 4 * S; ++p) sum += xchg_load(part + p * 64 + o * 32 + row);
            if (rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };

    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        const bool have_next = t + 1 < Tp;
        chaos_delay(a.coop_chaos, t, 0);
        // ---------------- layer 0 of every row tile: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            float xr[NG];
            NormMD mdn = md[r];
            if (have_next) {                                  // prefetch x(t+1)
                if (md_t[r]) mdn = md_t[r][t + 1];
#pragma unroll
                for (int i = 0; i < NG; ++i) xr[i] = goff[r][i] >= 0 ? gbase[goff[r][i] + (t + 1) * gstep] : 0.0f;
            }
            f32x16 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[n][q] = 0.0f;
            const float4* xw = Xs[cur][r] + lane;
            coopn_unrolled<KGX>(acc, ws, 0, [&](int k) -> float4 { return xw[k * 64]; });
            coopn_loop<KGH>(acc, ws, KGX, [&](int k) -> float4 { return nload_sc1(hs[r], prv * (HIMG * 16) + k * 1024); });
            float* img = reinterpret_cast<float*>(a.coop_hx + ((size_t)rt[r] * HXT + (size_t)cur * HIMG) * 4);
#pragma unroll
            for (int q = 0; q < 16; q += 2) {                  // two cells per pass: packed fp32 math (lstm_common.h)
                const f32x2 h2 = cell_pair(acc, q, b0, c0[r]);
                const int row = (q & 3) + 8 * (q >> 2) + rowbase;
                const bool corrupt = a.coop_corrupt != 0 && rt[r] == 0 && unit == 0 && row == 0 && t + 1 == a.coop_corrupt;     // test hook: published value only
                xchg_store(img + a_frag_index(row, unit), corrupt ? h2.x + 1.0f : h2.x);
                xchg_store(img + a_frag_index(row + 1, unit), h2.y);
            }
            if (have_next) {      // the other parity: last read in step t-1, before that step's barrier
                float* Xf = reinterpret_cast<float*>(Xs[prv][r]);
#pragma unroll
                for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[r][i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
            }
        }
        chaos_delay(a.coop_chaos, t, 1);
        if (!inter_wg_barrier((unsigned)S * (unsigned)(t + 1))) return;   // h0_t, h1_{t-1} and the FC partials of step t-1 are now visible
        chaos_delay(a.coop_chaos, t, 2);

        // ---------------- layer 1 of every row tile: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            if (t > 0) fc_epilogue(r, t - 1);
            f32x16 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[n][q] = 0.0f;
            coopn_loop<KGH>(acc, ws, KG0, [&](int k) -> float4 { return nload_sc1(hs[r], (2 + prv) * (HIMG * 16) + k * 1024); });
            coopn_loop<KGH>(acc, ws, KG0 + KGH, [&](int k) -> float4 { return nload_sc1(hs[r], cur * (HIMG * 16) + k * 1024); });
            float* img = reinterpret_cast<float*>(a.coop_hx + ((size_t)rt[r] * HXT + (size_t)(2 + cur) * HIMG) * 4);
            float* part = a.coop_hx + ((size_t)rt[r] * HXT + 4 * HIMG) * 4 + ((size_t)cur * (4 * S) + ub) * 64;
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const f32x2 h2 = cell_pair(acc, q, b1, c1[r]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float hval = e == 0 ? h2.x : h2.y;
                    const int row = (q & 3) + 8 * (q >> 2) + rowbase + e;
                    xchg_store(img + a_frag_index(row, unit), hval);
                    float p0 = hval * wfc0, p1 = hval * wfc1;        // partial Linear over this wave's 32 units
#pragma unroll
                    for (int m = 16; m > 0; m >>= 1) { p0 += __shfl_xor(p0, m); p1 += __shfl_xor(p1, m); }
                    if ((lane & 31) == 0) { xchg_store(part + row, p0); xchg_store(part + 32 + row, p1); }
                }
            }
        }
    }
    // last step's Linear: one more barrier so that every partial of step Tp-1 is visible
    if (!inter_wg_barrier((unsigned)S * (unsigned)(Tp + 1))) return;
#pragma unroll
    for (int r = 0; r < R; ++r) fc_epilogue(r, Tp - 1);
}

// ------------------------------------------------------------------------------------------------
size_t lstm_coopn_pack_floats(int H, int KX) { return (size_t)(H / 32) * (KX / 8 + 3 * (H / 8)) * 4 * 64 * 4; }

// [32-unit block ub][k-group (layer 0: x | h0, then layer 1: h1 | h0)][gate][lane][k-pair]
void lstm_coopn_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1,
                             const float* whh1, float* wpack) {
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    for (int ub = 0; ub < H / 32; ++ub)
        for (int g = 0; g < KGT; ++g)
            for (int gate = 0; gate < 4; ++gate)
                for (int lane = 0; lane < 64; ++lane)
                    for (int p = 0; p < 4; ++p) {
                        const int wrow = gate * H + ub * 32 + (lane & 31);
                        float v = 0.0f;
                        if (g < KG0) {
                            const int k = 8 * g + 2 * p + (lane >> 5);
                            if (k < KX) { if (k < NIN) v = wih0[(size_t)wrow * NIN + k]; }
                            else v = whh0[(size_t)wrow * H + (k - KX)];
                        } else {
                            const int k = 8 * (g - KG0) + 2 * p + (lane >> 5);
                            if (k < H) v = whh1[(size_t)wrow * H + k];
                            else v = wih1[(size_t)wrow * H + (k - H)];
                        }
                        wpack[((((size_t)ub * KGT + g) * 4 + gate) * 64 + lane) * 4 + p] = v;
                    }
}

// Plan for `row_tiles` 32-row tiles: R row tiles per group (1 or 2), G groups of S = 3 workgroups, all co-resident.
// Returns R (0 = not applicable: more than 2 row tiles per group would be slower than the row-tile kernel).
int lstm_coopn_plan(int H, int row_tiles, int num_cus, int* groups) {
    const int S = H / 128;
    const int gmax = num_cus / S;
    if (H % 128 != 0 || gmax <= 0 || row_tiles <= 0) return 0;
    const int R = cdiv(row_tiles, gmax);
    if (R > 2) return 0;
    *groups = cdiv(row_tiles, R);
    return R;
}

template <int R, bool GRU, int KX, int HID = 384>
static void launch_coopn_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    LstmWeights wv = w;
    wv.wpack = w.wpack_coopn;
    if (occ) {        // do not launch: how many workgroups of this instantiation fit one CU at once
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(lstm2_coopn_kernel<HID, KX, R, GRU>), 256, 0) != hipSuccess) *occ = 0;
        return;
    }
    const int grid = a.coop_xcd ? 8 * xcd_local_blocks_per_xcd(HID / 128, a.coop_groups, a.coop_xcd) : a.coop_groups * (HID / 128);
    // LstmArgs::coop_own_cu: claim the rest of the CU's LDS as (unused) dynamic LDS so that no LDS-using workgroup of a
    // concurrent kernel shares the CU (pipelined loop: the next forward's full-band GEMMs)
    auto kern = lstm2_coopn_kernel<HID, KX, R, GRU>;
    static PerDeviceOnce attr_once;
    static int static_lds = 0;
    attr_once.run([&] {
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)) == hipSuccess) static_lds = (int)fa.sharedSizeBytes;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - static_lds);
    });
    const int pad = a.coop_own_cu > 0 ? a.coop_own_cu - static_lds : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), pad > 0 ? pad : 0, s, wv, a);
}

template <int KX, int HID = 384>
static void launch_coopn_kx(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    if (w.gru) {
        if (a.coop_rows_per_group == 1) launch_coopn_inst<1, true, KX, HID>(w, a, s, occ);
        else launch_coopn_inst<2, true, KX, HID>(w, a, s, occ);
    } else {
        if (a.coop_rows_per_group == 1) launch_coopn_inst<1, false, KX, HID>(w, a, s, occ);
        else launch_coopn_inst<2, false, KX, HID>(w, a, s, occ);
    }
}
template <int HID>
static void launch_coopn_h(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    if (w.KX == 64) launch_coopn_kx<64, HID>(w, a, s, occ);
    else launch_coopn_kx<40, HID>(w, a, s, occ);
}

void launch_lstm_coopn(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.H == 256) { launch_coopn_h<256>(w, a, s, nullptr); return; }      // sb_model_hidden_size 256 / 512
    if (w.H == 512) { launch_coopn_h<512>(w, a, s, nullptr); return; }
    if (w.KX == 64) launch_coopn_kx<64>(w, a, s, nullptr);       // sub-band inputs of 41..64 features
    else launch_coopn_kx<40>(w, a, s, nullptr);
}
int lstm_coopn_occupancy(const LstmWeights& w, int rows_per_group) {
    LstmArgs a{};
    a.coop_rows_per_group = rows_per_group;
    int occ = 0;
    if (w.H == 256) launch_coopn_h<256>(w, a, nullptr, &occ);
    else if (w.H == 512) launch_coopn_h<512>(w, a, nullptr, &occ);
    else if (w.KX == 64) launch_coopn_kx<64>(w, a, nullptr, &occ);
    else launch_coopn_kx<40>(w, a, nullptr, &occ);
    return occ;
}

}  // namespace fsnp
