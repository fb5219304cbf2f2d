// lstm_coop.hip - column-split ("cooperative") two-layer LSTM + Linear for SMALL batches on gfx950.
//
// Same arithmetic as lstm.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123), different decomposition.  The row-tile
// kernel gives one CU 32 sequences and ALL 4H gate columns, so a step costs one CU ~200 us however few tiles exist:
// the reference CLI's batch of ONE utterance (257 sequences = 9 tiles) kept 247 CUs idle for 26 ms.  Here a 32-row
// tile is shared by S = H / (32 TW) workgroups:
//   * workgroup (rt, cs) owns hidden units [cs * 32 TW, (cs+1) * 32 TW) of both layers, i.e. 4 TW accumulator tiles;
//   * its 4 waves split K (each wave reduces a quarter of the k-groups into its own copy of the tiles; partial tiles
//     are summed through LDS; wave w then owns rows 8w..8w+7 for the cell update, c stays in registers);
//   * every step each workgroup publishes its 32 x 32TW slice of h0_t / h1_t into a per-tile, double-buffered
//     exchange image in global memory that is ALREADY in MFMA A-fragment order, so consumers read their A operands
//     straight from L2 with one coalesced 16-byte load per lane per k-group (no LDS copy);
//   * ONE inter-workgroup barrier per step (after h0_t is published) orders everything: h1_{t-1} was published
//     before its writer arrived.  The barrier is the MI355X hand-off recipe (MI355X_MICROARCH.md): every storing wave
//     drains vmcnt, __syncthreads, ONE lane: agent-scope release + asm vmcnt(0) + relaxed atomic arrive, relaxed
//     polling with s_sleep, ONE agent-scope acquire, __syncthreads, plain vector loads.  Spins are bounded; all
//     workgroups of a launch must be co-resident (the host only launches RT * S <= number of CUs).
//   * the Linear(H, 2) epilogue is a per-workgroup partial dot over its own units (from registers), published with
//     h1 and summed in a fixed order by workgroup cs == 0 one step later (deterministic, no float atomics).
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

// One layer's share of this wave: k-groups [lo, hi) of the layer, weights at stream group (wbase + g).
// A operand of group g comes from `src(g)` (LDS or the global exchange image).  Register pipeline D groups deep:
// a k-group lasts NT * 256 cycles, so D = 4 (one tile quad per wave) or 2 (two / three) keeps >= 4096 cycles of loads
// in flight without spilling.
template <int NT, typename ASrc>
__device__ __forceinline__ void coop_layer(f32x16 (&acc)[NT], const CoopStream& ws, int wbase, int lo, int hi, ASrc src) {
    constexpr int D = NT <= 4 ? 4 : 2;
    float4 a[D];
    float4 b[D][NT];
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (lo + k < hi) {
            a[k] = src(lo + k);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[k][n] = coop_wload<NT>(ws, wbase + lo + k, n);
        }
    for (int g0 = lo; g0 < hi; g0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (g0 + k < hi) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].x, b[k][n].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].y, b[k][n].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].z, b[k][n].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k].w, b[k][n].w, acc[n], 0, 0, 0);
                }
                if (g0 + k + D < hi) {
                    a[k] = src(g0 + k + D);
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[k][n] = coop_wload<NT>(ws, wbase + g0 + k + D, n);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace

template <int HID, int KX, int OUT, int TW>
__global__ __launch_bounds__(256) void lstm2_fc_coop_kernel(LstmWeights w, LstmArgs a) {
    static_assert(OUT == 2, "epilogue assumes output_size == 2");
    constexpr int NT = 4 * TW;
    constexpr int KGX = KX / 8, KGH = HID / 8, KG0 = KGX + KGH, KG1 = 2 * KGH;
    constexpr int S = HID / (32 * TW);
    constexpr int HIMG = KGH * 64;                 // float4 per exchange image (32 rows x HID)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);                     // [KGX][64] A image of x_t
    float* red = reinterpret_cast<float*>(Xs + KGX * 64);                 // [4 waves][NT][16][64]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(red + 4 * NT * 16 * 64); // [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = blockIdx.x / S, cs = blockIdx.x % S;
    const int slot0 = rt * 32;
    const int Tp = a.Tp;

    // exchange region of this row tile: [h0 p0][h0 p1][h1 p0][h1 p1] images + FC partials [2][S][64]
    float4* hx = reinterpret_cast<float4*>(a.coop_hx) + (size_t)rt * (4 * HIMG + 2 * S * 16);
    float4* h0img[2] = {hx, hx + HIMG};
    float4* h1img[2] = {hx + 2 * HIMG, hx + 3 * HIMG};
    float* fcp = reinterpret_cast<float*>(hx + 4 * HIMG);                  // [2][S][64]
    unsigned* bar = a.coop_bar + rt;

    for (int i = tid; i < KGX * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 32) rows_s[tid] = a.rows[slot0 + tid];
    __syncthreads();

    // ---- gather plan (as lstm.hip): thread owns row = tid & 31, features j = (tid >> 5) + 8 i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? w.NIN : a.FP;
    constexpr int NG = KGX;
    const int grow = tid & 31;
    int goff[NG];
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_row = nullptr;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = (tid >> 5) + 8 * i;
            int off = -1;
            if (rd.valid && j < w.NIN) {
                if (dense) off = rd.b * Tp * w.NIN + j;
                else {
                    const int base = rd.b * Tp * a.FP;
                    const int nsb = 2 * a.NSBN + 1;
                    off = (j < nsb) ? base + reflect_index(rd.f - a.NSBN + j, a.F)
                                    : a.fb_rel + (j - nsb) * a.fb_branch_stride + base + rd.f;
                }
            }
            goff[i] = off;
        }
        if (!dense && rd.valid) {
            if (a.md_row != nullptr) md_row = a.md_row + (size_t)(slot0 + grow) * Tp;
            else md = a.md_utt[rd.b];
        }
    }
    const int xdst0 = a_frag_index(grow, tid >> 5);
    float* Xf = reinterpret_cast<float*>(Xs);
    {
        const NormMD m0 = md_row ? md_row[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[i] >= 0 ? (gbase[goff[i]] - m0.m) / m0.d : 0.0f;
    }

    // ---- this wave's share of K and its weight stream (all waves of the workgroup share one packed slice)
    const int lo0 = KG0 * wave / 4, hi0 = KG0 * (wave + 1) / 4;
    const int lo1 = KG1 * wave / 4, hi1 = KG1 * (wave + 1) / 4;
    CoopStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)cs * (KG0 + KG1) * NT * 256, 0,
                                                (KG0 + KG1) * NT * 1024, 0x00020000);
    ws.voff = lane * 16;

    // ---- cell state: wave w owns rows 8w..8w+7 (accumulator registers 4w..4w+3), lanes = (row half, unit)
    float c0[TW][4], c1[TW][4];
#pragma unroll
    for (int s = 0; s < TW; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) { c0[s][q] = 0.f; c1[s][q] = 0.f; }
    float wfc0[TW], wfc1[TW];
    const float* __restrict__ bias_l0 = w.bias + cs * 32 * TW + (lane & 31);             // + gate * HID + s * 32
    const float* __restrict__ bias_l1 = w.bias + 4 * HID + cs * 32 * TW + (lane & 31);
#pragma unroll
    for (int s = 0; s < TW; ++s) {
        wfc0[s] = w.wfc[cs * 32 * TW + s * 32 + (lane & 31)];
        wfc1[s] = w.wfc[HID + cs * 32 * TW + s * 32 + (lane & 31)];
    }

    // sum the 4 waves' partial tiles through LDS; returns, for the 4 rows this wave owns, gate pre-activations
    auto reduce_tiles = [&](f32x16 (&acc)[NT], float (&g)[NT][4]) {
        __syncthreads();                       // previous use of `red` finished
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NT + n) * 16 + r) * 64 + lane] = acc[n][r];
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = wave * 4 + q;
                g[n][q] = red[((0 * NT + n) * 16 + r) * 64 + lane] + red[((1 * NT + n) * 16 + r) * 64 + lane] +
                          red[((2 * NT + n) * 16 + r) * 64 + lane] + red[((3 * NT + n) * 16 + r) * 64 + lane];
            }
    };
    // rows owned by this lane for register q of the wave's group: C layout row = (r&3) + 8 (r>>2) + 4 (lane>>5), r = 4w+q
    auto own_row = [&](int q) { return q + 8 * wave + 4 * (lane >> 5); };

    auto inter_wg_barrier = [&](unsigned target) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 24)) {                         // seconds: a peer is not resident - give up loudly
                    __hip_atomic_store(a.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    };

    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        // prefetch x(t+1)
        float xr[NG];
        NormMD mdn = md;
        const bool have_next = t + 1 < Tp;
        if (have_next) {
            if (md_row) mdn = md_row[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = goff[i] >= 0 ? gbase[goff[i] + (t + 1) * gstep] : 0.0f;
        }

        f32x16 acc[NT];
        float g[NT][4];
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        {
            const float4* h0p = h0img[prv] + lane;
            coop_layer<NT>(acc, ws, 0, lo0, hi0, [&](int gg) -> float4 {
                return gg < KGX ? Xs[gg * 64 + lane] : h0p[(gg - KGX) * 64];
            });
        }
        reduce_tiles(acc, g);
        float fc_part0 = 0.f, fc_part1 = 0.f;       // Linear partials of h1_{t-1} are produced in the layer-1 block below
        {
            float* img = reinterpret_cast<float*>(h0img[cur]);
#pragma unroll
            for (int s = 0; s < TW; ++s) {
                const int k = cs * 32 * TW + s * 32 + (lane & 31);
                const float bi = bias_l0[s * 32], bf = bias_l0[HID + s * 32], bg = bias_l0[2 * HID + s * 32], bo = bias_l0[3 * HID + s * 32];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ig = fast_sigmoid(g[s][q] + bi);
                    const float fg = fast_sigmoid(g[TW + s][q] + bf);
                    const float gg = fast_tanh(g[2 * TW + s][q] + bg);
                    const float og = fast_sigmoid(g[3 * TW + s][q] + bo);
                    const float cn = fg * c0[s][q] + ig * gg;
                    c0[s][q] = cn;
                    img[a_frag_index(own_row(q), k)] = og * fast_tanh(cn);
                }
            }
        }
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i) Xf[xdst0 + i * 256] = goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
        }
        inter_wg_barrier((unsigned)S * (unsigned)(t + 1));   // h0_t, h1_{t-1} and the FC partials of step t-1 are now visible

        // Linear epilogue of step t-1: workgroup cs == 0 sums the S partials in a fixed order
        if (cs == 0 && t > 0 && tid < 64) {
            const int row = tid & 31, o = tid >> 5;
            const RowDesc rd = rows_s[row];
            const float* part = fcp + (size_t)prv * S * 64;
            float sum = w.bfc[o];
            for (int p = 0; p < S; ++p) sum += part[p * 64 + o * 32 + row];
            if (rd.valid && t - 1 >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t - 1 - a.LA)] = apply_act(sum, a.act);
        }

        // ---------------- layer 1: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        {
            const float4* h1p = h1img[prv] + lane;
            const float4* h0c = h0img[cur] + lane;
            coop_layer<NT>(acc, ws, KG0, lo1, hi1, [&](int gg) -> float4 {
                return gg < KGH ? h1p[gg * 64] : h0c[(gg - KGH) * 64];
            });
        }
        reduce_tiles(acc, g);
        {
            float* img = reinterpret_cast<float*>(h1img[cur]);
            float p0[4] = {0.f, 0.f, 0.f, 0.f}, p1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TW; ++s) {
                const int k = cs * 32 * TW + s * 32 + (lane & 31);
                const float bi = bias_l1[s * 32], bf = bias_l1[HID + s * 32], bg = bias_l1[2 * HID + s * 32], bo = bias_l1[3 * HID + s * 32];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ig = fast_sigmoid(g[s][q] + bi);
                    const float fg = fast_sigmoid(g[TW + s][q] + bf);
                    const float gg = fast_tanh(g[2 * TW + s][q] + bg);
                    const float og = fast_sigmoid(g[3 * TW + s][q] + bo);
                    const float cn = fg * c1[s][q] + ig * gg;
                    c1[s][q] = cn;
                    const float h = og * fast_tanh(cn);
                    img[a_frag_index(own_row(q), k)] = h;
                    p0[q] += h * wfc0[s];
                    p1[q] += h * wfc1[s];
                }
            }
            // partial Linear over this workgroup's units: reduce over the 32 unit lanes of each half-wave
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) { p0[q] += __shfl_xor(p0[q], m); p1[q] += __shfl_xor(p1[q], m); }
                if ((lane & 31) == 0) {
                    float* part = fcp + ((size_t)cur * S + cs) * 64;
                    part[own_row(q)] = p0[q];
                    part[32 + own_row(q)] = p1[q];
                }
            }
        }
        (void)fc_part0; (void)fc_part1;
    }
    // last step's Linear: one more barrier so that every partial of step Tp-1 is visible
    inter_wg_barrier((unsigned)S * (unsigned)(Tp + 1));
    if (cs == 0 && tid < 64) {
        const int row = tid & 31, o = tid >> 5;
        const RowDesc rd = rows_s[row];
        const float* part = fcp + (size_t)((Tp - 1) & 1) * S * 64;
        float sum = w.bfc[o];
        for (int p = 0; p < S; ++p) sum += part[p * 64 + o * 32 + row];
        if (rd.valid && Tp - 1 >= a.LA)
            a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (Tp - 1 - a.LA)] = apply_act(sum, a.act);
    }
}

// ------------------------------------------------------------------------------------------------
size_t lstm_coop_pack_floats(int H, int KX, int TW) {
    const int S = H / (32 * TW), NT = 4 * TW;
    const int KGT = KX / 8 + 3 * (H / 8);
    return (size_t)S * KGT * NT * 64 * 4;
}

// [cs][k-group (layer 0: x | h0, then layer 1: h1 | h0)][tile n = gate*TW + s][lane][k-pair]
void lstm_coop_pack_weights(int H, int NIN, int KX, int TW, const float* wih0, const float* whh0, const float* wih1,
                            const float* whh1, float* wpack) {
    const int S = H / (32 * TW), NT = 4 * TW;
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    for (int cs = 0; cs < S; ++cs)
        for (int g = 0; g < KGT; ++g)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int p = 0; p < 4; ++p) {
                        const int gate = n / TW, s = n % TW;
                        const int wrow = gate * H + cs * 32 * TW + s * 32 + (lane & 31);
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
                        wpack[((((size_t)cs * KGT + g) * NT + n) * 64 + lane) * 4 + p] = v;
                    }
}

size_t lstm_coop_exchange_bytes(int H, int TW, int row_tiles) {
    const int S = H / (32 * TW);
    return (size_t)row_tiles * (4 * (size_t)(H / 8) * 64 + 2 * (size_t)S * 16) * 16;
}

template <int TW>
static void launch_lstm_coop_tw(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    constexpr int HID = 384, KX = 40, OUT = 2;
    constexpr int S = HID / (32 * TW), NT = 4 * TW;
    const size_t smem = (size_t)(KX / 8) * 64 * 16 + (size_t)4 * NT * 16 * 64 * 4 + 32 * sizeof(RowDesc);
    auto kern = lstm2_fc_coop_kernel<HID, KX, OUT, TW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    LstmWeights wv = w;
    wv.wpack = w.wpack_coop[TW - 1];
    hipLaunchKernelGGL(kern, dim3(a.num_tiles * S), dim3(256), smem, s, wv, a);
}

// Largest column split (fewest units per workgroup: 32 TW, TW in {1,2}) whose row_tiles * H/(32 TW) workgroups
// are all resident at once; 0 = use the row-tile kernel.  (TW = 3 would need 192 KB of LDS for the partial tiles.)
int lstm_coop_pick_tw(int H, int row_tiles, int num_cus) {
    for (int tw = 1; tw <= 2; ++tw)
        if (H % (32 * tw) == 0 && row_tiles * (H / (32 * tw)) <= num_cus) return tw;
    return 0;
}

void launch_lstm_coop(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (a.coop_tw == 1) launch_lstm_coop_tw<1>(w, a, s);
    else launch_lstm_coop_tw<2>(w, a, s);
}

}  // namespace fsnp
