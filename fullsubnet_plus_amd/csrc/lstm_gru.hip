// lstm_gru.hip - fused two-layer sub-band GRU + Linear, one 32-sequence tile per CU (gfx950).
//
// SequenceModel(sequence_model="GRU") of the reference (speech_enhance/audio_zen/model/module/sequence_model.py:39-46:
// nn.GRU(input_size, 384, num_layers=2, batch_first=True) + Linear(384, 2), forward :113-123) on the decomposition of
// lstm.hip: rows = 32 sequences per workgroup (M of v_mfma_f32_32x32x2_f32, exact fp32), wave w owns hidden units
// [w H/4, (w+1) H/4) of both layers, weights stream L2 -> registers in MFMA B-fragment order, x_t / h0 / h1 live in LDS
// in A-fragment order, the Linear epilogue writes out[b, o, f, t - look_ahead] directly.
//
// What differs from the LSTM: torch.nn.GRU
//     r = s(W_ir x + b_ir + W_hr h + b_hr)      z = s(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn))          h' = (1 - z) n + z h
// keeps the input and the hidden contribution of the n gate APART, so a unit has four accumulator columns (r, z, n_x, n_h)
// but only THREE of them take part in any k-group: (r, z, n_x) while K runs over the layer's input, (r, z, n_h) while it runs
// over its own previous h.  The column-split kernels (lstm_coop.hip / lstm_coopn.hip) run GRU on zero-padded four-slot
// columns - a quarter of their MFMAs multiply zero blocks; here the weight stream carries exactly the three live tiles per
// k-group, so a step costs 3/4 of the LSTM's matrix-pipe time.  The per-unit register state is h itself.
#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

struct GStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
// one 1 KiB weight fragment (tile n of k-group g, NL live tiles per group) of this wave's stream
template <int NL>
__device__ __forceinline__ float4 gload(const GStream& ws, int g, int n) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, (g * NL + n) * 1024, 0));
}

// `ngroups` k-groups of one K segment.  HID_PART = false: the segment is the layer's INPUT (live tiles r, z, n_x =
// accumulators [0, 3 ST)); true: its own previous h (r, z, n_h = accumulators [0, 2 ST) and [3 ST, 4 ST)).  b always holds
// the group about to be used and is refilled in place from the stream (same pipeline as lstm.hip: mfma_groups).
template <int ST, bool HID_PART>
__device__ __forceinline__ void gru_groups(f32x16 (&acc)[4 * ST], float4 (&b)[3 * ST], const float4* __restrict__ A, int ngroups,
                                           const GStream& ws, int& gnext, int groups_total) {
    constexpr int NL = 3 * ST;
    float4 a = A[0];
    for (int g = 0; g < ngroups; ++g) {
        const float4 an = A[(g + 1 < ngroups ? g + 1 : g) * 64];
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int ai = (n < 2 * ST) ? n : (HID_PART ? n + ST : n);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[n].x, acc[ai], 0, 0, 0);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[n].y, acc[ai], 0, 0, 0);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[n].z, acc[ai], 0, 0, 0);
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[n].w, acc[ai], 0, 0, 0);
            b[n] = gload<NL>(ws, gnext, n);
            __builtin_amdgcn_sched_barrier(0);
        }
        gnext = (gnext + 1 == groups_total) ? 0 : gnext + 1;
        a = an;
    }
}

// lane-local cell update of the wave's ST 32-unit blocks: acc = [r | z | n_x | n_h] pre-activations (biases included),
// hreg = h_{t-1} -> h_t, written to the A image Hs in fragment order
template <int ST, int UW>
__device__ __forceinline__ void gru_cell(f32x16 (&acc)[4 * ST], f32x16 (&hreg)[ST], float* __restrict__ Hs, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int k = wave * UW + s * 32 + (lane & 31);
        const int kbase = (((k >> 3) * 64) + ((k & 1) * 32)) * 4 + ((k >> 1) & 3);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                  // two cells per pass: packed fp32 math (lstm_common.h gru_cell_pair)
            auto rd = [](float v) { float o; asm("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v)); return o; };   // (see lstm.hip lstm_cell)
            const f32x2 h = gru_cell_pair(f32x2{rd(acc[s][r]), rd(acc[s][r + 1])}, f32x2{rd(acc[ST + s][r]), rd(acc[ST + s][r + 1])},
                                          f32x2{rd(acc[2 * ST + s][r]), rd(acc[2 * ST + s][r + 1])},
                                          f32x2{rd(acc[3 * ST + s][r]), rd(acc[3 * ST + s][r + 1])}, f32x2{hreg[s][r], hreg[s][r + 1]});
            hreg[s][r] = h.x; hreg[s][r + 1] = h.y;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Hs[kbase + row * 4] = h.x;
            Hs[kbase + (row + 1) * 4] = h.y;
        }
    }
}

}  // namespace

template <int HID, int KX, int OUT, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void gru2_fc_kernel(LstmWeights w, LstmArgs a) {
    static_assert(OUT == 2, "FC lane mapping assumes output_size == 2");
    constexpr int NTHR = 64 * NW;
    constexpr int UW = HID / NW, ST = UW / 32, NT = 4 * ST, NL = 3 * ST;
    static_assert(UW % 32 == 0 && UW * NW == HID, "hidden/NW must be a multiple of 32");
    constexpr int KGX = KX / 8, KGH = HID / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    static_assert(KGH % 4 == 0, "FC k-split");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);   // [KGX][64]  A image of x_t
    float4* H0s = Xs + KGX * 64;                         // [KGH][64]  h0
    float4* H1s = H0s + KGH * 64;                        // [KGH][64]  h1
    float4* Wfc4 = H1s + KGH * 64;                       // [OUT][KGH][2]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(Wfc4 + OUT * KGH * 2);  // [32]
    float* Bs = reinterpret_cast<float*>(rows_s + 32);                   // [2][NW][NT][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot0 = blockIdx.x * 32;
    const int Tp = a.Tp;

    for (int i = tid; i < (KGX + 2 * KGH) * 64; i += NTHR) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < OUT * KGH * 2; i += NTHR) {
        const int o = i / (KGH * 2), kg = (i >> 1) % KGH, kh = i & 1;
        const float* wr = w.wfc + (size_t)o * HID + kg * 8 + kh;
        Wfc4[i] = make_float4(wr[0], wr[2], wr[4], wr[6]);
    }
    if (tid < 32) rows_s[tid] = a.rows[slot0 + tid];
    for (int i = tid; i < 2 * NW * NT * 32; i += NTHR) {     // biases: four slots r, z, n_x, n_h per unit (fsnp_abi.hip: expand)
        const int col = i & 31, n = (i >> 5) % NT, wv = (i / (32 * NT)) % NW, layer = i / (32 * NT * NW);
        Bs[i] = w.bias[layer * 4 * HID + (n / ST) * HID + wv * UW + (n % ST) * 32 + col];
    }
    __syncthreads();

    // ---- gather plan (as lstm.hip): thread owns row = tid & 31, features j = (tid >> 5) + (NTHR / 32) i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? w.NIN : a.FP;
    constexpr int JSTEP = NTHR / 32, NG = (KX + JSTEP - 1) / JSTEP;
    const int grow = tid & 31;
    int goff[NG], xdst[NG];
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_row = nullptr;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = (tid >> 5) + JSTEP * i;
            int off = -2;                                   // -2: this thread has no element i; -1: zero
            if (j < KX) {
                off = -1;
                if (rd.valid && j < w.NIN)
                    off = dense ? rd.b * Tp * w.NIN + j
                                : sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            goff[i] = off;
            xdst[i] = a_frag_index(grow, j < KX ? j : 0);
        }
        if (!dense && rd.valid) {
            if (a.md_row != nullptr) md_row = a.md_row + (size_t)(slot0 + grow) * Tp;
            else md = a.md_utt[rd.b];
        }
    }
    float* Xf = reinterpret_cast<float*>(Xs);
    {
        const NormMD m0 = md_row ? md_row[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (goff[i] != -2) Xf[xdst[i]] = goff[i] >= 0 ? (gbase[goff[i]] - m0.m) / m0.d : 0.0f;
    }

    const float* __restrict__ bias_l0 = Bs + ((0 * NW + wave) * NT) * 32 + (lane & 31);
    const float* __restrict__ bias_l1 = Bs + ((1 * NW + wave) * NT) * 32 + (lane & 31);
    f32x16 h0r[ST], h1r[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) { h0r[s][r] = 0.0f; h1r[s][r] = 0.0f; }

    GStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)wave * KGT * NL * 256, 0, KGT * NL * 1024, 0x00020000);
    ws.voff = lane * 16;
    float4 breg[NL];
#pragma unroll
    for (int n = 0; n < NL; ++n) breg[n] = gload<NL>(ws, 0, n);
    int gnext = 1;

    // Linear: 8 rows x 2 outputs x 4 k-parts per wave (waves 0..3)
    const int fc_row = (wave & 3) * 8 + (lane & 7);
    const int fc_o = (lane >> 3) & 1;
    const int fc_kp = lane >> 4;
    const RowDesc fc_rd = rows_s[fc_row];
    auto fc_store = [&](int t_of_h) {
        if (wave < 4) {
            constexpr int KGP = KGH / 4;
            float sum = 0.0f;
#pragma unroll 4
            for (int kk = 0; kk < KGP; ++kk) {
                const int kg = fc_kp * KGP + kk;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const float4 h4 = H1s[kg * 64 + kh * 32 + fc_row];
                    const float4 w4 = Wfc4[(fc_o * KGH + kg) * 2 + kh];
                    sum += h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
                }
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            if (fc_kp == 0 && fc_rd.valid && t_of_h >= a.LA)
                a.out[(size_t)fc_rd.out_off + (size_t)fc_o * a.out_stride_o + (t_of_h - a.LA)] = apply_act(sum + w.bfc[fc_o], a.act);
        }
    };

    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        float xr[NG];
        NormMD mdn = md;
        const bool have_next = (t + 1 < Tp);
        if (have_next) {                                    // prefetch x(t+1) (consumed after the layer-0 MFMA phase)
            if (md_row) mdn = md_row[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = goff[i] >= 0 ? gbase[goff[i] + (t + 1) * gstep] : 0.0f;
        }

        f32x16 acc[NT];
        // ---------------- layer 0: input x_t (r, z, n_x), then h0_{t-1} (r, z, n_h) ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bias_l0[n * 32];
        gru_groups<ST, false>(acc, breg, Xs + lane, KGX, ws, gnext, KGT);
        gru_groups<ST, true>(acc, breg, H0s + lane, KGH, ws, gnext, KGT);
        __syncthreads();
        gru_cell<ST, UW>(acc, h0r, reinterpret_cast<float*>(H0s), wave, lane);
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if (goff[i] != -2) Xf[xdst[i]] = goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
        }
        if (t > 0) fc_store(t - 1);
        __syncthreads();
        // ---------------- layer 1: h1_{t-1} (r, z, n_h), then its input h0_t (r, z, n_x) ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bias_l1[n * 32];
        gru_groups<ST, true>(acc, breg, H1s + lane, KGH, ws, gnext, KGT);
        gru_groups<ST, false>(acc, breg, H0s + lane, KGH, ws, gnext, KGT);
        __syncthreads();
        gru_cell<ST, UW>(acc, h1r, reinterpret_cast<float*>(H1s), wave, lane);
    }
    __syncthreads();
    fc_store(Tp - 1);
}

// -------------------------------------------------------------------------------------------------
size_t gru_pack_floats(int H, int KX, int NW) {
    const int NL = 3 * (H / NW / 32);
    const int KGT = KX / 8 + H / 8 + 2 * (H / 8);
    return (size_t)NW * KGT * NL * 64 * 4;
}

// [wave][k-group][live tile][lane][k-pair] from the FOUR-SLOT matrices of fsnp_abi.hip's expand() (W_ih: slots r, z, n, 0;
// W_hh: r, z, 0, n - each [4H][cols]): live tile n of a k-group is slot n / ST for n < 2 ST, else slot 2 in an input segment
// and slot 3 in a hidden segment.  K order: layer 0 = [x (KX, zero padded) | h0], layer 1 = [h1 | h0] (as lstm.hip).
void gru_pack_weights(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1, const float* whh1,
                      float* wpack) {
    const int UW = H / NW, ST = UW / 32, NL = 3 * ST;
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    for (int wv = 0; wv < NW; ++wv)
        for (int g = 0; g < KGT; ++g) {
            // segment of this k-group: source matrix, its column count, first column, hidden / input
            const bool l0 = g < KG0;
            const bool hidden = l0 ? g >= KGX : g < KG0 + KGH;
            const float* src = l0 ? (hidden ? whh0 : wih0) : (hidden ? whh1 : wih1);
            const int cols = l0 ? (hidden ? H : NIN) : H;
            const int g0 = l0 ? (hidden ? KGX : 0) : (hidden ? KG0 : KG0 + KGH);
            for (int n = 0; n < NL; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int p = 0; p < 4; ++p) {
                        const int slot = n < 2 * ST ? n / ST : (hidden ? 3 : 2);
                        const int s = n % ST;
                        const int wrow = slot * H + wv * UW + s * 32 + (lane & 31);
                        const int k = 8 * (g - g0) + 2 * p + (lane >> 5);
                        float v = 0.0f;
                        if (k < cols) v = src[(size_t)wrow * cols + k];
                        wpack[((((size_t)wv * KGT + g) * NL + n) * 64 + lane) * 4 + p] = v;
                    }
        }
}

template <int KX>
static void launch_gru_kx(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    constexpr int HID = 384, OUT = 2, NW = 4;
    constexpr int KGX = KX / 8, KGH = HID / 8, NT = 4 * (HID / NW / 32);
    const size_t smem = (size_t)(KGX + 2 * KGH) * 64 * 16 + (size_t)OUT * KGH * 2 * 16 + 32 * sizeof(RowDesc) + (size_t)2 * NW * NT * 32 * 4;
    auto kern = gru2_fc_kernel<HID, KX, OUT, NW>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    LstmWeights wv = w;
    wv.wpack = w.wpack_gru;
    hipLaunchKernelGGL(kern, dim3(a.num_tiles), dim3(64 * NW), smem, s, wv, a);
}

// one 32-row tile per workgroup, any number of tiles (rounds of num_CUs run back to back); no VALU rows
void launch_gru(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (a.num_tiles <= 0) return;
    if (w.KX == 64) launch_gru_kx<64>(w, a, s);
    else launch_gru_kx<40>(w, a, s);
}

}  // namespace fsnp
