// lstm16.hip - the one-tile-per-CU sub-band LSTM + Linear on HALF tiles: 16 sequences per workgroup (gfx950).
//
// Same arithmetic and decomposition as lstm.hip (SequenceModel.forward's LSTM branch, speech_enhance/audio_zen/model/module/
// sequence_model.py:113-123): a workgroup owns a tile of independent sequences and ALL 4 H gate columns, wave w the hidden units
// [w H/4, (w+1) H/4) of both layers, weights stream L2 -> registers in MFMA B-fragment order, x_t / h0 / h1 live in LDS in
// A-fragment order, the Linear epilogue writes out[b, o, f, t - look_ahead] - but on v_mfma_f32_16x16x4_f32 (M = 16 rows, exact
// fp32, the same 64 FLOP / clk / SIMD as the 32x32x2 form), so a round of 256 workgroups covers 4096 sequences in HALF the
// matrix-pipe time of a 32-row round.  It exists for the batches between the column-split kernels' range and a chip-filling
// 32-row round: the reference's literal drop-band call at B = 32 is 128 row tiles - 85 + 42 + 1 on the column-split kernels
// (135 us per step) - but exactly 256 half tiles: one launch, no inter-workgroup exchange at all.  A wave = 96 units x 4 gates
// = 24 accumulator tiles of 16 columns (96 registers; i / f / g / o of a (row, unit) share lane and register index: lane-local
// cell update, c in registers).  Cost: one 1 KiB weight fragment feeds 4 MFMAs of 32 cycles instead of 4 of 64, so the
// fragment loads weigh twice as much per matrix-pipe cycle as in lstm.hip (measured: profiles/r02_column_split.md).
#include <cstring>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// float index of A element (row < 16, k) inside a half-tile A image: [k-group of 16][k & 3][row][(k >> 2) & 3] - lane l of a
// k-group reads the float4 at (l >> 4 = k & 3, l & 15 = row), whose components feed the group's four MFMAs (k = 16 g + 4 q + (l >> 4))
__host__ __device__ __forceinline__ int a16_index(int row, int k) {
    return ((((k >> 4) * 4) + (k & 3)) * 16 + row) * 4 + ((k >> 2) & 3);
}

struct S16 {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
template <int NT>
__device__ __forceinline__ float4 w16load(const S16& ws, int g, int n) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, (g * NT + n) * 1024, 0));
}

// `ngroups` k-groups of 16 (A already offset by lane); b holds the group about to be used and is refilled in place
template <int NT>
__device__ __forceinline__ void groups16(f32x4 (&acc)[NT], float4 (&b)[NT], const float4* __restrict__ A, int ngroups, const S16& ws,
                                         int& gnext, int groups_total) {
    float4 a = A[0];
    for (int g = 0; g < ngroups; ++g) {
        const float4 an = A[(g + 1 < ngroups ? g + 1 : g) * 64];
        // two tiles at a time with their MFMAs interleaved, so no MFMA reads the accumulator the one before it wrote
#pragma unroll
        for (int n = 0; n < NT; n += 2) {
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[n].x, acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[n + 1].x, acc[n + 1], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[n].y, acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[n + 1].y, acc[n + 1], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[n].z, acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[n + 1].z, acc[n + 1], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[n].w, acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[n + 1].w, acc[n + 1], 0, 0, 0);
            b[n] = w16load<NT>(ws, gnext, n);
            b[n + 1] = w16load<NT>(ws, gnext, n + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        gnext = (gnext + 1 == groups_total) ? 0 : gnext + 1;
        a = an;
    }
}

// ---- bf16 ih-GEMM segment (BASELINE.json configs[4], round 4: the half-tile kernel too, so that `bf16_ih` means the same thing at
// B = 16 / parity-mode B = 32 as at the chip-filling batch): layer 1's W_ih1 . h0_t as HID / 32 k-steps of ONE
// v_mfma_f32_16x16x32_bf16 per tile (fp32 accumulate into the same tiles), operands = a bf16 A image of h0_t in LDS and bf16 weight
// fragments that travel through the same 16-byte-per-lane register pipeline as the fp32 groups (a k-step of 32 is NT KiB per wave,
// exactly one fp32 k-group's worth).  A / B operand: lane l holds the 8 elements k = 32 ks + 8 (l >> 4) + j of row / column l & 15.
using bf16x8_16 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ unsigned short bf16_bits16(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);           // round to nearest even
}
// index (in 2-byte elements) of element (row < 16, k) inside the bf16 half-tile A image: [k-step of 32][k quarter of 8][row][8]
__host__ __device__ __forceinline__ int a16_index_bf16(int row, int k) {
    return (((k >> 5) * 64) + (((k >> 3) & 3) * 16) + row) * 8 + (k & 7);
}
// (round 6: the bf16 A image of h0_t is a hi + lo PAIR, ALO float4 apart - every weight fragment multiplies both: lstm.hip mfma_groups_bf16)
template <int NT, int ALO>
__device__ __forceinline__ void groups16_bf16(f32x4 (&acc)[NT], float4 (&b)[NT], const float4* __restrict__ A, int nsteps, const S16& ws,
                                              int& gnext, int groups_total) {
    float4 a = A[0], al = A[ALO];
    for (int g = 0; g < nsteps; ++g) {
        const float4 an = A[(g + 1 < nsteps ? g + 1 : g) * 64], aln = A[ALO + (g + 1 < nsteps ? g + 1 : g) * 64];
#pragma unroll
        for (int n = 0; n < NT; n += 2) {
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_16, a), __builtin_bit_cast(bf16x8_16, b[n]), acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_16, a), __builtin_bit_cast(bf16x8_16, b[n + 1]), acc[n + 1], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_16, al), __builtin_bit_cast(bf16x8_16, b[n]), acc[n], 0, 0, 0);
            acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_16, al), __builtin_bit_cast(bf16x8_16, b[n + 1]), acc[n + 1], 0, 0, 0);
            b[n] = w16load<NT>(ws, gnext, n);
            b[n + 1] = w16load<NT>(ws, gnext, n + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        gnext = (gnext + 1 == groups_total) ? 0 : gnext + 1;
        a = an; al = aln;
    }
}

// lane-local cell update: accumulator register r of tile (gate, s) <-> row 4 (lane >> 4) + r, unit wave UW + 16 s + (lane & 15)
// (Hb != nullptr: h is also written, rounded to bf16, into the A image of the bf16 ih-GEMM segment)
template <int SB, int UW, int HBLO = 0>
__device__ __forceinline__ void cell16(f32x4 (&acc)[4 * SB], f32x4 (&c)[SB], float* __restrict__ Hs, int wave, int lane,
                                       unsigned short* __restrict__ Hb = nullptr) {
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const int k = wave * UW + s * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r += 2) {                   // two cells per pass: packed fp32 math (lstm_common.h lstm_cell_pair)
            auto rd = [](float v) { float o; asm("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v)); return o; };
            f32x2 cc{c[s][r], c[s][r + 1]};
            const f32x2 h = lstm_cell_pair(f32x2{rd(acc[s][r]), rd(acc[s][r + 1])}, f32x2{rd(acc[SB + s][r]), rd(acc[SB + s][r + 1])},
                                           f32x2{rd(acc[2 * SB + s][r]), rd(acc[2 * SB + s][r + 1])},
                                           f32x2{rd(acc[3 * SB + s][r]), rd(acc[3 * SB + s][r + 1])}, cc);
            c[s][r] = cc.x; c[s][r + 1] = cc.y;
            Hs[a16_index(4 * (lane >> 4) + r, k)] = h.x;
            Hs[a16_index(4 * (lane >> 4) + r + 1, k)] = h.y;
            if (Hb) {           // hi + lo: h = hi + lo to ~16 mantissa bits (HBLO 2-byte elements apart)
                const unsigned short hx = bf16_bits16(h.x), hy = bf16_bits16(h.y);
                const int ix = a16_index_bf16(4 * (lane >> 4) + r, k), iy = a16_index_bf16(4 * (lane >> 4) + r + 1, k);
                Hb[ix] = hx; Hb[iy] = hy;
                Hb[HBLO + ix] = bf16_bits16(h.x - __uint_as_float((unsigned)hx << 16));
                Hb[HBLO + iy] = bf16_bits16(h.y - __uint_as_float((unsigned)hy << 16));
            }
        }
    }
}

}  // namespace

// OWN (the sampled exchange verification's two-workgroup launch beside other forwards, fsnp_set_verify_sample): the kernel CLAIMS the whole
// register file of its SIMDs.  Its waves stream fp32 MFMAs back to back, and a wave of another kernel that shares such a SIMD is starved
// (measured ~10 x): with ~300 registers allocated, the small full-band kernels of the following forwards DID land beside it
// and every launch of theirs waited for those stragglers (tcn_dwconv_kernel 7.7 -> 110 us, prologue 16 -> 400 us while a sample ran:
// profiles/r06_verify_sample.md).  Touching v255 / a255 makes the kernel descriptor ask for all 512 registers: nothing else fits the CU.
template <int HID, int KX, int OUT, bool BF, bool OWN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lstm2_fc16_kernel(LstmWeights w, LstmArgs a) {
    if constexpr (OWN) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
    static_assert(OUT == 2, "FC lane mapping assumes output_size == 2");
    constexpr int NW = 4, UW = HID / NW, SB = UW / 16, NT = 4 * SB;      // 96 units, 6 blocks of 16, 24 tiles per wave
    static_assert(UW % 16 == 0 && HID % 128 == 0, "tile shapes");
    constexpr int KGX = (KX + 15) / 16, KGH = HID / 16, KG0 = KGX + KGH;                         // k-groups of 16
    constexpr int KSB = HID / 32;                                                                  // bf16 k-steps of 32 (BF only)
    constexpr int KGT = KG0 + KGH + (BF ? KSB : KGH);                                              // weight-stream groups per step
    constexpr int XP = KGX * 16;
    static_assert(!BF || HID % 32 == 0, "bf16 k-steps");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);   // [KGX][64] A image of x_t
    float4* H0s = Xs + KGX * 64;                         // [KGH][64]
    float4* H1s = H0s + KGH * 64;                        // [KGH][64]
    float* Wfc = reinterpret_cast<float*>(H1s + KGH * 64);               // [OUT][HID]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(Wfc + OUT * HID);       // [16]
    float* Bs = reinterpret_cast<float*>(rows_s + 16);                   // [2][NW][NT][16]
    float4* H0b = reinterpret_cast<float4*>(Bs + 2 * NW * NT * 16);      // BF: [hi | lo][KSB][64] bf16 A images of h0_t

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot0 = blockIdx.x * 16;
    const int Tp = a.Tp;

    for (int i = tid; i < (KGX + 2 * KGH) * 64; i += 256) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < OUT * HID; i += 256) Wfc[i] = w.wfc[i];
    if (tid < 16) rows_s[tid] = a.rows[slot0 + tid];
    for (int i = tid; i < 2 * NW * NT * 16; i += 256) {
        const int col = i & 15, n = (i >> 4) % NT, wv = (i / (16 * NT)) % NW, layer = i / (16 * NT * NW);
        Bs[i] = w.bias[layer * 4 * HID + (n / SB) * HID + wv * UW + (n % SB) * 16 + col];
    }
    __syncthreads();

    // ---- gather plan: thread owns row = tid & 15, features j = (tid >> 4) + 16 i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? w.NIN : a.FP;
    constexpr int NG = XP / 16;
    const int grow = tid & 15;
    int goff[NG], xdst[NG];
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_row = nullptr;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = (tid >> 4) + 16 * i;
            int off = -1;
            if (rd.valid && j < w.NIN)
                off = dense ? rd.b * Tp * w.NIN + j
                            : sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            goff[i] = off;
            xdst[i] = a16_index(grow, j);
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
        for (int i = 0; i < NG; ++i) Xf[xdst[i]] = goff[i] >= 0 ? (gbase[goff[i]] - m0.m) / m0.d : 0.0f;
    }

    const float* __restrict__ bias_l0 = Bs + ((0 * NW + wave) * NT) * 16 + (lane & 15);
    const float* __restrict__ bias_l1 = Bs + ((1 * NW + wave) * NT) * 16 + (lane & 15);
    f32x4 c0[SB], c1[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) { c0[s][r] = 0.0f; c1[s][r] = 0.0f; }

    S16 ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)wave * KGT * NT * 256, 0, KGT * NT * 1024, 0x00020000);
    ws.voff = lane * 16;
    float4 breg[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) breg[n] = w16load<NT>(ws, 0, n);
    int gnext = 1;

    // Linear: wave w sums rows 4 w .. 4 w + 3 x 2 outputs over 8 k-parts of HID / 8 (all in-wave)
    const int fc_row = 4 * wave + (lane & 3);
    const int fc_o = (lane >> 2) & 1;
    const int fc_kp = lane >> 3;
    const RowDesc fc_rd = rows_s[fc_row];
    auto fc_store = [&](int t_of_h) {
        constexpr int GPP = KGH / 8;                       // k-groups of 16 per k-part
        float sum = 0.0f;
#pragma unroll
        for (int gg = 0; gg < GPP; ++gg) {
            const int g = fc_kp * GPP + gg;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 h4 = H1s[(g * 4 + k4) * 16 + fc_row];          // k = 16 g + 4 q + k4, q = component
                const float* wr = Wfc + fc_o * HID + 16 * g + k4;
                sum += h4.x * wr[0] + h4.y * wr[4] + h4.z * wr[8] + h4.w * wr[12];
            }
        }
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (fc_kp == 0 && fc_rd.valid && t_of_h >= a.LA)
            a.out[(size_t)fc_rd.out_off + (size_t)fc_o * a.out_stride_o + (t_of_h - a.LA)] = apply_act(sum + w.bfc[fc_o], a.act);
    };

    __syncthreads();

    for (int t = 0; t < Tp; ++t) {
        float xr[NG];
        NormMD mdn = md;
        const bool have_next = (t + 1 < Tp);
        if (have_next) {                                    // prefetch x(t+1)
            if (md_row) mdn = md_row[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = goff[i] >= 0 ? gbase[goff[i] + (t + 1) * gstep] : 0.0f;
        }
        f32x4 acc[NT];
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[n][r] = bias_l0[n * 16];
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[n]));
        groups16<NT>(acc, breg, Xs + lane, KGX, ws, gnext, KGT);
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[n]));
        groups16<NT>(acc, breg, H0s + lane, KGH, ws, gnext, KGT);
        __syncthreads();
        cell16<SB, UW, KSB * 64 * 8>(acc, c0, reinterpret_cast<float*>(H0s), wave, lane, BF ? reinterpret_cast<unsigned short*>(H0b) : nullptr);
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i) Xf[xdst[i]] = goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
        }
        if (t > 0) fc_store(t - 1);
        __syncthreads();
        // ---------------- layer 1: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[n][r] = bias_l1[n * 16];
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[n]));   // pin the tiles to AGPRs: left alone, hipcc moves them to VGPRs and back
                                                                        // inside one of the k-loops (124 v_accvgpr moves per k-group)
        groups16<NT>(acc, breg, H1s + lane, KGH, ws, gnext, KGT);
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" : "+a"(acc[n]));
        if constexpr (BF) groups16_bf16<NT, KSB * 64>(acc, breg, H0b + lane, KSB, ws, gnext, KGT);
        else groups16<NT>(acc, breg, H0s + lane, KGH, ws, gnext, KGT);
        __syncthreads();
        cell16<SB, UW>(acc, c1, reinterpret_cast<float*>(H1s), wave, lane);
    }
    __syncthreads();
    fc_store(Tp - 1);
}

// -------------------------------------------------------------------------------------------------
size_t lstm16_pack_floats(int H, int KX) {
    const int NT = 4 * (H / 4 / 16);
    const int KGT = (KX + 15) / 16 + 3 * (H / 16);
    return (size_t)4 * KGT * NT * 64 * 4;
}

// [wave][k-group of 16][tile][lane][q]: tile n = gate * SB + s holds columns unit = wv UW + 16 s + (lane & 15) of gate `gate`;
// component q of lane l is k = 16 g + 4 q + (l >> 4).  K order as lstm.hip: layer 0 = [x (zero padded to 16 KGX) | h0], layer 1 = [h1 | h0].
void lstm16_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* wpack) {
    const int UW = H / 4, SB = UW / 16, NT = 4 * SB;
    const int KGX = (KX + 15) / 16, KGH = H / 16, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    for (int wv = 0; wv < 4; ++wv)
        for (int g = 0; g < KGT; ++g)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int gate = n / SB, s = n % SB;
                        const int wrow = gate * H + wv * UW + s * 16 + (lane & 15);
                        float v = 0.0f;
                        if (g < KGX) {
                            const int k = 16 * g + 4 * q + (lane >> 4);
                            if (k < NIN) v = wih0[(size_t)wrow * NIN + k];
                        } else if (g < KG0) {
                            v = whh0[(size_t)wrow * H + 16 * (g - KGX) + 4 * q + (lane >> 4)];
                        } else if (g < KG0 + KGH) {
                            v = whh1[(size_t)wrow * H + 16 * (g - KG0) + 4 * q + (lane >> 4)];
                        } else {
                            v = wih1[(size_t)wrow * H + 16 * (g - KG0 - KGH) + 4 * q + (lane >> 4)];
                        }
                        wpack[((((size_t)wv * KGT + g) * NT + n) * 64 + lane) * 4 + q] = v;
                    }
}

// bf16-ih variant of the stream: layer 0 and the h1 part of layer 1 as above (fp32), then HID / 32 bf16 k-steps of W_ih1: lane l of
// step ks / tile n holds the 8 weights k = 32 ks + 8 (l >> 4) + j of its column, 2 bytes each (16 bytes per lane, like an fp32 group)
size_t lstm16_pack_floats_bf16ih(int H, int KX) {
    const int NT = 4 * (H / 4 / 16);
    const int KGT = (KX + 15) / 16 + 2 * (H / 16) + H / 32;
    return (size_t)4 * KGT * NT * 64 * 4;
}
static unsigned short host_bf16_rne16(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
void lstm16_pack_weights_bf16ih(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* wpack) {
    const int UW = H / 4, SB = UW / 16, NT = 4 * SB;
    const int KGX = (KX + 15) / 16, KGH = H / 16, KG0 = KGX + KGH, KSB = H / 32, KGT = KG0 + KGH + KSB;
    for (int wv = 0; wv < 4; ++wv)
        for (int g = 0; g < KGT; ++g)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int gate = n / SB, s = n % SB;
                    const int wrow = gate * H + wv * UW + s * 16 + (lane & 15);
                    float* dst = wpack + ((((size_t)wv * KGT + g) * NT + n) * 64 + lane) * 4;
                    if (g < KG0 + KGH) {
                        for (int q = 0; q < 4; ++q) {
                            float v = 0.0f;
                            if (g < KGX) {
                                const int k = 16 * g + 4 * q + (lane >> 4);
                                if (k < NIN) v = wih0[(size_t)wrow * NIN + k];
                            } else if (g < KG0) {
                                v = whh0[(size_t)wrow * H + 16 * (g - KGX) + 4 * q + (lane >> 4)];
                            } else {
                                v = whh1[(size_t)wrow * H + 16 * (g - KG0) + 4 * q + (lane >> 4)];
                            }
                            dst[q] = v;
                        }
                    } else {
                        const int ks = g - KG0 - KGH;
                        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
                        for (int j = 0; j < 8; ++j) d16[j] = host_bf16_rne16(wih1[(size_t)wrow * H + 32 * ks + 8 * (lane >> 4) + j]);
                    }
                }
}

template <bool BF>
static void launch_lstm16_bf(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    constexpr int HID = 384, KX = 40, OUT = 2;
    constexpr int KGX = (KX + 15) / 16, KGH = HID / 16, NT = 4 * (HID / 4 / 16);
    const size_t smem = (size_t)(KGX + 2 * KGH) * 64 * 16 + (size_t)OUT * HID * 4 + 16 * sizeof(RowDesc) + (size_t)2 * 4 * NT * 16 * 4 +
                        (BF ? (size_t)2 * (HID / 32) * 64 * 16 : 0);
    LstmWeights wv = w;
    wv.wpack = BF ? w.wpack16_bf : w.wpack16;
    if constexpr (!BF) {
        if (a.coop_own_cu != 0) {          // a launch that shares the chip with other kernels: own the CUs it runs on (see OWN above)
            auto ko = lstm2_fc16_kernel<HID, KX, OUT, false, true>;
            static PerDeviceOnce own_once;
            own_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ko), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
            hipLaunchKernelGGL(ko, dim3(a.num_tiles), dim3(256), smem, s, wv, a);
            return;
        }
    }
    auto kern = lstm2_fc16_kernel<HID, KX, OUT, BF>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    hipLaunchKernelGGL(kern, dim3(a.num_tiles), dim3(256), smem, s, wv, a);
}
// one 16-row tile per workgroup, any number of tiles (rounds of num_CUs run back to back); w.ih_bf16: the bf16 ih-GEMM variant
void launch_lstm16(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (a.num_tiles <= 0) return;
    if (w.ih_bf16 && w.wpack16_bf) launch_lstm16_bf<true>(w, a, s);
    else launch_lstm16_bf<false>(w, a, s);
}

}  // namespace fsnp
