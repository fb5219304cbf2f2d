// lstm_bf3.hip - the one-tile-per-CU sub-band LSTM + Linear with fp32 products EMULATED by three bf16 MFMAs (gfx950).
//
// OPTIONAL precision mode (fsnp_set_precision(h, 2), never the headline, whose arithmetic stays the reference's fp32).
// Same decomposition as lstm.hip (SequenceModel.forward's LSTM branch, speech_enhance/audio_zen/model/module/
// sequence_model.py:113-123): 32 sequences per workgroup, wave w owns hidden units [w H/4, (w+1) H/4) of both layers as
// 4 gates x 3 accumulator tiles, weights stream L2 -> registers, x_t / h0 / h1 live in LDS, the Linear epilogue writes
// out[b, o, f, t - look_ahead].  What changes is the matrix instruction: lstm.hip is bound by v_mfma_f32_32x32x2_f32
// (157 TFLOP/s on the chip: 91 % reached); v_mfma_f32_32x32x16_bf16 retires 8x the K per instruction in half the cycles.
// Every operand is split into two bf16 values, v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits
// together), and a product a b is accumulated as  a_hi b_hi + a_lo b_hi + a_hi b_lo  in fp32 (the dropped a_lo b_lo term is
// 2^-16 relative): 3 MFMAs of 32 cycles per 16 K where fp32 needs 8 of 64 - 5.3x less matrix-pipe time; the weights
// (hi + lo = 4 bytes per value) cost the same L2 bytes as fp32, which becomes the new bound.
// Measured error vs the fp32 path: tests/test_gpu_parity.py::test_bf16x3_variant.
#include <cstring>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned short bf3_bits(float v) {            // round-to-nearest-even bf16 bits
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// v = hi + lo, both bf16
__device__ __forceinline__ void bf3_split(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf3_bits(v);
    lo = bf3_bits(v - __uint_as_float((unsigned)hi << 16));
}
// index (2-byte elements) of element (row, k) inside a bf16 A image of v_mfma_f32_32x32x16_bf16:
// [k-step of 16][k half of 8][row][8] - lane l of a step reads the 16 bytes of (half l >> 5, row l & 31)
__host__ __device__ __forceinline__ int bf3_a_index(int row, int k) {
    return (((k >> 4) * 64) + (((k >> 3) & 1) * 32) + row) * 8 + (k & 7);
}

struct B3Stream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
// fragment (k-step s, tile n, part 0 = hi / 1 = lo) of this wave's stream: 1 KiB each
template <int NT>
__device__ __forceinline__ float4 b3load(const B3Stream& ws, int s, int n, int part) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, ((s * NT + n) * 2 + part) * 1024, 0));
}

// `nsteps` k-steps of 16: per tile three MFMAs (hi hi, lo hi, hi lo), both weight fragments refilled in place from the stream
template <int NT>
__device__ __forceinline__ void bf3_steps(f32x16 (&acc)[NT], float4 (&bh)[NT], float4 (&bl)[NT], const float4* __restrict__ Ahi,
                                          const float4* __restrict__ Alo, int nsteps, const B3Stream& ws, int& snext, int steps_total) {
    for (int s = 0; s < nsteps; ++s) {
        const float4 ah = Ahi[s * 64], al = Alo[s * 64];       // (no register double buffer: the SIMD's other waves cover the LDS read)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[n]), acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[n]), acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[n]), acc[n], 0, 0, 0);
            bh[n] = b3load<NT>(ws, snext, n, 0);
            bl[n] = b3load<NT>(ws, snext, n, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        snext = (snext + 1 == steps_total) ? 0 : snext + 1;
    }
}

// lane-local cell update (as lstm.hip: lstm_cell); h goes to the hi / lo bf16 images
template <int ST, int UW>
__device__ __forceinline__ void bf3_cell(f32x16 (&acc)[4 * ST], f32x16 (&c)[ST], unsigned short* __restrict__ Hhi,
                                         unsigned short* __restrict__ Hlo, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int k = wave * UW + s * 32 + (lane & 31);
        const int kbase = (((k >> 4) * 64) + (((k >> 3) & 1) * 32)) * 8 + (k & 7);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                  // two cells per pass: packed fp32 math (lstm_common.h lstm_cell_pair)
            f32x2 cc{c[s][r], c[s][r + 1]};
            const f32x2 h = lstm_cell_pair(f32x2{acc[s][r], acc[s][r + 1]}, f32x2{acc[ST + s][r], acc[ST + s][r + 1]},
                                           f32x2{acc[2 * ST + s][r], acc[2 * ST + s][r + 1]},
                                           f32x2{acc[3 * ST + s][r], acc[3 * ST + s][r + 1]}, cc);
            c[s][r] = cc.x; c[s][r + 1] = cc.y;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            unsigned short hi, lo;
            bf3_split(h.x, hi, lo);
            Hhi[kbase + row * 8] = hi;
            Hlo[kbase + row * 8] = lo;
            bf3_split(h.y, hi, lo);
            Hhi[kbase + (row + 1) * 8] = hi;
            Hlo[kbase + (row + 1) * 8] = lo;
        }
    }
}

}  // namespace

template <int HID, int KX, int OUT, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void lstm2_fc_bf3_kernel(LstmWeights w, LstmArgs a) {
    static_assert(OUT == 2, "FC lane mapping assumes output_size == 2");
    constexpr int NTHR = 64 * NW;
    constexpr int UW = HID / NW, ST = UW / 32, NT = 4 * ST;
    static_assert(UW % 32 == 0 && UW * NW == HID && HID % 16 == 0, "tile shapes");
    constexpr int KSX = (KX + 15) / 16, KSH = HID / 16;            // k-steps of 16: x (zero padded), one hidden vector
    constexpr int KS0 = KSX + KSH, KST = KS0 + 2 * KSH;             // layer 0: [x | h0], layer 1: [h1 | h0]
    constexpr int XP = KSX * 16;                                    // padded input width

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xhi = reinterpret_cast<float4*>(smem_raw);   // [KSX][64] bf16 A image of x_t, hi parts (16 B per lane)
    float4* Xlo = Xhi + KSX * 64;
    float4* H0hi = Xlo + KSX * 64;                        // [KSH][64]
    float4* H0lo = H0hi + KSH * 64;
    float4* H1hi = H0lo + KSH * 64;
    float4* H1lo = H1hi + KSH * 64;
    float* Wfc = reinterpret_cast<float*>(H1lo + KSH * 64);            // [OUT][HID]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(Wfc + OUT * HID);     // [32]
    float* Bs = reinterpret_cast<float*>(rows_s + 32);                 // [2][NW][NT][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot0 = blockIdx.x * 32;
    const int Tp = a.Tp;

    for (int i = tid; i < (2 * KSX + 4 * KSH) * 64; i += NTHR) Xhi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < OUT * HID; i += NTHR) Wfc[i] = w.wfc[i];
    if (tid < 32) rows_s[tid] = a.rows[slot0 + tid];
    for (int i = tid; i < 2 * NW * NT * 32; i += NTHR) {
        const int col = i & 31, n = (i >> 5) % NT, wv = (i / (32 * NT)) % NW, layer = i / (32 * NT * NW);
        Bs[i] = w.bias[layer * 4 * HID + (n / ST) * HID + wv * UW + (n % ST) * 32 + col];
    }
    __syncthreads();

    // ---- gather plan (as lstm.hip): thread owns row = tid & 31, features j = (tid >> 5) + (NTHR / 32) i
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? w.NIN : a.FP;
    constexpr int JSTEP = NTHR / 32, NG = (XP + JSTEP - 1) / JSTEP;
    const int grow = tid & 31;
    int goff[NG], xdst[NG];
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_row = nullptr;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = (tid >> 5) + JSTEP * i;
            int off = -2;                                   // -2: no element; -1: zero
            if (j < XP) {
                off = -1;
                if (rd.valid && j < w.NIN)
                    off = dense ? rd.b * Tp * w.NIN + j
                                : sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            goff[i] = off;
            xdst[i] = bf3_a_index(grow, j < XP ? j : 0);
        }
        if (!dense && rd.valid) {
            if (a.md_row != nullptr) md_row = a.md_row + (size_t)(slot0 + grow) * Tp;
            else md = a.md_utt[rd.b];
        }
    }
    unsigned short* Xh16 = reinterpret_cast<unsigned short*>(Xhi);
    unsigned short* Xl16 = reinterpret_cast<unsigned short*>(Xlo);
    auto put_x = [&](int i, float v) {
        unsigned short hi, lo;
        bf3_split(v, hi, lo);
        Xh16[xdst[i]] = hi;
        Xl16[xdst[i]] = lo;
    };
    {
        const NormMD m0 = md_row ? md_row[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (goff[i] != -2) put_x(i, goff[i] >= 0 ? (gbase[goff[i]] - m0.m) / m0.d : 0.0f);
    }

    const float* __restrict__ bias_l0 = Bs + ((0 * NW + wave) * NT) * 32 + (lane & 31);
    const float* __restrict__ bias_l1 = Bs + ((1 * NW + wave) * NT) * 32 + (lane & 31);
    f32x16 c0[ST], c1[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[s][r] = 0.0f; c1[s][r] = 0.0f; }

    B3Stream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)wave * KST * NT * 512, 0, KST * NT * 2048, 0x00020000);
    ws.voff = lane * 16;
    float4 bh[NT], bl[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[n] = b3load<NT>(ws, 0, n, 0); bl[n] = b3load<NT>(ws, 0, n, 1); }
    int snext = 1;

    // Linear from the hi + lo images of h1: 8 rows x 2 outputs x 4 k-parts per wave (waves 0..3)
    const int fc_row = (wave & 3) * 8 + (lane & 7);
    const int fc_o = (lane >> 3) & 1;
    const int fc_kp = lane >> 4;
    const RowDesc fc_rd = rows_s[fc_row];
    auto fc_store = [&](int t_of_h) {
        if (wave < 4) {
            constexpr int K8P = HID / 8 / 4;               // 8-element groups per k-part
            const unsigned short* hh = reinterpret_cast<const unsigned short*>(H1hi);
            const unsigned short* hl = reinterpret_cast<const unsigned short*>(H1lo);
            float sum = 0.0f;
#pragma unroll 4
            for (int kk = 0; kk < K8P; ++kk) {
                const int k0 = (fc_kp * K8P + kk) * 8;
                const int idx = bf3_a_index(fc_row, k0);
                const uint4 vh = *reinterpret_cast<const uint4*>(hh + idx);
                const uint4 vl = *reinterpret_cast<const uint4*>(hl + idx);
                const unsigned wh[4] = {vh.x, vh.y, vh.z, vh.w}, wl[4] = {vl.x, vl.y, vl.z, vl.w};
                const float* wr = Wfc + fc_o * HID + k0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float e0 = __uint_as_float(wh[q] << 16) + __uint_as_float(wl[q] << 16);
                    const float e1 = __uint_as_float(wh[q] & 0xFFFF0000u) + __uint_as_float(wl[q] & 0xFFFF0000u);
                    sum += e0 * wr[2 * q] + e1 * wr[2 * q + 1];
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
        if (have_next) {
            if (md_row) mdn = md_row[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = goff[i] >= 0 ? gbase[goff[i] + (t + 1) * gstep] : 0.0f;
        }
        f32x16 acc[NT];
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bias_l0[n * 32];
        bf3_steps<NT>(acc, bh, bl, Xhi + lane, Xlo + lane, KSX, ws, snext, KST);
        bf3_steps<NT>(acc, bh, bl, H0hi + lane, H0lo + lane, KSH, ws, snext, KST);
        __syncthreads();
        bf3_cell<ST, UW>(acc, c0, reinterpret_cast<unsigned short*>(H0hi), reinterpret_cast<unsigned short*>(H0lo), wave, lane);
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if (goff[i] != -2) put_x(i, goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f);
        }
        if (t > 0) fc_store(t - 1);
        __syncthreads();
        // ---------------- layer 1: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bias_l1[n * 32];
        bf3_steps<NT>(acc, bh, bl, H1hi + lane, H1lo + lane, KSH, ws, snext, KST);
        bf3_steps<NT>(acc, bh, bl, H0hi + lane, H0lo + lane, KSH, ws, snext, KST);
        __syncthreads();
        bf3_cell<ST, UW>(acc, c1, reinterpret_cast<unsigned short*>(H1hi), reinterpret_cast<unsigned short*>(H1lo), wave, lane);
    }
    __syncthreads();
    fc_store(Tp - 1);
}

// -------------------------------------------------------------------------------------------------
static unsigned short host_bf16_rne(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static float host_bf16_to_float(unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

size_t lstm_bf3_pack_floats(int H, int KX, int NW) {
    const int NT = 4 * (H / NW / 32);
    const int KST = (KX + 15) / 16 + 3 * (H / 16);
    return (size_t)NW * KST * NT * 2 * 64 * 4;
}

// [wave][k-step][tile][hi | lo][lane][8 x bf16]: lane l of step s / tile n holds the 8 weights k = 16 s' + 8 (l >> 5) + j of its
// column (gate n / ST, unit wv UW + 32 (n % ST) + (l & 31)); K order as lstm.hip: layer 0 = [x (zero padded to 16 KSX) | h0],
// layer 1 = [h1 | h0].  hi = bf16(w), lo = bf16(w - hi).
void lstm_bf3_pack_weights(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1, const float* whh1,
                           float* wpack) {
    const int UW = H / NW, ST = UW / 32, NT = 4 * ST;
    const int KSX = (KX + 15) / 16, KSH = H / 16, KS0 = KSX + KSH, KST = KS0 + 2 * KSH;
    for (int wv = 0; wv < NW; ++wv)
        for (int s = 0; s < KST; ++s)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int gate = n / ST, sb = n % ST;
                    const int wrow = gate * H + wv * UW + sb * 32 + (lane & 31);
                    unsigned short* hi = reinterpret_cast<unsigned short*>(wpack + (((((size_t)wv * KST + s) * NT + n) * 2 + 0) * 64 + lane) * 4);
                    unsigned short* lo = reinterpret_cast<unsigned short*>(wpack + (((((size_t)wv * KST + s) * NT + n) * 2 + 1) * 64 + lane) * 4);
                    for (int j = 0; j < 8; ++j) {
                        float v = 0.0f;
                        if (s < KSX) {
                            const int k = 16 * s + 8 * (lane >> 5) + j;
                            if (k < NIN) v = wih0[(size_t)wrow * NIN + k];
                        } else if (s < KS0) {
                            v = whh0[(size_t)wrow * H + 16 * (s - KSX) + 8 * (lane >> 5) + j];
                        } else if (s < KS0 + KSH) {
                            v = whh1[(size_t)wrow * H + 16 * (s - KS0) + 8 * (lane >> 5) + j];
                        } else {
                            v = wih1[(size_t)wrow * H + 16 * (s - KS0 - KSH) + 8 * (lane >> 5) + j];
                        }
                        hi[j] = host_bf16_rne(v);
                        lo[j] = host_bf16_rne(v - host_bf16_to_float(hi[j]));
                    }
                }
}

void launch_lstm_bf3(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (a.num_tiles <= 0) return;
    // 12 waves (three per SIMD, 32 hidden units x 4 gates each): with 32-cycle MFMAs the kernel leans on the weight stream from
    // L2, and two more waves per SIMD cover a wave's load latency; 4 waves x 12 tiles x (hi + lo) fragments also spill
    constexpr int HID = 384, KX = 40, OUT = 2, NW = 12;
    constexpr int KSX = (KX + 15) / 16, KSH = HID / 16, NT = 4 * (HID / NW / 32);
    const size_t smem = (size_t)(2 * KSX + 4 * KSH) * 64 * 16 + (size_t)OUT * HID * 4 + 32 * sizeof(RowDesc) + (size_t)2 * NW * NT * 32 * 4;
    auto kern = lstm2_fc_bf3_kernel<HID, KX, OUT, NW>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    LstmWeights wv = w;
    wv.wpack = w.wpack_bf3;
    hipLaunchKernelGGL(kern, dim3(a.num_tiles), dim3(64 * NW), smem, s, wv, a);
}

}  // namespace fsnp
