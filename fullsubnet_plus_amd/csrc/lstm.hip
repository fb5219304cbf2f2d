// lstm.hip - fused two-layer sub-band LSTM + Linear for gfx950 (MI355X).
//
// Replaces SequenceModel.forward's LSTM branch
//   (speech_enhance/audio_zen/model/module/sequence_model.py:113-123: permute -> nn.LSTM(34,384,
//    num_layers=2, batch_first) -> Linear(384,2) -> permute)
// together with the tensor plumbing in front of it in FullSubNet_Plus.forward
//   (speech_enhance/fullsubnet_plus/model/fullsubnet_plus.py:167-206: four unfolds, cat, second norm,
//    drop_band, reshape, and the final reshape/permute/look-ahead slice),
// none of which is materialised: every 32-sequence tile gathers its 34-dim input frame from the
// time-major [utt][t][freq] full-band buffers each step, and writes its 2 mask values straight into
// out[b, o, f, t - look_ahead].
//
// Mapping (one workgroup per CU; NW = 4 waves (one per SIMD, 512 registers) or NW = 12 (three per SIMD,
// 168 registers: measured on gfx950, a wave's own loads / LDS reads / VALU ops each ADD 7-23 cycles to its
// MFMA stream - tools/ubench/mfma_issue.hip - so only OTHER waves' MFMAs can cover them); described for NW = 4:
//   * rows      : 32 independent sequences per workgroup = the M of v_mfma_f32_32x32x2_f32.
//   * columns   : wave w owns hidden units [w*H/4, (w+1)*H/4) of BOTH layers, i.e. 4 gates x 96 units
//                 = 12 accumulator tiles of 32 columns (192 accumulator registers); the i/f/g/o values
//                 of one (row, unit) land in the same lane/register of 4 tiles, so the cell update is
//                 lane-local and c never leaves registers.
//   * K         : layer 0: [x_t (34, zero-padded to KX=40) | h0_{t-1} (384)], layer 1: [h1_{t-1} | h0_t].
//   * B operand : weights are pre-packed on the host in exact MFMA B-fragment order, one float4 per
//                 lane per (8-deep k-group, tile) = 4 MFMAs; each wave streams its private 1.8 MB/step
//                 slice straight from L2 into registers (no LDS: no other wave shares it), always one
//                 k-group (12 x 1 KiB loads) ahead, continuously across phase boundaries.
//   * A operand : x_t, h0, h1 live in LDS in A-fragment order ([k-group][k parity][row][4 k-pairs]) so
//                 one ds_read_b128 per k-group feeds 48 MFMAs.
//   * extra rows: a tile may carry up to EX more sequences (rows 32..32+EX-1).  Their gate pre-activations
//                 are plain v_fma_f32 on the VALU pipe, issued between the MFMAs and reusing the very same
//                 B registers (lane = (k parity, column), two half-wave partial sums combined once per
//                 layer).  The VALU is otherwise idle while the matrix pipe is busy, so the extra rows
//                 are almost free; they exist to kill the tail: B=32 gives 8224 = 257 x 32 sequences,
//                 i.e. one tile more than the chip has CUs - 33-row tiles finish in ONE round instead of two.
// fp32 throughout: v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain.
#include <cstring>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

// float index of extra-row element (e, k) inside an E-image ([k-group][k parity][EX][4 k-pairs])
template <int EX>
__host__ __device__ __forceinline__ int e_frag_index(int e, int k) {
    return ((((k >> 3) * 2) + (k & 1)) * EX + e) * 4 + ((k >> 1) & 3);
}

// Consume `ngroups` k-groups: A from LDS (A already offset by lane), B from the rotating register
// buffer `b` (always holding the group about to be used); refills b from the weight stream.
// AE (offset by (lane>>5)*EX) is the E-image of the extra rows; accx their per-lane partial sums.
// One 1 KiB weight fragment (tile n of k-group g) of this wave's stream: SRD in SGPRs, constant per-lane
// voffset (lane * 16), everything else in the scalar offset - no VALU address math in the hot loop, and the
// cheapest load form measured next to MFMAs (tools/ubench/mfma_issue.hip: 18 vs 23.5 pipe cycles).
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
template <int NT>
__device__ __forceinline__ float4 wload(const WStream& ws, int g, int n) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, (g * NT + n) * 1024, 0);
    return __builtin_bit_cast(float4, v);
}

template <int NT, int EX>
__device__ __forceinline__ void mfma_groups(f32x16 (&acc)[NT], float (&accx)[EX > 0 ? EX : 1][NT], float4 (&b)[NT],
                                            const float4* __restrict__ A, const float4* __restrict__ AE, int ngroups,
                                            const WStream& ws, int& gnext, int groups_total) {
    float4 a = A[0];
    float4 ae[EX > 0 ? EX : 1];
#pragma unroll
    for (int e = 0; e < EX; ++e) ae[e] = AE[e];
    for (int g = 0; g < ngroups; ++g) {
        const int gn = (g + 1 < ngroups ? g + 1 : g);
        const float4 an = A[gn * 64];
        float4 aen[EX > 0 ? EX : 1];
#pragma unroll
        for (int e = 0; e < EX; ++e) aen[e] = AE[gn * 2 * EX + e];
        if constexpr (EX == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[n].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[n].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[n].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[n].w, acc[n], 0, 0, 0);
                // refill the just-consumed registers with the same tile of the NEXT k-group, and pin the
                // (4 x MFMA, refill) order per tile: left alone hipcc hoists all 48 MFMAs above the refills,
                // needs 96 B registers, parks the refills in AGPRs and drains vmcnt(0) every group.
                b[n] = wload<NT>(ws, gnext, n);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // With VALU rows the group is processed in two halves: MFMA chains of the half, then ALL its
            // v_fma in one batch (a batched v_fma costs ~5 matrix-pipe cycles, an isolated one ~13 -
            // tools/ubench/mfma_issue.hip), then the half's refills (half a group = >1500 cycles ahead of use).
            constexpr int HALF = NT / 2;
#pragma unroll
            for (int hb = 0; hb < NT; hb += HALF) {
#pragma unroll
                for (int n = hb; n < hb + HALF; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[n].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[n].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[n].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[n].w, acc[n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = hb; n < hb + HALF; ++n)
#pragma unroll
                    for (int e = 0; e < EX; ++e) {
                        float v = accx[e][n];
                        v = fmaf(ae[e].x, b[n].x, v);
                        v = fmaf(ae[e].y, b[n].y, v);
                        v = fmaf(ae[e].z, b[n].z, v);
                        accx[e][n] = fmaf(ae[e].w, b[n].w, v);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = hb; n < hb + HALF; ++n) b[n] = wload<NT>(ws, gnext, n);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gnext = (gnext + 1 == groups_total) ? 0 : gnext + 1;
        a = an;
#pragma unroll
        for (int e = 0; e < EX; ++e) ae[e] = aen[e];
    }
}

// bf16 (round-to-nearest-even) bits of an fp32 value
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// index (in 2-byte elements) of element (row, k) inside the bf16 A image of v_mfma_f32_32x32x16_bf16:
// [k-step of 16][k half of 8][row][8]  - lane l of a step reads the 16 bytes of (half l>>5, row l&31)
__host__ __device__ __forceinline__ int a_frag_index_bf16(int row, int k) {
    return (((k >> 4) * 64) + (((k >> 3) & 1) * 32) + row) * 8 + (k & 7);
}

// HBLO: 2-byte elements between the hi and the lo bf16 image of h0_t (Hb != nullptr only)
template <int ST, int UW, int HBLO = 0>
__device__ __forceinline__ void lstm_cell(f32x16 (&acc)[4 * ST], f32x2 (&c)[ST][8], float* __restrict__ Hs, int wave,
                                          int lane, unsigned short* __restrict__ Hb = nullptr) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int k = wave * UW + s * 32 + (lane & 31);
        const int kbase = (((k >> 3) * 64) + ((k & 1) * 32)) * 4 + ((k >> 1) & 3);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                  // two cells per pass: packed fp32 math (lstm_common.h)
            // element-wise AGPR reads: handed `f32x2{acc[..][r], acc[..][r + 1]}` hipcc copies whole 16-register tiles to VGPRs
            // (ST > 1 = one wave per SIMD: the accumulators live in AGPRs; with 12 waves they are VGPRs already)
            auto rd = [](float v) {
                if constexpr (ST > 1) { float o; asm("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v)); return o; }
                else return v;
            };
            const f32x2 h = lstm_cell_pair(f32x2{rd(acc[s][r]), rd(acc[s][r + 1])}, f32x2{rd(acc[ST + s][r]), rd(acc[ST + s][r + 1])},
                                           f32x2{rd(acc[2 * ST + s][r]), rd(acc[2 * ST + s][r + 1])},
                                           f32x2{rd(acc[3 * ST + s][r]), rd(acc[3 * ST + s][r + 1])}, c[s][r >> 1]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);     // (r even: r + 1 is the next row)
            Hs[kbase + row * 4] = h.x;
            Hs[kbase + (row + 1) * 4] = h.y;
            if (Hb) {           // bf16 A image of h0_t as hi + lo (round 6): h = hi + lo to ~16 mantissa bits, two MFMAs per weight fragment
                const unsigned short hx = bf16_bits(h.x), hy = bf16_bits(h.y);
                Hb[a_frag_index_bf16(row, k)] = hx; Hb[a_frag_index_bf16(row + 1, k)] = hy;
                if constexpr (HBLO > 0) {
                    Hb[HBLO + a_frag_index_bf16(row, k)] = bf16_bits(h.x - __uint_as_float((unsigned)hx << 16));
                    Hb[HBLO + a_frag_index_bf16(row + 1, k)] = bf16_bits(h.y - __uint_as_float((unsigned)hy << 16));
                }
            }
        }
    }
}

// bf16 ih-GEMM segment (BASELINE.json configs[4]): `nsteps` k-steps of 16, ONE v_mfma_f32_32x32x16_bf16 per tile and
// step (fp32 accumulate into the same tiles), operands = bf16 A image of h0_t in LDS and bf16 weight fragments that
// travel through the same 16-byte-per-lane register pipeline as the fp32 groups.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
// ALO: float4 units between the hi and the lo bf16 A image (round 6): every weight fragment multiplies both - the activation's own
// quantisation error (half of the mode's error budget, profiles/r06_bf16_error.md) leaves the product, the weights stay bf16
template <int NT, int EX, int ALO>
__device__ __forceinline__ void mfma_groups_bf16(f32x16 (&acc)[NT], float (&accx)[EX > 0 ? EX : 1][NT], float4 (&b)[NT],
                                                 const float4* __restrict__ A, const float4* __restrict__ AE, int nsteps,
                                                 const WStream& ws, int& gnext, int groups_total, int lane) {
    float4 a = A[0], al = A[ALO];                 // (ALO == 0 - VALU-row tiles - : hi only)
    for (int g = 0; g < nsteps; ++g) {
        const float4 an = A[(g + 1 < nsteps ? g + 1 : g) * 64], aln = A[ALO + (g + 1 < nsteps ? g + 1 : g) * 64];
        // VALU rows: this lane's 8 weights of tile n are k = 16 g + 8 (lane>>5) + j, i.e. fp32 k-group 2g + (lane>>5)
        // of the E image (kh = 0: j even, kh = 1: j odd)
        float4 he[EX > 0 ? EX : 1][2];
        if constexpr (EX > 0) {
            const int kg = 2 * g + (lane >> 5);
#pragma unroll
            for (int e = 0; e < EX; ++e) { he[e][0] = AE[(kg * 2 + 0) * EX + e]; he[e][1] = AE[(kg * 2 + 1) * EX + e]; }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if constexpr (EX > 0) {
                const unsigned w0 = __float_as_uint(b[n].x), w1 = __float_as_uint(b[n].y), w2 = __float_as_uint(b[n].z),
                               w3 = __float_as_uint(b[n].w);
#pragma unroll
                for (int e = 0; e < EX; ++e) {
                    float v = accx[e][n];
                    v = fmaf(he[e][0].x, __uint_as_float(w0 << 16), v);
                    v = fmaf(he[e][1].x, __uint_as_float(w0 & 0xFFFF0000u), v);
                    v = fmaf(he[e][0].y, __uint_as_float(w1 << 16), v);
                    v = fmaf(he[e][1].y, __uint_as_float(w1 & 0xFFFF0000u), v);
                    v = fmaf(he[e][0].z, __uint_as_float(w2 << 16), v);
                    v = fmaf(he[e][1].z, __uint_as_float(w2 & 0xFFFF0000u), v);
                    v = fmaf(he[e][0].w, __uint_as_float(w3 << 16), v);
                    accx[e][n] = fmaf(he[e][1].w, __uint_as_float(w3 & 0xFFFF0000u), v);
                }
            }
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[n]),
                                                             acc[n], 0, 0, 0);
            if constexpr (ALO > 0)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, b[n]),
                                                                 acc[n], 0, 0, 0);
            b[n] = wload<NT>(ws, gnext, n);
            __builtin_amdgcn_sched_barrier(0);
        }
        gnext = (gnext + 1 == groups_total) ? 0 : gnext + 1;
        a = an; al = aln;
    }
}

// Extra rows: combine the two half-wave partial sums, add the bias, update c, write h into the E-image.
// (bias == nullptr: the bias rode in the product - bf16-ih variant, layer 0: folded into the weight column of a constant-1 input)
template <int ST, int UW, int EX, int NT>
__device__ __forceinline__ void lstm_cell_extra(float (&accx)[EX > 0 ? EX : 1][NT], float (&cx)[EX > 0 ? EX : 1][ST],
                                                const float* __restrict__ bias, float* __restrict__ He, int wave,
                                                int lane) {
#pragma unroll
    for (int e = 0; e < EX; ++e)
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            float g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = accx[e][q * ST + s];
                g4[q] = v + __shfl_xor(v, 32) + (bias ? bias[(q * ST + s) * 32] : 0.0f);
            }
            const float cn = fast_sigmoid(g4[1]) * cx[e][s] + fast_sigmoid(g4[0]) * fast_tanh(g4[2]);
            cx[e][s] = cn;
            const float h = fast_sigmoid(g4[3]) * fast_tanh(cn);
            if (lane < 32) He[e_frag_index<EX>(e, wave * UW + s * 32 + lane)] = h;
        }
}

template <int HID, int KX, int OUT, int EX, bool PROF, int NW, bool BF>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void lstm2_fc_kernel(LstmWeights w, LstmArgs a) {
    static_assert(OUT == 2, "FC lane mapping assumes output_size == 2");
    static_assert(EX >= 0 && EX <= 4, "at most 4 VALU rows per tile (one FC wave per extra row)");
    static_assert(NW % 4 == 0 && NW >= 4, "whole waves per SIMD");
    constexpr int NTHR = 64 * NW;
    constexpr int UW = HID / NW, ST = UW / 32, NT = 4 * ST;   // hidden units, 32-unit blocks, tiles per wave
    static_assert(UW % 32 == 0 && UW * NW == HID, "hidden/NW must be a multiple of 32");
    constexpr int KGX = KX / 8, KGH = HID / 8, KG0 = KGX + KGH;
    constexpr int KSB = HID / 16;                                   // bf16 k-steps of the layer-1 ih segment (BF only)
    constexpr int KG1 = BF ? KGH + KSB : 2 * KGH, KGT = KG0 + KG1;   // weight-stream groups per step
    static_assert(KGH % 4 == 0, "FC k-split");
    static_assert(!BF || HID % 16 == 0, "bf16 k-steps");
    constexpr int RT = 32 + EX;                 // row slots per tile
    constexpr int EXA = EX > 0 ? EX : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* Xs = reinterpret_cast<float4*>(smem_raw);   // [KGX][64]           A image of x_t (rows 0..31)
    float4* H0s = Xs + KGX * 64;                         // [KGH][64]           h0
    float4* H1s = H0s + KGH * 64;                        // [KGH][64]           h1
    float4* XEs = H1s + KGH * 64;                        // [KGX][2][EX]        E image of x_t (extra rows)
    float4* HE0s = XEs + KGX * 2 * EX;                   // [KGH][2][EX]
    float4* HE1s = HE0s + KGH * 2 * EX;                  // [KGH][2][EX]
    float4* Wfc4 = HE1s + KGH * 2 * EX;                  // [OUT][KGH][2]
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(Wfc4 + OUT * KGH * 2);  // [RT]
    // BF (bf16-ih variant): the layer-0 bias is FOLDED into the product - input slot k = NIN (one of the zero-padded columns of the x
    // image) holds the constant 1, the weight column k = NIN holds b_ih0 + b_hh0 (lstm_pack_weights_bf16ih) - so the table holds layer 1
    // only: the 6 KB that frees are what the second (lo) bf16 image of h0_t needs to fit a CU's 160 KB next to the fp32 images
    constexpr int BL = BF ? 1 : 2;                                       // bias layers in LDS
    constexpr bool HILO = BF && EX == 0;                                 // h0_t as bf16 hi + lo (VALU-row tiles: hi only, the E images take the room)
    float* Bs = reinterpret_cast<float*>(rows_s + RT);                   // [BL][NW][NT][32]
    float4* H0b = reinterpret_cast<float4*>(Bs + BL * NW * NT * 32);     // BF: [hi | lo][KSB][64] bf16 A images of h0_t (h = hi + lo)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot0 = blockIdx.x * RT;
    const int Tp = a.Tp;

    // which clock does this launch hold?  workgroup 0 stamps the shader-side counter and the 100 MHz wall clock on entry and on exit
    // (straight to host-mapped memory: nothing stays live across the time loop); bench.py reports the ratio (fsnp_debug_launch_clock)
    // ... and EVERY workgroup folds its own duration into a maximum / minimum (one 64-bit atomic each at the very end): the launch lasts as
    // long as its slowest workgroup, and the chip's XCDs do not hold the same clock (profiles/r06_box_variance.md)
    __shared__ unsigned long long clk_s[2];
    if (a.clk != nullptr && tid == 0) {
        clk_s[1] = __builtin_amdgcn_s_memrealtime();
        clk_s[0] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0) { a.clk[0] = clk_s[0]; a.clk[1] = clk_s[1]; a.clk[4] = 0ull; a.clk[5] = 0ull; a.clk[6] = ~0ull; }
    }

    for (int i = tid; i < (KGX + 2 * KGH) * (64 + 2 * EX); i += NTHR) Xs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < OUT * KGH * 2; i += NTHR) {
        const int o = i / (KGH * 2), kg = (i >> 1) % KGH, kh = i & 1;
        const float* wr = w.wfc + (size_t)o * HID + kg * 8 + kh;
        Wfc4[i] = make_float4(wr[0], wr[2], wr[4], wr[6]);
    }
    if (tid < RT) rows_s[tid] = a.rows[slot0 + tid];
    for (int i = tid; i < BL * NW * NT * 32; i += NTHR) {
        const int col = i & 31, n = (i >> 5) % NT, wv = (i / (32 * NT)) % NW, layer = BF ? 1 : i / (32 * NT * NW);
        Bs[i] = w.bias[layer * 4 * HID + (n / ST) * HID + wv * UW + (n % ST) * 32 + col];
    }
    __syncthreads();
    if constexpr (BF) {                   // the constant-1 input column (main rows and extra rows); never written again: its owner has no element
        if (tid < 32) reinterpret_cast<float*>(Xs)[a_frag_index(tid, w.NIN)] = 1.0f;
        if (EX > 0 && tid < EX) reinterpret_cast<float*>(XEs)[e_frag_index<EXA>(tid, w.NIN)] = 1.0f;
    }

    // ---- gather plan.  main rows: row = tid & 31, features j = (tid >> 5) + 8 i.
    //      extra rows: thread tid < EX*KX owns (e = tid / KX, j = tid % KX).
    // 32-bit float offsets from one uniform base (att_mag, or the dense input); -1 = zero.
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? w.NIN : a.FP;
    auto plan = [&](const RowDesc& rd, int j) -> int {
        if (!rd.valid || j >= w.NIN) return -1;
        if (dense) return rd.b * Tp * w.NIN + j;           // rd.b = sequence index in dense mode
        return sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
    };
    auto row_md = [&](const RowDesc& rd, int slot, NormMD& md, const NormMD*& md_row) {
        md.m = 0.0f; md.d = 1.0f; md_row = nullptr;
        if (dense || !rd.valid) return;
        if (a.md_row != nullptr) md_row = a.md_row + (size_t)slot * Tp;
        else md = a.md_utt[rd.b];
    };
    // main rows: NTHR/32 feature lanes; thread owns row = tid & 31, features j = (tid >> 5) + (NTHR/32) i
    constexpr int JSTEP = NTHR / 32, NG = (KX + JSTEP - 1) / JSTEP;
    const int grow = tid & 31;
    int goff[NG], xdst[NG];
    NormMD md; const NormMD* md_row;
    {
        const RowDesc rd = rows_s[grow];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int j = (tid >> 5) + JSTEP * i;
            goff[i] = (j < KX && !(BF && j == w.NIN)) ? plan(rd, j) : -2;          // -2: this thread has no element i (BF: the constant-1 column)
            xdst[i] = a_frag_index(grow, j < KX ? j : 0);
        }
        row_md(rd, slot0 + grow, md, md_row);
    }
    int goffx = -1, xdstx = 0;
    NormMD mdx = {0.0f, 1.0f}; const NormMD* mdx_row = nullptr;
    bool havex = EX > 0 && tid < EX * KX;
    if (BF && havex && tid % KX == w.NIN) havex = false;       // the constant-1 column of the extra rows
    if (havex) {
        const int e = tid / KX, j = tid % KX;
        const RowDesc rd = rows_s[32 + e];
        goffx = plan(rd, j);
        xdstx = e_frag_index<EXA>(e, j);
        row_md(rd, slot0 + 32 + e, mdx, mdx_row);
    }

    float* Xf = reinterpret_cast<float*>(Xs);
    float* XEf = reinterpret_cast<float*>(XEs);
    {   // x(0)
        const NormMD m0 = md_row ? md_row[0] : md;
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (goff[i] != -2) Xf[xdst[i]] = goff[i] >= 0 ? (gbase[goff[i]] - m0.m) / m0.d : 0.0f;
        if (havex) {
            const NormMD mx = mdx_row ? mdx_row[0] : mdx;
            XEf[xdstx] = goffx >= 0 ? (gbase[goffx] - mx.m) / mx.d : 0.0f;
        }
    }

    // ---- register state -----------------------------------------------------------------------------
    // biases sit in LDS in (layer, wave, tile, column) order: Bs[((layer*NW + wave)*NT + n)*32 + col]
    const float* __restrict__ bias_l0 = BF ? nullptr : Bs + ((0 * NW + wave) * NT) * 32 + (lane & 31);
    const float* __restrict__ bias_l1 = Bs + (((BL - 1) * NW + wave) * NT) * 32 + (lane & 31);
    f32x2 c0[ST][8], c1[ST][8];          // cell state as register PAIRS (rows r, r + 1): operands of the packed cell update
    float cx0[EXA][ST], cx1[EXA][ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { c0[s][r] = f32x2{0.0f, 0.0f}; c1[s][r] = f32x2{0.0f, 0.0f}; }
#pragma unroll
        for (int e = 0; e < EXA; ++e) { cx0[e][s] = 0.0f; cx1[e][s] = 0.0f; }
    }

    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack) + (size_t)wave * KGT * NT * 256, 0,
                                                KGT * NT * 1024, 0x00020000);
    ws.voff = lane * 16;
    float4 breg[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) breg[n] = wload<NT>(ws, 0, n);   // group 0
    int gnext = 1;
    auto run_groups = [&](f32x16 (&acc_)[NT], float (&accx_)[EXA][NT], const float4* A_, const float4* AE_, int ng) {
        mfma_groups<NT, EX>(acc_, accx_, breg, A_, AE_, ng, ws, gnext, KGT);
    };

    // FC lane mapping (rows 0..31): 8 rows x 2 outputs x 4 k-parts per wave
    const int fc_row = (wave & 3) * 8 + (lane & 7);
    const int fc_o = (lane >> 3) & 1;
    const int fc_kp = lane >> 4;
    const RowDesc fc_rd = rows_s[fc_row];
    // FC lane mapping (extra row e = wave): output = lane >> 5, k = (lane & 31) + 32 i
    const RowDesc fcx_rd = (EX > 0 && wave < EX) ? rows_s[32 + (wave < EX ? wave : 0)] : RowDesc{0, 0, 0, 0};

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
        if (EX > 0 && wave < EX) {
            const int o = lane >> 5;
            const float* he = reinterpret_cast<const float*>(HE1s);
            float sx = 0.0f;
#pragma unroll 4
            for (int i = 0; i < HID / 32; ++i) {
                const int k = (lane & 31) + 32 * i;
                sx += he[e_frag_index<EXA>(wave, k)] * w.wfc[o * HID + k];
            }
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) sx += __shfl_xor(sx, m);
            if ((lane & 31) == 0 && fcx_rd.valid && t_of_h >= a.LA)
                a.out[(size_t)fcx_rd.out_off + (size_t)o * a.out_stride_o + (t_of_h - a.LA)] = apply_act(sx + w.bfc[o], a.act);
        }
    };

    const float4* AEx = XEs + (lane >> 5) * EX;      // E images, offset by this lane's k parity
    const float4* AEh0 = HE0s + (lane >> 5) * EX;
    const float4* AEh1 = HE1s + (lane >> 5) * EX;

    __syncthreads();

    // optional phase profile (debug ABI): workgroup 0, thread 0 stamps s_memtime at 8 points per step
    // (compile-time variant: the stamp branches would otherwise wreck the production kernel's register allocation)
    // Branch-free: thread 0 of workgroup 0 stamps the real slots, every other thread a private dump slot.
    unsigned long long* prof = nullptr;
    int prof_stride = 0;
    if constexpr (PROF) {
        const bool rec = blockIdx.x == 0 && tid == 0;
        prof = a.prof + (rec ? 0 : (size_t)Tp * 8 + (size_t)blockIdx.x * 256 + tid);
        prof_stride = rec ? 8 : 0;
    }
#define FSNP_STAMP(i) do { if constexpr (PROF) prof[t * prof_stride + (prof_stride ? (i) : 0)] = __builtin_amdgcn_s_memtime(); } while (0)

    for (int t = 0; t < Tp; ++t) {
        FSNP_STAMP(0);
        // prefetch x(t+1) (consumed after the layer-0 MFMA phase)
        float xr[NG], xrx = 0.0f;
        NormMD mdn = md, mdxn = mdx;
        const bool have_next = (t + 1 < Tp);
        if (have_next) {
            if (md_row) mdn = md_row[t + 1];
#pragma unroll
            for (int i = 0; i < NG; ++i) xr[i] = goff[i] >= 0 ? gbase[goff[i] + (t + 1) * gstep] : 0.0f;
            if (havex) {
                if (mdx_row) mdxn = mdx_row[t + 1];
                xrx = goffx >= 0 ? gbase[goffx + (t + 1) * gstep] : 0.0f;
            }
        }

        f32x16 acc[NT];
        float accx[EXA][NT];
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = BF ? 0.0f : bias_l0[n * 32];
#pragma unroll
            for (int e = 0; e < EXA; ++e) accx[e][n] = 0.0f;
        }
        run_groups(acc, accx, Xs + lane, AEx, KG0);
        FSNP_STAMP(1);
        __syncthreads();
        FSNP_STAMP(2);
        lstm_cell<ST, UW, HILO ? KSB * 64 * 8 : 0>(acc, c0, reinterpret_cast<float*>(H0s), wave, lane, BF ? reinterpret_cast<unsigned short*>(H0b) : nullptr);
        if (EX > 0) lstm_cell_extra<ST, UW, EX, NT>(accx, cx0, bias_l0, reinterpret_cast<float*>(HE0s), wave, lane);      // (BF: bias_l0 == nullptr)
        if (have_next) {
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if (goff[i] != -2) Xf[xdst[i]] = goff[i] >= 0 ? (xr[i] - mdn.m) / mdn.d : 0.0f;
            if (havex) XEf[xdstx] = goffx >= 0 ? (xrx - mdxn.m) / mdxn.d : 0.0f;
        }
        if (t > 0) fc_store(t - 1);
        FSNP_STAMP(3);
        __syncthreads();
        FSNP_STAMP(4);
        // ---------------- layer 1: [h1_{t-1} | h0_t] ----------------
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = bias_l1[n * 32];
#pragma unroll
            for (int e = 0; e < EXA; ++e) accx[e][n] = 0.0f;
        }
        run_groups(acc, accx, H1s + lane, AEh1, KGH);
        if constexpr (BF) mfma_groups_bf16<NT, EX, HILO ? KSB * 64 : 0>(acc, accx, breg, H0b + lane, HE0s, KSB, ws, gnext, KGT, lane);
        else run_groups(acc, accx, H0s + lane, AEh0, KGH);
        FSNP_STAMP(5);
        __syncthreads();
        FSNP_STAMP(6);
        lstm_cell<ST, UW>(acc, c1, reinterpret_cast<float*>(H1s), wave, lane);
        if (EX > 0) lstm_cell_extra<ST, UW, EX, NT>(accx, cx1, bias_l1, reinterpret_cast<float*>(HE1s), wave, lane);
        FSNP_STAMP(7);
    }
#undef FSNP_STAMP
    __syncthreads();
    fc_store(Tp - 1);
    if (a.clk != nullptr && tid == 0) {
        const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) { a.clk[2] = m1; a.clk[3] = r1; }
        atomicMax(a.clk + 4, r1 - clk_s[1]);
        atomicMax(a.clk + 5, m1 - clk_s[0]);
        atomicMin(a.clk + 6, r1 - clk_s[1]);
    }
}

// -------------------------------------------------------------------------------------------------
size_t lstm_pack_floats(int H, int KX, int NW) {
    const int NT = 4 * (H / NW / 32);
    const int KGT = KX / 8 + H / 8 + 2 * (H / 8);
    return (size_t)NW * KGT * NT * 64 * 4;
}

// [wave][k-group][tile][lane][k-pair]: wave wv owns hidden units [wv*H/NW, (wv+1)*H/NW); tile n = gate*ST + s
// holds columns unit = wv*UW + 32 s + (lane & 31) of gate `gate`; k = 8 g + 2 p + (lane >> 5).
void lstm_pack_weights(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1,
                       const float* whh1, float* wpack) {
    const int UW = H / NW, ST = UW / 32, NT = 4 * ST;
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH;
    for (int wv = 0; wv < NW; ++wv)
        for (int g = 0; g < KGT; ++g)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int p = 0; p < 4; ++p) {
                        const int gate = n / ST, s = n % ST;
                        const int wrow = gate * H + wv * UW + s * 32 + (lane & 31);
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
                        wpack[((((size_t)wv * KGT + g) * NT + n) * 64 + lane) * 4 + p] = v;
                    }
}

// bf16-ih variant of the stream: layer 0 and the h1 part of layer 1 as above (fp32), then HID/16 bf16 k-steps of
// W_ih1: lane l of step ks / tile n holds the 8 weights k = 16 ks + 8 (l>>5) + j of its column, 2 bytes each.
size_t lstm_pack_floats_bf16ih(int H, int KX, int NW) {
    const int NT = 4 * (H / NW / 32);
    const int KGT = KX / 8 + H / 8 + H / 8 + H / 16;
    return (size_t)NW * KGT * NT * 64 * 4;
}

static unsigned short host_bf16(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// bias0 = b_ih0 + b_hh0 [4H]: packed as the weight column of input slot k = NIN, which the kernel feeds with the constant 1 (NIN < KX)
void lstm_pack_weights_bf16ih(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1,
                              const float* whh1, const float* bias0, float* wpack) {
    const int UW = H / NW, ST = UW / 32, NT = 4 * ST;
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KSB = H / 16, KGT = KG0 + KGH + KSB;
    for (int wv = 0; wv < NW; ++wv)
        for (int g = 0; g < KGT; ++g)
            for (int n = 0; n < NT; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int gate = n / ST, s = n % ST;
                    const int wrow = gate * H + wv * UW + s * 32 + (lane & 31);
                    float* dst = wpack + ((((size_t)wv * KGT + g) * NT + n) * 64 + lane) * 4;
                    if (g < KG0 + KGH) {
                        for (int p = 0; p < 4; ++p) {
                            float v = 0.0f;
                            if (g < KG0) {
                                const int k = 8 * g + 2 * p + (lane >> 5);
                                if (k < KX) { if (k < NIN) v = wih0[(size_t)wrow * NIN + k]; else if (k == NIN) v = bias0[wrow]; }
                                else v = whh0[(size_t)wrow * H + (k - KX)];
                            } else {
                                const int k = 8 * (g - KG0) + 2 * p + (lane >> 5);
                                v = whh1[(size_t)wrow * H + k];
                            }
                            dst[p] = v;
                        }
                    } else {
                        const int ks = g - KG0 - KGH;
                        unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
                        for (int j = 0; j < 8; ++j) d16[j] = host_bf16(wih1[(size_t)wrow * H + 16 * ks + 8 * (lane >> 5) + j]);
                    }
                }
}

template <int EX, int NW, bool BF, int KX, int HID = 384>
static void launch_lstm_ex(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    constexpr int OUT = 2;
    constexpr int KGX = KX / 8, KGH = HID / 8, NT = 4 * (HID / NW / 32);
    const size_t smem = (size_t)(KGX + 2 * KGH) * (64 + 2 * EX) * 16 + (size_t)OUT * KGH * 2 * 16 +
                        (32 + EX) * sizeof(RowDesc) + (size_t)(BF ? 1 : 2) * NW * NT * 32 * 4 + (BF ? (size_t)(EX == 0 ? 2 : 1) * (HID / 16) * 64 * 16 : 0);
    LstmWeights wv = w;
    wv.wpack = BF ? (NW == 12 ? w.wpack_bf[1] : w.wpack_bf[0]) : (NW == 12 ? w.wpack12 : w.wpack);
    if constexpr (KX == 40 && HID == 384) {          // the phase-profile variant exists for the default sizes only
        if (a.prof != nullptr) {
            auto kern = lstm2_fc_kernel<HID, KX, OUT, EX, true, NW, BF>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(kern, dim3(a.num_tiles), dim3(64 * NW), smem, s, wv, a);
            return;
        }
    }
    auto kern = lstm2_fc_kernel<HID, KX, OUT, EX, false, NW, BF>;
    static PerDeviceOnce attr_once;            // the attribute is per device: one process may drive several GPUs
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    hipLaunchKernelGGL(kern, dim3(a.num_tiles), dim3(64 * NW), smem, s, wv, a);
}

template <int NW, bool BF, int KX>
static void launch_lstm_nw(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    switch (a.ex) {
        case 0: launch_lstm_ex<0, NW, BF, KX>(w, a, s); break;
        case 1: launch_lstm_ex<1, NW, BF, KX>(w, a, s); break;
        case 2: launch_lstm_ex<2, NW, BF, KX>(w, a, s); break;
        default: launch_lstm_ex<4, NW, BF, KX>(w, a, s); break;
    }
}

void launch_lstm(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (a.num_tiles <= 0) return;
    // measured (profiles/r01_lstm_phase_ab.md): with VALU rows the 12-wave shape is 10 % faster, without them
    // both shapes tie and the 4-wave one needs no spills
    const int waves = w.waves != 0 ? w.waves : (a.ex > 0 ? 12 : 4);
    if (w.H == 256) {                  // sb_model_hidden_size = 256: 4 waves x 64 units, fp32, no VALU rows (planner: ex = 0)
        if (w.KX == 64) launch_lstm_ex<0, 4, false, 64, 256>(w, a, s);
        else launch_lstm_ex<0, 4, false, 40, 256>(w, a, s);
        return;
    }
    if (w.KX == 64) {                  // sub-band inputs of 41..64 features (fb_num_neighbors >= 2, ...): fp32 only
        if (waves == 12) launch_lstm_nw<12, false, 64>(w, a, s);
        else launch_lstm_nw<4, false, 64>(w, a, s);
    } else if (w.ih_bf16) {
        if (waves == 12) launch_lstm_nw<12, true, 40>(w, a, s);
        else launch_lstm_nw<4, true, 40>(w, a, s);
    } else {
        if (waves == 12) launch_lstm_nw<12, false, 40>(w, a, s);
        else launch_lstm_nw<4, false, 40>(w, a, s);
    }
}

// Tile plan: a tile = 32 MFMA rows + up to ex VALU rows.  All tiles cost the same time whatever their
// row count, so the makespan is rounds = ceil(tiles / CUs); pick the smallest ex in {0,1,2,4} that
// minimises rounds, then spread the rows evenly over rounds * CUs tiles.
LstmPlan plan_lstm_tiles(int num_rows, int num_cus) {
    LstmPlan p{};
    if (num_rows <= 32 * num_cus) {
        p.ex = 0; p.num_tiles = cdiv(num_rows, 32);
    } else {
        const int cand[4] = {0, 1, 2, 4};
        int best_rounds = 1 << 30;
        for (int i = 0; i < 4; ++i) {
            const int rounds = cdiv(cdiv(num_rows, 32 + cand[i]), num_cus);
            if (rounds < best_rounds) { best_rounds = rounds; p.ex = cand[i]; }
        }
        p.num_tiles = best_rounds * num_cus;
    }
    p.rows_per_slot_tile = 32 + p.ex;
    return p;
}

}  // namespace fsnp
