// tcn.hip - the three full-band TCN sequence models on gfx950, all branches in each launch.
//
// Replaces SequenceModel.forward's TCN branch (speech_enhance/audio_zen/model/module/sequence_model.py:
// 106-112; stack :48-57) and TCNBlock.forward (speech_enhance/audio_zen/model/module/causal_conv.py:96-108):
//     y = conv1x1(x) -> PReLU -> GroupNorm(1, 512, eps=1e-8) -> depthwise dilated conv (k=3, non-causal)
//       -> PReLU -> GroupNorm -> sconv (1x1) ;  x <- x + y            (x 8, dilations 1,2,5,9,1,2,5,9)
//     fb = act(Linear(ReLU(x)))
//
// Layout: activations are time-major [branch][utt][t][channel], so a 1x1 conv over all frames of all
// utterances is ONE row-major GEMM  C[M = B*T'][N] = A[M][K] * W[N][K]^T  on v_mfma_f32_32x32x2_f32
// (exact fp32).  GroupNorm(1, C) needs statistics over the whole (C x T') plane of one utterance: the
// producing kernel's epilogue accumulates (sum, sum of squares) in fp64 and adds them to a per-utterance
// fp64 slot with one atomic pair per workgroup (row tiles never straddle utterances); the consuming
// kernel applies the per-utterance scalars + per-channel affine while it loads its operand, so no
// normalised tensor is ever written.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "fsnp_common.h"

namespace fsnp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { PRO_NONE = 0, PRO_GN = 1, PRO_RELU = 2 };
enum { EPI_PRELU_STATS = 0, EPI_RESIDUAL = 1, EPI_ACT = 2 };

struct GemmArgs {
    const float* A; long a_bs; int lda;        // A[branch][utt][t][lda]
    long a_us; int a_cols;                     // optional: utterance stride (0 = Tp * lda) and readable floats per row
                                               // (0 = lda); the STFT reads OVERLAPPING frames: lda = hop < K = n_fft
    const float* W; long w_bs; int ldw;        // W[branch][Npad][ldw], zero padded
    const float* bias; long bias_bs;           // [branch][Npad]
    float* C; long c_bs; int ldc;              // C[branch][utt][t][ldc]
    const float* R; long r_bs; int ldr;        // residual (EPI_RESIDUAL)
    const double* gn_in;                       // [branch][utt][kGnStride]: {sum, sum of squares} in a 128-byte line of their own   (PRO_GN)
    const float* gamma; const float* beta; long gb_bs;  // [branch][K] (PRO_GN)
    double* gn_out;                            // [branch][utt][kGnStride]: {sum, sum of squares} in a 128-byte line of their own   (EPI_PRELU_STATS)
    const float* prelu; long prelu_bs;         // [branch] scalar slope (EPI_PRELU_STATS)
    int K, N, Tp, B, act;
    double gn_count;                           // elements per GroupNorm plane (PRO_GN)
    float gn_eps;
    int ntiles_n, row_tiles, row_tiles_all;    // launch geometry: column tiles, row tiles per branch, row tiles of all branches
    const float* c2; long c2_bs;               // tcn_gemm_dma_kernel<EPI_RESIDUAL>: [branch][Npad] sum_k gamma_k W[n][k] (GroupNorm folded)
    int relu_out;                              // EPI_RESIDUAL: 1 = store max(x, 0) (the last block: only ReLU -> Linear reads it, sequence_model.py:109-111)
};

// XCD-aware workgroup order (cdna_hip_programming.md T1).  The dispatcher places workgroup id L on XCD L % 8, and every XCD
// has its own 4 MiB L2.  With a plain (column tile, row tile) grid the 5-8 workgroups that share one 128-row A tile land on
// 5-8 different XCDs and each of them pulls that tile from Infinity Cache / HBM again (B = 32: 100-125 MB per GEMM for a
// 12-25 MB operand).  Here ids are decoded so that ALL column tiles of a row tile sit on ONE XCD, back to back in dispatch
// order: the A tile is fetched once per XCD-local L2 and re-read from it.  `units` row tiles are padded to a multiple of 8;
// returns false for the padding ids.
__device__ __forceinline__ bool xcd_decode(int id, int inner, int units, int& unit, int& in_idx) {
    const int xcd = id & 7, j = id >> 3;
    in_idx = j % inner;
    unit = (j / inner) * 8 + xcd;
    return unit < units;
}
static int xcd_grid(int inner, int units) { return inner * ((units + 7) / 8 * 8); }

constexpr int BM = 128, BK = 16;   // BN (32-column accumulator tiles per wave) is a template parameter

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup = 4 waves, wave w owns rows [32 w, 32 w + 32) of the 128-row tile and all BN columns
// (BN / 32 accumulators of v_mfma_f32_32x32x2_f32).  BN = 96 makes the N = 257 GEMMs 3 column tiles (288, 11 %
// padding) instead of 5 x 64 (25 %).
// PF = register prefetch distance in k-tiles: the global loads of k-tile kt + PF are issued while tile kt is multiplied,
// i.e. they have PF - 1 whole iterations (x 1024 MFMA cycles x the waves sharing the SIMD) to land before they are
// normalised and stored to LDS.  With PF = 1 (round 1) a k-tile of 16 MFMAs per wave had to cover an L2 / Infinity
// Cache round trip by itself and every iteration ended in a vmcnt(0) stall.
template <int PRO, int EPI, int BN, int PF>
__global__ __launch_bounds__(256) void tcn_gemm_kernel(GemmArgs g) {
    constexpr int NTILE = BN / 32;
    constexpr int BLD = (BN * 4 + 255) / 256;   // float4 B loads per thread per k-tile
    constexpr int STAGE = 2 * 2 * BM * 4 + 2 * 2 * BN * 4;      // floats of one (A, B) k-tile stage
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 16];   // double buffered: ONE barrier per k-tile
    double* red = reinterpret_cast<double*>(smem + 2 * STAGE);  // [8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int row_tile_all, ntile;
    if (!xcd_decode(blockIdx.x, g.ntiles_n, g.row_tiles_all, row_tile_all, ntile)) return;
    const int branch = row_tile_all / g.row_tiles, row_tile = row_tile_all % g.row_tiles;
    const int tiles_per_utt = cdiv(g.Tp, BM);
    const int utt = row_tile / tiles_per_utt;
    const int t0 = (row_tile % tiles_per_utt) * BM;
    const int n0 = ntile * BN;

    const float* __restrict__ A = g.A + branch * g.a_bs + (g.a_us ? (long)utt * g.a_us : ((long)utt * g.Tp) * g.lda);
    const int a_cols = g.a_cols ? g.a_cols : g.lda;
    const float* __restrict__ W = g.W + branch * g.w_bs + (long)n0 * g.ldw;

    float mean = 0.f, rstd = 1.f;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    if constexpr (PRO == PRO_GN) {
        const double* st = g.gn_in + ((long)branch * g.B + utt) * kGnStride;
        const double m = st[0] / g.gn_count;
        const double var = st[1] / g.gn_count - m * m;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)g.gn_eps));
        gamma = g.gamma + branch * g.gb_bs;
        beta = g.beta + branch * g.gb_bs;
    }

    // staging assignment: A rows ar0, ar0+64 ; k quad kq ; B columns (tid + 256 i) >> 2
    const int kq = (tid & 3) * 4;
    const int ar0 = tid >> 2;
    float4 areg[PF][2], breg[PF][BLD], ga4[PF], be4[PF];

    // Phase 1: ISSUE the global loads of k-tile k0 (raw values only - nothing here consumes them, so they stay in flight
    // while the MFMAs of the previous tile run).  Phase 2 (finish_tiles, just before the LDS store) applies the operand
    // prologue: GroupNorm + affine / ReLU / zeroing of the K tail.
    auto load_tiles = [&](float4 (&ar)[2], float4 (&br)[BLD], float4& ga, float4& be, int k0) {
        const int k = k0 + kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = t0 + ar0 + 64 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < g.Tp && k < g.K) {
                const float* p = A + (long)t * g.lda + k;
                if (k + 4 <= a_cols) v = *reinterpret_cast<const float4*>(p);
                else { v.x = p[0]; if (k + 1 < a_cols) v.y = p[1]; if (k + 2 < a_cols) v.z = p[2]; }
            }
            ar[i] = v;
        }
        if constexpr (PRO == PRO_GN) {
            if (k < g.K) {
                ga = *reinterpret_cast<const float4*>(gamma + k);   // K % 4 == 0 here
                be = *reinterpret_cast<const float4*>(beta + k);
            }
        }
#pragma unroll
        for (int i = 0; i < BLD; ++i) {
            const int bc = (tid + 256 * i) >> 2;
            br[i] = bc < BN ? *reinterpret_cast<const float4*>(W + (long)bc * g.ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto finish_tiles = [&](float4 (&ar)[2], const float4& ga, const float4& be, int k0) {
        const int k = k0 + kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = t0 + ar0 + 64 * i;
            float4 v = ar[i];
            if (t < g.Tp && k < g.K) {
                if constexpr (PRO == PRO_GN) {
                    v.x = (v.x - mean) * rstd * ga.x + be.x;
                    v.y = (v.y - mean) * rstd * ga.y + be.y;
                    v.z = (v.z - mean) * rstd * ga.z + be.z;
                    v.w = (v.w - mean) * rstd * ga.w + be.w;
                }
                if constexpr (PRO == PRO_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (k + 1 >= g.K) v.y = 0.f;
                if (k + 2 >= g.K) v.z = 0.f;
                if (k + 3 >= g.K) v.w = 0.f;
            }
            ar[i] = v;
        }
    };
    // a float4 of 4 consecutive k = (kh0,p) (kh1,p) (kh0,p+1) (kh1,p+1) with p = (kq>>1)&3  ->  two 8-byte LDS stores
    auto store_frag = [&](float* base, int ld, int row, const float4& v) {
        const int kg = kq >> 3, p = (kq >> 1) & 3;
        *reinterpret_cast<float2*>(base + ((kg * 2 + 0) * ld + row) * 4 + p) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(base + ((kg * 2 + 1) * ld + row) * 4 + p) = make_float2(v.y, v.w);
    };
    auto store_tiles = [&](const float4 (&ar)[2], const float4 (&br)[BLD], int stage) {
        float* As = smem + stage * STAGE;               // [kg 2][kh 2][BM][4]
        float* Bs = As + 2 * 2 * BM * 4;                // [kg 2][kh 2][BN][4]
#pragma unroll
        for (int i = 0; i < 2; ++i) store_frag(As, BM, ar0 + 64 * i, ar[i]);
#pragma unroll
        for (int i = 0; i < BLD; ++i) {
            const int bc = (tid + 256 * i) >> 2;
            if (bc < BN) store_frag(Bs, BN, bc, br[i]);
        }
    };

    f32x16 acc[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int ktiles = g.ldw / BK;
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < ktiles) load_tiles(areg[j], breg[j], ga4[j], be4[j], j * BK);
    finish_tiles(areg[0], ga4[0], be4[0], 0);
    store_tiles(areg[0], breg[0], 0);
    __syncthreads();
    // one k-tile; J = its register slot (compile time: the slot arrays must stay in registers)
    auto k_iteration = [&](auto Jc, int kt) {
        constexpr int J = decltype(Jc)::value;
        constexpr int J1 = (J + 1) % PF;       // NB: PF == 1 -> the slot that was just refilled (round-1 schedule)
        // register slot J held k-tile kt, which went to LDS one iteration ago: refill it with tile kt + PF while tile kt
        // is multiplied out of LDS stage kt & 1.  Tile kt + 1 (slot J1, issued PF - 1 iterations ago) is then normalised
        // and stored to the OTHER stage (last read in iteration kt - 1, before that iteration's barrier).
        if (kt + PF < ktiles) load_tiles(areg[J], breg[J], ga4[J], be4[J], (kt + PF) * BK);
        const float4* As4 = reinterpret_cast<const float4*>(smem + (kt & 1) * STAGE);
        const float4* Bs4 = As4 + 2 * 2 * BM;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            const float4 a4 = As4[(kg * 2 + (lane >> 5)) * BM + wave * 32 + (lane & 31)];
#pragma unroll
            for (int jn = 0; jn < NTILE; ++jn) {
                const float4 b4 = Bs4[(kg * 2 + (lane >> 5)) * BN + jn * 32 + (lane & 31)];
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[jn], 0, 0, 0);
            }
        }
        if (kt + 1 < ktiles) {
            finish_tiles(areg[J1], ga4[J1], be4[J1], (kt + 1) * BK);
            store_tiles(areg[J1], breg[J1], (kt + 1) & 1);
        }
        __syncthreads();
    };
    for (int kt0 = 0; kt0 < ktiles; kt0 += PF) {
        k_iteration(std::integral_constant<int, 0>{}, kt0);
        if constexpr (PF > 1) { if (kt0 + 1 < ktiles) k_iteration(std::integral_constant<int, 1 % PF>{}, kt0 + 1); }
        if constexpr (PF > 2) { if (kt0 + 2 < ktiles) k_iteration(std::integral_constant<int, 2 % PF>{}, kt0 + 2); }
        if constexpr (PF > 3) { if (kt0 + 3 < ktiles) k_iteration(std::integral_constant<int, 3 % PF>{}, kt0 + 3); }
    }

    // ---- epilogue: C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    float* C = g.C + branch * g.c_bs + ((long)utt * g.Tp) * g.ldc;
    double s = 0.0, q = 0.0;
    float slope = 0.f;
    if constexpr (EPI == EPI_PRELU_STATS) slope = g.prelu[branch * g.prelu_bs];
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        const bool col_ok = col < g.N;
        const float bias = g.bias[branch * g.bias_bs + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (t < g.Tp && col_ok) {
                float v = acc[j][r] + bias;
                if constexpr (EPI == EPI_PRELU_STATS) {
                    v = v >= 0.f ? v : slope * v;
                    s += (double)v;
                    q += (double)v * (double)v;
                }
                if constexpr (EPI == EPI_RESIDUAL) {
                    v += g.R[branch * g.r_bs + ((long)utt * g.Tp + t) * g.ldr + col];
                    if (g.relu_out) v = fmaxf(v, 0.f);
                }
                if constexpr (EPI == EPI_ACT) {
                    if (g.act == FSNP_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (g.act == FSNP_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                    else if (g.act == FSNP_ACT_TANH) v = tanhf(v);
                }
                C[(long)t * g.ldc + col] = v;
            }
        }
    }
    if constexpr (EPI == EPI_PRELU_STATS) {
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
        __syncthreads();
        if (tid == 0) {
            double* out = g.gn_out + ((long)branch * g.B + utt) * kGnStride;
            atomicAdd(out, red[0] + red[2] + red[4] + red[6]);
            atomicAdd(out + 1, red[1] + red[3] + red[5] + red[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ---- epilogues shared by tcn_gemm_dma_kernel and tcn_gemm_sk_kernel: per-plane scalars, one output element, the statistics tail
struct EpiCtx { float slope, rstd, mr; };
template <int EPI>
__device__ __forceinline__ EpiCtx epi_ctx(const GemmArgs& g, int branch, int utt) {
    EpiCtx e{0.f, 1.f, 0.f};
    if constexpr (EPI == EPI_PRELU_STATS) e.slope = g.prelu[branch * g.prelu_bs];
    if constexpr (EPI == EPI_RESIDUAL) {
        const double* stt = g.gn_in + ((long)branch * g.B + utt) * kGnStride;
        const double m = stt[0] / g.gn_count;
        const double var = stt[1] / g.gn_count - m * m;
        const double rs = 1.0 / sqrt((var > 0 ? var : 0) + (double)g.gn_eps);
        e.rstd = (float)rs;
        e.mr = (float)(m * rs);
    }
    return e;
}
// one fp64 atomic pair per workgroup into the plane's {sum, sum of squares} slot
__device__ __forceinline__ void epi_stats_finish(const GemmArgs& g, double s, double q2, double* red, int branch, int utt, int tid, int lane, int wave) {
    s = wave_sum(s);
    q2 = wave_sum(q2);
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q2; }
    __syncthreads();
    if (tid == 0) {
        double* out = g.gn_out + ((long)branch * g.B + utt) * kGnStride;
        atomicAdd(out, red[0] + red[2] + red[4] + red[6]);
        atomicAdd(out + 1, red[1] + red[3] + red[5] + red[7]);
    }
}

// ---- float4 epilogue of the DMA GEMM kernels (round 4).  The accumulator layout of v_mfma_f32_32x32x2_f32 gives a lane ONE column
// and 16 scattered rows: stored from there, a tile leaves as 32 (sconv: 32 + 32 residual loads, each load -> add -> store a dependent
// round trip because the residual aliases the output: the in-place x += ...) 4-byte instructions per lane.  Here a wave first
// transposes its R rows x 64 columns through a private LDS slice (row-major, 256 B per row: ds_write_b32 of 32 consecutive columns
// and ds_read_b128 of whole rows are both conflict-free) so that a lane owns float4s along the row: ALL residual loads are issued
// up front as 16-byte loads, waited for ONCE, and the tile leaves in 16-byte stores of whole 256-byte row segments.
// Lane l -> column quad (l & 15), row (l >> 4) + 4 i of the slice.
template <int EPI, int ROWS>      // ROWS = rows of the wave's slice (32: 128-row kernel, 8: split-K kernel)
struct EpiF4 {
    static constexpr int NV = ROWS / 4;
    float4 rv[NV];
    int colv, er;
    bool f4_ok;
    __device__ __forceinline__ void init(const GemmArgs& g, int n0, int lane) {
        er = lane >> 4;
        colv = n0 + (lane & 15) * 4;
        f4_ok = colv < g.ldc;                   // ldc is a float4 multiple: columns [N, ldc) are the zero pad the next GEMM's DMA reads
    }
    // the residual operand of rows [trow0, trow0 + ROWS) - issued long before it is needed (EPI_RESIDUAL only)
    __device__ __forceinline__ void load_residual(const GemmArgs& g, int branch, int utt, int trow0) {
        if constexpr (EPI == EPI_RESIDUAL) {
            const float* Rp = g.R + branch * g.r_bs + ((long)utt * g.Tp) * g.ldr + colv;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int t = trow0 + i * 4 + er;
                rv[i] = (t < g.Tp && f4_ok) ? *reinterpret_cast<const float4*>(Rp + (long)t * g.ldr) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // slice: the wave's [ROWS][64] floats, already holding the finished values EXCEPT the residual / ReLU
    __device__ __forceinline__ void store(const GemmArgs& g, const float* slice, float* C, int trow0, int lane) {
        const float4* s4 = reinterpret_cast<const float4*>(slice);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = trow0 + i * 4 + er;
            float4 v = s4[(i * 4 + er) * 16 + (lane & 15)];
            if constexpr (EPI == EPI_RESIDUAL) {
                v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
                if (g.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (colv + 3 >= g.N) {               // the tile that holds column N - 1: pad columns are written as zeros
                if (colv >= g.N) v.x = 0.f;
                if (colv + 1 >= g.N) v.y = 0.f;
                if (colv + 2 >= g.N) v.z = 0.f;
                v.w = 0.f;
            }
            if (t < g.Tp && f4_ok) *reinterpret_cast<float4*>(C + (long)t * g.ldc + colv) = v;
        }
    }
};
// one accumulator element in the register layout: everything but the residual (added in the float4 phase)
template <int EPI>
__device__ __forceinline__ float epi_stage_value(const EpiCtx& e, float acc, float cb, int act, bool counted, double& s, double& q2) {
    float v;
    if constexpr (EPI == EPI_PRELU_STATS) {
        v = acc + cb;
        v = v >= 0.f ? v : e.slope * v;
        if (counted) { s += (double)v; q2 += (double)v * (double)v; }
    } else if constexpr (EPI == EPI_RESIDUAL) {
        v = e.rstd * acc + cb;                   // cb = c1[n] - m r c2[n]  (GroupNorm folded, see tcn_gemm_dma_kernel)
    } else {
        v = acc + cb;
        if (act == FSNP_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == FSNP_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
        else if (act == FSNP_ACT_TANH) v = tanhf(v);
    }
    return v;
}

// tcn_gemm_dma_kernel: the two GEMMs of a TCNBlock with the k-loop stripped to what the matrix pipe needs.
// tcn_gemm_kernel above issues ~200 VALU/SALU instructions per 16 MFMAs (bounds checks, the GroupNorm prologue, the
// fragment-order shuffle, LDS stores): on this chip they do not overlap a wave's fp32 MFMAs, so it ran at 45 % of the
// matrix pipe (profiles/r02_tcn_gemm.md).  Here
//  * operands go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write, no VALU): three
//    16-byte pieces per lane and k-tile with loop-invariant per-lane offsets and the k position in the scalar offset;
//    rows beyond the plane and k beyond the row are out of the descriptor's range (-> zeros), so there are no bounds checks;
//  * the LDS image is plain row-major [row][16 k], XOR-swizzled on BOTH sides (the DMA destination is lane-linear: lane l of a
//    piece fetches the k-quad that belongs in slot l; readers apply the same involution): ds_read_b128 is conflict-free;
//  * a lane's float4 is 4 consecutive k, i.e. MFMA j multiplies k = k0 + 8 kg + j (lanes 0-31) and + 4 (lanes 32-63) - a
//    permutation of the 16 k of the tile that A and B share, so no fragment shuffle is needed;
//  * GroupNorm on the sconv operand is folded out of the loop: sum_k ((a - m) r g_k + b_k) W[n][k] =
//    r (sum_k a g_k W[n][k]) + c1[n] - r m c2[n] with W pre-scaled by gamma and c1 = bias + sum_k b_k W, c2 = sum_k g_k W
//    packed at fsnp_create (fp64 sums); m, r are per-plane scalars and a row tile never straddles planes.
// Requirements (checked by launch_gemm_dma, else the general kernel runs): lda % 4 == 0, 16-byte aligned planes, K padded
// to 16 with W zero-padded, ldw - lda < 16, pad columns [K, lda) of A finite (zero: this kernel's epilogue writes them).
template <int EPI>
__global__ __launch_bounds__(256) void tcn_gemm_dma_kernel(GemmArgs g) {
    constexpr int BN = 64;
    constexpr int A_SLOTS = BM * 4, B_SLOTS = BN * 4, STAGE = A_SLOTS + B_SLOTS;      // float4 slots per stage (12 KiB)
    constexpr int EPI_SLOTS = BM * BN / 4;                                             // the epilogue's transposition slices: 4 waves x [32][64] floats
    __shared__ __attribute__((aligned(16))) float4 smem[2 * STAGE > EPI_SLOTS ? 2 * STAGE : EPI_SLOTS];      // 32 KiB
    __shared__ double red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int row_tile_all, ntile;
    if (!xcd_decode(blockIdx.x, g.ntiles_n, g.row_tiles_all, row_tile_all, ntile)) return;
    const int branch = row_tile_all / g.row_tiles, row_tile = row_tile_all % g.row_tiles;
    const int tiles_per_utt = cdiv(g.Tp, BM);
    const int utt = row_tile / tiles_per_utt;
    const int t0 = (row_tile % tiles_per_utt) * BM;
    const int n0 = ntile * BN;
    const int rows_valid = min(BM, g.Tp - t0);
    // the plane's scalars (EPI_RESIDUAL: two loads + an fp64 rsqrt / divide chain) and the columns' constants: fetched and computed
    // here, under the first DMA round trips, instead of at the head of the epilogue
    const EpiCtx ec = epi_ctx<EPI>(g, branch, utt);
    float cb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        cb[j] = g.bias[branch * g.bias_bs + col];
        if constexpr (EPI == EPI_RESIDUAL) cb[j] -= ec.mr * g.c2[branch * g.c2_bs + col];
    }

    const float* A = g.A + branch * g.a_bs + ((long)utt * g.Tp + t0) * g.lda;
    const float* W = g.W + branch * g.w_bs + (long)n0 * g.ldw;
    const int a_bytes = rows_valid * g.lda * 4, w_bytes = BN * g.ldw * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, w_bytes, 0x00020000);

    // DMA pieces: wave w moves A slots [128 w, 128 w + 128) (two pieces) and B slots [64 w, 64 w + 64).  Slot s holds row
    // s >> 2, k-quad (s & 3) ^ swz(row); swz(row) = (row >> 2) & 3 spreads 16 consecutive rows of one k-quad over all banks.
    const int ktiles = g.ldw / BK;
    int va[2], va_last[2], vb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = (wave * 2 + i) * 64 + lane, row = s >> 2, kq = (s & 3) ^ ((row >> 2) & 3);
        va[i] = row * g.lda * 4 + kq * 16;                                    // rows >= rows_valid: beyond a_bytes -> zeros
        va_last[i] = ((ktiles - 1) * BK + kq * 4 < g.lda) ? va[i] : a_bytes;   // last k-tile: k-quads beyond the row -> zeros
    }
    {
        const int s = wave * 64 + lane, n = s >> 2, kq = (s & 3) ^ ((n >> 2) & 3);
        vb = n * g.ldw * 4 + kq * 16;
    }
    using lds_ptr = __attribute__((address_space(3))) void*;
    auto issue = [&](int kt, int stage) {
        float4* st = smem + stage * STAGE;
        const bool last = kt == ktiles - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + (wave * 2 + 0) * 64), 16, last ? va_last[0] : va[0], kt * (BK * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + (wave * 2 + 1) * 64), 16, last ? va_last[1] : va[1], kt * (BK * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(st + A_SLOTS + wave * 64), 16, vb, kt * (BK * 4), 0, 0);
    };
    // readers: lane = (r = lane & 31, kh = lane >> 5) takes k-quad 2 kg + kh of row 32 wave + r (A) / column 32 jn + r (B)
    const int r = lane & 31, kh = lane >> 5, sw = (r >> 2) & 3;
    int aoff[2], boff[2][2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        aoff[kg] = (wave * 32 + r) * 4 + ((kg * 2 + kh) ^ sw);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) boff[kg][jn] = A_SLOTS + (jn * 32 + r) * 4 + ((kg * 2 + kh) ^ sw);
    }

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    // one k-tile out of `stage`: ALL fragment reads first, then the DMA of the next tile into the other stage (hipcc waits
    // vmcnt(0) before any ds_read that follows a DMA in program order - issued before the reads, the DMA would be waited for
    // at once), then the 16 MFMAs that cover its flight, then the barrier (which carries the vmcnt(0)).
    auto k_tile = [&](int stage, int kt_next) {
        const float4* st = smem + stage * STAGE;
        float4 a4[2], b4[2][2];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            a4[kg] = st[aoff[kg]];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) b4[kg][jn] = st[boff[kg][jn]];
        }
        if (kt_next < ktiles) issue(kt_next, stage ^ 1);
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].x, b4[kg][jn].x, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].y, b4[kg][jn].y, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].z, b4[kg][jn].z, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].w, b4[kg][jn].w, acc[jn], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);          // else hipcc hoists the vmcnt(0) + s_barrier above 15 of the 16 MFMAs
        __syncthreads();
    };
    issue(0, 0);
    __syncthreads();
    asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));     // keep the accumulators in AGPRs across the loop (see lstm16.hip)
    for (int kt = 0; kt < ktiles; kt += 2) {          // two k-tiles per trip: the stage is a compile-time constant
        k_tile(0, kt + 1);
        if (kt + 1 < ktiles) k_tile(1, kt + 2);
        asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
    }

    // ---- epilogue (EpiF4): residual loads first, the wave's 32 x 64 slice transposed through LDS (the k-loop's last barrier has
    // passed: no wave reads the operand stages any more; a wave touches only its own 8 KiB), 16-byte stores
    float* C = g.C + branch * g.c_bs + ((long)utt * g.Tp) * g.ldc;
    double s = 0.0, q2 = 0.0;
    EpiF4<EPI, 32> ef;
    ef.init(g, n0, lane);
    ef.load_residual(g, branch, utt, t0 + wave * 32);
    float* slice = reinterpret_cast<float*>(smem) + wave * (32 * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool col_ok = n0 + j * 32 + (lane & 31) < g.N;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rl = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const bool counted = col_ok && t0 + wave * 32 + rl < g.Tp;
            slice[rl * 64 + j * 32 + (lane & 31)] = epi_stage_value<EPI>(ec, acc[j][q], cb[j], g.act, counted, s, q2);
        }
    }
    __builtin_amdgcn_wave_barrier();            // (LDS operations of one wave execute in order; this only pins the compiler's order)
    ef.store(g, slice, C, t0 + wave * 32, lane);
    if constexpr (EPI == EPI_PRELU_STATS) epi_stats_finish(g, s, q2, red, branch, utt, tid, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// tcn_gemm_sk_kernel: the DMA GEMM for SMALL problems (round 3) - the reference CLI's B = 1 up to B = 16.  There a launch of tcn_gemm_dma_kernel is 15-24 workgroups whose serial k-loop (17 / 32 k-tiles x 0.43 us of MFMAs)
// IS the launch: conv1x1 17.6 us, sconv 25.6 us at B = 1 with 232 CUs idle (profiles/r03_fullband.md).  Here a workgroup owns a
// 32 x 64 output tile and its four waves split K (wave w multiplies k-tiles w, w + 4, ...): four times the workgroups, a quarter of
// the k-loop each.  A wave stages ITS k-tiles in a private double buffer (same slot layout / swizzle as above with 32 rows), so
// the loop has no workgroup barrier at all - the wave waits for its own DMA only; the four partial tiles are added in a fixed order
// (wave 0 + 1 + 2 + 3) through the staging LDS and wave w finishes rows [8 w, 8 w + 8) with the same epilogues.  K is summed in another
// order than by tcn_gemm_dma_kernel: same tolerance, not bit-identical to it.
constexpr int BMS = 32;
template <int EPI>
__global__ __launch_bounds__(256) void tcn_gemm_sk_kernel(GemmArgs g) {
    constexpr int BN = 64;
    constexpr int A_SLOTS = BMS * 4, B_SLOTS = BN * 4, STAGE = A_SLOTS + B_SLOTS;     // float4 slots per stage of ONE wave (6 KiB)
    __shared__ __attribute__((aligned(16))) float4 smem[4 * 2 * STAGE];                // 48 KiB; re-used for the partial tiles (32 KiB)
    __shared__ double red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int row_tile_all, ntile;
    if (!xcd_decode(blockIdx.x, g.ntiles_n, g.row_tiles_all, row_tile_all, ntile)) return;
    const int branch = row_tile_all / g.row_tiles, row_tile = row_tile_all % g.row_tiles;
    const int tiles_per_utt = cdiv(g.Tp, BMS);
    const int utt = row_tile / tiles_per_utt;
    const int t0 = (row_tile % tiles_per_utt) * BMS;
    const int n0 = ntile * BN;
    const int rows_valid = min(BMS, g.Tp - t0);

    const float* A = g.A + branch * g.a_bs + ((long)utt * g.Tp + t0) * g.lda;
    const float* W = g.W + branch * g.w_bs + (long)n0 * g.ldw;
    const int a_bytes = rows_valid * g.lda * 4, w_bytes = BN * g.ldw * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, w_bytes, 0x00020000);

    // DMA pieces of one k-tile (per wave): A slots [0, 128) = two pieces, B slots [0, 256) = four.  Slot s = row s >> 2, k-quad (s & 3) ^ swz(row)
    const int ktiles = g.ldw / BK;
    int va[2], va_last[2], vb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = i * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        va[i] = row * g.lda * 4 + kq * 16;                                    // rows >= rows_valid: beyond a_bytes -> zeros
        va_last[i] = ((ktiles - 1) * BK + kq * 4 < g.lda) ? va[i] : a_bytes;   // last k-tile: k-quads beyond the row -> zeros
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = i * 64 + lane, n = sl >> 2, kq = (sl & 3) ^ ((n >> 2) & 3);
        vb[i] = n * g.ldw * 4 + kq * 16;
    }
    using lds_ptr = __attribute__((address_space(3))) void*;
    float4* mine = smem + wave * 2 * STAGE;
    auto issue = [&](int kt, int stage) {
        float4* st = mine + stage * STAGE;
        const bool last = kt == ktiles - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + 0), 16, last ? va_last[0] : va[0], kt * (BK * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + 64), 16, last ? va_last[1] : va[1], kt * (BK * 4), 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(st + A_SLOTS + i * 64), 16, vb[i], kt * (BK * 4), 0, 0);
    };
    const int r = lane & 31, kh = lane >> 5, sw = (r >> 2) & 3;
    int aoff[2], boff[2][2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        aoff[kg] = r * 4 + ((kg * 2 + kh) ^ sw);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) boff[kg][jn] = A_SLOTS + (jn * 32 + r) * 4 + ((kg * 2 + kh) ^ sw);
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    // the wave's k-tiles: kt = wave, wave + 4, ... (hipcc waits vmcnt(0) in front of the ds_reads that follow a DMA: exactly the
    // wave's own DMA here - no other wave touches this buffer, so no barrier)
    if (wave < ktiles) issue(wave, 0);
    int stage = 0;
    for (int kt = wave; kt < ktiles; kt += 4) {
        const float4* st = mine + stage * STAGE;
        float4 a4[2], b4[2][2];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            a4[kg] = st[aoff[kg]];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) b4[kg][jn] = st[boff[kg][jn]];
        }
        if (kt + 4 < ktiles) issue(kt + 4, stage ^ 1);
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].x, b4[kg][jn].x, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].y, b4[kg][jn].y, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].z, b4[kg][jn].z, acc[jn], 0, 0, 0);
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].w, b4[kg][jn].w, acc[jn], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        stage ^= 1;
    }
    // ---- add the four partial tiles: part[wave][j][q][lane]; wave w then owns accumulator registers q in [4 w, 4 w + 4) = rows 8 w ... 8 w + 7
    // (the epilogue's scalars, column constants and residual rows are fetched first: their round trips pass under the reduction)
    const EpiCtx ec = epi_ctx<EPI>(g, branch, utt);
    float cb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        cb[j] = g.bias[branch * g.bias_bs + col];
        if constexpr (EPI == EPI_RESIDUAL) cb[j] -= ec.mr * g.c2[branch * g.c2_bs + col];
    }
    EpiF4<EPI, 8> ef;
    ef.init(g, n0, lane);
    ef.load_residual(g, branch, utt, t0 + wave * 8);
    __syncthreads();                                  // every wave is done with its staging buffers
    float* part = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) part[((wave * 2 + j) * 16 + q) * 64 + lane] = acc[j][q];
    __syncthreads();
    float sum[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = wave * 4 + qq;
            float v = part[((0 * 2 + j) * 16 + q) * 64 + lane];
            v += part[((1 * 2 + j) * 16 + q) * 64 + lane];
            v += part[((2 * 2 + j) * 16 + q) * 64 + lane];
            v += part[((3 * 2 + j) * 16 + q) * 64 + lane];
            sum[j][qq] = v;
        }

    // ---- epilogue (EpiF4, as tcn_gemm_dma_kernel): the wave's 8 x 64 slice transposed through LDS behind the partial tiles (32 KiB
    // of the 48; that area was wave 2's staging buffer - idle since the barriers above), 16-byte residual loads and stores
    float* C = g.C + branch * g.c_bs + ((long)utt * g.Tp) * g.ldc;
    double s = 0.0, q2 = 0.0;
    float* slice = reinterpret_cast<float*>(smem) + 4 * 2 * 16 * 64 + wave * (8 * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool col_ok = n0 + j * 32 + (lane & 31) < g.N;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int rl = qq + 4 * (lane >> 5);
            const bool counted = col_ok && t0 + wave * 8 + rl < g.Tp;
            slice[rl * 64 + j * 32 + (lane & 31)] = epi_stage_value<EPI>(ec, sum[j][qq], cb[j], g.act, counted, s, q2);
        }
    }
    __builtin_amdgcn_wave_barrier();
    ef.store(g, slice, C, t0 + wave * 8, lane);
    if constexpr (EPI == EPI_PRELU_STATS) epi_stats_finish(g, s, q2, red, branch, utt, tid, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// tcn_gemm_dma64_kernel (round 4): the sconv GEMM [M][512] x [257][512]^T on 64 x 64 tiles.  N = 257 on 64-wide column tiles is FIVE
// tiles, the last one all padding but one column: at B = 32 the 128-row kernel launches 96 x 5 = 480 workgroups on 256 CUs - two
// rounds of a 13.6 us k-loop for 1.9 rounds of work, 20 % of it multiplying zeros.  Here
//  * the lone column N - 1 leaves the matrix pipe: the workgroups of the LAST full column tile also form its dot products on the
//    VALU out of the A tile they hold in LDS anyway (8 FMAs and two ds_read_b128 pairs per thread and k-tile; the column's weights
//    are staged in LDS once) - FOUR column tiles;
//  * tiles are 64 rows: 192 x 4 = 768 workgroups = exactly 3 per CU at B = 32 (three rounds of a 6.8 us k-loop: 20.4 us of MFMA time
//    instead of 27.2); the four waves sit 2 x 2 on the tile (one 32 x 32 accumulator each);
//  * a k-tile is 32 deep so that a wave still issues 16 MFMAs per barrier (as the 128-row kernel); the LDS image is [row][8 k-quads],
//    XOR-swizzled by (row >> 1) & 7 on both sides (16 consecutive rows of one k-quad cover all 64 banks); 4 DMA pieces per wave and k-tile.
// Same operands, GroupNorm fold and float4 epilogue as tcn_gemm_dma_kernel<EPI_RESIDUAL>; the k-sum of a column runs over the same k
// in another association, column N - 1 is summed by fp32 FMAs: equal to the 128-row kernel within rounding, not bit for bit.
// Requirements (launch_gemm_dma64): N % 64 == 1, K % 32 == 0, lda == ldw == K (no k tail), float4-aligned C / R rows.
constexpr int BM64 = 64, BK64 = 32;
template <int EPI>
__global__ __launch_bounds__(256) void tcn_gemm_dma64_kernel(GemmArgs g) {
    constexpr int BN = 64;
    constexpr int A_SLOTS = BM64 * 8, B_SLOTS = BN * 8, STAGE = A_SLOTS + B_SLOTS;     // float4 slots per stage (16 KiB)
    __shared__ __attribute__((aligned(16))) float4 smem[2 * STAGE];                     // 32 KiB; the epilogue's 64 x 64 tile re-uses 16
    __shared__ __attribute__((aligned(16))) float wx[1024];                             // weights of column N - 1 (K <= 1024)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    int row_tile_all, ntile;
    if (!xcd_decode(blockIdx.x, g.ntiles_n, g.row_tiles_all, row_tile_all, ntile)) return;
    const int branch = row_tile_all / g.row_tiles, row_tile = row_tile_all % g.row_tiles;
    const int tiles_per_utt = cdiv(g.Tp, BM64);
    const int utt = row_tile / tiles_per_utt;
    const int t0 = (row_tile % tiles_per_utt) * BM64;
    const int n0 = ntile * BN;
    const int rows_valid = min(BM64, g.Tp - t0);
    const bool xcol = ntile == g.ntiles_n - 1;          // this workgroup also owns column N - 1 = ntiles_n * 64

    const EpiCtx ec = epi_ctx<EPI>(g, branch, utt);
    const int mycol = n0 + wc * 32 + (lane & 31);
    float cb = g.bias[branch * g.bias_bs + mycol];
    if constexpr (EPI == EPI_RESIDUAL) cb -= ec.mr * g.c2[branch * g.c2_bs + mycol];

    const float* A = g.A + branch * g.a_bs + ((long)utt * g.Tp + t0) * g.lda;
    const float* W = g.W + branch * g.w_bs + (long)n0 * g.ldw;
    const int a_bytes = rows_valid * g.lda * 4, w_bytes = BN * g.ldw * 4;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, w_bytes, 0x00020000);

    // DMA pieces: wave w moves A slots [128 w, 128 w + 128) and B slots [128 w, 128 w + 128).  Slot s holds row s >> 3, k-quad (s & 7) ^ swz(row)
    const int ktiles = g.ldw / BK64;
    int va[2], vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = (wave * 2 + i) * 64 + lane, row = sl >> 3, kq = (sl & 7) ^ ((row >> 1) & 7);
        va[i] = row * g.lda * 4 + kq * 16;               // rows >= rows_valid: beyond a_bytes -> zeros
        vb[i] = row * g.ldw * 4 + kq * 16;
    }
    using lds_ptr = __attribute__((address_space(3))) void*;
    auto issue = [&](int kt, int stage) {
        float4* st = smem + stage * STAGE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + (wave * 2 + 0) * 64), 16, va[0], kt * (BK64 * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + (wave * 2 + 1) * 64), 16, va[1], kt * (BK64 * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(st + A_SLOTS + (wave * 2 + 0) * 64), 16, vb[0], kt * (BK64 * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(st + A_SLOTS + (wave * 2 + 1) * 64), 16, vb[1], kt * (BK64 * 4), 0, 0);
    };
    // readers: lane = (r = lane & 31, kh = lane >> 5) takes k-quad 2 kg + kh of row 32 wr + r (A) / column 32 wc + r (B)
    const int r = lane & 31, kh = lane >> 5;
    int aoff[4], boff[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
        const int ar = wr * 32 + r, br = wc * 32 + r;
        aoff[kg] = ar * 8 + ((kg * 2 + kh) ^ ((ar >> 1) & 7));
        boff[kg] = A_SLOTS + br * 8 + ((kg * 2 + kh) ^ ((br >> 1) & 7));
    }
    // column N - 1: thread -> A slots tid and tid + 256 of a stage = rows tid >> 3 and 32 + (tid >> 3), the k-quad that sits in position tid & 7
    const int xrow = tid >> 3, xq0 = (tid & 7) ^ ((xrow >> 1) & 7), xq1 = (tid & 7) ^ (((xrow + 32) >> 1) & 7);
    float xacc0 = 0.f, xacc1 = 0.f;
    if (xcol) {
        const float* wl = g.W + branch * g.w_bs + (long)(g.ntiles_n * BN) * g.ldw;
        for (int k = tid; k < g.K; k += 256) wx[k] = wl[k];
    }

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

    auto k_tile = [&](int stage, int kt, int kt_next) {
        const float4* st = smem + stage * STAGE;
        float4 a4[4], b4[4];
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) { a4[kg] = st[aoff[kg]]; b4[kg] = st[boff[kg]]; }
        float4 xa0, xa1, xw0, xw1;
        if (xcol) {
            xa0 = st[tid]; xa1 = st[tid + 256];
            xw0 = *reinterpret_cast<const float4*>(wx + kt * BK64 + xq0 * 4);
            xw1 = *reinterpret_cast<const float4*>(wx + kt * BK64 + xq1 * 4);
        }
        if (kt_next < ktiles) issue(kt_next, stage ^ 1);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].x, b4[kg].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].y, b4[kg].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].z, b4[kg].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kg].w, b4[kg].w, acc, 0, 0, 0);
        }
        if (xcol) {
            xacc0 = fmaf(xa0.x, xw0.x, xacc0); xacc0 = fmaf(xa0.y, xw0.y, xacc0); xacc0 = fmaf(xa0.z, xw0.z, xacc0); xacc0 = fmaf(xa0.w, xw0.w, xacc0);
            xacc1 = fmaf(xa1.x, xw1.x, xacc1); xacc1 = fmaf(xa1.y, xw1.y, xacc1); xacc1 = fmaf(xa1.z, xw1.z, xacc1); xacc1 = fmaf(xa1.w, xw1.w, xacc1);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    issue(0, 0);
    __syncthreads();                                   // (also: wx is staged)
    asm volatile("" : "+a"(acc));
    for (int kt = 0; kt < ktiles; kt += 2) {
        k_tile(0, kt, kt + 1);
        if (kt + 1 < ktiles) k_tile(1, kt + 1, kt + 2);
        asm volatile("" : "+a"(acc));
    }

    // ---- epilogue: the 64 x 64 tile through LDS (all four waves write, then wave w owns rows 16 w ... 16 w + 15 as float4s)
    float* C = g.C + branch * g.c_bs + ((long)utt * g.Tp) * g.ldc;
    double s = 0.0, q2 = 0.0;
    EpiF4<EPI, 16> ef;
    ef.init(g, n0, lane);
    ef.load_residual(g, branch, utt, t0 + wave * 16);
    // column N - 1: the 8 threads that share a row add their partial dot products (lanes 8 i ... 8 i + 7 of a wave)
    float xr0 = 0.f, xr1 = 0.f;
    const int xcolumn = g.ntiles_n * BN;
    if (xcol) {
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { xacc0 += __shfl_xor(xacc0, m); xacc1 += __shfl_xor(xacc1, m); }
        if constexpr (EPI == EPI_RESIDUAL) {
            if ((tid & 7) == 0) {
                const float* Rp = g.R + branch * g.r_bs + ((long)utt * g.Tp) * g.ldr + xcolumn;
                if (t0 + xrow < g.Tp) xr0 = Rp[(long)(t0 + xrow) * g.ldr];
                if (t0 + xrow + 32 < g.Tp) xr1 = Rp[(long)(t0 + xrow + 32) * g.ldr];
            }
        }
    }
    float* tile = reinterpret_cast<float*>(smem);
    {
        const bool col_ok = mycol < g.N;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rl = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const bool counted = col_ok && t0 + wr * 32 + rl < g.Tp;
            tile[(wr * 32 + rl) * 64 + wc * 32 + (lane & 31)] = epi_stage_value<EPI>(ec, acc[q], cb, g.act, counted, s, q2);
        }
    }
    __syncthreads();
    ef.store(g, tile + wave * 16 * 64, C, t0 + wave * 16, lane);
    if (xcol && (tid & 7) == 0) {
        float cx = g.bias[branch * g.bias_bs + xcolumn];
        if constexpr (EPI == EPI_RESIDUAL) cx -= ec.mr * g.c2[branch * g.c2_bs + xcolumn];
        double sd = 0.0, qd = 0.0;
        float v0 = epi_stage_value<EPI>(ec, xacc0, cx, g.act, false, sd, qd) + xr0;
        float v1 = epi_stage_value<EPI>(ec, xacc1, cx, g.act, false, sd, qd) + xr1;
        if constexpr (EPI == EPI_RESIDUAL) { if (g.relu_out) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); } }
        // the float4 at column N - 1 = {value, 0, 0, 0}: columns [N, ldc) are the zero pad the next GEMM's DMA reads
        if (t0 + xrow < g.Tp) *reinterpret_cast<float4*>(C + (long)(t0 + xrow) * g.ldc + xcolumn) = make_float4(v0, 0.f, 0.f, 0.f);
        if (t0 + xrow + 32 < g.Tp) *reinterpret_cast<float4*>(C + (long)(t0 + xrow + 32) * g.ldc + xcolumn) = make_float4(v1, 0.f, 0.f, 0.f);
    }
}

// launched (true) when the 64-row kernel applies AND beats the 128-row kernel's round count
template <int EPI>
static bool launch_gemm_dma64(const GemmArgs& g, int n, int num_cus, hipStream_t s, int branches) {
    if constexpr (EPI != EPI_RESIDUAL) return false;      // (sconv only: conv1x1 and the final Linear have K = 257 -> a 16-deep k tail)
    else {
    if (n % 64 != 1 || g.K % BK64 || g.lda != g.K || g.ldw != g.K || g.K > 1024 || g.a_us || g.a_cols) return false;
    if (g.ldc % 4 || g.ldc < n + 3 || g.ldr % 4) return false;            // the float4 {column N - 1, three pad columns}
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W) | reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.R)) & 15) return false;
    if ((g.a_bs * 4) % 16 || (g.w_bs * 4) % 16 || (g.c_bs * 4) % 16 || (g.r_bs * 4) % 16) return false;
    if ((long)BM64 * g.lda * 4 >= (1L << 31) || (long)64 * g.ldw * 4 >= (1L << 31)) return false;
    GemmArgs ga = g;
    ga.ntiles_n = n / 64;
    ga.row_tiles = cdiv(g.Tp, BM64) * g.B; ga.row_tiles_all = ga.row_tiles * branches;
    // rounds of k-loop time, in units of a 64-row tile: the 128-row kernel's workgroups cost two
    const long wg64 = (long)ga.ntiles_n * ga.row_tiles_all, wg128 = (long)cdiv(n, 64) * cdiv(g.Tp, BM) * g.B * branches;
    const long cost64 = (wg64 + num_cus - 1) / num_cus, cost128 = 2 * ((wg128 + num_cus - 1) / num_cus);
    if (cost64 > cost128) return false;
    hipLaunchKernelGGL((tcn_gemm_dma64_kernel<EPI>), dim3(xcd_grid(ga.ntiles_n, ga.row_tiles_all)), dim3(256), 0, s, ga);
    return true;
    }
}

// true (and launched) when the DMA kernel's requirements hold
template <int EPI>
static bool launch_gemm_dma(const GemmArgs& g, int n, int row_tiles, int num_cus, hipStream_t s, int branches, bool allow_splitk, bool allow_bm64 = false) {
    if (g.a_us || g.a_cols || g.lda % 4 || g.ldw % BK || g.ldw < g.K || g.lda < g.K) return false;
    if (g.ldw - g.lda >= BK) return false;                 // only the LAST k-tile may reach beyond a row of A
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15 || (g.a_bs * 4) % 16 || (g.w_bs * 4) % 16) return false;
    if ((long)BM * g.lda * 4 >= (1L << 31) || (long)64 * g.ldw * 4 >= (1L << 31)) return false;
    // float4 epilogue: 16-byte rows of C (and of the residual)
    if (g.ldc % 4 || (reinterpret_cast<uintptr_t>(g.C) & 15) || (g.c_bs * 4) % 16) return false;
    if (EPI == EPI_RESIDUAL && (g.ldr % 4 || (reinterpret_cast<uintptr_t>(g.R) & 15) || (g.r_bs * 4) % 16)) return false;
    GemmArgs ga = g;
    ga.ntiles_n = cdiv(n, 64); ga.row_tiles = row_tiles; ga.row_tiles_all = row_tiles * branches;
    // small problems: 32-row tiles whose four waves split K (tcn_gemm_sk_kernel) while that launch has at most 6 workgroups per CU
    // (B <= 16 at 2 s clips; measured B = 1 ... 32, profiles/r03_fullband.md: full-band stage B = 1 0.47 -> 0.27 ms, B = 4 0.53 -> 0.32,
    // B = 8 0.56 -> 0.41, B = 16 0.66 -> 0.65, B = 32 0.92 -> 1.05: not there)
    constexpr int sk = 6;
    const int row_tiles32 = cdiv(g.Tp, BMS) * g.B;
    if (allow_splitk && sk && (long)ga.ntiles_n * row_tiles32 * branches <= (long)sk * num_cus && row_tiles == cdiv(g.Tp, BM) * g.B) {
        ga.row_tiles = row_tiles32; ga.row_tiles_all = row_tiles32 * branches;
        hipLaunchKernelGGL((tcn_gemm_sk_kernel<EPI>), dim3(xcd_grid(ga.ntiles_n, ga.row_tiles_all)), dim3(256), 0, s, ga);
        return true;
    }
    if (allow_bm64 && launch_gemm_dma64<EPI>(g, n, num_cus, s, branches)) return true;
    hipLaunchKernelGGL((tcn_gemm_dma_kernel<EPI>), dim3(xcd_grid(ga.ntiles_n, ga.row_tiles_all)), dim3(256), 0, s, ga);
    return true;
}

// Column-tile width for one GEMM launch.  Workgroups are MFMA-bound, so the launch takes about
// ceil(blocks / CUs) * BN; pick the BN in {64, 96, 128} that minimises it (ties -> the narrower tile: more,
// smaller workgroups balance better).  Weights are zero-padded to a multiple of 384 rows so any choice is valid.
static int pick_bn(int n, int row_tiles, int num_cus, int branches) {
    int best = 64;
    long best_cost = -1;
    const int cand[3] = {64, 96, 128};
    for (int c : cand) {
        const long blocks = (long)cdiv(n, c) * row_tiles * branches;
        const long cost = ((blocks + num_cus - 1) / num_cus) * c;
        if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
    return best;
}

template <int PRO, int EPI, int PF>
static void launch_gemm_pf(const GemmArgs& g, int bn, const dim3& grid, hipStream_t s) {
    if (bn == 128) hipLaunchKernelGGL((tcn_gemm_kernel<PRO, EPI, 128, PF>), grid, dim3(256), 0, s, g);
    else if (bn == 96) hipLaunchKernelGGL((tcn_gemm_kernel<PRO, EPI, 96, PF>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((tcn_gemm_kernel<PRO, EPI, 64, PF>), grid, dim3(256), 0, s, g);
}

template <int PRO, int EPI>
static void launch_gemm(const GemmArgs& g, int n, int row_tiles, int num_cus, hipStream_t s, int branches = 3) {
    // measured (profiles/r02_tcn_gemm.md): the prefetch distance makes no difference (1.99 / 1.90 / 1.93 / 1.93 ms full-band
    // stage for PF = 1..4): the kernel was never latency-bound.  2 is kept.
    const int bn = pick_bn(n, row_tiles, num_cus, branches);
    GemmArgs ga = g;
    ga.ntiles_n = cdiv(n, bn); ga.row_tiles = row_tiles; ga.row_tiles_all = row_tiles * branches;
    const dim3 grid(xcd_grid(ga.ntiles_n, ga.row_tiles_all));
    launch_gemm_pf<PRO, EPI, 2>(ga, bn, grid, s);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm1 -> depthwise dilated conv (k=3, zero padding = dilation) -> PReLU2 (+ GroupNorm2 statistics)
struct DwArgs {
    const float* Y1; float* Y2; long y_bs;      // [branch][utt][t][CH]
    const double* gn_in; double* gn_out;        // [branch][utt][kGnStride]: {sum, sum of squares} in a 128-byte line of their own
    const float* gamma; const float* beta;      // [branch][CH]
    const float* w; const float* b;             // [branch][3][CH], [branch][CH]
    const float* prelu;                         // [branch]
    long cb_bs, w_bs, prelu_bs;
    int CH, Tp, B, dil;
    double gn_count; float gn_eps;
    int chunks, planes;                          // DW_ROWS-row chunks per (utterance, branch) plane; planes = B * branches
};
constexpr int DW_ROWS = 8;

__global__ __launch_bounds__(256) void tcn_dwconv_kernel(DwArgs g) {
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // all chunks of one (utterance, branch) plane on one XCD: the +-dilation halo rows are then L2 hits (xcd_decode)
    int plane, chunk;
    if (!xcd_decode(blockIdx.x, g.chunks, g.planes, plane, chunk)) return;
    const int branch = plane / g.B, utt = plane % g.B, t0 = chunk * DW_ROWS;
    const double* st = g.gn_in + ((long)branch * g.B + utt) * kGnStride;
    const double m = st[0] / g.gn_count;
    const double var = st[1] / g.gn_count - m * m;
    const float mean = (float)m;
    const float rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)g.gn_eps));
    const float slope = g.prelu[branch * g.prelu_bs];
    const float* __restrict__ Y1 = g.Y1 + branch * g.y_bs + (long)utt * g.Tp * g.CH;
    float* __restrict__ Y2 = g.Y2 + branch * g.y_bs + (long)utt * g.Tp * g.CH;
    const int quads = g.CH / 4;
    double s = 0.0, q = 0.0;
    for (int item = tid; item < DW_ROWS * quads; item += 256) {
        const int c = (item % quads) * 4;
        const int t = t0 + item / quads;
        if (t >= g.Tp) continue;
        const float4 ga = *reinterpret_cast<const float4*>(g.gamma + branch * g.cb_bs + c);
        const float4 be = *reinterpret_cast<const float4*>(g.beta + branch * g.cb_bs + c);
        float4 acc = *reinterpret_cast<const float4*>(g.b + branch * g.cb_bs + c);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tt = t + (j - 1) * g.dil;
            if (tt < 0 || tt >= g.Tp) continue;     // zero padding of the NORMALISED tensor
            const float4 y = *reinterpret_cast<const float4*>(Y1 + (long)tt * g.CH + c);
            const float4 wj = *reinterpret_cast<const float4*>(g.w + branch * g.w_bs + (long)j * g.CH + c);
            acc.x += wj.x * ((y.x - mean) * rstd * ga.x + be.x);
            acc.y += wj.y * ((y.y - mean) * rstd * ga.y + be.y);
            acc.z += wj.z * ((y.z - mean) * rstd * ga.z + be.z);
            acc.w += wj.w * ((y.w - mean) * rstd * ga.w + be.w);
        }
        acc.x = acc.x >= 0.f ? acc.x : slope * acc.x;
        acc.y = acc.y >= 0.f ? acc.y : slope * acc.y;
        acc.z = acc.z >= 0.f ? acc.z : slope * acc.z;
        acc.w = acc.w >= 0.f ? acc.w : slope * acc.w;
        *reinterpret_cast<float4*>(Y2 + (long)t * g.CH + c) = acc;
        s += (double)acc.x + (double)acc.y + (double)acc.z + (double)acc.w;
        q += (double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z + (double)acc.w * acc.w;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        double* out = g.gn_out + ((long)branch * g.B + utt) * kGnStride;
        atomicAdd(out, red[0] + red[2] + red[4] + red[6]);
        atomicAdd(out + 1, red[1] + red[3] + red[5] + red[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// `branches` sequence models run side by side (grid.z): 3 for the full-band mag / real / imag models, 1 when the same
// stack is the SUB-BAND model (sequence_model="TCN", sequence_model.py:47-58: d.B = sub-band sequences, d.F = 34 channels).
void launch_tcn(const Dims& d, int fb_act, const TcnWeights& w, const TcnBuffers& buf, hipStream_t s, int branches) {
    const long x_bs = (long)d.B * d.Tp * d.FP;
    const long y_bs = (long)d.B * d.Tp * d.CH;
    const int row_tiles = cdiv(d.Tp, BM) * d.B;
    const double gn_count = (double)d.CH * d.Tp;
    auto gn_slot = [&](int blk, int which) { return buf.gn + ((long)(blk * 2 + which) * branches) * d.B * kGnStride; };
    // DMA GEMMs: the full-band stacks only (their input `att` has zero pad columns; the sub-band TCN's buffers make no such promise)
    const bool dma = branches == 3 && w.gemm_dma;
    const bool relu_fused = dma && w.NB > 0 && !(w.NB == 1 && buf.dbg_tcn0);     // the last sconv stores max(x, 0) for the final Linear

    for (int blk = 0; blk < w.NB; ++blk) {
        const float* xin = blk == 0 ? buf.att : buf.x;
        {   // conv1x1 + PReLU1 (+ GN1 stats): y1[M][CH] = x[M][F] * W1^T
            GemmArgs g{};
            g.A = xin; g.a_bs = x_bs; g.lda = d.FP;
            g.W = w.w1 + (long)blk * w.N1P * w.K1P; g.w_bs = (long)w.NB * w.N1P * w.K1P; g.ldw = w.K1P;
            g.bias = w.b1 + (long)blk * w.N1P; g.bias_bs = (long)w.NB * w.N1P;
            g.C = buf.y1; g.c_bs = y_bs; g.ldc = d.CH;
            g.gn_out = gn_slot(blk, 0);
            g.prelu = w.a1 + blk; g.prelu_bs = w.NB;
            g.K = d.F; g.N = d.CH; g.Tp = d.Tp; g.B = d.B;
            if (!(dma && launch_gemm_dma<EPI_PRELU_STATS>(g, d.CH, row_tiles, w.num_cus, s, branches, w.gemm_dma == 1)))
                launch_gemm<PRO_NONE, EPI_PRELU_STATS>(g, d.CH, row_tiles, w.num_cus, s, branches);
        }
        {   // GN1 -> depthwise -> PReLU2 (+ GN2 stats)
            DwArgs g{};
            g.Y1 = buf.y1; g.Y2 = buf.y2; g.y_bs = y_bs;
            g.gn_in = gn_slot(blk, 0); g.gn_out = gn_slot(blk, 1);
            g.gamma = w.g1w + (long)blk * d.CH; g.beta = w.g1b + (long)blk * d.CH;
            g.b = w.db + (long)blk * d.CH; g.cb_bs = (long)w.NB * d.CH;
            g.w = w.dw + (long)blk * 3 * d.CH; g.w_bs = (long)w.NB * 3 * d.CH;
            g.prelu = w.a2 + blk; g.prelu_bs = w.NB;
            g.CH = d.CH; g.Tp = d.Tp; g.B = d.B; g.dil = w.dilation[blk];
            g.gn_count = gn_count; g.gn_eps = 1e-8f;
            g.chunks = cdiv(d.Tp, DW_ROWS); g.planes = d.B * branches;
            hipLaunchKernelGGL(tcn_dwconv_kernel, dim3(xcd_grid(g.chunks, g.planes)), dim3(256), 0, s, g);
        }
        {   // GN2 (on load) -> sconv + residual: x[M][F] = xin + GN2(y2)[M][CH] * W2^T
            GemmArgs g{};
            g.A = buf.y2; g.a_bs = y_bs; g.lda = d.CH;
            g.W = w.w2 + (long)blk * w.N2P * w.K2P; g.w_bs = (long)w.NB * w.N2P * w.K2P; g.ldw = w.K2P;
            g.bias = w.b2 + (long)blk * w.N2P; g.bias_bs = (long)w.NB * w.N2P;
            g.C = buf.x; g.c_bs = x_bs; g.ldc = d.FP;
            g.R = xin; g.r_bs = x_bs; g.ldr = d.FP;
            g.gn_in = gn_slot(blk, 1);
            g.gamma = w.g2w + (long)blk * d.CH; g.beta = w.g2b + (long)blk * d.CH; g.gb_bs = (long)w.NB * d.CH;
            g.K = d.CH; g.N = d.F; g.Tp = d.Tp; g.B = d.B;
            g.gn_count = gn_count; g.gn_eps = 1e-8f;
            // the last block's output is only read through ReLU (-> Linear): store it ReLU'd, so that the Linear's operand needs no
            // prologue and can go global -> LDS by DMA like the others (final Linear 39 -> 29.5 us at B = 32)
            g.relu_out = (relu_fused && blk == w.NB - 1) ? 1 : 0;
            GemmArgs gf = g;                      // GroupNorm folded into the weights (tcn_gemm_dma_kernel)
            gf.W = w.w2g + (long)blk * w.N2P * w.K2P;
            gf.bias = w.c1 + (long)blk * w.N2P;
            gf.c2 = w.c2 + (long)blk * w.N2P; gf.c2_bs = (long)w.NB * w.N2P;
            if (!(dma && w.w2g && launch_gemm_dma<EPI_RESIDUAL>(gf, d.F, row_tiles, w.num_cus, s, branches, w.gemm_dma == 1, w.gemm_dma != 3)))
                launch_gemm<PRO_GN, EPI_RESIDUAL>(g, d.F, row_tiles, w.num_cus, s, branches);
        }
        if (blk == 0 && buf.dbg_tcn0)
            (void)hipMemcpyAsync(buf.dbg_tcn0, buf.x, (size_t)x_bs * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    {   // ReLU (on load) -> Linear(F, F) -> activation
        GemmArgs g{};
        g.A = w.NB > 0 ? buf.x : buf.att; g.a_bs = x_bs; g.lda = d.FP;
        g.W = w.wf; g.w_bs = (long)w.N2P * w.K1P; g.ldw = w.K1P;
        g.bias = w.bf; g.bias_bs = w.N2P;
        g.C = buf.fb; g.c_bs = x_bs; g.ldc = d.FP;
        g.K = d.F; g.N = d.F; g.Tp = d.Tp; g.B = d.B; g.act = fb_act;
        // (PRO_RELU of the general kernel is idempotent on an operand that was stored ReLU'd)
        if (!(relu_fused && launch_gemm_dma<EPI_ACT>(g, d.F, row_tiles, w.num_cus, s, branches, w.gemm_dma == 1)))
            launch_gemm<PRO_RELU, EPI_ACT>(g, d.F, row_tiles, w.num_cus, s, branches);
    }
}

// C[utt][t][0..N) = act(A[utt][t][0..K) * W^T + bias): the Linear(512, 257) + ReLU after the full-band LSTM of the
// original FullSubNet (SequenceModel.forward, sequence_model.py:119-122).  W is [N pad 384][ldw], zero padded.
void launch_linear_act(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int K,
                       int N, int B, int Tp, int act, int num_cus, hipStream_t s, long a_utt_stride, int a_cols) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.a_us = a_utt_stride; g.a_cols = a_cols;
    g.K = K; g.N = N; g.Tp = Tp; g.B = B; g.act = act;
    launch_gemm<PRO_NONE, EPI_ACT>(g, N, cdiv(Tp, BM) * B, num_cus, s, 1);
}

}  // namespace fsnp
