// tcn_fused.hip - the TCNBlock stack of the full-band sequence models in ONE launch (or one launch per block), for the batches where a
// launch is latency, not throughput (the reference CLI's B = 1 and the small serving batches).
//
// Replaces, per block, the three launches of tcn.hip (TCNBlock.forward, speech_enhance/audio_zen/model/module/causal_conv.py:96-108):
//     y1 = PReLU(conv1x1(x))            GroupNorm(1, CH) statistics of y1 over the whole (CH x T') plane
//     y2 = PReLU(dwconv_d(GN1(y1)))     statistics of y2
//     x  = x + sconv(GN2(y2))           (GroupNorm-2 folded into the sconv weights, as tcn_gemm_dma_kernel does)
// At B = 1 each of those launches is at the floor of a dependent launch (4.7 us) plus a 2 - 3 us k-loop: 24 launches = 190 of the
// full-band stage's 245 us (profiles/r05_fullband.md).  Here a block is two plane-wide hand-offs, and the next block follows behind a
// third one in the same launch.
//
// Decomposition (per (branch, utterance) plane and 128-row time chunk): G = max(CH / 16, 4 * ceil(F / 32)) workgroups, ONE PER CU (the
// waves own the CU's whole register file: every operand of a phase is in flight at once - a phase costs one memory round trip, not one
// per pipeline refill; a first version with a 4 / 8 k-tile pipeline and four workgroups per CU took 22.8 us per block, as long as the
// three launches it replaced).
//   phase 1  workgroup cs < CH / 16 owns ALL rows of the chunk x 16 channels of y1: `v_mfma_f32_16x16x4_f32`, operands L2 -> registers
//            (x rows: A, W1 rows: B; a lane's float4 = 4 consecutive k, both sides alike, so MFMA j of a k-tile multiplies k = 16 i + 4 kq + j),
//            wave w rows [32 w, 32 w + 32).  The tile stays in LDS: the depthwise conv runs along time, i.e. inside the workgroup - y1 never
//            leaves the CU.  The sconv weights of phase 3 (the wave's 16 columns x K = CH: 128 registers) are requested right behind.
//   hand-off 1: (sum, sum of squares) of the owned rows as fp64 atomics into the plane's slot, arrival counter in the same 128-byte line;
//            every phase-1 workgroup of the plane waits for all of them (bounded, launch-wide abort: lstm_common.h).
//   phase 2  GN1 -> 3-tap dilated depthwise conv (zero padding of the NORMALISED tensor) -> PReLU: from LDS, float4 along channels;
//            y2 leaves as write-through 16-byte stores ([t][CH] rows of the plane: the sconv operand).
//   hand-off 2: statistics of y2 + arrival; ALL G workgroups wait.
//   phase 3  workgroup cs -> 32 rows x 32 columns of the sconv output; wave -> one 16 x 16 tile over the whole K = CH (two accumulators by
//            k-tile parity: a dependent 16x16x4 chain costs 15 %), y2 by sc1 loads (other XCDs wrote it), epilogue from the registers:
//            rstd acc + c1[n] - mean rstd c2[n] + residual, pad columns [F, FP) written as zeros (the next conv1x1 reads them as K padding).
//            The output tile IS the next block's residual tile: it stays in registers.
//   hand-off 3 (between blocks of one launch): x left as write-through stores; the phase-1 workgroups wait for the plane's phase-3 tiles.
// Clips longer than 128 frames: chunks of 110 owned rows + 9 halo rows either side (>= the largest dilation); halo rows of y1 are
// recomputed, counted once.  Hand-offs are per PLANE and workgroup ids are plane-major, so a launch needs only one plane's workgroups
// (G x chunks) co-resident, whatever the batch.
// Summation order differs from tcn.hip's kernels (K in one pass per lane group): same tolerance against the oracle, not bit-identical
// to them; the statistics are fp64 atomics as there.
#include <algorithm>
#include <cstdint>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FT_ROWS = 128;     // rows of y1 a workgroup computes (owned + halo)
constexpr int FT_HALO = 9;       // >= the largest dilation (sequence_model.py:48-57: 1, 2, 5, 9)
constexpr int FT_LDY = 20;       // LDS row stride of the y1 tile (16 channels + 4: the accumulator writes of the four lane groups hit 4 x 16 different banks)
constexpr unsigned FT_OOB = 0x7fff0000u;   // buffer offset beyond every descriptor range here: the load returns zeros

struct FusedArgs {
    const float* xin; float* xout; long x_bs; int FP;        // [branch][utt][Tp][FP]: input of block b0 (att for block 0), running activation
    float* y2; long y_bs;                                     // [branch][utt][Tp][CH]
    // weights of ALL blocks, [branch][NB][...] (TcnWeights): block blk of branch br at br * bs + blk * per-block size
    const float* w1; long w1_bs; int K1P, N1P;
    const float* b1; long b1_bs;
    const float* a1; const float* a2; long a_bs;              // PReLU slopes [branch][NB]
    const float* g1w; const float* g1b; const float* db; long cb_bs;
    const float* dw; long dw_bs;                              // [branch][NB][3][CH]
    const float* w2g; long w2_bs; int K2P, N2P;
    const float* c1; const float* c2; long c_bs;
    double* gn; long gn_blk;                                  // [NB][2][branches][B][kGnStride]: {sum, sumsq, arrival counter at [2], hand-off 3 counter at [3]}
    unsigned* abort_dev; unsigned* err_host;
    unsigned long long* prof;                                 // PROF: 8 s_memtime stamps of workgroup 0, block b0 + 1 (or b0)
    int F, CH, Tp, B, nchunks, CR, G, relu_last, b0, b1_, NB, branches;
    int dil[16];
    double gn_count; float gn_eps;
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int AUX>
__device__ __forceinline__ float4 ld128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
// Thread 0 of a workgroup: add the workgroup's (s, q) to the plane's slot and arrive (contribute), wait for `target` arrivals, read the
// totals.  false = the launch was aborted (a peer timed out).
__device__ __forceinline__ bool plane_handoff(double* slot, bool contribute, double s, double q, unsigned target, const FusedArgs& g, float* bc) {
    unsigned* ctr = reinterpret_cast<unsigned*>(slot + 2);
    if (contribute) {
        __hip_atomic_fetch_add(slot, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(slot + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // both atomics (and this thread's write-through stores) are done
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!xchg_wait(ctr, target, g.abort_dev, g.err_host)) return false;
    const double sum = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double sq = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m = sum / g.gn_count;
    const double var = sq / g.gn_count - m * m;
    const double rs = 1.0 / sqrt((var > 0 ? var : 0) + (double)g.gn_eps);
    bc[0] = (float)m; bc[1] = (float)rs; bc[2] = (float)(m * rs);
    return true;
}

template <int AUX>
__device__ __forceinline__ void st32(__amdgpu_buffer_rsrc_t r, float v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, AUX);
}

template <int KT1, int KT2, bool PROF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tcn_block_fused_kernel(FusedArgs g) {
    __shared__ __attribute__((aligned(16))) float y1s[FT_ROWS * FT_LDY];
    __shared__ double red[8];
    __shared__ float bc[4];
    __shared__ int ok_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_plane = g.G * g.nchunks;
    const int plane = blockIdx.x / per_plane, rem = blockIdx.x % per_plane;
    const int chunk = rem / g.G, cs = rem % g.G;
    const int branch = plane / g.B, utt = plane % g.B;
    const int own0 = chunk * g.CR, own1 = min(g.Tp, own0 + g.CR);
    const int row_lo = own0 - (g.nchunks > 1 ? FT_HALO : 0);
    const int r16 = lane & 15, kq = lane >> 4;
    const bool p1 = cs < g.CH / 16;
    const int nct = (g.F + 31) / 32;
    const bool p3 = cs < 4 * nct;
    const unsigned target = (unsigned)(g.CH / 16) * g.nchunks, target3 = (unsigned)(4 * nct) * g.nchunks;
    const bool prof_wg = PROF && blockIdx.x == 0 && tid == 0;
    unsigned long long stamp[8] = {};
    if (tid == 0) ok_s = 1;

    float* Xo = g.xout + branch * g.x_bs + (long)utt * g.Tp * g.FP;
    float* Y2 = g.y2 + branch * g.y_bs + (long)utt * g.Tp * g.CH;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y2, 0, g.Tp * g.CH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxo = __builtin_amdgcn_make_buffer_rsrc(Xo, 0, g.Tp * g.FP * 4, 0x00020000);

    // ---- phase 3 geometry; the residual rows of the first block (later blocks: the tile this lane wrote)
    const int rs3 = cs / nct, ct3 = cs % nct;
    const int col3 = 32 * ct3 + 16 * (wave & 1) + r16;                    // this lane's output column / W2g row
    const int row3 = own0 + 32 * rs3 + 16 * (wave >> 1);                  // first row of the wave's 16 x 16 tile
    float resid[4] = {0.f, 0.f, 0.f, 0.f};
    if (p3) {
        const float* X0 = g.xin + branch * g.x_bs + (long)utt * g.Tp * g.FP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = row3 + 4 * kq + j;
            if (t < own1 && col3 < g.F) resid[j] = X0[(long)t * g.FP + col3];
        }
    }
    // ---- operand addressing (the same in every block)
    int va1[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int grow = row_lo + 32 * wave + 16 * sub + r16;
        va1[sub] = (grow >= 0 && grow < g.Tp) ? (grow * g.FP + 4 * kq) * 4 : (int)FT_OOB;
    }
    const int ch1 = 16 * cs + r16;
    const int vb1 = (ch1 * g.K1P + 4 * kq) * 4;
    const int arow3 = row3 + r16;
    const int va3 = arow3 < own1 ? (arow3 * g.CH + 4 * kq) * 4 : (int)FT_OOB;
    const int vb3 = (col3 * g.K2P + 4 * kq) * 4;
    const int cq2 = tid & 3;

    float4 wb[KT1];                    // W1 operand of the coming phase 1 (requested one block ahead)
    auto load_w1 = [&](int blk) {
        const float* W1 = g.w1 + branch * g.w1_bs + (long)blk * g.N1P * g.K1P;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W1), 0, g.N1P * g.K1P * 4, 0x00020000);
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) wb[kt] = ld128<0>(rw, vb1, kt * 64);
    };
    if (p1) load_w1(g.b0);

    for (int blk = g.b0; blk < g.b1_; ++blk) {
        const bool first = blk == g.b0, last = blk == g.b1_ - 1;
        const bool prof_blk = prof_wg && blk == min(g.b0 + 1, g.b1_ - 1);
        double* slot1 = g.gn + (long)(blk * 2 + 0) * g.gn_blk + ((long)branch * g.B + utt) * kGnStride;
        double* slot2 = g.gn + (long)(blk * 2 + 1) * g.gn_blk + ((long)branch * g.B + utt) * kGnStride;
        if (prof_blk) stamp[0] = __builtin_amdgcn_s_memtime();
        // sconv operand of THIS block's phase 3 and its column constants: requested now, used after two hand-offs
        float4 w2[KT2];
        float c1v = 0.f, c2v = 0.f;
        auto load_w2 = [&]() {
            const float* W2 = g.w2g + branch * g.w2_bs + (long)blk * g.N2P * g.K2P;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W2), 0, g.N2P * g.K2P * 4, 0x00020000);
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) w2[kt] = ld128<0>(rw, vb3, kt * 64);
            c1v = g.c1[branch * g.c_bs + (long)blk * g.N2P + col3];
            c2v = g.c2[branch * g.c_bs + (long)blk * g.N2P + col3];
        };

        if (p1) {
            // ================= phase 1: y1[128 rows][16 channels] = PReLU(x W1^T + b1); every operand in flight at once
            const float* X = (first ? g.xin : g.xout) + branch * g.x_bs + (long)utt * g.Tp * g.FP;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, g.Tp * g.FP * 4, 0x00020000);
            float4 xa0[KT1], xa1[KT1];
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt) {       // (sc1: later blocks read what other XCDs wrote in this launch)
                xa0[kt] = ld128<kSc1>(rx, va1[0], kt * 64);
                xa1[kt] = ld128<kSc1>(rx, va1[1], kt * 64);
            }
            if (p3) load_w2();
            // parameters of phase 2
            const long cbase = branch * g.cb_bs + (long)blk * g.CH + 16 * cs + 4 * cq2;
            const float4 ga = *reinterpret_cast<const float4*>(g.g1w + cbase);
            const float4 be = *reinterpret_cast<const float4*>(g.g1b + cbase);
            const float4 dbv = *reinterpret_cast<const float4*>(g.db + cbase);
            float4 wt[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) wt[j] = *reinterpret_cast<const float4*>(g.dw + branch * g.dw_bs + ((long)blk * 3 + j) * g.CH + 16 * cs + 4 * cq2);
            const float slope1 = g.a1[branch * g.a_bs + blk], slope2 = g.a2[branch * g.a_bs + blk];
            const float bias1 = g.b1[branch * g.b1_bs + (long)blk * g.N1P + ch1];
            const int dil = g.dil[blk];

            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT1; ++kt) {
                const float4 x0 = xa0[kt], x1 = xa1[kt], w = wb[kt];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, w.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, w.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, w.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, w.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w.w, acc1, 0, 0, 0);
            }
            // accumulator register j of lane (channel r16, group kq) = row 4 kq + j of the 16-row sub-tile
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lrow = 32 * wave + 16 * sub + 4 * kq + j;
                    float v = (sub ? acc1[j] : acc0[j]) + bias1;
                    v = v >= 0.f ? v : slope1 * v;
                    y1s[lrow * FT_LDY + r16] = v;
                    const int grow = row_lo + lrow;
                    if (grow >= own0 && grow < own1) { s += (double)v; q += (double)v * (double)v; }
                }
            }
            s = wave_sum_d(s); q = wave_sum_d(q);
            if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
            __syncthreads();
            if (prof_blk) stamp[1] = __builtin_amdgcn_s_memtime();
            if (tid == 0 && !plane_handoff(slot1, true, red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], target, g, bc)) ok_s = 0;
            __syncthreads();
            if (!ok_s) return;
            if (prof_blk) stamp[2] = __builtin_amdgcn_s_memtime();
            const float mean1 = bc[0], rstd1 = bc[1];

            // ================= phase 2: GN1 -> depthwise conv -> PReLU2, 16-byte items (row, channel quad)
            double s2 = 0.0, q2 = 0.0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lt = (tid >> 2) + 64 * i, gt = row_lo + lt;
                if (gt < own0 || gt >= own1) continue;
                float4 acc = dbv;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int tt = gt + (j - 1) * dil;
                    if (tt < 0 || tt >= g.Tp) continue;               // zero padding of the NORMALISED tensor
                    const float4 y = *reinterpret_cast<const float4*>(y1s + (tt - row_lo) * FT_LDY + 4 * cq2);
                    acc.x += wt[j].x * ((y.x - mean1) * rstd1 * ga.x + be.x);
                    acc.y += wt[j].y * ((y.y - mean1) * rstd1 * ga.y + be.y);
                    acc.z += wt[j].z * ((y.z - mean1) * rstd1 * ga.z + be.z);
                    acc.w += wt[j].w * ((y.w - mean1) * rstd1 * ga.w + be.w);
                }
                acc.x = acc.x >= 0.f ? acc.x : slope2 * acc.x;
                acc.y = acc.y >= 0.f ? acc.y : slope2 * acc.y;
                acc.z = acc.z >= 0.f ? acc.z : slope2 * acc.z;
                acc.w = acc.w >= 0.f ? acc.w : slope2 * acc.w;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc), ry,
                                                       (gt * g.CH + 16 * cs + 4 * cq2) * 4, 0, kSc1);
                s2 += (double)acc.x + (double)acc.y + (double)acc.z + (double)acc.w;
                q2 += (double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z + (double)acc.w * acc.w;
            }
            s2 = wave_sum_d(s2); q2 = wave_sum_d(q2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
            if (lane == 0) { red[wave * 2] = s2; red[wave * 2 + 1] = q2; }
            __syncthreads();
            if (prof_blk) stamp[3] = __builtin_amdgcn_s_memtime();
        } else if (p3) {
            load_w2();
        }
        // ---- hand-off 2: every workgroup of the plane waits for y2 and its statistics; the phase-1 workgroups contribute
        if (tid == 0 && !plane_handoff(slot2, p1, p1 ? red[0] + red[2] + red[4] + red[6] : 0.0, p1 ? red[1] + red[3] + red[5] + red[7] : 0.0, target, g, bc)) ok_s = 0;
        __syncthreads();
        if (!ok_s) return;
        if (prof_blk) stamp[4] = __builtin_amdgcn_s_memtime();
        const float rstd2 = bc[1], mr2 = bc[2];

        // ================= phase 3: x[32 rows][32 columns] += sconv(GN2(y2)); this wave: one 16 x 16 tile, K = CH
        if (p3) {
            float4 ya[KT2];
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) ya[kt] = ld128<kSc1>(ry, va3, kt * 64);
            if (p1 && !last) load_w1(blk + 1);                        // the next block's conv1x1 operand, behind them
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT2; kt += 2) {
                const float4 x0 = ya[kt], w0 = w2[kt], x1 = ya[kt + 1], w1 = w2[kt + 1];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, w0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, w0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, w0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, w0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w1.w, acc1, 0, 0, 0);
            }
            if (prof_blk) stamp[5] = __builtin_amdgcn_s_memtime();
            const float cb = c1v - mr2 * c2v;
            const bool relu = g.relu_last && blk == g.NB - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = row3 + 4 * kq + j;
                if (t >= own1 || col3 >= g.FP) continue;
                float v = rstd2 * (acc0[j] + acc1[j]) + cb + resid[j];
                if (relu) v = fmaxf(v, 0.f);
                v = col3 < g.F ? v : 0.f;                                 // pad columns: zeros
                resid[j] = v;
                st32<kSc1>(rxo, v, (t * g.FP + col3) * 4);
            }
        }
        if (last) break;
        // ---- hand-off 3: the plane's x tiles of this block are out (write-through); phase 1 of the next block reads all of them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (prof_blk) stamp[6] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned* ctr3 = reinterpret_cast<unsigned*>(slot2 + 3);
            if (p3) __hip_atomic_fetch_add(ctr3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p1 && !xchg_wait(ctr3, target3, g.abort_dev, g.err_host)) ok_s = 0;
        }
        __syncthreads();
        if (!ok_s) return;
        if (prof_blk) stamp[7] = __builtin_amdgcn_s_memtime();
    }
    if (PROF && prof_wg && g.prof) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g.prof[i] = stamp[i];
    }
}

}  // namespace

// true = the whole stack of blocks can run on the fused kernel (launch_tcn falls back to the three-launch path otherwise)
constexpr int FT_KT1 = 17, FT_KT2 = 32;         // the one instantiation: num_freqs 257 (K1P = 272), 512 channels
bool tcn_fused_available(const Dims& d, const TcnWeights& w) {
    if (!w.w2g || w.NB <= 0 || w.NB > 16) return false;
    if (w.K1P != FT_KT1 * 16 || w.K2P != FT_KT2 * 16) return false;
    if (d.CH % 16 || w.K2P < d.CH || w.K1P < d.F || d.FP % 4 || d.FP < d.F) return false;
    if (w.K1P - d.FP >= 16) return false;                                   // only the last k-tile reaches beyond a row of x
    if (32 * ((d.F + 31) / 32) > w.N2P || d.CH > w.N1P) return false;
    for (int i = 0; i < w.NB; ++i) if (w.dilation[i] < 1 || w.dilation[i] > FT_HALO) return false;
    if ((long)d.Tp * d.CH * 4 >= (long)FT_OOB || (long)d.Tp * d.FP * 4 >= (long)FT_OOB) return false;
    if ((long)w.N1P * w.K1P * 4 >= (long)FT_OOB || (long)w.N2P * w.K2P * 4 >= (long)FT_OOB) return false;
    return true;
}
int tcn_fused_chunks(const Dims& d) { return d.Tp <= FT_ROWS ? 1 : cdiv(d.Tp, FT_ROWS - 2 * FT_HALO); }
int tcn_fused_workgroups_per_plane(const Dims& d) { return std::max(d.CH / 16, 4 * ((d.F + 31) / 32)) * tcn_fused_chunks(d); }

// blocks [b0, b1) of the three full-band stacks in one launch; relu_last: block NB - 1 stores max(x, 0) (tcn.hip: relu_fused)
void launch_tcn_fused(const Dims& d, const TcnWeights& w, const TcnBuffers& buf, int b0, int b1, bool relu_last, hipStream_t s) {
    FusedArgs g{};
    constexpr int branches = 3;
    g.xin = b0 == 0 ? buf.att : buf.x; g.xout = buf.x; g.x_bs = (long)d.B * d.Tp * d.FP; g.FP = d.FP;
    g.y2 = buf.y2; g.y_bs = (long)d.B * d.Tp * d.CH;
    g.w1 = w.w1; g.w1_bs = (long)w.NB * w.N1P * w.K1P; g.K1P = w.K1P; g.N1P = w.N1P;
    g.b1 = w.b1; g.b1_bs = (long)w.NB * w.N1P;
    g.a1 = w.a1; g.a2 = w.a2; g.a_bs = w.NB;
    g.g1w = w.g1w; g.g1b = w.g1b; g.db = w.db; g.cb_bs = (long)w.NB * d.CH;
    g.dw = w.dw; g.dw_bs = (long)w.NB * 3 * d.CH;
    g.w2g = w.w2g; g.w2_bs = (long)w.NB * w.N2P * w.K2P; g.K2P = w.K2P; g.N2P = w.N2P;
    g.c1 = w.c1; g.c2 = w.c2; g.c_bs = (long)w.NB * w.N2P;
    g.gn = buf.gn; g.gn_blk = (long)branches * d.B * kGnStride;
    g.abort_dev = buf.fused_abort; g.err_host = buf.fused_err; g.prof = buf.fused_prof;
    g.F = d.F; g.CH = d.CH; g.Tp = d.Tp; g.B = d.B;
    g.nchunks = tcn_fused_chunks(d);
    g.CR = g.nchunks == 1 ? FT_ROWS : FT_ROWS - 2 * FT_HALO;
    g.G = std::max(d.CH / 16, 4 * ((d.F + 31) / 32));
    g.relu_last = relu_last ? 1 : 0; g.b0 = b0; g.b1_ = b1; g.NB = w.NB; g.branches = branches;
    for (int i = 0; i < w.NB; ++i) g.dil[i] = w.dilation[i];
    g.gn_count = (double)d.CH * d.Tp; g.gn_eps = 1e-8f;
    const dim3 grid(g.G * g.nchunks * branches * d.B);
    if (buf.fused_prof) hipLaunchKernelGGL((tcn_block_fused_kernel<FT_KT1, FT_KT2, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((tcn_block_fused_kernel<FT_KT1, FT_KT2, false>), grid, dim3(256), 0, s, g);
}

}  // namespace fsnp
