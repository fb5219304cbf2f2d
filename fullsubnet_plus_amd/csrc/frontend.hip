// frontend.hip - input normalisation + TSSE channel attention for the three full-band branches.
//
// Replaces, for each of (mag, real, imag):
//   functional.pad(look_ahead)                       fullsubnet_plus/model/fullsubnet_plus.py:137-139
//   self.norm(x)                                     fullsubnet_plus.py:144,157,162  (base_model.py:210-330)
//   ChannelTimeSenseSELayer.forward                  audio_zen/model/module/attention_model.py:78-98
// (paths relative to /root/reference/speech_enhance).
//
// Pipeline (all tiny, HBM/latency bound):
//   repack : strided [B,1,F,T] view (torch.stft layout or any other) -> raw[branch][utt][t][FP], zero padded; on the way it
//            accumulates, per branch and utterance, the column sums over t and the plane's (sum, sum of squares) in fp64
//   offline norms (m, d constant per utterance; round 5: THREE launches instead of six):
//     gate   : (m, d) from the plane's totals, written for every frame; S_f = (column sum - T' m) / d; TSSE squeeze + MLP.
//              conv -> AdaptiveAvgPool is linear, so mean_t(conv_K(x))[f] = sum_j w[f,j] * (S_f - prefix_j - suffix_{K-1-j}) / (T'-K+1) + b[f]
//     apply  : att = normalised * gate[f]
//   cumulative norms (m_t, d_t per frame):
//     frame  : per-frame (sum, sumsq) over F in fp64;  scan : prefix over frames -> (m_t, d_t);  fsum : per-frequency sum over t of
//              the normalised input, fp64 atomics;  gate, apply as above
#include "fsnp_common.h"

namespace fsnp {

struct StridedIn {
    const float* p[3];
    long sb[3], sf[3], st[3];
};

// Statistics of one 32 x 32 tile (rows = frames, columns = bins; zeros outside the clip / the bins): column sums -> colsum[f],
// the tile's (sum, sum of squares) -> tot[2], all fp64 atomics.  tile(r, c) = value of frame t0 + r, bin f0 + c.
template <typename TileFn>
__device__ __forceinline__ void repack_tile_stats(TileFn tile, double* __restrict__ colsum, double* __restrict__ tot, int f0, int F,
                                                  double (*red)[33], double* wred) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const double v = tile(ty + 8 * i, tx); s += v; q += v * v; }
    red[ty][tx] = s;
    double ts = s, tq = q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ts += __shfl_xor(ts, o); tq += __shfl_xor(tq, o); }
    if ((threadIdx.x & 63) == 0) { wred[(threadIdx.x >> 6) * 2] = ts; wred[(threadIdx.x >> 6) * 2 + 1] = tq; }
    __syncthreads();
    if (ty == 0 && f0 + tx < F) {
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += red[k][tx];
        atomicAdd(colsum + f0 + tx, c);
    }
    if (threadIdx.x == 0) {
        atomicAdd(tot, wred[0] + wred[2] + wred[4] + wred[6]);
        atomicAdd(tot + 1, wred[1] + wred[3] + wred[5] + wred[7]);
    }
}

__global__ __launch_bounds__(256) void fe_repack_kernel(StridedIn in, float* __restrict__ raw, double* __restrict__ colsum,
                                                        double* __restrict__ tot, int B, int T, int Tp, int F, int FP) {
    __shared__ float tile[32][33];
    __shared__ double red[8][33];
    __shared__ double wred[8];
    const int branch = blockIdx.z / B, b = blockIdx.z % B;
    const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ src = in.p[branch] + (long)b * in.sb[branch];
    const long sF = in.sf[branch], sT = in.st[branch];
    const bool f_fast = sF <= sT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        if (f_fast) {
            const int f = f0 + tx, t = t0 + r;
            tile[r][tx] = (f < F && t < T) ? src[f * sF + t * sT] : 0.0f;
        } else {
            const int t = t0 + tx, f = f0 + r;
            tile[tx][r] = (f < F && t < T) ? src[f * sF + t * sT] : 0.0f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        const int t = t0 + r, f = f0 + tx;
        if (t < Tp && f < FP) raw[(((long)branch * B + b) * Tp + t) * FP + f] = tile[r][tx];
    }
    if (colsum != nullptr) {
        const long ub = (long)branch * B + b;
        repack_tile_stats([&](int r, int c) -> double { return (double)tile[r][c]; }, colsum + ub * FP, tot + ub * 2, f0, F, red, wred);
    }
}

// the same repack straight from the interleaved complex64 STFT buffer (torch.stft's output; strides
// in complex elements): real and imag are the two halves of each element, mag = |X| is derived here instead of by a
// torch op (audio_zen/acoustics/feature.py:24-31 `mag_phase`, inferencer.py:143-147).  nbr = 3: [mag, real, imag]
// planes; nbr = 1: magnitude only (original FullSubNet).
__global__ __launch_bounds__(256) void fe_repack_complex_kernel(const float2* __restrict__ x, long sb, long sf, long st,
                                                                float* __restrict__ raw, double* __restrict__ colsum,
                                                                double* __restrict__ tot, int nbr, int B, int T, int Tp,
                                                                int F, int FP) {
    __shared__ float2 tile[32][33];
    __shared__ double red[8][33];
    __shared__ double wred[8];
    const int b = blockIdx.z;
    const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float2* __restrict__ src = x + (long)b * sb;
    const bool f_fast = sf <= st;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        if (f_fast) {
            const int f = f0 + tx, t = t0 + r;
            tile[r][tx] = (f < F && t < T) ? src[f * sf + t * st] : make_float2(0.f, 0.f);
        } else {
            const int t = t0 + tx, f = f0 + r;
            tile[tx][r] = (f < F && t < T) ? src[f * sf + t * st] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    const long plane = (long)B * Tp * FP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        const int t = t0 + r, f = f0 + tx;
        if (t < Tp && f < FP) {
            const float2 v = tile[r][tx];
            const long o = ((long)b * Tp + t) * FP + f;
            raw[o] = hypotf(v.x, v.y);
            if (nbr == 3) { raw[plane + o] = v.x; raw[2 * plane + o] = v.y; }
        }
    }
    if (colsum != nullptr) {           // [mag | real | imag] planes: branch-major [branch][utt]
        repack_tile_stats([&](int r, int c) -> double { const float2 v = tile[r][c]; return (double)hypotf(v.x, v.y); },
                          colsum + (long)b * FP, tot + (long)b * 2, f0, F, red, wred);
        if (nbr == 3) {
            __syncthreads();
            repack_tile_stats([&](int r, int c) -> double { return (double)tile[r][c].x; }, colsum + ((long)B + b) * FP, tot + ((long)B + b) * 2, f0, F, red, wred);
            __syncthreads();
            repack_tile_stats([&](int r, int c) -> double { return (double)tile[r][c].y; }, colsum + (2L * B + b) * FP, tot + (2L * B + b) * 2, f0, F, red, wred);
        }
    }
}

// one wave per (branch, utt, t)
__global__ __launch_bounds__(64) void fe_frame_kernel(const float* __restrict__ raw, double* __restrict__ frame, int B,
                                                      int Tp, int F, int FP) {
    const long row = ((long)blockIdx.z * B + blockIdx.y) * Tp + blockIdx.x;
    const float* __restrict__ p = raw + row * FP;
    double s = 0.0, q = 0.0;
    for (int f = threadIdx.x; f < F; f += 64) {
        const double v = p[f];
        s += v;
        q += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (threadIdx.x == 0) { frame[row * 2] = s; frame[row * 2 + 1] = q; }
}

// (norm_md - running (sum, sumsq, count) -> (m, d) for the four norm types - lives in fsnp_common.h: stages.hip uses it too)

// one workgroup per (branch, utt): chunked prefix scan over frames
__global__ __launch_bounds__(256) void fe_scan_kernel(const double* __restrict__ frame, NormMD* __restrict__ md,
                                                      int B, int Tp, int F, int norm_type) {
    __shared__ double cs[256], cq[256];
    const long base = ((long)blockIdx.y * B + blockIdx.x) * Tp;
    const int tid = threadIdx.x;
    const int chunk = cdiv(Tp, 256);
    const int lo = tid * chunk, hi = min(lo + chunk, Tp);
    double s = 0.0, q = 0.0;
    for (int t = lo; t < hi; ++t) { s += frame[(base + t) * 2]; q += frame[(base + t) * 2 + 1]; }
    cs[tid] = s; cq[tid] = q;
    __syncthreads();
    const bool cumulative = norm_type == FSNP_NORM_CUMULATIVE_LAPLACE || norm_type == FSNP_NORM_CUMULATIVE_LAYER;
    if (!cumulative) {
        double ts = 0.0, tq = 0.0;
        for (int i = 0; i < 256; ++i) { ts += cs[i]; tq += cq[i]; }
        const NormMD r = norm_md(norm_type, ts, tq, (double)F * Tp);
        for (int t = lo; t < hi; ++t) md[base + t] = r;
    } else {
        double ps = 0.0, pq = 0.0;
        for (int i = 0; i < tid; ++i) { ps += cs[i]; pq += cq[i]; }
        for (int t = lo; t < hi; ++t) {
            ps += frame[(base + t) * 2];
            pq += frame[(base + t) * 2 + 1];
            md[base + t] = norm_md(norm_type, ps, pq, (double)F * (t + 1));
        }
    }
}

// frames per workgroup of fe_fsum_kernel: 32 with many utterances, fewer with few (B = 1: 12 workgroups took 13 us)
static int fsum_rows_per_wg(int B) { return B >= 8 ? 32 : B >= 4 ? 16 : B >= 2 ? 8 : 4; }
// (blockDim = the smallest multiple of 64 that covers F up to 512: see sb_offline_stats_kernel)
__global__ __launch_bounds__(512) void fe_fsum_kernel(const float* __restrict__ raw, const NormMD* __restrict__ md,
                                                      double* __restrict__ fsum, int B, int Tp, int F, int FP, int rows) {
    const long ub = (long)blockIdx.z * B + blockIdx.y;
    const int t0 = blockIdx.x * rows, t1 = min(t0 + rows, Tp);
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        double s = 0.0;
        for (int t = t0; t < t1; ++t) {
            const NormMD r = md[ub * Tp + t];
            s += (double)((raw[(ub * Tp + t) * FP + f] - r.m) / r.d);
        }
        atomicAdd(fsum + ub * FP + f, s);
    }
}

struct GateArgs {
    FrontendWeights w;
    const float* raw; NormMD* md; const double* fsum; const double* tot; float* gate;
    int B, Tp, F, FP;
    int offline_norm;          // FSNP_NORM_OFFLINE_*: (m, d) come from `tot` here and are WRITTEN to md for every frame; fsum = raw column
                               // sums.  -1: cumulative norms - md and fsum (of the normalised input) were produced by fe_scan / fe_fsum
};

// one workgroup per (branch, utt).  The (kmax-1) first and last normalised frames are staged in LDS so that every
// thread's prefix / suffix sums come from shared memory instead of 2*(kmax-1) dependent global round trips.
// Round 5: 1024 threads - the kernel is a chain of latency-bound phases on ONE workgroup per (branch, utt) (28 us whatever the
// batch), so every phase is cut into as many independent items as the workgroup has threads: the squeeze per (conv, bin), fc1 and fc2
// per (K slice, output).
constexpr int kGateThreads = 1024;
__global__ __launch_bounds__(kGateThreads) void fe_gate_kernel(GateArgs g) {
    constexpr int NTHR = kGateThreads;
    extern __shared__ float sh[];
    const int F = g.F, Tp = g.Tp, Fr = F / 2, FP = g.FP;
    constexpr int MAXK = 16;
    float* sq = sh;                    // [FP]  squeeze
    float* hid = sh + FP;              // [FP]  (F/2 used)
    float* edge = sh + 2 * FP;         // [2][MAXK][FP] normalised first / last frames
    float* feat3 = sh + (2 + 2 * MAXK) * FP;   // [3][FP] TSSE: the three conv branches' pooled features
    const int branch = blockIdx.y, b = blockIdx.x;
    const long ub = (long)branch * g.B + b;
    const int tid = threadIdx.x;
    const int kmax = max(max(g.w.ksize[0], g.w.ksize[1]), g.w.ksize[2]);

    const int att = g.w.attention;
    float* sqmax = edge;               // [FP] CBAM: max over t (edge is unused then)
    // offline norms: one (m, d) per branch and utterance, from the plane's totals the repack kernel accumulated; every later
    // consumer (fe_apply_kernel) reads the per-frame table, written here
    const bool offline = g.offline_norm >= 0;
    NormMD mu{0.0f, 1.0f};
    if (offline) {
        mu = norm_md(g.offline_norm, g.tot[ub * 2], g.tot[ub * 2 + 1], (double)F * Tp);
        for (int t = tid; t < Tp; t += NTHR) g.md[ub * Tp + t] = mu;
    }
    if (att == FSNP_ATT_TSSE) {
        for (int i = tid; i < 2 * (kmax - 1) * F; i += NTHR) {
            const int side = i / ((kmax - 1) * F), r = (i / F) % (kmax - 1), f = i % F;
            const int t = side == 0 ? r : Tp - 1 - r;                    // r-th frame from the start / from the end
            const NormMD m = offline ? mu : g.md[ub * Tp + t];
            edge[(side * MAXK + r) * FP + f] = (g.raw[(ub * Tp + t) * FP + f] - m.m) / m.d;
        }
        __syncthreads();
    }
    // sum over t of the normalised input: sum_t (x - m) / d = (column sum - T' m) / d for the offline norms
    auto total_of = [&](int f) -> double {
        return offline ? (g.fsum[ub * FP + f] - (double)Tp * (double)mu.m) / (double)mu.d : g.fsum[ub * FP + f];
    };
    if (att != FSNP_ATT_TSSE) {
        for (int f = tid; f < F; f += NTHR) {
            // SE / ECA / CBAM squeeze = mean over time (attention_model.py:31, :349, :320)
            sq[f] = (float)(total_of(f) / (double)Tp);
            if (att == FSNP_ATT_CBAM) {                      // + max over time (attention_model.py:321)
                float mx = -3.4e38f;
                for (int t = 0; t < Tp; ++t) {
                    const NormMD m = offline ? mu : g.md[ub * Tp + t];
                    mx = fmaxf(mx, (g.raw[(ub * Tp + t) * FP + f] - m.m) / m.d);
                }
                sqmax[f] = mx;
            }
        }
    } else {
        for (int item = tid; item < 3 * F; item += NTHR) {   // one (conv branch, bin) per thread
            const int c = item / F, f = item - c * F;
            const double S = total_of(f);
            const float* first = edge + f;                   // first[r * FP] = r-th normalised frame
            const float* last = edge + (long)MAXK * FP + f;  // last[r * FP]  = r-th frame from the end
            const int K = g.w.ksize[c];
            const float* wk = g.w.conv_w[branch][c] + (long)f * K;
            // tap j sees frames [j, j + T' - K]: everything but the first j and the last K-1-j frames
            double pj = 0.0, sj = 0.0, acc = 0.0;
            for (int r = 0; r < K - 1; ++r) sj += (double)last[r * FP];
            for (int j = 0; j < K; ++j) {
                acc += (double)wk[j] * (S - pj - sj);
                if (j + 1 < K) { pj += (double)first[j * FP]; sj -= (double)last[(K - 2 - j) * FP]; }
            }
            const float feat = (float)(acc / (double)(Tp - K + 1)) + g.w.conv_b[branch][c][f];
            feat3[c * FP + f] = fmaxf(feat, 0.f);
        }
        __syncthreads();
        for (int f = tid; f < F; f += NTHR) {
            float squeeze = g.w.cat_b[branch][0];
            for (int c = 0; c < 3; ++c) squeeze += g.w.cat_w[branch][c] * feat3[c * FP + f];
            sq[f] = squeeze;
        }
    }
    __syncthreads();
    if (att == FSNP_ATT_ECA && branch == 0 && g.w.subband_num > 1) {
        // subband_num > 1 (fullsubnet_plus.py:146-153), magnitude branch only: the normalised magnitude is reflect-padded
        // at the high-frequency end by pad = sn - F % sn rows and viewed as [C = (F + pad) / sn channels][sn * T']; the ECA
        // layer pools, convolves and gates those C channels, then the first F rows are kept.  So frequency f takes the gate
        // of channel f / sn, whose squeeze is the mean over the sn rows sn c .. sn c + sn - 1 (row F + k = row F - 2 - k).
        const int sn = g.w.subband_num, pad = sn - F % sn, C = (F + pad) / sn;
        const float w0 = g.w.cat_w[0][0], w1 = g.w.cat_w[0][1], w2 = g.w.cat_w[0][2];
        for (int c = tid; c < C; c += NTHR) {
            float m = 0.f;
            for (int r = 0; r < sn; ++r) { const int row = sn * c + r; m += sq[row < F ? row : 2 * (F - 1) - row]; }
            hid[c] = m / (float)sn;
        }
        __syncthreads();
        for (int o = tid; o < F; o += NTHR) {
            const int c = o / sn;
            const float y = w0 * (c > 0 ? hid[c - 1] : 0.f) + w1 * hid[c] + w2 * (c + 1 < C ? hid[c + 1] : 0.f);
            g.gate[ub * FP + o] = 1.0f / (1.0f + expf(-y));
        }
        return;
    }
    if (att == FSNP_ATT_ECA) {
        // Conv1d(1, 1, 3, padding=1, bias=False) ALONG THE CHANNEL AXIS of the pooled vector, then sigmoid
        const float w0 = g.w.cat_w[branch][0], w1 = g.w.cat_w[branch][1], w2 = g.w.cat_w[branch][2];
        for (int o = tid; o < F; o += NTHR) {
            const float y = w0 * (o > 0 ? sq[o - 1] : 0.f) + w1 * sq[o] + w2 * (o + 1 < F ? sq[o + 1] : 0.f);
            g.gate[ub * FP + o] = 1.0f / (1.0f + expf(-y));
        }
        return;
    }
    // fc1 + ReLU, fc2 + sigmoid over TRANSPOSED weights (coalesced, independent loads; a wave-per-output dot product is a chain of
    // dependent L2 round trips here: measured 120 us).  Round 3: the kernel is latency-bound (one workgroup per utterance and
    // branch), so every thread works - fc1's K is cut into NTHR / (F / 2) slices per output, fc2's into NTHR / F - and every chain
    // keeps 16 loads in flight on 4 independent accumulators.
    auto dot = [&](const float* wT, long stride, const float* x, int k0, int k1) -> float {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = k0;
        for (; k + 16 <= k1; k += 16) {
            float wv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[j] = wT[(long)(k + j) * stride];
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                a0 += wv[j] * x[k + j]; a1 += wv[j + 1] * x[k + j + 1]; a2 += wv[j + 2] * x[k + j + 2]; a3 += wv[j + 3] * x[k + j + 3];
            }
        }
        for (; k < k1; ++k) a0 += wT[(long)k * stride] * x[k];
        return (a0 + a1) + (a2 + a3);
    };
    // [slices][Fr] partial sums in `edge` (the staged edge frames are no longer needed; CBAM keeps its max vector in edge[0, FP): its
    // partials sit behind it).  part2 follows part's ns1 * Fr floats (a fixed + 256 aliased them from num_freqs = 514 up); the
    // slice count is capped so that FP + 2 ns1 Fr floats always fit edge's 32 FP (tiny num_freqs)
    const int ns1 = Fr >= NTHR ? 1 : min(NTHR / Fr, (31 * FP) / (2 * Fr)), chunk = cdiv(F, ns1);
    float* part = att == FSNP_ATT_CBAM ? edge + FP : edge;
    float* part2 = part + ns1 * Fr;
    __syncthreads();                                         // every thread is done with `edge`
    for (int item = tid; item < Fr * ns1; item += NTHR) {
        const int o = item % Fr, sl = item / Fr;
        const int k0 = sl * chunk, k1 = min(F, k0 + chunk);
        part[sl * Fr + o] = dot(g.w.fc1_wT[branch] + o, Fr, sq, k0, k1);
        if (att == FSNP_ATT_CBAM) part2[sl * Fr + o] = dot(g.w.fc1_wT[branch] + o, Fr, sqmax, k0, k1);
    }
    __syncthreads();
    for (int o = tid; o < Fr; o += NTHR) {
        float acc = g.w.fc1_b[branch][o];
        for (int sl = 0; sl < ns1; ++sl) acc += part[sl * Fr + o];
        float h = fmaxf(acc, 0.f);
        if (att == FSNP_ATT_CBAM) {                          // relu(fc1(mean)) + relu(fc1(max))
            float acc2 = g.w.fc1_b[branch][o];
            for (int sl = 0; sl < ns1; ++sl) acc2 += part2[sl * Fr + o];
            h += fmaxf(acc2, 0.f);
        }
        hid[o] = h;
    }
    __syncthreads();
    // fc2: K = F / 2 cut into ns2 slices per output ([ns2][F] partials over the fc1 partials, which hid has consumed), then bias +
    // sigmoid per output
    const int ns2 = F >= NTHR ? 1 : max(1, min(min(NTHR / F, (30 * FP) / F), Fr / 16)), chunk2 = cdiv(Fr, ns2);
    float* part3 = edge + FP;
    for (int item = tid; item < F * ns2; item += NTHR) {
        const int sl = item / F, o = item - sl * F;
        const int k0 = sl * chunk2, k1 = min(Fr, k0 + chunk2);
        part3[sl * F + o] = dot(g.w.fc2_wT[branch] + o, F, hid, k0, k1);
    }
    __syncthreads();
    for (int o = tid; o < F; o += NTHR) {
        float acc = g.w.fc2_b[branch][o];
        for (int sl = 0; sl < ns2; ++sl) acc += part3[sl * F + o];
        g.gate[ub * FP + o] = 1.0f / (1.0f + expf(-acc));
    }
}

__global__ __launch_bounds__(256) void fe_apply_kernel(const float* __restrict__ raw, const NormMD* __restrict__ md,
                                                       const float* __restrict__ gate, float* __restrict__ att,
                                                       long rows, int Tp, int F, int FP) {
    // one thread per (row, f); row = (branch*B + utt)*Tp + t
    const long total = rows * FP;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / FP;
        const int f = (int)(i - row * FP);
        float v = 0.0f;
        if (f < F) {
            const NormMD r = md[row];
            v = ((raw[i] - r.m) / r.d) * gate[(row / Tp) * FP + f];
        }
        att[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// decompress_cIRM (mask.py:60-63) + complex multiply (inferencer.py:152-157)
__global__ __launch_bounds__(256) void apply_cirm_kernel(const float* __restrict__ mask, const float2* __restrict__ noisy,
                                                         long sb, long sf, long st, float2* __restrict__ out, long ob,
                                                         long of, long ot, int B, int F, int T) {
    // thread per (b, t, f) with f fastest: matches torch.stft's memory order for the complex operands
    const long total = (long)B * T * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i % F);
        const int t = (int)((i / F) % T);
        const int b = (int)(i / ((long)F * T));
        float m[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float v = mask[(((long)b * 2 + o) * F + f) * T + t];
            const float lim = 9.9f, K = 10.0f;
            v = v >= lim ? lim : (v <= -lim ? -lim : v);
            m[o] = -K * logf((K - v) / (K + v));
        }
        const float2 x = noisy[b * sb + f * sf + t * st];
        out[b * ob + f * of + t * ot] = make_float2(m[0] * x.x - m[1] * x.y, m[1] * x.x + m[0] * x.y);
    }
}

void launch_apply_cirm(const float* mask, const float* noisy, const int64_t strides[3], float* out,
                       const int64_t out_strides[3], int B, int F, int T, hipStream_t s) {
    const long total = (long)B * T * F;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(apply_cirm_kernel, dim3(blocks), dim3(256), 0, s, mask, reinterpret_cast<const float2*>(noisy),
                       strides[0], strides[1], strides[2], reinterpret_cast<float2*>(out), out_strides[0], out_strides[1],
                       out_strides[2], B, F, T);
}

void launch_frontend(const Dims& d, int norm_type, const float* const in[3], const int64_t strides[3][3], bool is_complex,
                     const FrontendWeights& w, const FrontendBuffers& buf, hipStream_t s) {
    const bool offline = norm_type == FSNP_NORM_OFFLINE_LAPLACE || norm_type == FSNP_NORM_OFFLINE_GAUSSIAN;
    double* colsum = offline ? buf.fsum : nullptr;       // (cumulative norms: fsum is the sum of the NORMALISED input, fe_fsum_kernel)
    if (is_complex) {
        hipLaunchKernelGGL(fe_repack_complex_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), d.B), dim3(256), 0, s,
                           reinterpret_cast<const float2*>(in[0]), (long)strides[0][0], (long)strides[0][1],
                           (long)strides[0][2], buf.raw, colsum, buf.tot, 3, d.B, d.T, d.Tp, d.F, d.FP);
    } else {
        StridedIn si;
        for (int i = 0; i < 3; ++i) {
            si.p[i] = in[i];
            si.sb[i] = strides[i][0]; si.sf[i] = strides[i][1]; si.st[i] = strides[i][2];
        }
        hipLaunchKernelGGL(fe_repack_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), 3 * d.B), dim3(256), 0, s, si, buf.raw, colsum, buf.tot,
                           d.B, d.T, d.Tp, d.F, d.FP);
    }
    if (!offline) {
        hipLaunchKernelGGL(fe_frame_kernel, dim3(d.Tp, d.B, 3), dim3(64), 0, s, buf.raw, buf.frame, d.B, d.Tp, d.F, d.FP);
        hipLaunchKernelGGL(fe_scan_kernel, dim3(d.B, 3), dim3(256), 0, s, buf.frame, buf.md, d.B, d.Tp, d.F, norm_type);
        const int frows = fsum_rows_per_wg(d.B);
        const int fthreads = d.F <= 256 ? 256 : d.F >= 512 ? 512 : (d.F + 63) / 64 * 64;
        hipLaunchKernelGGL(fe_fsum_kernel, dim3(cdiv(d.Tp, frows), d.B, 3), dim3(fthreads), 0, s, buf.raw, buf.md, buf.fsum,
                           d.B, d.Tp, d.F, d.FP, frows);
    }
    GateArgs g;
    g.w = w; g.raw = buf.raw; g.md = buf.md; g.fsum = buf.fsum; g.tot = buf.tot; g.gate = buf.gate;
    g.B = d.B; g.Tp = d.Tp; g.F = d.F; g.FP = d.FP;
    g.offline_norm = offline ? norm_type : -1;
    // 37 FP floats of dynamic LDS: beyond 64 KiB (num_freqs > 440) the kernel needs the opt-in; beyond a CU's LDS the launch fails
    // loudly (hipGetLastError in fsnp_forward)
    static PerDeviceOnce gate_once;
    gate_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fe_gate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    hipLaunchKernelGGL(fe_gate_kernel, dim3(d.B, 3), dim3(kGateThreads), (size_t)(2 + 2 * 16 + 3) * d.FP * sizeof(float), s, g);
    const long rows = 3L * d.B * d.Tp;
    const long total = rows * d.FP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(fe_apply_kernel, dim3(blocks), dim3(256), 0, s, buf.raw, buf.md, buf.gate, buf.att, rows, d.Tp,
                       d.F, d.FP);
}

// speech_enhance/fullsubnet/model/fullsubnet.py:82-89: pad the look-ahead, norm(noisy_mag) - one branch, no attention.
void launch_frontend_mag(const Dims& d, int norm_type, const float* mag, const int64_t strides[3], bool is_complex,
                         const FrontendBuffers& buf, hipStream_t s) {
    if (is_complex) {
        hipLaunchKernelGGL(fe_repack_complex_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), d.B), dim3(256), 0, s,
                           reinterpret_cast<const float2*>(mag), (long)strides[0], (long)strides[1], (long)strides[2],
                           buf.raw, nullptr, nullptr, 1, d.B, d.T, d.Tp, d.F, d.FP);
    } else {
        StridedIn si;
        for (int i = 0; i < 3; ++i) {
            si.p[i] = mag;
            si.sb[i] = strides[0]; si.sf[i] = strides[1]; si.st[i] = strides[2];
        }
        hipLaunchKernelGGL(fe_repack_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), d.B), dim3(256), 0, s, si, buf.raw, nullptr, nullptr,
                           d.B, d.T, d.Tp, d.F, d.FP);
    }
    hipLaunchKernelGGL(fe_frame_kernel, dim3(d.Tp, d.B, 1), dim3(64), 0, s, buf.raw, buf.frame, d.B, d.Tp, d.F, d.FP);
    hipLaunchKernelGGL(fe_scan_kernel, dim3(d.B, 1), dim3(256), 0, s, buf.frame, buf.md, d.B, d.Tp, d.F, norm_type);
}

// ------------------------------------------------------------------------------------------------
// Stage-level entry (fsnp_channel_attention): ONE branch's channel attention on a caller's [B, F, T] tensor, as the reference calls the
// submodule - `self.channel_attention(fb_input)` (fullsubnet_plus.py:160-165; ChannelTimeSenseSELayer.forward attention_model.py:78-101,
// or the SE / ECA / CBAM layer the handle was configured with).  The input is whatever the caller normalised (or not): the kernels run on
// their per-frame (m, d) path with the identity table.  `w` holds the branch's weights in slot 0.  (With subband_num > 1 the FORWARD
// regroups the magnitude branch's channels around this call, fullsubnet_plus.py:146-153; the module itself never does: subband_num = 1 here.)
__global__ void fe_identity_md_kernel(NormMD* __restrict__ md, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) md[i] = NormMD{0.0f, 1.0f};
}
void launch_attention_stage(const Dims& d, const FrontendWeights& w, const float* in, const int64_t strides[3],
                            const FrontendBuffers& buf, hipStream_t s) {
    StridedIn si;
    for (int i = 0; i < 3; ++i) { si.p[i] = in; si.sb[i] = strides[0]; si.sf[i] = strides[1]; si.st[i] = strides[2]; }
    hipLaunchKernelGGL(fe_repack_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), d.B), dim3(256), 0, s, si, buf.raw, nullptr, nullptr,
                       d.B, d.T, d.Tp, d.F, d.FP);
    const long nmd = (long)d.B * d.Tp;
    hipLaunchKernelGGL(fe_identity_md_kernel, dim3((unsigned)((nmd + 255) / 256)), dim3(256), 0, s, buf.md, nmd);
    const int frows = fsum_rows_per_wg(d.B);
    const int fthreads = d.F <= 256 ? 256 : d.F >= 512 ? 512 : (d.F + 63) / 64 * 64;
    hipLaunchKernelGGL(fe_fsum_kernel, dim3(cdiv(d.Tp, frows), d.B, 1), dim3(fthreads), 0, s, buf.raw, buf.md, buf.fsum, d.B, d.Tp, d.F, d.FP, frows);
    GateArgs g;
    g.w = w;
    g.w.subband_num = 1;
    g.raw = buf.raw; g.md = buf.md; g.fsum = buf.fsum; g.tot = buf.tot; g.gate = buf.gate;
    g.B = d.B; g.Tp = d.Tp; g.F = d.F; g.FP = d.FP;
    g.offline_norm = -1;
    static PerDeviceOnce gate_once;
    gate_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fe_gate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    hipLaunchKernelGGL(fe_gate_kernel, dim3(d.B, 1), dim3(kGateThreads), (size_t)(2 + 2 * 16 + 3) * d.FP * sizeof(float), s, g);
    const long rows = (long)d.B * d.Tp, total = rows * d.FP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(fe_apply_kernel, dim3(blocks), dim3(256), 0, s, buf.raw, buf.md, buf.gate, buf.att, rows, d.Tp, d.F, d.FP);
}
// the time-major repack alone ([B, F, T] strided -> [B][T'][FP], pad columns and look-ahead rows zero): fsnp_fullband_model's input
void launch_repack_plane(const Dims& d, const float* in, const int64_t strides[3], float* raw, hipStream_t s) {
    StridedIn si;
    for (int i = 0; i < 3; ++i) { si.p[i] = in; si.sb[i] = strides[0]; si.sf[i] = strides[1]; si.st[i] = strides[2]; }
    hipLaunchKernelGGL(fe_repack_kernel, dim3(cdiv(d.FP, 32), cdiv(d.Tp, 32), d.B), dim3(256), 0, s, si, raw, nullptr, nullptr,
                       d.B, d.T, d.Tp, d.F, d.FP);
}

}  // namespace fsnp
