// subband.hip - statistics of the sub-band input tensor without materialising it.
//
// The reference builds sb_input[b, f, j, t] = cat(unfold(att_mag, 15), fb_mag, fb_real, fb_imag)
// (speech_enhance/fullsubnet_plus/model/fullsubnet_plus.py:167-188; BaseModel.unfold
// speech_enhance/audio_zen/model/base_model.py:15-47) - a [B,257,34,T'] tensor, 143 MB at B=32 - and then
// applies self.norm to it (fullsubnet_plus.py:189).  The original FullSubNet builds cat(unfold(noisy_mag, 15),
// fb_output) with ONE full-band branch (speech_enhance/fullsubnet/model/fullsubnet.py:92-105).  Here:
//   offline norms   : sum / sumsq over the tensor == sum_r w_r * rowstat(att_mag[:, r]) + rowstats(fb*),
//                     where w_r = number of (f, j) pairs whose reflect-padded neighbour index is r;
//   cumulative norms: the reshape to [B*257, 34, T'] (base_model.py:237-238, 288-289) makes the running
//                     statistics per sub-band sequence -> one (m_t, d_t) table row per sequence.
// The LSTM kernel applies (x - m) / d while it gathers its input frame.
#include "fsnp_common.h"

namespace fsnp {

#define FSNP_EPS 1.1920928955078125e-07f

__device__ __forceinline__ NormMD sb_norm_md(int norm_type, double sum, double sq, double count) {
    NormMD r;
    const double mean = sum / count;
    if (norm_type == FSNP_NORM_OFFLINE_LAPLACE) { r.m = 0.f; r.d = (float)mean + 1e-5f; }
    else if (norm_type == FSNP_NORM_CUMULATIVE_LAPLACE) { r.m = 0.f; r.d = (float)mean + FSNP_EPS; }
    else if (norm_type == FSNP_NORM_OFFLINE_GAUSSIAN) {
        double var = (sq - count * mean * mean) / (count - 1.0);
        if (var < 0) var = 0;
        r.m = (float)mean; r.d = (float)sqrt(var) + 1e-5f;
    } else {
        const double var = (sq - 2.0 * mean * sum) / count + mean * mean;
        r.m = (float)mean; r.d = (float)sqrt(var + (double)FSNP_EPS);
    }
    return r;
}

// frames per workgroup: 16 with many utterances; fewer with few, so that a small batch still launches >= ~64 workgroups (B = 1: 8
// workgroups took 25 us, latency-bound)
static int sb_rows_per_wg(int B) { return B >= 8 ? 16 : B >= 4 ? 8 : B >= 2 ? 4 : 2; }
// blockDim = the smallest multiple of 64 that covers F up to 512 (F = 257: 320): with 256 threads, frequency 256 was a second pass of
// thread 0 alone and the whole workgroup waited for it (24 us at B = 1 for 17 MB of reads)
__global__ __launch_bounds__(512) void sb_offline_stats_kernel(const float* __restrict__ att_mag,
                                                               const float* __restrict__ fb, long fb_bs, int nfb,
                                                               const float* __restrict__ refl_w,
                                                               const float* __restrict__ refl_wfb,
                                                               double* __restrict__ acc, int Tp, int F, int FP, int rows) {
    __shared__ double red[16];
    const int b = blockIdx.y, t0 = blockIdx.x * rows, t1 = min(t0 + rows, Tp);
    double s = 0.0, q = 0.0;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const double wr = refl_w[f], wfb = refl_wfb[f];
        for (int t = t0; t < t1; ++t) {
            const long i = ((long)b * Tp + t) * FP + f;
            const double a = att_mag[i];
            s += wr * a;
            q += wr * a * a;
            for (int k = 0; k < nfb; ++k) {
                const double v = fb[k * fb_bs + i];
                s += wfb * v;
                q += wfb * v * v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tq = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += red[w * 2]; tq += red[w * 2 + 1]; }
        atomicAdd(acc + b * 2, ts);
        atomicAdd(acc + b * 2 + 1, tq);
    }
}

__global__ void sb_offline_final_kernel(const double* __restrict__ acc, NormMD* __restrict__ md_utt, int B,
                                        double count, int norm_type) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) md_utt[b] = sb_norm_md(norm_type, acc[b * 2], acc[b * 2 + 1], count);
}

// cumulative norms: one workgroup per sub-band sequence.  Every thread sums the NIN features of its frames
// (t = tid, tid + 256, ...) in fp64, then a workgroup-wide inclusive scan over t turns the per-frame (sum, sumsq) into the
// running statistics of base_model.py:237-258 / 288-316.  (A thread per sequence, serial in t, took 2.8 ms for 10 s clips.)
__global__ __launch_bounds__(256) void sb_cumulative_kernel(const float* __restrict__ att_mag,
                                                            const float* __restrict__ fb, long fb_bs,
                                                            const RowDesc* __restrict__ rows, NormMD* __restrict__ md_row,
                                                            int num_slots, int Tp, int F, int FP, int nsbn, int nfbn, int nin,
                                                            int norm_type) {
    __shared__ double wsum[4][2];
    __shared__ double carry[2];
    const int row = blockIdx.x;
    const RowDesc rd = rows[row];
    if (!rd.valid) return;                               // uniform per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsb = 2 * nsbn + 1;
    if (tid == 0) { carry[0] = 0.0; carry[1] = 0.0; }
    __syncthreads();
    for (int t0 = 0; t0 < Tp; t0 += 256) {
        const int t = t0 + tid;
        double s = 0.0, q = 0.0;
        if (t < Tp) {
            const long base = ((long)rd.b * Tp + t) * FP;
            for (int j = 0; j < nsb; ++j) {
                const double v = att_mag[base + reflect_index(rd.f - nsbn + j, F)];
                s += v; q += v * v;
            }
            for (int j = nsb; j < nin; ++j) {
                const double v = att_mag[sb_feature_offset(j, rd.f, 0, F, nsbn, nfbn, (int)(fb - att_mag), (int)fb_bs) + base];
                s += v; q += v * v;
            }
        }
        // inclusive scan over the 256 frames of this chunk: within the wave, then across the 4 waves, plus the carry
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double us = __shfl_up(s, o), uq = __shfl_up(q, o);
            if (lane >= o) { s += us; q += uq; }
        }
        if (lane == 63) { wsum[wave][0] = s; wsum[wave][1] = q; }
        __syncthreads();
        double cs = carry[0], cq = carry[1];
        for (int wv = 0; wv < wave; ++wv) { cs += wsum[wv][0]; cq += wsum[wv][1]; }
        s += cs; q += cq;
        if (t < Tp) md_row[(long)row * Tp + t] = sb_norm_md(norm_type, s, q, (double)nin * (t + 1));
        __syncthreads();
        if (tid == 255) { carry[0] = s; carry[1] = q; }
        __syncthreads();
    }
}

void launch_subband_stats(const Dims& d, int norm_type, const SubbandBuffers& buf, const RowDesc* rows,
                          int num_slots, hipStream_t s) {
    const long fb_bs = (long)d.B * d.Tp * d.FP;
    if (norm_type == FSNP_NORM_OFFLINE_LAPLACE || norm_type == FSNP_NORM_OFFLINE_GAUSSIAN) {
        const int rows = sb_rows_per_wg(d.B);
        const int threads = d.F <= 256 ? 256 : d.F >= 512 ? 512 : (d.F + 63) / 64 * 64;
        hipLaunchKernelGGL(sb_offline_stats_kernel, dim3(cdiv(d.Tp, rows), d.B), dim3(threads), 0, s, buf.att_mag, buf.fb,
                           fb_bs, (d.NIN - d.NSB) / (2 * buf.NFBN + 1), buf.refl_w, buf.refl_wfb, buf.acc, d.Tp, d.F, d.FP, rows);
        hipLaunchKernelGGL(sb_offline_final_kernel, dim3(cdiv(d.B, 64)), dim3(64), 0, s, buf.acc, buf.md_utt, d.B,
                           (double)d.F * d.NIN * d.Tp, norm_type);
    } else {
        hipLaunchKernelGGL(sb_cumulative_kernel, dim3(num_slots), dim3(256), 0, s, buf.att_mag, buf.fb, fb_bs,
                           rows, buf.md_row, num_slots, d.Tp, d.F, d.FP, (d.NSB - 1) / 2, buf.NFBN, d.NIN, norm_type);
    }
}

// ---- sequence_model="TCN" (sequence_model.py:47-58): the sub-band model is a TCN stack over [N, 34, T'], so the
// sub-band input of fullsubnet_plus.py:167-202 IS materialised here, time-major [slot][t][xstride], normalised.
__global__ __launch_bounds__(256) void sb_gather_kernel(SbGatherArgs a) {
    const int slot = blockIdx.x;
    const RowDesc rd = a.rows[slot];
    for (int i = threadIdx.x; i < a.Tp * a.xstride; i += 256) {
        const int t = i / a.xstride, j = i % a.xstride;
        float v = 0.0f;
        if (rd.valid && j < a.NIN) {
            const int off = sb_feature_offset(j, rd.f, (rd.b * a.Tp + t) * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            const NormMD m = a.md_row ? a.md_row[(size_t)slot * a.Tp + t] : a.md_utt[rd.b];
            v = (a.att_mag[off] - m.m) / m.d;
        }
        a.x[((size_t)slot * a.Tp + t) * a.xstride + j] = v;
    }
}

__global__ void sb_scatter_kernel(const float* __restrict__ y, int ystride, const RowDesc* __restrict__ rows,
                                  float* __restrict__ out, long out_stride_o, int num_slots, int Tp, int LA, int OC) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (slot, o, t), o < OC = output_size
    const int T = Tp - LA;
    if (i >= (long)num_slots * OC * T) return;
    const int t = (int)(i % T), o = (int)((i / T) % OC), slot = (int)(i / ((long)OC * T));
    const RowDesc rd = rows[slot];
    if (rd.valid) out[(size_t)rd.out_off + (size_t)o * out_stride_o + t] = y[((size_t)slot * Tp + t + LA) * ystride + o];
}

void launch_sb_gather(const SbGatherArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(sb_gather_kernel, dim3(a.num_slots), dim3(256), 0, s, a);
}

void launch_sb_scatter(const float* y, int ystride, const RowDesc* rows, float* out, long out_stride_o, int num_slots,
                       int Tp, int LA, int out_channels, hipStream_t s) {
    const long n = (long)num_slots * out_channels * (Tp - LA);
    hipLaunchKernelGGL(sb_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, ystride, rows, out,
                       out_stride_o, num_slots, Tp, LA, out_channels);
}

}  // namespace fsnp
