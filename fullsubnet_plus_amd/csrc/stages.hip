// stages.hip - BaseModel's public helpers as stage-level entry points (include/fsnp.h: fsnp_norm, fsnp_unfold).
//
// The reference model exposes `self.norm = self.norm_wrapper(norm_type)` (fullsubnet_plus.py:115, fullsubnet.py:61),
// `BaseModel.norm_wrapper` (audio_zen/model/base_model.py:318-330) and `BaseModel.unfold` (base_model.py:15-47) on every model object.
// Inside the forward neither exists as a kernel here - the norms are folded into the consumers' (m, d) tables and the unfold is never
// materialised (csrc/lstm*.hip gather the sub-band input on the fly) - so the module protocol is served by these two small, generic
// kernels on arbitrary [B, C, F, T] device tensors.  Memory-bound elementwise work: coalesced along T (the innermost axis of the
// reference's layout), statistics in fp64 like the forward's own (csrc/frontend.hip), no workspace kept (stream-ordered scratch).
#include "fsnp_common.h"

namespace fsnp {
void set_error(const char* fmt, ...);

namespace {

struct Strides4 { long b, c, f, t; };

// per (bc, t): sum and sum of squares over F.  One workgroup per (t-block of 64, bc): thread = frame (coalesced along t for the
// contiguous layout; strided inputs - torch.stft views are [B][T][F] in memory - still read every element once)
__global__ __launch_bounds__(64) void stage_frame_kernel(const float* __restrict__ in, Strides4 st, double* __restrict__ frame, int C, int F, int T) {
    const int t = blockIdx.x * 64 + threadIdx.x, bc = blockIdx.y;
    if (t >= T) return;
    const float* p = in + (long)(bc / C) * st.b + (long)(bc % C) * st.c + (long)t * st.t;
    double s = 0.0, q = 0.0;
    for (int f = 0; f < F; ++f) { const double v = (double)p[(long)f * st.f]; s += v; q += v * v; }
    frame[((long)bc * T + t) * 2] = s;
    frame[((long)bc * T + t) * 2 + 1] = q;
}

// offline norms: one (m, d) per utterance from the totals over (C, F, T) (base_model.py:211-226, 261-275); cumulative norms: a prefix
// over the frames of every (b, c) row with count F (t + 1) (base_model.py:228-258, 278-316).  One workgroup per utterance.
__global__ __launch_bounds__(256) void stage_scan_kernel(const double* __restrict__ frame, NormMD* __restrict__ md, int C, int F, int T, int norm_type) {
    __shared__ double cs[256], cq[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool cumulative = norm_type == FSNP_NORM_CUMULATIVE_LAPLACE || norm_type == FSNP_NORM_CUMULATIVE_LAYER;
    if (!cumulative) {
        const long n = (long)C * T;
        double s = 0.0, q = 0.0;
        for (long i = tid; i < n; i += 256) { s += frame[((long)b * n + i) * 2]; q += frame[((long)b * n + i) * 2 + 1]; }
        cs[tid] = s; cq[tid] = q;
        __syncthreads();
        double ts = 0.0, tq = 0.0;
        for (int i = 0; i < 256; ++i) { ts += cs[i]; tq += cq[i]; }
        const NormMD r = norm_md(norm_type, ts, tq, (double)C * F * T);
        for (long i = tid; i < n; i += 256) md[(long)b * n + i] = r;
        return;
    }
    for (int c = 0; c < C; ++c) {
        const long base = ((long)b * C + c) * T;
        const int chunk = cdiv(T, 256);
        const int lo = tid * chunk, hi = min(lo + chunk, T);
        double s = 0.0, q = 0.0;
        for (int t = lo; t < hi; ++t) { s += frame[(base + t) * 2]; q += frame[(base + t) * 2 + 1]; }
        __syncthreads();
        cs[tid] = s; cq[tid] = q;
        __syncthreads();
        double ps = 0.0, pq = 0.0;
        for (int i = 0; i < tid; ++i) { ps += cs[i]; pq += cq[i]; }
        for (int t = lo; t < hi; ++t) {
            ps += frame[(base + t) * 2];
            pq += frame[(base + t) * 2 + 1];
            md[base + t] = norm_md(norm_type, ps, pq, (double)F * (t + 1));
        }
    }
}

__global__ __launch_bounds__(256) void stage_apply_kernel(const float* __restrict__ in, Strides4 st, const NormMD* __restrict__ md, float* __restrict__ out,
                                                          int C, int F, int T, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T), f = (int)((i / T) % F);
        const long bc = i / ((long)T * F);
        const NormMD r = md[bc * T + t];
        out[i] = (in[(bc / C) * st.b + (bc % C) * st.c + (long)f * st.f + (long)t * st.t] - r.m) / r.d;
    }
}

// out[b][f][c][j][t] = in[b][c][reflect(f - N + j)][t], j < 2 N + 1 (functional.pad(mode="reflect") + functional.unfold, base_model.py:32-45)
__global__ __launch_bounds__(256) void stage_unfold_kernel(const float* __restrict__ in, Strides4 st, float* __restrict__ out, int C, int F, int T, int N, long total) {
    const int NS = 2 * N + 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T), j = (int)((i / T) % NS), c = (int)((i / ((long)T * NS)) % C);
        const int f = (int)((i / ((long)T * NS * C)) % F);
        const long b = i / ((long)T * NS * C * F);
        int fs = f - N + j;
        if (fs < 0) fs = -fs;
        if (fs >= F) fs = 2 * (F - 1) - fs;
        out[i] = in[b * st.b + (long)c * st.c + (long)fs * st.f + (long)t * st.t];
    }
}

// [B][T'][FP] time-major plane (the layout of every full-band stage buffer) -> the caller's contiguous [B, F, T] (frames t < T)
__global__ __launch_bounds__(256) void stage_tm_to_bft_kernel(const float* __restrict__ tm, float* __restrict__ out, int T, int Tp, int F, int FP) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, f0 = blockIdx.x * 32, t0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, f = f0 + tx;
        tile[ty + 8 * i][tx] = (t < T && f < F) ? tm[((long)b * Tp + t) * FP + f] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = f0 + ty + 8 * i, t = t0 + tx;
        if (f < F && t < T) out[((long)b * F + f) * T + t] = tile[tx][ty + 8 * i];
    }
}

}  // namespace

int launch_norm_stage(int norm_type, const float* in, const int64_t strides[4], float* out, int B, int C, int F, int T, hipStream_t s) {
    const Strides4 st{(long)strides[0], (long)strides[1], (long)strides[2], (long)strides[3]};
    const size_t rows = (size_t)B * C * T;
    unsigned char* work = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&work), rows * (16 + sizeof(NormMD)), s));
    double* frame = reinterpret_cast<double*>(work);
    NormMD* md = reinterpret_cast<NormMD*>(work + rows * 16);
    hipLaunchKernelGGL(stage_frame_kernel, dim3(cdiv(T, 64), B * C), dim3(64), 0, s, in, st, frame, C, F, T);
    hipLaunchKernelGGL(stage_scan_kernel, dim3(B), dim3(256), 0, s, frame, md, C, F, T, norm_type);
    const long total = (long)rows * F;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(stage_apply_kernel, dim3(blocks), dim3(256), 0, s, in, st, md, out, C, F, T, total);
    FSNP_HIP_CHECK(hipGetLastError());
    FSNP_HIP_CHECK(hipFreeAsync(work, s));
    return 0;
}

void launch_unfold_stage(const float* in, const int64_t strides[4], float* out, int B, int C, int F, int T, int num_neighbor, hipStream_t s) {
    const Strides4 st{(long)strides[0], (long)strides[1], (long)strides[2], (long)strides[3]};
    const int N = num_neighbor < 1 ? 0 : num_neighbor;       // < 1: input.permute(0, 2, 1, 3) with a unit sub-band axis (base_model.py:29-31)
    const long total = (long)B * F * C * (2 * N + 1) * T;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(stage_unfold_kernel, dim3(blocks), dim3(256), 0, s, in, st, out, C, F, T, N, total);
}

void launch_tm_to_bft(const float* tm, float* out, int B, int T, int Tp, int F, int FP, hipStream_t s) {
    hipLaunchKernelGGL(stage_tm_to_bft_kernel, dim3(cdiv(F, 32), cdiv(T, 32), B), dim3(256), 0, s, tm, out, T, Tp, F, FP);
}

}  // namespace fsnp

using namespace fsnp;

extern "C" {

int fsnp_norm(int32_t norm_type, const float* in, const int64_t strides[4], float* out, int32_t batch, int32_t channels, int32_t freqs,
              int32_t frames, void* hip_stream) {
    if (!in || !out || !strides) { set_error("fsnp_norm: null argument"); return 1; }
    if (norm_type < 0 || norm_type > 3) { set_error("fsnp_norm: unknown norm_type %d", norm_type); return 2; }
    if (batch <= 0 || channels <= 0 || freqs <= 0 || frames <= 0) { set_error("fsnp_norm: empty input [%d, %d, %d, %d]", batch, channels, freqs, frames); return 2; }
    return launch_norm_stage(norm_type, in, strides, out, batch, channels, freqs, frames, static_cast<hipStream_t>(hip_stream));
}

int fsnp_unfold(const float* in, const int64_t strides[4], float* out, int32_t batch, int32_t channels, int32_t freqs, int32_t frames,
                int32_t num_neighbor, void* hip_stream) {
    if (!in || !out || !strides) { set_error("fsnp_unfold: null argument"); return 1; }
    if (batch <= 0 || channels <= 0 || freqs <= 0 || frames <= 0) { set_error("fsnp_unfold: empty input [%d, %d, %d, %d]", batch, channels, freqs, frames); return 2; }
    if (num_neighbor >= freqs) { set_error("fsnp_unfold: num_neighbor %d needs more than %d frequency bins (reflect pad)", num_neighbor, freqs); return 2; }
    launch_unfold_stage(in, strides, out, batch, channels, freqs, frames, num_neighbor, static_cast<hipStream_t>(hip_stream));
    FSNP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
