// fsnp_stft_abi.hip - the STFT / iSTFT / waveform entry points of the C ABI: torch.stft / torch.istft of
// audio_zen/acoustics/feature.py:10-56 as DFT GEMMs (kernels: stft.hip, tcn.hip) and the inferencer's inner loop
// (fullsubnet_plus/inferencer/inferencer.py:142-158) as one call.  Host code only.
#include <vector>

#include "fsnp_handle.h"

// ---------------------------------------------------------------------------------------------- STFT / iSTFT (f-3)
namespace fsnp {
StftPlan stft_plan(const fsnp_handle* h) {
    StftPlan p{};
    p.F = h->F; p.n_fft = 2 * (h->F - 1); p.hop = p.n_fft / 2; p.N2 = 2 * p.F; p.sp = (int)align_up(p.N2, 4);
    p.inv_ld = (int)align_up(p.N2, 16);
    p.o_fwd = 0;
    p.o_inv = p.o_fwd + align_up(p.N2, 384) * (size_t)p.n_fft;
    p.o_win = p.o_inv + align_up(p.n_fft, 384) * (size_t)p.inv_ld;
    p.o_zero = p.o_win + align_up(p.n_fft, 64);
    p.total = p.o_zero + align_up(p.N2 > p.n_fft ? p.N2 : p.n_fft, 384);
    return p;
}
int ensure_stft(fsnp_handle* h) {
    if (h->d_stft) return 0;
    if (h->F < 3 || ((h->F - 1) & (h->F - 2)) != 0) { set_error("STFT: num_freqs - 1 must be a power of two (n_fft = 2 (num_freqs - 1))"); return 2; }
    const StftPlan p = stft_plan(h);
    std::vector<float> host(p.total, 0.0f);
    stft_build_matrices(p.n_fft, host.data() + p.o_fwd, host.data() + p.o_inv, host.data() + p.o_win);
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_stft), p.total * sizeof(float)));
    FSNP_HIP_CHECK(hipMemcpy(h->d_stft, host.data(), p.total * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
int ensure_io(fsnp_handle* h, size_t bytes, hipStream_t s) {      // stream-ordered, like ensure_workspace
    if (bytes <= h->io_bytes) return 0;
    if (order_after_last_forward(h, s)) return 4;
    unsigned char* nio = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&nio), bytes, s));
    if (h->io) FSNP_HIP_CHECK(hipFreeAsync(h->io, s));
    h->io = nio;
    h->io_bytes = bytes;
    return 0;
}
__global__ void spec_repack_kernel(const float2* __restrict__ in, long sb, long sf, long st, float2* __restrict__ out, int B,
                                   int F, int T, int spc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (b, t, f) over the padded row of spc complex elements
    if (i >= (long)B * T * spc) return;
    const int f = (int)(i % spc), t = (int)((i / spc) % T), b = (int)(i / ((long)spc * T));
    out[i] = f < F ? in[b * sb + f * sf + t * st] : make_float2(0.f, 0.f);
}
// wav [B][L] -> spectrum rows [B][T][ldc] (interleaved complex), T = 1 + L / hop
static void stft_into(fsnp_handle* h, const StftPlan& p, const float* wav, long wav_stride, float* xp, long xs, float* spec,
                      int ldc, int B, int L, hipStream_t s) {
    const int T = 1 + L / p.hop;
    launch_stft_pad(wav, wav_stride, xp, xs, B, L, p.n_fft, s);
    launch_linear_act(xp, p.hop, h->d_stft + p.o_fwd, p.n_fft, h->d_stft + p.o_zero, spec, ldc, p.n_fft, p.N2, B, T,
                      FSNP_ACT_NONE, h->num_cus, s, xs, p.n_fft);
}
// spectrum rows [B][T][sp] -> wav [B][L]
static void istft_from(fsnp_handle* h, const StftPlan& p, const float* spec, float* frames, float* wav, long wav_stride, int B,
                       int T, int L, hipStream_t s) {
    launch_linear_act(spec, p.sp, h->d_stft + p.o_inv, p.inv_ld, h->d_stft + p.o_zero, frames, p.n_fft, p.N2, p.n_fft, B, T,
                      FSNP_ACT_NONE, h->num_cus, s, (long)T * p.sp, p.N2);
    launch_istft_ola(frames, h->d_stft + p.o_win, wav, wav_stride, B, T, L, p.n_fft, s);
}
}  // namespace fsnp

extern "C" {

int fsnp_stft(fsnp_handle* h, const float* wav, int64_t wav_stride, float* spec, int32_t batch, int32_t samples, void* hip_stream) {
    if (!h || !wav || !spec) { set_error("fsnp_stft: null argument"); return 1; }
    if (batch <= 0) { set_error("fsnp_stft: empty input"); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if (samples <= p.hop) { set_error("fsnp_stft: need more than n_fft/2 = %d samples (reflect padding)", p.hop); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const long xs = (long)align_up((size_t)samples + p.n_fft, 4);
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, (size_t)batch * xs * 4, s)) return 4;
    stft_into(h, p, wav, wav_stride, reinterpret_cast<float*>(h->io), xs, spec, p.N2, batch, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

int fsnp_istft(fsnp_handle* h, const float* spec, const int64_t strides[3], float* wav, int64_t wav_stride, int32_t batch,
               int32_t frames, int32_t samples, void* hip_stream) {
    if (!h || !spec || !strides || !wav) { set_error("fsnp_istft: null argument"); return 1; }
    if (batch <= 0 || frames <= 0 || samples <= 0) { set_error("fsnp_istft: empty input"); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if ((long)(frames - 1) * p.hop + p.n_fft < (long)samples + p.hop) { set_error("fsnp_istft: %d frames do not cover %d samples", frames, samples); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const size_t spec_b = align_up((size_t)batch * frames * p.sp * 4 + 256, 256), fr_b = (size_t)batch * frames * p.n_fft * 4;
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, spec_b + fr_b, s)) return 4;
    float* sp = reinterpret_cast<float*>(h->io);
    float* fr = reinterpret_cast<float*>(h->io + spec_b);
    const long n = (long)batch * frames * (p.sp / 2);
    hipLaunchKernelGGL(spec_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2*>(spec),
                       (long)strides[0], (long)strides[1], (long)strides[2], reinterpret_cast<float2*>(sp), batch, p.F, frames, p.sp / 2);
    istft_from(h, p, sp, fr, wav, wav_stride, batch, frames, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

int fsnp_enhance_wave(fsnp_handle* h, const float* wav, int64_t wav_stride, float* out, int64_t out_stride, int32_t batch,
                      int32_t samples, void* hip_stream) {
    if (!h || !wav || !out) { set_error("fsnp_enhance_wave: null argument"); return 1; }
    if (batch <= 0) { set_error("fsnp_enhance_wave: empty input"); return 2; }
    if (!h->committed) { set_error("fsnp_enhance_wave: weights not committed (call fsnp_commit_weights)"); return 2; }
    // the cIRM epilogue multiplies by exactly two mask planes (decompress_cIRM, audio_zen/acoustics/mask.py:60-63) and the io area is
    // sized for them: a model built with another output_size has no waveform path
    if (h->cfg.output_size != 2) { set_error("fsnp_enhance_wave: the cIRM epilogue needs output_size = 2 (this handle: %d)", h->cfg.output_size); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if (samples <= p.hop) { set_error("fsnp_enhance_wave: need more than n_fft/2 = %d samples (reflect padding)", p.hop); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const int T = 1 + samples / p.hop;
    const long xs = (long)align_up((size_t)samples + p.n_fft, 4);
    const size_t xp_b = align_up((size_t)batch * xs * 4, 256), spec_b = align_up((size_t)batch * T * p.sp * 4 + 256, 256);
    const size_t mask_b = align_up((size_t)batch * 2 * p.F * T * 4, 256), fr_b = (size_t)batch * T * p.n_fft * 4;
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, xp_b + 2 * spec_b + mask_b + fr_b, s)) return 4;
    float* xp = reinterpret_cast<float*>(h->io);
    float* noisy = reinterpret_cast<float*>(h->io + xp_b);
    float* enh = reinterpret_cast<float*>(h->io + xp_b + spec_b);
    float* mask = reinterpret_cast<float*>(h->io + xp_b + 2 * spec_b);
    float* fr = reinterpret_cast<float*>(h->io + xp_b + 2 * spec_b + mask_b);
    // inferencer.py:142-158: stft -> (mag, real, imag) -> model -> decompress_cIRM, complex multiply -> istft(length)
    stft_into(h, p, wav, wav_stride, xp, xs, noisy, p.sp, batch, samples, s);
    const int64_t cst[3] = {(int64_t)T * (p.sp / 2), 1, p.sp / 2};       // complex-element strides of [B][T][sp/2] as (b, f, t)
    const int rc = fsnp_forward_complex(h, noisy, cst, mask, batch, T, FSNP_MODE_FULL, 0, batch, hip_stream);
    if (rc) return rc;
    // pipelined mode: the forward left chunks of the sub-band plan on the side stream (at B = 1 the whole plan); `mask` is read
    // right here and lives in the single-buffered io area, so `s` waits for them now (fsnp_flush) - nothing of this call is deferred
    if (h->pipeline && fsnp_flush(h, hip_stream)) return 4;
    // the pad column of every row of `enh` is never written by apply_cirm and multiplies zero weights: clear it once
    FSNP_HIP_CHECK(hipMemsetAsync(enh, 0, spec_b, s));
    launch_apply_cirm(mask, noisy, cst, enh, cst, batch, p.F, T, s);
    istft_from(h, p, enh, fr, out, out_stride, batch, T, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

}  // extern "C"
