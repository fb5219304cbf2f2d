// stft.hip - SURVEY.md 8(f-3), second half: the STFT / iSTFT around the model, so that the reference inferencer's
// whole inner loop (speech_enhance/fullsubnet_plus/inferencer/inferencer.py:142-158: torch_stft -> model -> cIRM ->
// torch_istft) is ONE C-ABI call on device buffers (fsnp_enhance_wave).
//
// Replaces audio_zen/acoustics/feature.py:10-31 (torch.stft(y, n_fft, hop, win, window=hann, return_complex=True), i.e.
// center=True, pad_mode="reflect", onesided, not normalised) and :34-56 (torch.istft(..., length=L)).
// With n_fft = 512 a DFT is a [frames x 512] x [512 x 514] fp32 GEMM - 66 MFLOP per 2 s clip, nothing next to the model -
// so both transforms run on the MFMA GEMM of tcn.hip with the (periodic hann) window folded into the DFT matrices:
//   STFT : xp = reflect-pad(wav, n_fft/2);  X[t][2f + {0,1}] = sum_n xp[t*hop + n] * w[n] * {cos, -sin}(2 pi f n / N)
//          (the A operand is the padded signal itself read with row stride hop: frames overlap in memory, no copy);
//          the output IS torch.stft's memory layout ([B][T][F] complex64).
//   iSTFT: fr[t][n] = w[n]/N * sum_f c_f (Re X cos - Im X sin),  c_0 = c_{N/2} = 1, else 2 (C2R semantics);
//          wav[i] = (sum_t fr[t][i + N/2 - t*hop]) / (sum_t w^2[i + N/2 - t*hop])   (torch.istft's envelope division).
#include <cmath>

#include "fsnp_common.h"

namespace fsnp {

__global__ __launch_bounds__(256) void stft_pad_kernel(const float* __restrict__ wav, long wav_stride, float* __restrict__ xp,
                                                       long xp_stride, int L, int half, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // index into [B][L + 2 half (+ tail)]
    if (i >= total) return;
    const int b = (int)(i / xp_stride);
    const int p = (int)(i - (long)b * xp_stride);
    float v = 0.0f;
    if (p < L + 2 * half) {
        int j = p - half;                                      // torch "reflect": no edge repeat
        if (j < 0) j = -j;
        if (j >= L) j = 2 * (L - 1) - j;
        v = wav[(long)b * wav_stride + j];
    }
    xp[i] = v;
}

void launch_stft_pad(const float* wav, long wav_stride, float* xp, long xp_stride, int B, int L, int n_fft, hipStream_t s) {
    const long total = (long)B * xp_stride;
    hipLaunchKernelGGL(stft_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wav, wav_stride, xp, xp_stride,
                       L, n_fft / 2, total);
}

// hop = n_fft / 2: every output sample is covered by at most two frames
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ wav, long wav_stride, int T, int L, int n_fft,
                                                        long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (b, sample)
    if (i >= total) return;
    const int b = (int)(i / L), n = (int)(i % L);
    const int hop = n_fft / 2;
    const int p = n + hop;                                     // position in the centre-padded signal
    const int t1 = p / hop, t0 = t1 - 1;
    float num = 0.0f, den = 0.0f;
    if (t1 < T) {
        const int k = p - t1 * hop;
        num += frames[((long)b * T + t1) * n_fft + k];
        den += window[k] * window[k];
    }
    if (t0 >= 0 && t0 < T) {
        const int k = p - t0 * hop;
        num += frames[((long)b * T + t0) * n_fft + k];
        den += window[k] * window[k];
    }
    wav[(long)b * wav_stride + n] = den > 1e-11f ? num / den : 0.0f;
}

void launch_istft_ola(const float* frames, const float* window, float* wav, long wav_stride, int B, int T, int L, int n_fft,
                      hipStream_t s) {
    const long total = (long)B * L;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, window, wav,
                       wav_stride, T, L, n_fft, total);
}

// Host: GEMM operands (zero padded as tcn_gemm_kernel expects: rows to a multiple of 384, K to a multiple of 16).
//   fwd [2F pad 384][n_fft]      : row 2f = w[n] cos(2 pi f n / N), row 2f+1 = -w[n] sin(2 pi f n / N)
//   inv [n_fft pad 384][2F pad 16]: row n, column 2f = c_f w[n] cos(2 pi f n / N) / N, column 2f+1 = -c_f w[n] sin(...) / N
void stft_build_matrices(int n_fft, float* fwd, float* inv, float* window) {
    const int F = n_fft / 2 + 1, N2 = 2 * F;
    const int kp = (N2 + 15) / 16 * 16;
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < n_fft; ++n) window[n] = (float)(0.5 - 0.5 * cos(two_pi * n / n_fft));   // torch.hann_window (periodic)
    for (int f = 0; f < F; ++f)
        for (int n = 0; n < n_fft; ++n) {
            const long ph = ((long)f * n) % n_fft;             // exact phase reduction
            const double c = cos(two_pi * ph / n_fft), sn = sin(two_pi * ph / n_fft);
            const double w = 0.5 - 0.5 * cos(two_pi * n / n_fft);
            fwd[(size_t)(2 * f) * n_fft + n] = (float)(w * c);
            fwd[(size_t)(2 * f + 1) * n_fft + n] = (float)(-w * sn);
            const double cf = (f == 0 || f == n_fft / 2) ? 1.0 : 2.0;
            inv[(size_t)n * kp + 2 * f] = (float)(cf * w * c / n_fft);
            inv[(size_t)n * kp + 2 * f + 1] = (float)(-cf * w * sn / n_fft);
        }
}

}  // namespace fsnp
