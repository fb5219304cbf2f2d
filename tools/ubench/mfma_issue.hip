// mfma_issue.hip - micro-benchmark: what does v_mfma_f32_32x32x2_f32 issue cost on gfx950 with ONE wave per
// SIMD, as a function of the accumulator-chain pattern and of the instructions placed between MFMAs?
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_issue.hip -o mfma_issue ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define MF16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)

constexpr int ITERS = 256;

template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void kern(const float* __restrict__ gin, float* __restrict__ gout, unsigned long long* __restrict__ ticks) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = gin[i];
    __syncthreads();
    f32x16 acc[12];
    f32x4 acc4[24];
#pragma unroll
    for (int n = 0; n < 12; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int n = 0; n < 24; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[n][r] = 0.f;
    float a = gin[tid], b = gin[tid + 256];
    float4 la = reinterpret_cast<const float4*>(lds)[tid];
    float4 gb = reinterpret_cast<const float4*>(gin)[tid];
    float vx[4] = {0.f, 0.f, 0.f, 0.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk0 = {0.f, 0.f}, pa = {gin[tid], gin[tid + 1]}, pb = {gin[tid + 2], gin[tid + 3]};
    f32x4 m4[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) m4[n][r] = 0.f;
    const float4* lp = reinterpret_cast<const float4*>(lds) + (tid & 63);
    const float4* gp = reinterpret_cast<const float4*>(gin) + tid;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gin), 0, 1 << 20, 0x00020000);
    const int voff = tid * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (P == 0) {          // A: 12 accumulators x chains of 4 (the LSTM kernel's pattern), nothing else
#pragma unroll
            for (int n = 0; n < 12; ++n) { MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB(); }
        } else if constexpr (P == 1) {   // B: one accumulator, chain of 48
#pragma unroll
            for (int n = 0; n < 48; ++n) { MF(acc[0], a, b); SB(); }
        } else if constexpr (P == 2) {   // C: 12 accumulators round robin (no back-to-back dependence)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 12; ++n) { MF(acc[n], a, b); SB(); }
        } else if constexpr (P == 3) {   // D: A + one ds_read_b128 after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                la = lp[(it + n) & 63]; SB();
            }
            a += la.x * 1e-30f;
        } else if constexpr (P == 4) {   // E: A + one global_load_dwordx4 after each chain (consumed next iteration)
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                gb = gp[((it + n) & 7) * 256]; SB();
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 5) {   // F: A + 4 dependent v_fmac after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); SB();
            }
        } else if constexpr (P == 6) {   // G: 4 v_fmac after the FIRST MFMA of each chain (inside the chain)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); SB();
                vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); vx[0] = fmaf(a, b, vx[0]); SB();
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
            }
        } else if constexpr (P == 7) {   // H: chain of 48 with one ds_read_b128 every 4 MFMAs (inside the chain)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[0], a, b); MF(acc[0], a, b); MF(acc[0], a, b); MF(acc[0], a, b); SB();
                la = lp[(it + n) & 63]; SB();
            }
            a += la.x * 1e-30f;
        } else if constexpr (P == 8) {   // I: 6 accumulators x chains of 8
#pragma unroll
            for (int n = 0; n < 6; ++n) {
#pragma unroll
                for (int k = 0; k < 8; ++k) MF(acc[n], a, b);
                SB();
            }
        } else if constexpr (P == 9) {   // J: 4 independent v_fmac (different registers) after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                vx[0] = fmaf(a, b, vx[0]); vx[1] = fmaf(a, b, vx[1]); vx[2] = fmaf(a, b, vx[2]); vx[3] = fmaf(a, b, vx[3]); SB();
            }
        } else if constexpr (P == 10) {  // K: 16x16x4 : 24 accumulators x chains of 4 (same FLOPs as A: 96 MFMAs)
#pragma unroll
            for (int n = 0; n < 24; ++n) { MF16(acc4[n], a, b); MF16(acc4[n], a, b); MF16(acc4[n], a, b); MF16(acc4[n], a, b); SB(); }
        } else if constexpr (P == 11) {  // L: 16x16x4 : two interleaved chains of 48 (96 MFMAs)
#pragma unroll
            for (int n = 0; n < 48; ++n) { MF16(acc4[0], a, b); MF16(acc4[1], a, b); SB(); }
        } else if constexpr (P == 12) {  // M: 16x16x4 round robin over 24 accumulators (96 MFMAs)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 24; ++n) { MF16(acc4[n], a, b); SB(); }
        } else if constexpr (P == 13) {  // N: A + (s_waitcnt-free) global load + ds_read + 4 fmac after each chain
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                gb = gp[((it + n) & 7) * 256];
                vx[0] = fmaf(a, b, vx[0]); vx[1] = fmaf(a, b, vx[1]); vx[2] = fmaf(a, b, vx[2]); vx[3] = fmaf(a, b, vx[3]); SB();
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 14) {  // O: 2 accumulators alternating every MFMA (A/B ping-pong chains of 24 each)
#pragma unroll
            for (int n = 0; n < 24; ++n) { MF(acc[0], a, b); MF(acc[1], a, b); SB(); }
        } else if constexpr (P == 16) {  // R: A + raw_buffer_load_b128 (SRD in SGPRs, constant voffset, scalar soffset)
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((it + n) & 7) * 4096, 0));
                gb = make_float4(t[0], t[1], t[2], t[3]); SB();
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 17) {  // S: A + global_load via uniform base pointer + lane offset (saddr form hoped for)
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                const float4* base = reinterpret_cast<const float4*>(gin) + (((it + n) & 7) * 256);   // uniform
                gb = base[tid]; SB();
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 18) {  // T: A + 2 x global_load_dwordx2 after each chain (same bytes as E)
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                const float2* q = reinterpret_cast<const float2*>(gp + ((it + n) & 7) * 256);
                const float2 u0 = q[0]; SB(); const float2 u1 = q[1]; SB();
                gb = make_float4(u0.x, u0.y, u1.x, u1.y);
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 19) {  // U: A + one global_load_dword (4 B/lane) after each chain
            float prevf = gb.x;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                gb.x = reinterpret_cast<const float*>(gp)[((it + n) & 7) * 256]; SB();
            }
            b += prevf * 1e-30f;
        } else if constexpr (P == 20) {  // V: A + raw_buffer_load_b128 ... lds?  no: plain buffer load with voffset varying per n
            const float4 prev = gb;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ((it + n) & 7) * 4096, 0, 0));
                gb = make_float4(t[0], t[1], t[2], t[3]); SB();
            }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 21) {  // W: A + ds_read_b64 after each chain
            float2 l2;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                l2 = reinterpret_cast<const float2*>(lp)[(it + n) & 63]; SB();
            }
            a += l2.x * 1e-30f;
        } else if constexpr (P == 22) {  // X: A + ds_read_b32 after each chain
            float l1;
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                l1 = reinterpret_cast<const float*>(lp)[(it + n) & 63]; SB();
            }
            a += l1 * 1e-30f;
        } else if constexpr (P == 23) {  // 16x16x4 round robin over 24 acc + 1 independent v_fmac after EVERY mfma
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 24; ++n) { MF16(acc4[n], a, b); SB(); vx[n & 3] = fmaf(a, b, vx[n & 3]); SB(); }
        } else if constexpr (P == 24) {  // 16x16x4 round robin + 1 v_fmac after every 2nd mfma
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 24; ++n) { MF16(acc4[n], a, b); SB(); if (n & 1) { vx[n & 3] = fmaf(a, b, vx[n & 3]); SB(); } }
        } else if constexpr (P == 25) {  // 16x16x4 round robin + one raw_buffer_load_b128 per 8 mfma (same bytes/flop as LSTM)
            const float4 prev = gb;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 24; ++n) {
                    MF16(acc4[n], a, b); SB();
                    if ((n & 7) == 7) {
                        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((it + n + p) & 7) * 4096, 0));
                        gb = make_float4(t[0], t[1], t[2], t[3]); SB();
                    }
                }
            b += prev.x * 1e-30f;
        } else if constexpr (P == 26) {  // 32x32x2: 12 acc round robin + 1 v_fmac after every mfma
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < 12; ++n) { MF(acc[n], a, b); SB(); vx[n & 3] = fmaf(a, b, vx[n & 3]); SB(); }
        } else if constexpr (P == 27) {  // 16x16x4: 2 interleaved chains + 1 v_fmac after every pair
#pragma unroll
            for (int n = 0; n < 48; ++n) { MF16(acc4[0], a, b); MF16(acc4[1], a, b); SB(); vx[n & 3] = fmaf(a, b, vx[n & 3]); SB(); }
        } else if constexpr (P == 28) {  // A + 2 v_pk_fma_f32 after each chain (same FLOPs as 4 v_fmac)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk0) : "v"(pa), "v"(pb));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk0) : "v"(pa), "v"(pb)); SB();
            }
        } else if constexpr (P == 29) {  // A + 4 x v_mfma_f32_4x4x1_16b_f32 after each chain (4 extra rows x 64 (k,col))
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                m4[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[n & 3], 0, 0, 0);
                m4[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[n & 3], 0, 0, 0);
                m4[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[n & 3], 0, 0, 0);
                m4[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[n & 3], 0, 0, 0); SB();
            }
        } else if constexpr (P == 30) {  // A + 4 x 4x4x1 on 4 DIFFERENT accumulators after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                m4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[0], 0, 0, 0);
                m4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[1], 0, 0, 0);
                m4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[2], 0, 0, 0);
                m4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[3], 0, 0, 0); SB();
            }
        } else if constexpr (P == 31) {  // A + 1 x 4x4x1 after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                m4[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[n & 3], 0, 0, 0); SB();
            }
        } else if constexpr (P == 32) {  // 48 v_fmac in ONE batch after the 48 MFMAs
#pragma unroll
            for (int n = 0; n < 12; ++n) { MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB(); }
#pragma unroll
            for (int n = 0; n < 48; ++n) vx[n & 3] = fmaf(a, b, vx[n & 3]);
            SB();
        } else if constexpr (P == 33) {  // 6 chains, then 12 v_pk_fma in one batch, twice per iteration (= 48 fma)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int n = 0; n < 6; ++n) { MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); }
                SB();
#pragma unroll
                for (int n = 0; n < 12; ++n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk0) : "v"(pa), "v"(pb));
                SB();
            }
        } else if constexpr (P == 34) {  // 6 chains, then 24 v_fmac in one batch, twice per iteration
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int n = 0; n < 6; ++n) { MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); MF(acc[hb * 6 + n], a, b); }
                SB();
#pragma unroll
                for (int n = 0; n < 24; ++n) vx[n & 3] = fmaf(a, b, vx[n & 3]);
                SB();
            }
        } else if constexpr (P == 15) {  // Q: 12 accumulators x chains of 4, with 1 s_nop after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB();
                asm volatile("s_nop 0"); SB();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a + b + la.y + gb.y + vx[0] + vx[1] + vx[2] + vx[3] + pk0[0] + pk0[1] + m4[0][0] + m4[1][1] + m4[2][2] + m4[3][3];
#pragma unroll
    for (int n = 0; n < 12; ++n) s += acc[n][0] + acc[n][7];
#pragma unroll
    for (int n = 0; n < 24; ++n) s += acc4[n][0];
    gout[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int P> void run(const char* name, int mfmas, int cyc_each, const float* din, float* dout, unsigned long long* dt, int blocks) {
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(blocks);
    hipMemcpy(t.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : t) avg += (double)v; avg /= blocks;
    const double per_iter = avg / ITERS;
    printf("%-72s blocks=%3d cycles/iter=%8.1f  ideal=%6d  overhead=%+7.1f (%+5.1f%%)\n", name, blocks, per_iter,
           mfmas * cyc_each, per_iter - mfmas * cyc_each, 100.0 * (per_iter / (mfmas * cyc_each) - 1));
}

int main() {
    float *din, *dout; unsigned long long* dt;
    hipMalloc(&din, 1 << 20); hipMalloc(&dout, 1 << 20); hipMalloc(&dt, 4096 * 8);
    std::vector<float> h(1 << 18, 0.001f);
    hipMemcpy(din, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int blocks : {256}) {
        run<0>("A  12 acc x chain4 (LSTM pattern), bare", 48, 64, din, dout, dt, blocks);
        run<1>("B  1 acc x chain48", 48, 64, din, dout, dt, blocks);
        run<2>("C  12 acc round-robin (no dependent back-to-back)", 48, 64, din, dout, dt, blocks);
        run<8>("I  6 acc x chain8", 48, 64, din, dout, dt, blocks);
        run<14>("O  2 acc ping-pong", 48, 64, din, dout, dt, blocks);
        run<15>("Q  A + s_nop after each chain", 48, 64, din, dout, dt, blocks);
        run<3>("D  A + ds_read_b128 after each chain", 48, 64, din, dout, dt, blocks);
        run<4>("E  A + global_load_dwordx4 after each chain", 48, 64, din, dout, dt, blocks);
        run<5>("F  A + 4 dependent v_fmac after each chain", 48, 64, din, dout, dt, blocks);
        run<9>("J  A + 4 independent v_fmac after each chain", 48, 64, din, dout, dt, blocks);
        run<6>("G  4 v_fmac after the FIRST mfma of each chain", 48, 64, din, dout, dt, blocks);
        run<7>("H  chain48 + ds_read_b128 every 4 mfma (inside chain)", 48, 64, din, dout, dt, blocks);
        run<16>("R  A + raw_buffer_load_b128 (scalar soffset) after each chain", 48, 64, din, dout, dt, blocks);
        run<20>("V  A + raw_buffer_load_b128 (vector offset) after each chain", 48, 64, din, dout, dt, blocks);
        run<17>("S  A + global_load x4 from uniform base + tid", 48, 64, din, dout, dt, blocks);
        run<18>("T  A + 2 x global_load_dwordx2", 48, 64, din, dout, dt, blocks);
        run<19>("U  A + global_load_dword (4 B/lane)", 48, 64, din, dout, dt, blocks);
        run<21>("W  A + ds_read_b64", 48, 64, din, dout, dt, blocks);
        run<22>("X  A + ds_read_b32", 48, 64, din, dout, dt, blocks);
        run<26>("Y  32x32x2 round-robin + 1 v_fmac after every mfma (48 fmac)", 48, 64, din, dout, dt, blocks);
        run<23>("Z1 16x16x4 round-robin + 1 v_fmac after every mfma (96 fmac)", 96, 32, din, dout, dt, blocks);
        run<24>("Z2 16x16x4 round-robin + 1 v_fmac after every 2nd mfma (48 fmac)", 96, 32, din, dout, dt, blocks);
        run<27>("Z3 16x16x4 2 chains + 1 v_fmac per pair (48 fmac)", 96, 32, din, dout, dt, blocks);
        run<25>("Z4 16x16x4 round-robin + 12 buffer_load_b128 per 96 mfma", 96, 32, din, dout, dt, blocks);
        run<28>("P1 A + 2 v_pk_fma_f32 after each chain", 48, 64, din, dout, dt, blocks);
        run<29>("P2 A + 4 x mfma_4x4x1 (same acc) after each chain", 48, 64, din, dout, dt, blocks);
        run<30>("P3 A + 4 x mfma_4x4x1 (4 acc) after each chain", 48, 64, din, dout, dt, blocks);
        run<31>("P4 A + 1 x mfma_4x4x1 after each chain", 48, 64, din, dout, dt, blocks);
        run<32>("P5 A + 48 v_fmac in one batch per 48 mfma", 48, 64, din, dout, dt, blocks);
        run<33>("P6 half-group: 6 chains + 12 v_pk_fma batch (x2)", 48, 64, din, dout, dt, blocks);
        run<34>("P7 half-group: 6 chains + 24 v_fmac batch (x2)", 48, 64, din, dout, dt, blocks);
        run<13>("N  A + global_load + 4 indep v_fmac after each chain", 48, 64, din, dout, dt, blocks);
        run<10>("K  16x16x4: 24 acc x chain4", 96, 32, din, dout, dt, blocks);
        run<11>("L  16x16x4: 2 interleaved chains", 96, 32, din, dout, dt, blocks);
        run<12>("M  16x16x4: 24 acc round-robin", 96, 32, din, dout, dt, blocks);
    }
    return 0;
}
