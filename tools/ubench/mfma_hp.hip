// mfma_hp.hip - micro-benchmark for csrc/lstm_hp.hip: does its MFMA pass pattern (v_mfma_f32_16x16x4_f32 on three accumulators
// round robin, B operands from resident AGPR / VGPR weights, A operands from ds_read_b128 one group ahead) issue at 32 cycles
// per MFMA?  One wave per SIMD, 256 workgroups.  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_hp.hip -o mfma_hp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
constexpr int ITERS = 64, G = 24;

template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void kern(const float* __restrict__ gin, float* __restrict__ gout, unsigned long long* __restrict__ ticks) {
    __shared__ __attribute__((aligned(16))) float4 lds[2 * G * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * G * 64; i += 256) lds[i] = reinterpret_cast<const float4*>(gin)[i & 1023];
    __syncthreads();
    float4 wa[2 * G], wv[G];          // "layer 1" weights (AGPR in P >= 1), "layer 0" weights (VGPR)
#pragma unroll
    for (int i = 0; i < 2 * G; ++i) wa[i] = reinterpret_cast<const float4*>(gin)[(tid + i * 64) & 4095];
#pragma unroll
    for (int i = 0; i < G; ++i) wv[i] = reinterpret_cast<const float4*>(gin)[(tid + i * 64 + 7) & 4095];
    if constexpr (P >= 1) {
#pragma unroll
        for (int i = 0; i < 2 * G; ++i) asm volatile("" : "+a"(wa[i].x), "+a"(wa[i].y), "+a"(wa[i].z), "+a"(wa[i].w));
    }
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    const float4* P1 = lds + lane;
    const float4* P0 = lds + G * 64 + lane;
    float4 pc = gin[tid] > 5.f ? P1[0] : make_float4(gin[tid], gin[tid + 1], gin[tid + 2], gin[tid + 3]), qc = pc;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        float4 p = pc, q = qc;
        if constexpr (P >= 3) { p = P1[0]; q = P0[0]; }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 pn = p, qn = q;
            if constexpr (P >= 2) { pn = P1[(g + 1 < G ? g + 1 : g) * 64]; qn = P0[(g + 1 < G ? g + 1 : g) * 64]; SB(); }
            MF(a1, p.x, wa[g].x); MF(a0, q.x, wv[g].x); MF(a2, q.x, wa[G + g].x);
            MF(a1, p.y, wa[g].y); MF(a0, q.y, wv[g].y); MF(a2, q.y, wa[G + g].y);
            MF(a1, p.z, wa[g].z); MF(a0, q.z, wv[g].z); MF(a2, q.z, wa[G + g].z);
            MF(a1, p.w, wa[g].w); MF(a0, q.w, wv[g].w); MF(a2, q.w, wa[G + g].w);
            SB();
            if constexpr (P >= 3) { p = pn; q = qn; }
            else if constexpr (P == 2) { pc.x += pn.x * 1e-30f; qc.x += qn.x * 1e-30f; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    gout[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + pc.x + qc.x;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int P> void run(const char* name, const float* din, float* dout, unsigned long long* dt) {
    const int blocks = 256;
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> t(blocks);
    (void)hipMemcpy(t.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : t) avg += (double)v; avg /= blocks;
    const double per = avg / ITERS / (12.0 * G);
    printf("%-100s ticks per MFMA = %6.2f (ideal 32 matrix-pipe cycles)\n", name, per);
}

int main() {
    float *din, *dout; unsigned long long* dt;
    (void)hipMalloc(&din, 1 << 20); (void)hipMalloc(&dout, 1 << 20); (void)hipMalloc(&dt, 4096 * 8);
    std::vector<float> h(1 << 18, 0.001f);
    (void)hipMemcpy(din, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<0>("0  288 MFMA 16x16x4, three accumulators round robin, A and B from VGPRs", din, dout, dt);
    run<1>("1  the same with two thirds of the B operands from AGPRs (resident layer-1 weights)", din, dout, dt);
    run<2>("2  1 + two ds_read_b128 per 12 MFMAs (results unused by the MFMAs)", din, dout, dt);
    run<3>("3  1 + two ds_read_b128 per 12 MFMAs, one group ahead, feeding the MFMAs (the kernel's loop)", din, dout, dt);
    return 0;
}
