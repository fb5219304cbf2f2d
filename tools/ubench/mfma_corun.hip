// mfma_corun.hip - does work issued by ANOTHER wave on the same SIMD slow a pure-MFMA wave down?
// 8 waves per workgroup (2 per SIMD): waves 0-3 run a pure v_mfma_f32_32x32x2_f32 stream and time it; waves 4-7
// (one per SIMD, co-resident) run pattern P for the whole duration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
constexpr int ITERS = 256;

template <int P>
__global__ __launch_bounds__(512) void kern(const float* __restrict__ gin, float* __restrict__ gout,
                                            unsigned long long* __restrict__ ticks, volatile int* flag) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    __shared__ int done;
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = gin[i];
    if (tid == 0) done = 0;
    __syncthreads();
    float a = gin[tid], b = gin[tid + 512];
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[12];
#pragma unroll
        for (int n = 0; n < 12; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int n = 0; n < 12; ++n) { MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB(); }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int n = 0; n < 12; ++n) s += acc[n][0];
        if ((tid & 63) == 0) { ticks[blockIdx.x * 4 + wave] = t1 - t0; atomicAdd(&done, 1); }
    } else {
        float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
        const float4* lp = reinterpret_cast<const float4*>(lds) + (tid & 63);
        const float4* gp = reinterpret_cast<const float4*>(gin) + tid;
        float4 q = make_float4(0, 0, 0, 0);
        int it = 0;
        while (*(volatile int*)&done < 4) {
            if constexpr (P == 1) {        // dense independent v_fmac
#pragma unroll
                for (int k = 0; k < 64; ++k) { v0 = fmaf(a, b, v0); v1 = fmaf(a, b, v1); v2 = fmaf(a, b, v2); v3 = fmaf(a, b, v3); }
            } else if constexpr (P == 2) { // ds_read_b128 stream
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float4 t = lp[(it + k) & 63]; q.x += t.x; q.y += t.w; }
            } else if constexpr (P == 3) { // global_load_dwordx4 stream (L2 resident)
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float4 t = gp[((it + k) & 15) * 512]; q.x += t.x; q.y += t.w; }
            } else if constexpr (P == 4) { // sparse VALU: 4 fmac then s_sleep
                v0 = fmaf(a, b, v0); v1 = fmaf(a, b, v1); v2 = fmaf(a, b, v2); v3 = fmaf(a, b, v3);
                __builtin_amdgcn_s_sleep(1);
            } else {                       // P == 0: idle partner
                __builtin_amdgcn_s_sleep(8);
            }
            ++it;
        }
        s = v0 + v1 + v2 + v3 + q.x + q.y + (float)it;
        if (P != 0 && (tid & 63) == 0) ticks[4096 + blockIdx.x * 4 + (wave - 4)] = (unsigned long long)it;
    }
    gout[blockIdx.x * 512 + tid] = s;
}

template <int P> void run(const char* name, const float* din, float* dout, unsigned long long* dt, int blocks, int per_iter_ops) {
    hipMemset(dt, 0, 8192 * 8);
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(512), 0, 0, din, dout, dt, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(8192);
    hipMemcpy(t.data(), dt, 8192 * 8, hipMemcpyDeviceToHost);
    double avg = 0, its = 0; for (int i = 0; i < blocks * 4; ++i) { avg += (double)t[i]; its += (double)t[4096 + i]; }
    avg /= blocks * 4; its /= blocks * 4;
    const double per = avg / ITERS;
    printf("%-44s blocks=%3d mfma-wave cycles/48mfma=%8.1f (%+5.1f%%)  partner iters=%9.0f -> partner ops per 48-mfma window=%.1f\n",
           name, blocks, per, 100.0 * (per / 3072 - 1), its, its * per_iter_ops / ITERS);
}

int main() {
    float *din, *dout; unsigned long long* dt;
    hipMalloc(&din, 1 << 22); hipMalloc(&dout, 1 << 22); hipMalloc(&dt, 8192 * 8);
    std::vector<float> h(1 << 20, 0.001f);
    hipMemcpy(din, h.data(), 1 << 22, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        run<0>("partner idle (s_sleep)", din, dout, dt, blocks, 0);
        run<1>("partner: dense v_fmac (256/iter)", din, dout, dt, blocks, 256);
        run<4>("partner: 4 v_fmac + s_sleep", din, dout, dt, blocks, 4);
        run<2>("partner: ds_read_b128 x16/iter", din, dout, dt, blocks, 16);
        run<3>("partner: global_load_dwordx4 x16/iter", din, dout, dt, blocks, 16);
    }
    return 0;
}
