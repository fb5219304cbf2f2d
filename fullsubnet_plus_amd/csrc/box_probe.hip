// box_probe.hip - what THIS box's matrix pipes sustain: a calibration probe for the bench line (include/fsnp_debug.h).
//
// The dominant kernel of the path (lstm2_fc_kernel, csrc/lstm.hip) is bound by the issue rate of v_mfma_f32_32x32x2_f32: 64 cycles
// per instruction and SIMD, 4096 FLOP each, 1024 SIMDs -> 65,536 FLOP per shader cycle = 157.3 TFLOP/s at the 2.4 GHz of the data
// sheet.  The clock a box actually HOLDS under that load depends on its power cap / DVFS state; the round-5 driver box ran the
// unchanged kernel 4 % slower than every other box and nothing in the bench line could say why.  This probe runs nothing but that
// instruction on every SIMD of the chip (one wave per SIMD, twelve independent accumulators round robin - pattern C of
// tools/ubench/mfma_issue.hip, which issues at 64.08 cycles per MFMA) on non-trivial operands for a caller-chosen time and reports
//   * the fp32 MFMA rate reached (hipEvents around the launch)                          -> box.mfma_peak_tflops
//   * the shader clock that rate implies (64 cycles per MFMA) per workgroup: mean / min / max over the chip (wall time of each
//     workgroup from s_memrealtime, the 100 MHz constant clock)
//   * s_memtime ticks per s_memrealtime tick (x 100 MHz = the rate of the s_memtime counter)
// No handle, no weights: it can run before and after the timed region of bench.py.
#include <cstring>
#include <vector>

#include "fsnp_common.h"
#include "../../include/fsnp_debug.h"

namespace fsnp {
void set_error(const char* fmt, ...);

using f32x16p = __attribute__((ext_vector_type(16))) float;

constexpr int kProbeChain = 12;          // independent accumulators (192 registers)
constexpr int kProbeRounds = 4;          // MFMAs per accumulator and iteration: 48 MFMAs = 3072 matrix-pipe cycles per iteration

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mfma_probe_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
    extern __shared__ unsigned char probe_lds[];       // claimed, not used: keeps a second workgroup off the CU
    const int tid = threadIdx.x;
    float a = in[tid], b = in[256 + tid];
    f32x16p acc[kProbeChain];
#pragma unroll
    for (int n = 0; n < kProbeChain; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = in[(n * 16 + r + tid) & 1023];
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < kProbeRounds; ++p)
#pragma unroll
            for (int n = 0; n < kProbeChain; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if (tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x * 2 + 0] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
    float s = 0.0f;
#pragma unroll
    for (int n = 0; n < kProbeChain; ++n) s += acc[n][0] + acc[n][5] + acc[n][15];
    out[blockIdx.x * 256 + tid] = s;
}

}  // namespace fsnp

using namespace fsnp;

extern "C" int fsnp_debug_box_probe(double target_ms, double out[FSNP_BOX_PROBE_VALUES], void* hip_stream) {
    if (!out || !(target_ms > 0.0) || target_ms > 2000.0) { set_error("fsnp_debug_box_probe: bad argument (0 < target_ms <= 2000)"); return 1; }
    for (int i = 0; i < FSNP_BOX_PROBE_VALUES; ++i) out[i] = 0.0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("fsnp_debug_box_probe: no HIP device"); return 3; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_error("fsnp_debug_box_probe: device %d is %s, not gfx950", dev, prop.gcnArchName); return 3; }
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    float* buf = nullptr;
    unsigned long long* stamps = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto fail = [&](const char* what) { set_error("fsnp_debug_box_probe: %s failed", what); rc = 4; };
    const size_t in_floats = 1024, out_floats = (size_t)cus * 256;
    if (hipMalloc(reinterpret_cast<void**>(&buf), (in_floats + out_floats) * 4) != hipSuccess) { fail("hipMalloc"); return rc; }
    if (hipMalloc(reinterpret_cast<void**>(&stamps), (size_t)cus * 16) != hipSuccess) { (void)hipFree(buf); fail("hipMalloc"); return rc; }
    std::vector<float> host(in_floats);
    unsigned lcg = 12345u;
    for (auto& v : host) { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 8) & 0xFFFF) / 65536.0f - 0.5f; }   // operands in [-0.5, 0.5)
    constexpr int lds_claim = 96 * 1024;      // more than half of a CU's 160 KB: one workgroup per CU
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_claim); });
    do {
        if (hipMemcpyAsync(buf, host.data(), in_floats * 4, hipMemcpyHostToDevice, s) != hipSuccess) { fail("hipMemcpyAsync"); break; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { fail("hipEventCreate"); break; }
        // iterations for the asked time at the data-sheet clock; a first short launch pages the code in and wakes the clocks
        const double mfma_per_ms = 2.4e6 / 64.0;
        const int iters = (int)(target_ms * mfma_per_ms / (kProbeChain * kProbeRounds)) + 1;
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(256), lds_claim, s, buf, buf + in_floats, stamps, iters / 16 + 1);
        if (hipEventRecord(e0, s) != hipSuccess) { fail("hipEventRecord"); break; }
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(256), lds_claim, s, buf, buf + in_floats, stamps, iters);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { fail("probe launch"); break; }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.0f)) { fail("hipEventElapsedTime"); break; }
        std::vector<unsigned long long> st((size_t)cus * 2);
        if (hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { fail("hipMemcpy"); break; }
        const double mfmas = (double)iters * kProbeChain * kProbeRounds;          // per wave
        const double flops = mfmas * 4096.0 * 4.0 * cus;                           // 32 x 32 x 2 x 2 per MFMA, 4 waves per workgroup
        double mhz_sum = 0, mhz_min = 1e30, mhz_max = 0, ratio_sum = 0, cyc_sum = 0;
        for (int i = 0; i < cus; ++i) {
            const double mt = (double)st[i * 2], rt = (double)st[i * 2 + 1];     // rt: 100 MHz ticks = 10 ns
            const double mhz = rt > 0 ? mfmas * 64.0 / (rt * 0.01) : 0.0;          // cycles / microseconds
            mhz_sum += mhz; mhz_min = mhz < mhz_min ? mhz : mhz_min; mhz_max = mhz > mhz_max ? mhz : mhz_max;
            ratio_sum += rt > 0 ? mt / rt : 0.0;
            cyc_sum += mt / mfmas;
        }
        out[0] = flops / (ms * 1e-3) / 1e12;        // TFLOP/s over the launch (hipEvents: includes launch ramp and tail)
        out[1] = mhz_sum / cus;                     // shader clock implied by 64 cycles per MFMA, mean over workgroups
        out[2] = mhz_min; out[3] = mhz_max;
        out[4] = ratio_sum / cus * 100.0;           // rate of the s_memtime counter in MHz (s_memrealtime = 100 MHz)
        out[5] = cyc_sum / cus;                     // s_memtime ticks per MFMA (64.1 if s_memtime counts shader cycles)
        out[6] = ms; out[7] = cus;
        out[8] = out[1] * 1e6 * 65536.0 * (cus / 256.0) / 1e12;   // TFLOP/s inside the kernels' own wall time (no launch overhead)
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(stamps);
    (void)hipFree(buf);
    return rc;
}
