// mfma_trans.hip - micro-benchmark (round 3): what do the instructions of the LSTM CELL phase
// (v_exp_f32 / v_rcp_f32 = quarter rate, v_pk_*_f32, v_accvgpr_read) cost when they are issued INSIDE a stream of
// v_mfma_f32_32x32x2_f32 instead of in a phase of their own?  If a transcendental's 16 cycles ran beside the matrix pipe, moving
// the cell update of one unit tile under the MFMAs of the next would hide most of it; if every VALU cycle is a matrix-pipe bubble
// (what mfma_issue.hip found for v_fmac), it cannot.  One wave per SIMD, 4 waves per CU, 256 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_trans.hip -o mfma_trans ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define VRCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define VFMA(x, a, b) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define VPK(x, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))

constexpr int ITERS = 256;

template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void kern(const float* __restrict__ gin, float* __restrict__ gout, unsigned long long* __restrict__ ticks) {
    const int tid = threadIdx.x;
    f32x16 acc[12];
#pragma unroll
    for (int n = 0; n < 12; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int n = 0; n < 12; ++n) asm volatile("" : "+a"(acc[n]));
    float a = gin[tid], b = gin[tid + 256];
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = gin[tid + i] * 1e-3f;
    f32x2 pk[4], pa = {gin[tid], gin[tid + 1]}, pb = {gin[tid + 2], gin[tid + 3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = f32x2{0.f, 0.f};
    float rd[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#define CHAIN(n) MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); MF(acc[n], a, b); SB()
        if constexpr (P == 0) {            // 48 MFMAs, bare
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); }
        } else if constexpr (P == 1) {     // 48 v_exp alone (8 independent registers)
#pragma unroll
            for (int n = 0; n < 48; ++n) { VEXP(e[n & 7]); }
            SB();
        } else if constexpr (P == 2) {     // 48 v_rcp alone
#pragma unroll
            for (int n = 0; n < 48; ++n) { VRCP(e[n & 7]); }
            SB();
        } else if constexpr (P == 3) {     // 48 v_fmac alone
#pragma unroll
            for (int n = 0; n < 48; ++n) { VFMA(e[n & 7], a, b); }
            SB();
        } else if constexpr (P == 4) {     // 48 v_pk_fma alone
#pragma unroll
            for (int n = 0; n < 48; ++n) { VPK(pk[n & 3], pa, pb); }
            SB();
        } else if constexpr (P == 5) {     // 48 v_accvgpr_read alone
#pragma unroll
            for (int n = 0; n < 48; ++n) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[n & 3]) : "a"(acc[n % 12][n & 15])); }
            SB();
        } else if constexpr (P == 10) {    // MFMAs + 4 v_exp after each chain (48)
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); VEXP(e[0]); VEXP(e[1]); VEXP(e[2]); VEXP(e[3]); SB(); }
        } else if constexpr (P == 11) {    // MFMAs + 1 v_exp after every MFMA (48)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); SB(); VEXP(e[0]); SB(); MF(acc[n], a, b); SB(); VEXP(e[1]); SB();
                MF(acc[n], a, b); SB(); VEXP(e[2]); SB(); MF(acc[n], a, b); SB(); VEXP(e[3]); SB();
            }
        } else if constexpr (P == 12) {    // MFMAs, then 48 v_exp in one batch
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); }
#pragma unroll
            for (int n = 0; n < 48; ++n) { VEXP(e[n & 7]); }
            SB();
        } else if constexpr (P == 13) {    // MFMAs + 4 v_rcp after each chain (48)
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); VRCP(e[0]); VRCP(e[1]); VRCP(e[2]); VRCP(e[3]); SB(); }
        } else if constexpr (P == 14) {    // MFMAs + 2 v_exp after every MFMA (96)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); SB(); VEXP(e[0]); VEXP(e[4]); SB(); MF(acc[n], a, b); SB(); VEXP(e[1]); VEXP(e[5]); SB();
                MF(acc[n], a, b); SB(); VEXP(e[2]); VEXP(e[6]); SB(); MF(acc[n], a, b); SB(); VEXP(e[3]); VEXP(e[7]); SB();
            }
        } else if constexpr (P == 15) {    // MFMAs + 4 v_pk_fma after each chain (48)
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); VPK(pk[0], pa, pb); VPK(pk[1], pa, pb); VPK(pk[2], pa, pb); VPK(pk[3], pa, pb); SB(); }
        } else if constexpr (P == 16) {    // MFMAs on tiles 0..7 + 4 v_accvgpr_read of tiles 8..11 (finished tiles) after each chain
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                CHAIN(n & 7);
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[0]) : "a"(acc[8 + (n & 3)][0]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[1]) : "a"(acc[8 + (n & 3)][1]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[2]) : "a"(acc[8 + (n & 3)][2]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd[3]) : "a"(acc[8 + (n & 3)][3])); SB();
            }
        } else if constexpr (P == 17) {    // MFMAs + the mix of ONE cell pair per chain: 4 accvgpr reads... 8 trans + 6 packed ops per 2 cells -> per chain: 4 exp/rcp + 3 pk
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); VEXP(e[0]); VPK(pk[0], pa, pb); VEXP(e[1]); VPK(pk[1], pa, pb); VRCP(e[2]); VPK(pk[2], pa, pb); VRCP(e[3]); SB(); }
        } else if constexpr (P == 18) {    // MFMAs + 8 v_exp after each chain (96): is the trans unit's cost linear?
#pragma unroll
            for (int n = 0; n < 12; ++n) { CHAIN(n); VEXP(e[0]); VEXP(e[1]); VEXP(e[2]); VEXP(e[3]); VEXP(e[4]); VEXP(e[5]); VEXP(e[6]); VEXP(e[7]); SB(); }
        } else if constexpr (P == 19) {    // MFMAs + dependent pair exp -> rcp on the same register after each MFMA pair (24 + 24)
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                MF(acc[n], a, b); MF(acc[n], a, b); SB(); VEXP(e[n & 7]); VRCP(e[n & 7]); SB();
                MF(acc[n], a, b); MF(acc[n], a, b); SB(); VEXP(e[(n + 4) & 7]); VRCP(e[(n + 4) & 7]); SB();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a + b + rd[0] + rd[1] + rd[2] + rd[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1];
#pragma unroll
    for (int n = 0; n < 12; ++n) s += acc[n][0] + acc[n][7];
    gout[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

static double ticks_per_cycle = 0;
template <int P> void run(const char* name, int mfmas, int extra, const float* din, float* dout, unsigned long long* dt) {
    const int blocks = 256;
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    hipLaunchKernelGGL(kern<P>, dim3(blocks), dim3(256), 0, 0, din, dout, dt);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(blocks);
    hipMemcpy(t.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : t) avg += (double)v; avg /= blocks;
    // s_memtime ticks at 100 MHz on this part: convert with the bare-MFMA run (64 cycles per MFMA) measured first
    const double per_iter_ticks = avg / ITERS;
    if (P == 0) ticks_per_cycle = per_iter_ticks / (48 * 64.0);
    const double cyc = per_iter_ticks / ticks_per_cycle;
    const double over = cyc - mfmas * 64.0;
    printf("%-78s cycles/iter=%8.1f  mfma=%5d  other=%+8.1f", name, cyc, mfmas * 64, over);
    if (extra) printf("  = %5.1f cycles per extra instruction (%d)", over / extra, extra);
    printf("\n");
}

int main() {
    float *din, *dout; unsigned long long* dt;
    hipMalloc(&din, 1 << 20); hipMalloc(&dout, 1 << 20); hipMalloc(&dt, 4096 * 8);
    std::vector<float> h(1 << 18, 0.001f);
    hipMemcpy(din, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<0>("0  48 MFMA 32x32x2 f32 (12 acc x chain 4), bare  [defines the cycle]", 48, 0, din, dout, dt);
    run<1>("1  48 v_exp_f32 alone", 0, 48, din, dout, dt);
    run<2>("2  48 v_rcp_f32 alone", 0, 48, din, dout, dt);
    run<3>("3  48 v_fmac_f32 alone", 0, 48, din, dout, dt);
    run<4>("4  48 v_pk_fma_f32 alone", 0, 48, din, dout, dt);
    run<5>("5  48 v_accvgpr_read_b32 alone", 0, 48, din, dout, dt);
    run<10>("10 MFMAs + 4 v_exp after each chain", 48, 48, din, dout, dt);
    run<11>("11 MFMAs + 1 v_exp after EVERY mfma", 48, 48, din, dout, dt);
    run<12>("12 MFMAs, then 48 v_exp in one batch", 48, 48, din, dout, dt);
    run<13>("13 MFMAs + 4 v_rcp after each chain", 48, 48, din, dout, dt);
    run<14>("14 MFMAs + 2 v_exp after EVERY mfma", 48, 96, din, dout, dt);
    run<18>("18 MFMAs + 8 v_exp after each chain", 48, 96, din, dout, dt);
    run<15>("15 MFMAs + 4 v_pk_fma after each chain", 48, 48, din, dout, dt);
    run<16>("16 MFMAs (8 tiles) + 4 v_accvgpr_read of finished tiles after each chain", 48, 48, din, dout, dt);
    run<17>("17 MFMAs + cell mix (2 exp, 2 rcp, 3 pk_fma) after each chain", 48, 84, din, dout, dt);
    run<19>("19 MFMAs + dependent exp -> rcp after every 2nd mfma", 48, 48, din, dout, dt);
    return 0;
}
