// lstm_common.h - device helpers shared by lstm.hip (row-tile kernel) and lstm_coop.hip (column-split kernel)
#pragma once
#include "fsnp_common.h"

namespace fsnp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// ---- LSTM cell update on TWO cells at a time (packed fp32 VALU: v_pk_mul / v_pk_add / v_pk_fma) with 8 transcendentals
// per cell instead of 10:   sigmoid(i) tanh(g) = (1 - eg) / ((1 + ei)(1 + eg)),   sigmoid(o) tanh(c') likewise - one v_rcp per
// product.  e* = exp2(-x log2 e) (sigmoid) / exp2(-2 x log2 e) (tanh).  The tanh exponents are clamped to 2^64: (1 - e) then
// stays finite (inf * 0 would be NaN) AND the product of the two denominators can only overflow when the sigmoid's own
// e exceeds 2^64, i.e. when that sigmoid is < 6e-20 and the true product is 0 to fp32 anyway (a clamp at 2^126 let
// (1 + eo)(1 + ec) overflow for ordinary gates next to a saturated cell: h = 0 instead of -sigmoid(o)); tanh is exactly
// -+1 in fp32 from |x| = 9 on, the clamp acts at |x| = 22.  The cell phases are ~5 % of the row-tile kernel's time and cannot overlap its fp32
// MFMAs (profiles/r01_lstm_phase_ab.md); v_exp / v_rcp issue at a quarter of the packed-math rate.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 exp2_pair(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ f32x2 rcp_pair(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f32x2 min_pair(f32x2 x, float m) { return f32x2{fminf(x.x, m), fminf(x.y, m)}; }
// c <- sigmoid(f) c + sigmoid(i) tanh(g);  returns h = sigmoid(o) tanh(c)
__device__ __forceinline__ f32x2 lstm_cell_pair(f32x2 xi, f32x2 xf, f32x2 xg, f32x2 xo, f32x2& c) {
    constexpr float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
    const f32x2 ei = exp2_pair(xi * kS), ef = exp2_pair(xf * kS), eo = exp2_pair(xo * kS);
    const f32x2 eg = exp2_pair(min_pair(xg * kT, 64.0f));
    const f32x2 igg = (1.0f - eg) * rcp_pair((1.0f + ei) * (1.0f + eg));
    const f32x2 cn = rcp_pair(1.0f + ef) * c + igg;
    c = cn;
    const f32x2 ec = exp2_pair(min_pair(cn * kT, 64.0f));
    return (1.0f - ec) * rcp_pair((1.0f + eo) * (1.0f + ec));
}

// nn.GRU cell on two cells at a time: r = s(a_r), z = s(a_z), n = tanh(a_nx + r a_nh), h' = (1 - z) n + z h.  With
// e_z = exp2(-a_z log2 e), e_n = exp2(-2 (a_nx + r a_nh) log2 e):  h' = [e_z (1 - e_n) + h (1 + e_n)] / ((1 + e_z)(1 + e_n)) -
// 5 transcendentals instead of 6, packed VALU math.  Both exponents are clamped to 2^60: numerator and denominator stay below
// 2^121 (finite) and the clamped factors are exact to fp32 (z < 9e-19 counts as 0, tanh = -1).
__device__ __forceinline__ f32x2 gru_cell_pair(f32x2 ar, f32x2 az, f32x2 anx, f32x2 anh, f32x2 h) {
    constexpr float kS = -1.4426950408889634f, kT = -2.8853900817779268f;
    const f32x2 r = rcp_pair(1.0f + exp2_pair(ar * kS));
    const f32x2 ez = exp2_pair(min_pair(az * kS, 60.0f));
    const f32x2 en = exp2_pair(min_pair((anx + r * anh) * kT, 60.0f));
    const f32x2 num = ez * (1.0f - en) + h * (1.0f + en);
    return num * rcp_pair((1.0f + ez) * (1.0f + en));
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == FSNP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == FSNP_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    if (act == FSNP_ACT_TANH) return tanhf(v);
    return v;
}

// float index of A element (row, k) inside an A-fragment-ordered LDS matrix
__host__ __device__ __forceinline__ int a_frag_index(int row, int k) {
    return (((k >> 3) * 64) + ((k & 1) * 32) + row) * 4 + ((k >> 1) & 3);
}


// ---- per-row-tile exchange region of the column-split kernels, in float4 units (HIMG = (HID / 8) * 64 = one 32 x HID image
// in A-fragment order):  [h0 parity 0][h0 parity 1][h1 parity 0][h1 parity 1][Linear partials 2 x (HID / 8) x 16][h0 third]
// The third h0 image is used by the layer-skewed K-split kernel only (lstm_coop.hip: lstm2_coop_skew_kernel).
// ---- arrival counters of a column-split plan: one 256-byte slot per row tile (word 0 = first counter, word 32 = second), so
// that no two counters share a 128-byte line.  Every workgroup of a tile adds to its counters and polls them every step; packed
// ([T] + [T] words, rounds 1-2) the counters of ALL tiles of a launch sat in one line and the launch's 200+ workgroups
// serialised on it: the half-tile ping-pong kernel's pass went 4.7 -> 6.1 (9 tiles) -> 8.4 us (10 tiles) until they were padded
// (profiles/r03_column_split.md section 6).  LstmArgs::coop_bar_stride = 0: the packed layout (full-band LSTM of FullSubNet).
constexpr int kCoopCounterStride = 64;      // words per row tile
#define FSNP_COOP_BAR(a, tile, which) ((a).coop_bar_stride ? (a).coop_bar + (size_t)(tile) * (a).coop_bar_stride + (which) * ((a).coop_bar_stride / 2) \
                                                           : ((which) ? (a).coop_bar2 : (a).coop_bar) + (tile))
__host__ __device__ constexpr size_t coop_counter_bytes(int tiles) { return (size_t)tiles * kCoopCounterStride * 4; }
__host__ __device__ constexpr int coop_tile_f4(int HID) { return 5 * (HID / 8) * 64 + 2 * (HID / 8) * 16; }

// ---- XCD-local placement of the column-split kernels' workgroups.  The S workgroups that share a row tile (a group)
// exchange h through global memory every step; the dispatcher places workgroup id i on XCD i % 8, so with consecutive ids
// they sit on S different XCDs.  Decoding ids XCD-major keeps a tile's S workgroups on ONE XCD: measured 20.7 -> 17.8 us per
// step at 16 units per workgroup, 30.2 -> 26.4 at 32, 56.5 -> 53.1 at 64 (profiles/r02_column_split.md).  Lx = tiles whose S
// workgroups fit one XCD's CUs; tiles beyond 8 Lx are "spread" over the XCDs as before, so capacities do not change
// (9 tiles at S = 24: 8 local + 1 spread, 27 workgroups per XCD).  Placement is a speed matter only: nothing depends on it.
__host__ __device__ __forceinline__ int xcd_local_tiles(int S, int T, int cus_per_xcd) {
    const int fit = cus_per_xcd / S, need = (T + 7) / 8;
    return fit < need ? fit : need;
}
__host__ __device__ __forceinline__ int xcd_local_blocks_per_xcd(int S, int T, int cus_per_xcd) {
    const int Lx = xcd_local_tiles(S, T, cus_per_xcd);
    const int t_local = T < 8 * Lx ? T : 8 * Lx;
    return Lx * S + ((T - t_local) * S + 7) / 8;
}
__device__ __forceinline__ bool xcd_local_decode(int id, int S, int T, int cus_per_xcd, int& tile, int& cs) {
    const int Lx = xcd_local_tiles(S, T, cus_per_xcd);
    const int t_local = T < 8 * Lx ? T : 8 * Lx;
    const int xcd = id & 7, j = id >> 3;
    if (j < Lx * S) { tile = (j / S) * 8 + xcd; cs = j % S; return tile < t_local; }
    const int q = (j - Lx * S) * 8 + xcd;
    tile = t_local + q / S; cs = q % S;
    return tile < T;
}

// ---- inter-workgroup exchange of the column-split kernels (lstm_coop.hip, lstm_coopn.hip): WRITE-THROUGH protocol.
// Producers store h / Linear partials with sc1 (write-through) stores - a relaxed agent-scope atomic store of 4 bytes is
// exactly `global_store_dword ... sc1` - every storing wave drains vmcnt, __syncthreads, ONE lane arrives on the counter;
// consumers poll relaxed, __syncthreads, and read with sc1 loads (buffer aux = 16 / relaxed agent atomic loads), which
// bypass the CU's L1.  No release / acquire fence: `buffer_wbl2 sc1` + `buffer_inv sc1` issued by ~200 workgroups
// every step cost 4.6 us of a 23 us step at B = 1 (measured by deleting them).  cdna_hip_programming.md, Guideline 16 R1.
constexpr int kSc1 = 16;   // buffer-load aux bit: sc1
__device__ __forceinline__ void xchg_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xchg_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane's wait on an arrival counter.  Bounded by WALL-CLOCK time (s_memrealtime, 100 MHz), not by a poll count: a
// legitimate wait can be as long as a foreign kernel that still occupies the CUs the peers of this launch need (tens of
// ms), a dead one (peers that can never become resident next to another spinning launch) must end in seconds.  The
// first waiter that times out raises the launch's device-side abort word (every other waiter sees it within 256 polls
// and leaves too - the whole launch drains in microseconds instead of timing out once per remaining step) and the
// host-mapped error word, which fails the call loudly (fsnp_poll_errors / the next call on the handle).
constexpr long long kXchgTimeoutTicks = 200000000LL;    // 2 s of the 100 MHz constant clock
__device__ __forceinline__ bool xchg_wait(const unsigned* bar, unsigned target, unsigned* abort_dev, unsigned* err_host) {
    unsigned spins = 0;
    long long t0 = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load(abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > kXchgTimeoutTicks) {
                __hip_atomic_store(abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
    }
    return true;
}

// ---- drift injection (LstmArgs::coop_chaos, fsnp_debug_set_chaos).  On an otherwise idle chip the workgroups of a column-split
// launch run in near lockstep, so a hand-off whose correctness silently depends on that lockstep (a buffer overwritten while a
// slow peer still reads it, a counter target off by one phase) passes every short test and shows only when workgroups drift -
// clocks ramping after an idle period, a neighbour kernel, thousands of steps.  With a seed set, every workgroup sleeps a
// pseudo-random time (hash of seed, workgroup, step, phase; uniform per workgroup) at each phase boundary: nothing on 7 of 8
// boundaries, 3 ... 24 us otherwise, ~200 us once in 1024 - drifts of many whole steps.  Results must not change by one bit
// (tests/test_gpu_parity.py::test_column_split_kernels_under_drift).
__device__ __forceinline__ void chaos_delay(int seed, int t, int phase) {
    if (seed == 0) return;
    unsigned h = (unsigned)seed * 2654435761u ^ (unsigned)blockIdx.x * 40503u ^ (unsigned)t * 2246822519u ^ (unsigned)phase * 3266489917u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    int n = 0;
    if ((h & 7u) == 0) n = 1 + (int)((h >> 3) & 7u);
    if ((h & 1023u) == 1u) n = 64;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles ~ 3.4 us
}

}  // namespace fsnp
