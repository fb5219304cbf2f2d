// weight_watch.h - device side of fsnp_watch_weights (fsnp_weights.hip), shared with the forward's prologue kernel (fsnp_abi.hip).
// A 64-bit fingerprint of the caller's source tensors: sum over all elements of bits(x_i) * (2 i + 1) mod 2^64 (i = position in the
// concatenation): any single changed element changes it, the sum is order-independent (integer adds), so blocks accumulate with one
// atomic each and the LAST block to finish compares with the baseline taken at registration.
#pragma once
#include "fsnp_handle.h"

namespace fsnp {

// (256 threads per block; `block` of `nblocks` - the blocks may be part of a larger launch: fsnp_abi.hip prologue_kernel)
__device__ __forceinline__ void weight_watch_block(const WatchSeg* __restrict__ segs, int nseg, unsigned long long* acc, int baseline, unsigned* err_host,
                                   int block, int nblocks) {
    // (segments are <= kWatchSeg = 8192 elements)
    unsigned long long sum = 0;
    for (int sg = block; sg < nseg; sg += nblocks) {
        const WatchSeg g = segs[sg];
        unsigned done = 0;
        if ((reinterpret_cast<unsigned long long>(g.p) & 15ull) == 0 && g.n >= 4) {
            const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(g.p);
            const unsigned n4 = g.n / 4;                       // <= 2048 (kWatchSeg): EIGHT loads per thread, all issued before the first use -
            uint4 v[8];                                        // as a loop over q hipcc waited for every load in turn (~50 us for 35 MB)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const unsigned q = threadIdx.x + 256u * k; v[k] = p4[q < n4 ? q : 0u]; }      // (load always, select after: no branch per load)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned long long i = 2ull * (g.first + 4ull * (threadIdx.x + 256u * k)) + 1ull;
                if (threadIdx.x + 256u * k < n4) sum += (unsigned long long)v[k].x * i + (unsigned long long)v[k].y * (i + 2ull) + (unsigned long long)v[k].z * (i + 4ull) + (unsigned long long)v[k].w * (i + 6ull);
            }
            done = n4 * 4;
        }
        for (unsigned i = done + threadIdx.x; i < g.n; i += 256) sum += (unsigned long long)g.p[i] * (2ull * (g.first + i) + 1ull);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
    __shared__ unsigned long long part[4];
    __shared__ int is_last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    // The block's partial sum leaves as ONE write-through store, then ONE relaxed ticket atomic behind a drained vmcnt - the hand-off
    // recipe of lstm_common.h.  No __threadfence(): a release fence is `buffer_wbl2` - it writes back whatever the previous kernels
    // (and the sibling blocks that zero the workspace) left dirty in the L2 - and 512 of them made this kernel take 32-50 us for the
    // default model's 35 MB instead of the ~10 us its loads need (measured: profiles/r05_fullband.md).  The last block adds the
    // partials (write-through-coherent loads).
    if (threadIdx.x == 0) {
        __hip_atomic_store(acc + 8 + block, part[0] + part[1] + part[2] + part[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = __hip_atomic_fetch_add(acc + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)nblocks - 1;
    }
    __syncthreads();
    if (is_last) {
        unsigned long long total = 0;
        for (int b = threadIdx.x; b < nblocks; b += 256) total += __hip_atomic_load(acc + 8 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) total += __shfl_xor(total, m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
            total = part[0] + part[1] + part[2] + part[3];
            if (baseline) acc[2] = total;
            else if (total != acc[2]) __hip_atomic_fetch_or(err_host, kErrStaleWeights, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(acc + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace fsnp
