// exchange_litmus.hip - litmus test of the fence-free write-through hand-off the column-split recurrent kernels use
// (csrc/lstm_common.h: sc1 stores, every storing wave drains vmcnt, barrier, ONE relaxed agent-scope arrival; relaxed poll, barrier,
// sc1 loads), with EVERY WORD CHECKED - the kernels themselves only show a stale read as a corrupted mask many steps later.
//
// Round 4: the one wrong answer the path ever returned (GPUTEST_r03.json, B = 3 x 126 s) never reproduced; a rare stale read in this
// exchange was the hypothesis left standing.  This program runs the exchange alone, millions of hand-offs per second:
//   T row tiles x S workgroups (default 5 x 48 = 240: the 8-unit K split of the failing plan; flat ids, so a tile's workgroups sit
//   on all 8 XCDs).  Step t: every workgroup writes its slice (256 words) of image t & 1 - 4-byte sc1 stores (mode 0: what
//   lstm_coop.hip / lstm_coopn.hip do) or 16-byte sc1 buffer stores (mode 1: lstm_hp.hip) - drains, barrier, one arrival; waits for
//   S (t + 1) arrivals, barrier; reads the WHOLE image (S x 256 words) with 16-byte sc1 buffer loads and compares every word with
//   its tag(tile, t, word).  A mismatch is counted and the first few are recorded (tile, step, word, got, want -> how stale).
// Options: steps, launches, mode, busy (a pseudo-random per-workgroup delay per step: uneven load).
//   hipcc --offload-arch=gfx950 -O3 -o exchange_litmus exchange_litmus.hip && ./exchange_litmus [steps=400000] [launches=4] [mode=0] [busy=0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef LITMUS_S
#define LITMUS_S 48                    // workgroups per row tile: 48 (8 hidden units per workgroup) or 12 (32 units: -DLITMUS_S=12)
#endif
constexpr int kS = LITMUS_S, kWordsPerSlice = 12288 / kS, kImgWords = kS * kWordsPerSlice;     // one image = 32 rows x 384 units = 48 KiB
constexpr int kCounterStride = 64;                                                 // words: one 256-byte slot per tile

__device__ __forceinline__ unsigned tag(int tile, int t, int word) {
    unsigned h = (unsigned)t * 2654435761u ^ (unsigned)word * 40503u ^ (unsigned)tile * 3266489917u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return h;
}

struct Args {
    unsigned* img;        // [T][2][kImgWords]
    unsigned* bar;        // [T][kCounterStride]
    unsigned* err;        // [0] mismatching words, [1] records taken, [2] time-outs; records from [8]: {tile, step, word, got, want, reader slice, -, -}
    int steps, mode, busy;
};

__global__ __launch_bounds__(256) void litmus_kernel(Args a) {
    __shared__ int abort_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / kS, cs = blockIdx.x % kS;
    unsigned* img = a.img + (size_t)tile * 2 * kImgWords;
    unsigned* bar = a.bar + (size_t)tile * kCounterStride;
    if (tid == 0) abort_s = 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(img, 0, 2 * kImgWords * 4, 0x00020000);
    unsigned bad = 0;
    for (int t = 0; t < a.steps; ++t) {
        const int cur = t & 1;
        if (a.busy) {                                                  // uneven load: 0 ... busy x 0.4 us, per workgroup and step
            unsigned h = tag(blockIdx.x, t, 12345);
            const int n = (int)(h % (unsigned)(a.busy + 1));
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
        }
        // ---- publish this workgroup's slice of image `cur`
        if (a.mode == 0) {
            for (int i = tid; i < kWordsPerSlice; i += 256)
                __hip_atomic_store(img + (size_t)cur * kImgWords + cs * kWordsPerSlice + i, tag(tile, t, cs * kWordsPerSlice + i), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);         // global_store_dword ... sc1
        } else if (wave == 0) {                                        // 16 bytes per lane, by ONE wave (lstm_hp.hip)
            for (int i = lane; i < kWordsPerSlice / 4; i += 64) {
                const int w0 = cs * kWordsPerSlice + i * 4;
                const __attribute__((ext_vector_type(4))) unsigned v = {tag(tile, t, w0), tag(tile, t, w0 + 1), tag(tile, t, w0 + 2), tag(tile, t, w0 + 3)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (cur * kImgWords + w0) * 4, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)kS * (unsigned)(t + 1);
            long long t0 = 0;
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0) {
                    const long long now = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > 300000000LL || __hip_atomic_load(a.err + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        abort_s = 1;
                        __hip_atomic_fetch_add(a.err + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (abort_s) return;
        // ---- read the whole image with 16-byte sc1 loads, check every word
#pragma unroll 4
        for (int i = tid; i < kImgWords / 4; i += 256) {
            const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(rs, (cur * kImgWords + i * 4) * 4, 0, 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned want = tag(tile, t, i * 4 + j);
                if (v[j] != want) {
                    ++bad;
                    const unsigned slot = __hip_atomic_fetch_add(a.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (slot < 16) {
                        unsigned* r = a.err + 8 + slot * 8;
                        r[0] = tile; r[1] = t; r[2] = i * 4 + j; r[3] = v[j]; r[4] = want; r[5] = cs;
                        r[6] = v[j] == tag(tile, t - 2, i * 4 + j) ? 2u : (t >= 4 && v[j] == tag(tile, t - 4, i * 4 + j)) ? 4u : 0u;   // how many steps stale
                    }
                }
            }
        }
        // (image cur is next written in step t + 2, behind the barrier of step t + 1, which every workgroup reaches after this read)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (bad) __hip_atomic_fetch_add(a.err, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 400000, launches = argc > 2 ? atoi(argv[2]) : 4, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int busy = argc > 4 ? atoi(argv[4]) : 0, tiles = argc > 5 ? atoi(argv[5]) : 5;
    Args a{};
    a.steps = steps; a.mode = mode; a.busy = busy;
    const size_t img_b = (size_t)tiles * 2 * kImgWords * 4, bar_b = (size_t)tiles * kCounterStride * 4, err_b = (8 + 16 * 8) * 4;
    CHECK(hipMalloc(&a.img, img_b));
    CHECK(hipMalloc(&a.bar, bar_b));
    CHECK(hipMalloc(&a.err, err_b));
    CHECK(hipMemset(a.err, 0, err_b));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double total_ms = 0.0;
    for (int l = 0; l < launches; ++l) {
        CHECK(hipMemset(a.img, 0xff, img_b));
        CHECK(hipMemset(a.bar, 0, bar_b));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(litmus_kernel, dim3(tiles * kS), dim3(256), 0, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
    }
    std::vector<unsigned> err(8 + 16 * 8);
    CHECK(hipMemcpy(err.data(), a.err, err_b, hipMemcpyDeviceToHost));
    const double handoffs = (double)tiles * steps * launches;
    printf("mode %d (%s stores), %d tiles x %d workgroups, %d steps x %d launches, busy %d: %.3g tile hand-offs, %.3g words checked, %.1f ms per launch (%.2f us per step)\n",
           mode, mode ? "16-byte sc1 buffer" : "4-byte sc1", tiles, kS, steps, launches, busy, handoffs, handoffs * kS * (double)kImgWords, total_ms / launches,
           total_ms / launches * 1e3 / steps);
    printf("  mismatching words: %u   time-outs: %u\n", err[0], err[2]);
    for (unsigned i = 0; i < err[1] && i < 16; ++i) {
        const unsigned* r = err.data() + 8 + i * 8;
        printf("  tile %u step %u word %u (slice %u) read by slice %u: got %08x want %08x  (%s)\n", r[0], r[1], r[2], r[2] / kWordsPerSlice, r[5], r[3], r[4],
               r[6] == 2 ? "the value of step t - 2: a STALE read of the same image" : r[6] == 4 ? "the value of step t - 4" : r[3] == 0xffffffffu ? "never written" : "unrelated");
    }
    return err[0] || err[2] ? 2 : 0;
}
