// fsnp_abi.hip - C ABI of libfsnp_hip.so (include/fsnp.h): handle life cycle, workspace management, the forward orchestration and
// the tuning / test hooks.  Host code only; kernels live in frontend.hip / tcn.hip / subband.hip / lstm*.hip; the planner in
// planner.cpp, weight packing in fsnp_weights.hip, the STFT entry points in fsnp_stft_abi.hip (shared declarations: fsnp_handle.h).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "fsnp_common.h"
#include "lstm_common.h"
#include "planner.h"
#include "fsnp_handle.h"
#include "weight_watch.h"

namespace fsnp {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

}  // namespace fsnp

using namespace fsnp;

namespace fsnp {

// Decodes and clears the host-mapped error word that finished launches set (fsnp_handle.h: kErr*).  0 = clean.
// The word is TAKEN with one atomic exchange (a bit the device ORs in between a read and a separate clearing store would be lost:
// ADVICE r05) and every condition that was set is named in the message; the return code is the gravest one (5 > 7 > 6).
int take_device_errors(fsnp_handle* h, const char* where) {
    unsigned* e = reinterpret_cast<unsigned*>(h->d_err);
    if (__atomic_load_n(e, __ATOMIC_ACQUIRE) == 0) return 0;
    const unsigned bits = __atomic_exchange_n(e, 0u, __ATOMIC_ACQ_REL);
    if (bits == 0) return 0;
    std::string msg = std::string(where) + ":";
    int rc = 0;
    if (bits & kErrTimeout) {
        msg += " an inter-workgroup wait timed out in a column-split LSTM kernel (its workgroups were not co-resident - is the GPU "
               "shared with another process?); the result of that forward is invalid.  FSNP_LSTM_COOP=0 avoids these kernels.";
        rc = 5;
    }
    if (bits & kErrVerify) {
        // where the two kernels first disagreed: a 64-bit key in DEVICE memory (utterance << 44 | bin << 24 | frame; atomic min), read
        // back here - on the error path only - once the comparing kernel has finished
        unsigned long long key = ~0ull;
        if (h->verify_key || h->verify_key_sampled) {
            fsnp::DeviceGuard guard(h->device);
            (void)hipDeviceSynchronize();
            for (unsigned long long* src : {h->verify_key, h->verify_key_sampled}) {
                unsigned long long k = ~0ull;
                if (src && hipMemcpy(&k, src, 8, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); k = ~0ull; }
                if (k < key) key = k;
            }
        }
        char at[160];
        if (key != ~0ull) snprintf(at, sizeof(at), ", first at utterance %llu, bin %llu, frame %llu", key >> 44, (key >> 24) & 0xFFFFFull, key & 0xFFFFFFull);
        else at[0] = 0;
        msg += std::string(rc ? " ALSO:" : "") + " exchange verification failed (fsnp_set_verify): the column-split kernels and the exchange-free "
               "kernel disagree" + at + "; the result of that forward is invalid (a sampled check - fsnp_set_verify_sample - reports a forward "
               "of up to a few calls ago).  FSNP_LSTM_COOP=0 avoids the column-split kernels.";
        if (!rc) rc = 7;
    }
    if (bits & kErrStaleWeights) {
        msg += std::string(rc ? " ALSO:" : "") + " the watched source tensors no longer match the packed weights (a parameter was modified in "
               "place through .data after packing): forwards since that edit ran on the OLD weights - re-pack (fsnp_set_weight / "
               "fsnp_commit_weights / fsnp_watch_weights; Python: model.refresh_weights()).";
        if (!rc) rc = 6;
    }
    if (!rc) { msg += " unknown device error bits"; rc = 4; }
    set_error("%s", msg.c_str());
    return rc;
}

// what the planner needs to know of a handle
static PlannerCtx pctx(const fsnp_handle* h) {
    PlannerCtx c;
    c.H = h->H; c.NIN = h->NIN; c.num_cus = h->num_cus; c.num_cus_real = h->num_cus_real;
    c.gru = h->gru != 0; c.sb_tcn = h->sb_tcn != 0; c.generic_sb = h->generic_sb; c.rowtile_ok = h->rowtile_ok; c.lstm16_ok = h->lstm16_ok;
    c.hp_ok = h->hp_ok; c.coop_hp = h->coop_hp; c.coopw_ok = h->coopw_ok; c.coop_w = h->coop_w; c.ih_bf16 = h->ih_bf16; c.lstm_coop = h->lstm_coop; c.coop_occ = h->coop_occ;
    for (int i = 0; i < 4; ++i) c.occ_ksplit[i] = h->occ_ksplit[i];
    for (int i = 0; i < 2; ++i) c.occ_coopn[i] = h->occ_coopn[i];
    c.pipeline = h->pipeline; c.composite_gain = h->composite_gain;
    c.cost = h->cost;
    return c;
}


// Row slots of the sub-band problem.  Tile i owns `rt` slots (32 MFMA rows + ex VALU rows) and gets
// base (+1 for the first rem tiles) consecutive sequences; slot -> (utterance, frequency, output offset).
__global__ void build_rows_kernel(RowDesc* rows, int num_rows, int num_tiles, int rt, int F, int T, int mode,
                                  int batch_offset, int global_batch, int dense_out, int n_base, int groups, int OC) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= num_tiles * rt) return;
    const int tile = slot / rt, sl = slot % rt;
    const int base = num_rows / num_tiles, rem = num_rows % num_tiles;
    const int cnt = base + (tile < rem ? 1 : 0);
    const int n = n_base + tile * base + (tile < rem ? tile : rem) + sl;      // n_base: first sequence of this chunk
    RowDesc r{0, 0, 0, 0};
    if (sl < cnt) {
        r.valid = 1;
        if (dense_out) {               // fsnp_lstm2_fc: x[n][t][:] -> out[n][o][t], o < OC = output_size (fullsubnet_plus.py:104,206)
            r.b = n; r.f = 0; r.out_off = n * OC * T;
        } else if (mode == FSNP_MODE_FULL) {
            r.b = n / F; r.f = n % F;
            r.out_off = ((r.b * OC) * F + r.f) * T;
        } else {                       // drop_band (feature.py:254-285) with G = num_groups_in_drop_band groups:
            // global sample s keeps bins p + G i (p = s % G, i < (F - F % G) / G); output rows = group 0's samples, group 1's, ...
            const int G = groups, Fh = F / G;
            r.b = n / Fh;
            const int i = n % Fh;
            const int s = batch_offset + r.b, p = s % G;
            int orow = s / G;
            for (int q = 0; q < p; ++q) orow += (global_batch - q + G - 1) / G;     // samples of the groups in front
            r.f = p + G * i;
            r.out_off = ((orow * OC) * Fh + i) * T;
        }
    }
    rows[slot] = r;
}

// Zeroes the accumulator / exchange / barrier region of the workspace.  A kernel rather than hipMemsetAsync: a memset
// node captured into the hipGraph was NOT re-executed reliably on replay (ROCm 7.2: stale barrier counters and
// accumulators after the first launch - tests/test_gpu_parity.py::test_b32_batch_independence caught it).
__global__ __launch_bounds__(256) void zero_region_kernel(uint4* __restrict__ p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static void launch_zero_region(void* p, size_t bytes, hipStream_t s) {      // bytes is a multiple of 256
    const size_t n16 = bytes / 16;
    if (n16 == 0) return;
    const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_region_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n16);
}

// ---- the forward's prologue as ONE launch (round 5; it used to be a zeroing kernel, a build_rows launch per chunk and, with a weight
// watch, the fingerprint kernel - three to six dependent launches of ~5 us each in front of a 250 us full-band stage at B = 1):
// blocks [0, zb) zero the accumulator / exchange / counter region, [zb, zb + rb) describe the sub-band rows of every chunk, the rest
// fingerprint the watched source tensors (fsnp_watch_weights).
struct PrologueChunk { int slot0, nrows, tiles, rt, row0, blocks; };
struct PrologueArgs {
    uint4* zero; size_t n16; int zero_blocks;
    RowDesc* rows; PrologueChunk chunk[8]; int nchunks, rows_blocks;
    int F, T, mode, batch_offset, global_batch, groups, OC;
    const WatchSeg* segs; int nseg, watch_blocks; unsigned long long* watch_acc; unsigned* err_host;
};
__device__ __forceinline__ void build_rows_slot(RowDesc* rows, int slot, int num_rows, int num_tiles, int rt, int F, int T, int mode,
                                                int batch_offset, int global_batch, int n_base, int groups, int OC) {
    const int tile = slot / rt, sl = slot % rt;
    const int base = num_rows / num_tiles, rem = num_rows % num_tiles;
    const int cnt = base + (tile < rem ? 1 : 0);
    const int n = n_base + tile * base + (tile < rem ? tile : rem) + sl;
    RowDesc r{0, 0, 0, 0};
    if (sl < cnt) {
        r.valid = 1;
        if (mode == FSNP_MODE_FULL) {
            r.b = n / F; r.f = n % F;
            r.out_off = ((r.b * OC) * F + r.f) * T;
        } else {                       // drop_band: see build_rows_kernel
            const int G = groups, Fh = F / G;
            r.b = n / Fh;
            const int i = n % Fh;
            const int s = batch_offset + r.b, p = s % G;
            int orow = s / G;
            for (int q = 0; q < p; ++q) orow += (global_batch - q + G - 1) / G;
            r.f = p + G * i;
            r.out_off = ((orow * OC) * Fh + i) * T;
        }
    }
    rows[slot] = r;
}
__global__ __launch_bounds__(256) void prologue_kernel(PrologueArgs a) {
    const int b = blockIdx.x;
    if (b < a.zero_blocks) {
        const size_t stride = (size_t)a.zero_blocks * 256;
        for (size_t i = (size_t)b * 256 + threadIdx.x; i < a.n16; i += stride) a.zero[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    int rb = b - a.zero_blocks;
    if (rb < a.rows_blocks) {
        for (int c = 0; c < a.nchunks; ++c) {
            const PrologueChunk k = a.chunk[c];
            if (rb < k.blocks) {
                const int slot = rb * 256 + threadIdx.x;
                if (slot < k.tiles * k.rt)
                    build_rows_slot(a.rows + k.slot0, slot, k.nrows, k.tiles, k.rt, a.F, a.T, a.mode, a.batch_offset, a.global_batch, k.row0, a.groups, a.OC);
                return;
            }
            rb -= k.blocks;
        }
        return;
    }
    weight_watch_block(a.segs, a.nseg, a.watch_acc, 0, a.err_host, b - a.zero_blocks - a.rows_blocks, a.watch_blocks);
}

// Column-split launches need all their workgroups co-resident.  Two of them running at once (two handles / two streams
// of one process) could each hold part of the chip and wait for peers that cannot be scheduled, so within a process
// they are chained per device: each one waits for the previous one's completion event.  Other kernels always finish,
// so they cannot close a cycle; a foreign PROCESS still can - that case ends in the kernels' wall-clock timeout.
// Round 6: two such launches of ONE handle may run side by side when their workgroups - one per CU each - fit on the chip together
// (`wgs` of both known, sum <= num_cus - 8): the pipelined loop's deferred remainder chunk of forward i (48 workgroups that own their CUs)
// and the original FullSubNet's full-band LSTM of forward i + 1 (64 workgroups) used to wait for each other through this chain, which is
// what kept FullSubNet at B = 32 above its 29 ms bar.  Both are resident at once whatever the dispatch order, so neither can wait for a CU
// the other holds; everything else (other handles, unknown sizes, sums beyond the chip) is chained as before.
static std::mutex g_coop_mu;
struct CoopSlot { hipEvent_t ev = nullptr; bool used = false; const void* owner = nullptr; int wgs = 0; unsigned long long seq = 0; };
static CoopSlot g_coop[64][2];
static unsigned long long g_coop_seq = 0;
template <typename F>
static void launch_coop_chained(int dev, hipStream_t s, F launch, const void* owner = nullptr, int wgs = 0, int num_cus = 0) {
    if (dev < 0 || dev >= 64) { launch(); return; }
    std::lock_guard<std::mutex> lk(g_coop_mu);
    CoopSlot* sl = g_coop[dev];
    const int order[2] = {sl[0].seq <= sl[1].seq ? 0 : 1, sl[0].seq <= sl[1].seq ? 1 : 0};      // older first
    bool waited[2] = {false, false};
    int beside = wgs;
    for (int o = 0; o < 2; ++o) {
        CoopSlot& c = sl[order[o]];
        if (!c.used) continue;
        const bool fits = owner && c.owner == owner && wgs > 0 && c.wgs > 0 && num_cus > 0 && beside + c.wgs <= num_cus - 8;
        if (fits) beside += c.wgs;
        else { (void)hipStreamWaitEvent(s, c.ev, 0); waited[order[o]] = true; }
    }
    launch();
    // the slot this launch takes: a free one, else one it waited for (that launch is behind it now), else one that has finished, else
    // the older one - after waiting for it
    int k = -1;
    for (int i = 0; i < 2 && k < 0; ++i) if (!sl[i].used) k = i;
    for (int o = 0; o < 2 && k < 0; ++o) if (waited[order[o]]) k = order[o];
    for (int o = 0; o < 2 && k < 0; ++o) {
        if (hipEventQuery(sl[order[o]].ev) == hipSuccess) k = order[o];
        else (void)hipGetLastError();
    }
    if (k < 0) { k = order[0]; (void)hipStreamWaitEvent(s, sl[k].ev, 0); }
    if (!sl[k].ev && hipEventCreateWithFlags(&sl[k].ev, hipEventDisableTiming) != hipSuccess) { sl[k].ev = nullptr; sl[k].used = false; return; }
    sl[k].used = hipEventRecord(sl[k].ev, s) == hipSuccess;
    sl[k].owner = owner; sl[k].wgs = wgs; sl[k].seq = ++g_coop_seq;
}

// The planner itself (cost table, launch shapes, shortest path over tile counts) is host-only code: planner.h / planner.cpp.
static PlannerCtx pctx(const fsnp_handle* h);
// gather_bytes: the furthest byte a recurrent kernel's input gather reaches from its base pointer.  The half-tile ping-pong kernel
// (lstm_hp.hip) addresses its input through ONE buffer descriptor with 32-bit BYTE offsets (2 GiB), the other kernels with 32-bit
// FLOAT offsets from a 64-bit pointer (8 GiB): beyond 2 GiB the planner is told not to use it (its loads would
// otherwise return zeros - out of the descriptor's range - and the sequences run on zero input without any error).
static SbPlan plan_sb(const fsnp_handle* h, int num_rows, double gather_bytes = 0.0) {
    PlannerCtx c = pctx(h);
    if (gather_bytes >= 2147483644.0) c.coop_hp = 0;
    return plan_sb(c, num_rows);
}
// forward: rows are gathered from att_mag [B][Tp][FP] and the full-band planes behind it (plan_workspace: att, fb adjacent)
static double forward_gather_bytes(const fsnp_handle* h, int B, int T) {
    const double nbr = h->model == FSNP_MODEL_FULLSUBNET ? 1 : 3;
    return 2.0 * ((double)align_up((size_t)(nbr * B * ((double)T + h->cfg.look_ahead) * h->FP * 4), 256));
}
static int chunk_workgroups(const fsnp_handle* h, const SbChunk& c) { return chunk_workgroups(pctx(h), c); }
// Launches chunks [first, last) of the plan on stream s.  `bar` = per-tile arrival counters followed (at bar +
// plan.coop_tiles, 64-byte aligned by the caller) by the launch-abort word.
static void launch_sb_lstm(const fsnp_handle* h, const SbPlan& plan, const LstmArgs& a, float* hx, unsigned* bar, unsigned* abort_word,
                           hipStream_t s, hipEvent_t after_first = nullptr, int first_chunk = 0, int last_chunk = -1) {
    const size_t hx_floats_per_tile = lstm_coop_exchange_bytes(h->H, 1) / 4;
    const int nchunks = (int)plan.chunks.size();
    if (last_chunk < 0 || last_chunk > nchunks) last_chunk = nchunks;
    for (int ci = first_chunk; ci < last_chunk; ++ci) {
        const SbChunk& c = plan.chunks[ci];
        if (ci == 1 && after_first) { (void)hipEventRecord(after_first, s); after_first = nullptr; }
        LstmArgs ca = a;
        ca.rows = a.rows + c.slot0;
        ca.num_rows = c.nrows; ca.num_tiles = c.num_tiles; ca.ex = c.ex;
        if (a.md_row) ca.md_row = a.md_row + (size_t)c.slot0 * a.Tp;
        if (c.kind == 0) {
            if (h->gru) launch_gru(h->lw, ca, s);
            else { ca.clk = h->d_clk; launch_lstm(h->lw, ca, s); }
            continue;
        }
        if (c.kind == 4) { launch_lstm16(h->lw, ca, s); continue; }
        if (c.kind == 7) { ca.coop_rows_per_group = c.rpg; launch_lstm_generic(h->lw, ca, false, s); continue; }
        ca.coop_hx = hx + (size_t)c.coop_tile0 * hx_floats_per_tile;
        ca.coop_bar = bar + (size_t)c.coop_tile0 * kCoopCounterStride;
        ca.coop_bar_stride = kCoopCounterStride;
        ca.coop_bar2 = nullptr;
        ca.coop_skew = h->coop_skew;
        ca.coop_chaos = h->coop_chaos;
        // pipelined loop: a deferred K-split chunk shares the chip with the next forward's full-band GEMMs; a GEMM workgroup that
        // lands on one of its CUs runs at ~0.6x (and each GEMM launch lasts as long as its slowest workgroup: stage 1.25 -> 1.85 ms),
        // so the chunk claims its CUs' whole LDS and the GEMM workgroups go to the other CUs
        // (never for a launch planned with two workgroups per CU: they would no longer be co-resident)
        ca.coop_own_cu = (h->side_stream && s == h->side_stream && chunk_workgroups(h, c) <= h->num_cus_real) ? 160 * 1024 - 256 : 0;
        ca.coop_err = h->d_err;
        ca.coop_abort = abort_word;
        ca.coop_units = c.units; ca.coop_groups = c.groups; ca.coop_rows_per_group = c.rpg;
        // XCD-local workgroup placement (lstm_common.h), unless the launch was planned with two workgroups per CU
        constexpr int xcd_local = 1;
        {
            const int S = c.kind == 8 ? h->H / 16 : (c.kind == 1 || c.kind == 9) ? h->H / c.units : h->H / 128;
            const int T = c.kind == 2 ? c.groups : c.num_tiles, cpx = h->num_cus_real / 8;
            ca.coop_xcd = xcd_local && h->num_cus_real % 8 == 0 && xcd_local_blocks_per_xcd(S, T, cpx) <= cpx ? cpx : 0;
        }
        launch_coop_chained(h->device, s, [&] {
            if (c.kind == 8) { if (h->lw.hp_wave && lstm_hpw_available(h->lw)) launch_lstm_hpw(h->lw, ca, s); else launch_lstm_hp(h->lw, ca, s); }
            else if (c.kind == 9) launch_lstm_coopw(h->lw, ca, s);
            else if (c.kind == 1) launch_lstm_coop(h->lw, ca, s);
            else launch_lstm_coopn(h->lw, ca, s);
        }, h, chunk_workgroups(h, c), h->num_cus_real);
    }
    if (after_first && last_chunk == nchunks) (void)hipEventRecord(after_first, s);
}
static void launch_build_rows(const SbPlan& plan, RowDesc* rows, int F, int T, int mode, int batch_offset, int global_batch,
                              int dense_out, int groups, int out_channels, hipStream_t s) {
    for (const SbChunk& c : plan.chunks) {
        const int slots = c.num_tiles * c.rps;
        hipLaunchKernelGGL(build_rows_kernel, dim3(cdiv(slots, 256)), dim3(256), 0, s, rows + c.slot0, c.nrows, c.num_tiles, c.rps,
                           F, T, mode, batch_offset, global_batch, dense_out, c.row0, groups, out_channels);
    }
}
// full-band LSTM of the original FullSubNet: B sequences, always the cooperative kernel (units in {8, 16, 32})
static int fb_row_tiles(int B) { return cdiv(B, 32); }
static int fb_coop_units(const fsnp_handle* h, int B) {
    const int u = lstm_coop_pick_units(h->CH, fb_row_tiles(B), h->num_cus_real, 8);
    return u > 32 ? 0 : u;
}

// Pipelined serving loop (fsnp_set_pipeline): which chunks of a plan go to the side stream, where they overlap the NEXT forward's
// full-band stages.  Deferred column-split launches own their CUs (LstmArgs::coop_own_cu), so the overlapped stages run on what is
// left: that pays while the deferred launches leave at least 32 CUs free (B = 32: 48 workgroups, 28.2 -> 27.5 ms; B = 1: 216, 1.97 ->
// 1.77) and LOSES when they fill the chip (B = 40: a 66-tile remainder, 36.9 -> 37.5 ms; B = 21: the overlapped stage took 5.4 ms
// instead of 0.66 - profiles/r04_bench_configs.md): the planner defers only in the first case.
//   returns: first deferred chunk (== chunks.size(): nothing is deferred; 0: the whole plan)
static int plan_first_deferred(const fsnp_handle* h, const SbPlan& plan) {
    const int n = (int)plan.chunks.size();
    auto fills_chip = [](const SbChunk& c) { return c.kind == 0 || c.kind == 4; };       // one (half) tile per CU, no exchange
    if (n == 0 || h->sb_tcn) return n;
    int first = n;
    if (n > 1 && fills_chip(plan.chunks[0])) {
        first = 1;
        while (first < n && fills_chip(plan.chunks[first])) ++first;
    } else if (h->defer_small && !fills_chip(plan.chunks[0])) {
        first = 0;
    }
    int busiest = 0;
    for (int i = first; i < n; ++i) busiest = std::max(busiest, chunk_workgroups(h, plan.chunks[i]));
    return (first < n && busiest <= h->num_cus_real - 32) ? first : n;
}

static int rows_per_utt(const fsnp_handle* h, int mode) {
    return mode == FSNP_MODE_PARITY ? h->F / h->cfg.num_groups_in_drop_band : h->F;
}

static Workspace plan_workspace(const fsnp_handle* h, int B, int T, int mode) {
    Workspace w{};
    const size_t Tp = (size_t)T + h->cfg.look_ahead;
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    const size_t nbr = fsn ? 1 : 3;
    const size_t xb = nbr * B * Tp * h->FP * 4;
    const size_t yb = nbr * B * Tp * align_up(h->CH, 4) * 4;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    w.att = take(xb);                     // FullSubNet: the padded raw magnitude lives here (no attention stage)
    w.fb = take(xb);
    w.raw = take(fsn ? 0 : xb);
    w.x = take(fsn ? 0 : xb);
    w.y1 = take(yb);                      // FullSubNet: h1 sequence of the full-band LSTM [B][Tp][CH]
    w.y2 = take(fsn ? 0 : yb);
    w.gate = take(fsn ? 0 : (size_t)3 * B * h->FP * 4);
    w.md = take(nbr * B * Tp * sizeof(NormMD));
    w.md_utt = take((size_t)B * sizeof(NormMD));
    const SbPlan plan = plan_sb(h, B * rows_per_utt(h, mode), forward_gather_bytes(h, B, T));
    const size_t nrows_pad = (size_t)plan.total_slots;
    const bool cumulative = h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAPLACE || h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAYER;
    w.md_row = take(cumulative ? nrows_pad * Tp * sizeof(NormMD) : 0);
    w.rows = take(nrows_pad * sizeof(RowDesc));
    w.fb_rows = take(fsn ? (size_t)fb_row_tiles(B) * 32 * sizeof(RowDesc) : 0);
    w.frame = take(nbr * B * Tp * 2 * 8);
    const size_t sbt_x = h->sb_tcn ? nrows_pad * Tp * h->XS * 4 : 0, sbt_y = h->sb_tcn ? nrows_pad * Tp * h->CH * 4 : 0;
    w.sbt_x0 = take(sbt_x); w.sbt_x = take(sbt_x); w.sbt_fb = take(sbt_x); w.sbt_y1 = take(sbt_y); w.sbt_y2 = take(sbt_y);
    w.zero_begin = o;
    w.fsum = take(fsn ? 0 : (size_t)3 * B * h->FP * 8);
    w.fe_tot = take(fsn ? 0 : (size_t)3 * B * 2 * 8);
    w.gn = take(fsn ? 0 : (size_t)h->NB * 2 * 3 * B * kGnStride * 8);
    w.sb_acc = take((size_t)B * 2 * 8);
    w.coop_hx = take(lstm_coop_exchange_bytes(h->H, plan.coop_tiles));
    w.coop_bar = take(coop_counter_bytes(plan.coop_tiles));      // two arrival counters per row tile (+ the padded copies: lstm_common.h)
    w.coop_abort = take(256);             // [0] sub-band launches, [16] full-band LSTM (FullSubNet)
    w.fb_hx = take(fsn ? lstm_coop_exchange_bytes(h->CH, fb_row_tiles(B)) : 0);
    w.fb_bar = take(fsn ? (size_t)fb_row_tiles(B) * 4 : 0);
    w.sbt_gn = take(h->sb_tcn ? (size_t)8 * 2 * nrows_pad * kGnStride * 8 : 0);
    w.zero_end = o;
    w.dbg_tcn0 = take(h->debug ? (size_t)B * Tp * h->FP * 4 : 0);
    w.total = o;
    return w;
}

// h->ws holds h->ws_slots (1, or 2 in pipelined mode) halves of h->ws_bytes each.  Growth is STREAM-ORDERED on the caller's
// stream (hipMallocAsync / hipFreeAsync from the device's default pool): no device-wide synchronisation, so a serving loop
// whose clip lengths vary never stalls other streams while it climbs to its high-water mark (fsnp_reserve jumps there at once).
// The old buffer is released behind everything that may still read it: earlier forwards on `s` (in order), their deferred
// chunks on the side stream (an event), forwards on another stream (forward_impl orders `s` behind them before it gets here).
static int ensure_workspace(fsnp_handle* h, size_t bytes, hipStream_t s) {
    bytes = align_up(bytes, 4096);
    if (bytes <= h->ws_bytes) return 0;
    unsigned char* nw = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&nw), bytes * h->ws_slots, s));
    // a fresh workspace is all zeros: tcn_gemm_dma_kernel DMAs the pad columns [K, lda) of its operand planes (they meet zero
    // weights, but NaN bit patterns left by an earlier owner of the memory would survive that); every kernel that writes a
    // plane writes its pad columns as zeros too, so this only matters for the very first use of a region
    FSNP_HIP_CHECK(hipMemsetAsync(nw, 0, bytes * h->ws_slots, s));
    if (h->ws) {
        for (int k = 0; k < 2; ++k)
            if (h->side_used[k]) FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_side[k], 0));
        FSNP_HIP_CHECK(hipFreeAsync(h->ws, s));
    }
    h->ws = nw;
    h->ws_bytes = bytes;
    h->side_used[0] = h->side_used[1] = false;     // `s` is ordered behind the side stream's work on the old buffer
    h->have_last = false;
    return 0;
}
// Orders `s` behind the last forward of this handle if that ran on ANOTHER stream (the workspace is shared by all of them)
int order_after_last_forward(fsnp_handle* h, hipStream_t s) {
    if (h->done_valid && h->done_stream != s) FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_done, 0));
    return 0;
}
int mark_forward_done(fsnp_handle* h, hipStream_t s) {
    if (!h->ev_done) FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    FSNP_HIP_CHECK(hipEventRecord(h->ev_done, s));
    h->done_stream = s; h->done_valid = true;
    return 0;
}

// hipEvents for the per-forward timing records come from a pool; records that nobody reads are folded into the
// accumulators once 256 have piled up (they completed long ago), so timing never grows without bound.
static int drain_timing(fsnp_handle* h) {
    for (auto& r : h->timing_recs) {
        float fb = 0, lstm = 0, all = 0, first = 0;
        // a forward that failed half way leaves events unrecorded: such a record is dropped, not counted
        const bool ok = hipEventSynchronize(r.e[2]) == hipSuccess && hipEventSynchronize(r.e[3]) == hipSuccess &&
                        hipEventElapsedTime(&fb, r.e[0], r.e[1]) == hipSuccess && hipEventElapsedTime(&lstm, r.e[1], r.e[2]) == hipSuccess &&
                        hipEventElapsedTime(&all, r.e[0], r.e[2]) == hipSuccess && hipEventElapsedTime(&first, r.e[1], r.e[3]) == hipSuccess;
        if (ok) {
            h->acc_ms[0] += lstm; h->acc_ms[1] += fb; h->acc_ms[2] += all; h->acc_ms[3] += first;
            for (int i = 0; i < 4; ++i) h->acc_cnt[i] += 1;
        } else {
            (void)hipGetLastError();
        }
        for (auto& e : r.e) h->event_pool.push_back(e);
    }
    h->timing_recs.clear();
    return 0;
}
static int take_timing_rec(fsnp_handle* h, TimingRec& rec) {
    if (h->timing_recs.size() >= 256 && drain_timing(h)) return 1;
    for (auto& e : rec.e) {
        if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
        else FSNP_HIP_CHECK(hipEventCreate(&e));
    }
    return 0;
}


static double lstm_flops_per_step(const fsnp_handle* h) {
    if (h->sb_tcn) return 8 * (2.0 * h->NIN * h->CH + 2.0 * h->CH * 3 + 2.0 * h->CH * h->NIN) + 2.0 * h->NIN * h->cfg.output_size;
    const double H = h->H, NIN = h->NIN, OUT = h->cfg.output_size, G = h->NG;
    return 2.0 * G * H * (NIN + H) + 2.0 * G * H * (2 * H) + 2.0 * H * OUT;
}
static double tcn_flops_per_frame(const fsnp_handle* h) {
    const double F = h->F, CH = h->CH;
    return h->NB * (2.0 * F * CH + 2.0 * CH * 3 + 2.0 * CH * F) + 2.0 * F * F;
}
static double fb_lstm_flops_per_frame(const fsnp_handle* h) {   // original FullSubNet: LSTM(F, CH) x 2 + Linear(CH, F)
    const double F = h->F, CH = h->CH;
    return 2.0 * h->NG * CH * (F + CH) + 2.0 * h->NG * CH * (2 * CH) + 2.0 * CH * F;
}

// The fused recurrent model + Linear on a dense input x [num_seq][steps][NIN] with a given plan (fsnp_lstm2_fc, calibration).
static int run_dense_plan(fsnp_handle* h, const SbPlan& plan, const float* x, float* out, int num_seq, int steps, hipStream_t s) {
    const int num_slots = plan.total_slots;
    const bool coop = plan.coop_tiles != 0;
    const size_t coop_off = align_up((size_t)num_slots * sizeof(RowDesc), 256);
    const size_t coop_hx_bytes = coop ? align_up(lstm_coop_exchange_bytes(h->H, plan.coop_tiles), 256) : 0;
    const size_t coop_bar_bytes = align_up(coop_counter_bytes(plan.coop_tiles), 256);
    const size_t coop_bytes = coop ? coop_hx_bytes + coop_bar_bytes + 256 : 0;       // images, counters, abort word
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_workspace(h, coop_off + coop_bytes, s)) return 4;
    if (h->pipeline && h->side_stream) FSNP_HIP_CHECK(hipStreamSynchronize(h->side_stream));   // slot 0 may still be read
    RowDesc* rows = reinterpret_cast<RowDesc*>(h->ws);
    h->have_last = false;   // the workspace no longer holds a forward's stages
    if (coop) FSNP_HIP_CHECK(hipMemsetAsync(h->ws + coop_off, 0, coop_bytes, s));
    launch_build_rows(plan, rows, 1, steps, 0, 0, 1, 1, 2, h->cfg.output_size, s);
    LstmArgs a{};
    a.rows = rows; a.dense = x; a.dense_stride = h->NIN; a.out = out; a.out_stride_o = steps;
    a.num_rows = num_seq; a.Tp = steps; a.LA = 0; a.FP = 0; a.F = 1; a.NSBN = 0; a.act = h->cfg.sb_act;
    launch_sb_lstm(h, plan, a, reinterpret_cast<float*>(h->ws + coop_off),
                   reinterpret_cast<unsigned*>(h->ws + coop_off + coop_hx_bytes),
                   reinterpret_cast<unsigned*>(h->ws + coop_off + coop_hx_bytes + coop_bar_bytes), s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

// ---- calibration of the planner's cost table: once per process and (device, cell, sizes), on the first call that plans.
// Every launch shape the planner can pick is run on zeros at two step counts; the per-step cost is the slope (launch and
// prologue cancel).  ~0.1 s and ~60 MB of scratch, then cached for every later handle of the same kind.  FSNP_CALIBRATE=0
// keeps the built-in table (the round-1 measurements).
struct CalKey {
    int dev, H, KX, gru, occ, cus;
    bool operator<(const CalKey& o) const {
        const int a[6] = {dev, H, KX, gru, occ, cus}, b[6] = {o.dev, o.H, o.KX, o.gru, o.occ, o.cus};
        for (int i = 0; i < 6; ++i) if (a[i] != b[i]) return a[i] < b[i];
        return false;
    }
};
static std::mutex g_cal_mu;
static std::map<CalKey, CostTable> g_cal_cache;

static int calibrate_costs(fsnp_handle* h, bool adopt = true, CostTable* measured = nullptr) {
    if (adopt && (h->cost.calibrated || !h->calibrate)) return 0;
    if (h->sb_tcn || h->generic_sb || !h->committed) return 0;
    int occ_sig = h->coop_occ;
    for (int i = 0; i < 4; ++i) occ_sig = occ_sig * 4 + h->occ_ksplit[i];
    for (int i = 0; i < 2; ++i) occ_sig = occ_sig * 4 + h->occ_coopn[i];
    occ_sig = occ_sig * 2 + h->coop_skew;
    const CalKey key{h->device, h->H, h->KX, h->gru, occ_sig, h->num_cus_real};
    {
        std::lock_guard<std::mutex> lk(g_cal_mu);
        auto it = g_cal_cache.find(key);
        if (it != g_cal_cache.end()) {
            if (measured) *measured = it->second;
            if (adopt) h->cost = it->second;
            return 0;
        }
    }
    // the timing launches overwrite the handle's workspace from a private stream: nothing of an earlier forward may still be
    // in flight on the caller's streams (ensure_workspace only synchronises when it grows)
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    const int S3 = h->H / 128, occ = h->coop_occ >= 2 ? 2 : 1;
    const int steps_a = 8, steps_b = 40;
    const int max_tiles = std::max(h->num_cus_real, (h->num_cus_real * occ / S3) * 2);
    const size_t x_floats = (size_t)max_tiles * 32 * steps_b * h->NIN, o_floats = (size_t)max_tiles * 32 * 2 * steps_b;
    float* scratch = nullptr;
    hipStream_t cs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scratch), (x_floats + o_floats) * 4));
    int rc = 0;
    auto cleanup = [&] {
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(scratch);
    };
#define FSNP_CAL_CHECK(expr) do { if ((expr) != hipSuccess) { set_error("calibration: %s failed", #expr); cleanup(); return 4; } } while (0)
    FSNP_CAL_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    FSNP_CAL_CHECK(hipEventCreate(&e0));
    FSNP_CAL_CHECK(hipEventCreate(&e1));
    FSNP_CAL_CHECK(hipMemsetAsync(scratch, 0, (x_floats + o_floats) * 4, cs));
    const bool had_pipeline = h->pipeline != 0;
    CostTable t = h->cost;
    if (h->rowtile_ok) {            // ramp the clocks: short launches on an idle chip run at a lower power state than a forward does
        SbPlan warm;
        warm.chunks = {SbChunk{0, 0, h->num_cus_real * 32, h->num_cus_real, 0, 32, 0, 0, 0, 0, 0}};
        warm.total_slots = h->num_cus_real * 32;
        for (int k = 0; k < 2; ++k)         // (the scratch buffers hold steps_b steps of max_tiles tiles: stay inside them)
            (void)run_dense_plan(h, warm, scratch, scratch + x_floats, h->num_cus_real * 32, steps_b, cs);
    }
    // one shape: returns its per-step cost in microseconds (< 0 on failure)
    auto time_shape = [&](SbChunk c) -> double {
        c.row0 = 0; c.nrows = c.num_tiles * c.rps; c.slot0 = 0; c.coop_tile0 = 0;
        SbPlan plan;
        plan.chunks = {c}; plan.total_slots = c.num_tiles * c.rps; plan.coop_tiles = (c.kind == 1 || c.kind == 2 || c.kind == 8 || c.kind == 9) ? c.num_tiles : 0;
        double ms[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            const int steps = k == 0 ? steps_a : steps_b;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {            // rep 0 warms caches / code / clocks; the faster of reps 1, 2 counts
                if (hipEventRecord(e0, cs) != hipSuccess) return -1.0;
                if (run_dense_plan(h, plan, scratch, scratch + x_floats, c.nrows, steps, cs)) return -1.0;
                if (hipEventRecord(e1, cs) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
                float f = 0;
                if (hipEventElapsedTime(&f, e0, e1) != hipSuccess) return -1.0;
                if (rep > 0 && f < best) best = f;
            }
            ms[k] = best;
        }
        return (ms[1] - ms[0]) * 1000.0 / (steps_b - steps_a);
    };
    (void)had_pipeline;
    for (int o = 0; o < occ && rc == 0; ++o) {
        const int slots = h->num_cus_real * (o + 1);
        for (int ui = 0; ui < 4 && rc == 0; ++ui) {
            const int u = 8 << ui, S = h->H / u;
            int tiles = slots / S;
            if (o == 1 && (tiles * S <= h->num_cus_real || h->occ_ksplit[ui] < 2)) continue;   // never planned two per CU
            if (tiles <= 0) continue;
            const double us = time_shape(SbChunk{1, 0, 0, tiles, 0, 32, u, 0, 0, 0, 0});
            if (us < 0) rc = 4; else t.ksplit[ui][o] = us;
            if (o == 0 && rc == 0) {
                const double u1 = tiles > 1 ? time_shape(SbChunk{1, 0, 0, 1, 0, 32, u, 0, 0, 0, 0}) : us;
                if (u1 < 0) rc = 4; else t.ksplit1[ui] = u1 < us ? u1 : us;
            }
        }
        for (int rpg = 1; rpg <= 2 && rc == 0; ++rpg) {
            const int groups = slots / S3;
            if (groups <= 0 || (o == 1 && h->occ_coopn[rpg - 1] < 2)) continue;
            const double us = time_shape(SbChunk{2, 0, 0, groups * rpg, 0, 32, 0, groups, rpg, 0, 0});
            if (us < 0) rc = 4; else t.coopn[rpg - 1][o] = us;
        }
    }
    if (rc == 0 && h->rowtile_ok) {
        const double us0 = time_shape(SbChunk{0, 0, 0, h->num_cus_real, 0, 32, 0, 0, 0, 0, 0});
        if (us0 < 0) rc = 4;
        else t.rowtile = us0;           // the VALU-row surcharge keeps its measured ratio (0.11 per row)
    }
    if (rc == 0 && h->hp_ok && h->num_cus_real / (h->H / 16) > 0) {
        const int cap = h->num_cus_real / (h->H / 16);
        const double u1 = time_shape(SbChunk{8, 0, 0, 1, 0, 32, 16, 0, 0, 0, 0}), uf = time_shape(SbChunk{8, 0, 0, cap, 0, 32, 16, 0, 0, 0, 0});
        if (u1 < 0 || uf < 0) rc = 4; else { t.hp[0] = u1; t.hp[1] = uf; }
    }
    for (int nt = 1; nt <= 3 && rc == 0 && h->coopw_ok; ++nt) {
        const int u = 32 * nt, cap = h->num_cus_real / (h->H / u);
        if (cap <= 0) continue;
        const double u1 = time_shape(SbChunk{9, 0, 0, 1, 0, 32, u, 0, 0, 0, 0}), uf = cap > 1 ? time_shape(SbChunk{9, 0, 0, cap, 0, 32, u, 0, 0, 0, 0}) : u1;
        if (u1 < 0 || uf < 0) rc = 4; else { t.coopw[nt - 1][0] = u1; t.coopw[nt - 1][1] = uf; }
    }
    if (rc == 0 && h->lstm16_ok) {
        const double us16 = time_shape(SbChunk{4, 0, 0, h->num_cus_real, 0, 16, 0, 0, 0, 0, 0});
        if (us16 < 0) rc = 4; else t.rowtile16 = us16;
    }
    cleanup();
#undef FSNP_CAL_CHECK
    if (rc) { if (g_last_error.empty()) set_error("calibration of the sub-band planner failed"); return rc; }
    if ((*reinterpret_cast<volatile unsigned*>(h->d_err) & kErrTimeout) != 0) {
        // a calibration launch gave up (its workgroups were not all resident): never plan two workgroups per CU
        *reinterpret_cast<volatile unsigned*>(h->d_err) &= ~kErrTimeout;
        t = default_costs();
        h->coop_occ = 1;
    }
    t.calibrated = 1;
    if (measured) *measured = t;
    if (adopt) h->cost = t;
    std::lock_guard<std::mutex> lk(g_cal_mu);
    g_cal_cache[key] = t;
    return 0;
}

// ---- fsnp_set_verify: the sequences a plan hands to column-split kernels, run again on the one-tile-per-CU kernel (no inter-workgroup
// exchange at all) into a scratch mask and compared on the device.  The kernels sum K in different orders, so "equal" is a tolerance:
// |a - b| <= 1e-4 + 1e-3 |b| (kernel-to-kernel differences are ~1e-6; one corrupted exchange element moves the mask by 1e-2 and more).
__global__ __launch_bounds__(256) void verify_compare_kernel(const float* __restrict__ out, const float* __restrict__ ref, const RowDesc* __restrict__ rows,
                                                             int num_slots, int T, int OC, long stride_o, unsigned* err_host,
                                                             unsigned long long* first_key) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)num_slots * OC * T) return;
    const int t = (int)(i % T), o = (int)((i / T) % OC), slot = (int)(i / ((long)OC * T));
    const RowDesc rd = rows[slot];
    if (!rd.valid) return;
    const size_t at = (size_t)rd.out_off + (size_t)o * stride_o + t;
    const float a = out[at], b = ref[at];
    if (!(fabsf(a - b) <= 1e-4f + 1e-3f * fabsf(b))) {
        // 20 bits of utterance, 20 of bin, 24 of frame: nothing the entry points accept wraps (the 8 / 10 / 14-bit key did)
        const unsigned long long key = ((unsigned long long)(unsigned)rd.b << 44) | ((unsigned long long)((unsigned)rd.f & 0xFFFFFu) << 24) |
                                       (unsigned long long)((unsigned)t & 0xFFFFFFu);
        __hip_atomic_fetch_min(first_key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(err_host, kErrVerify, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static bool plan_has_exchange(const SbPlan& plan) {
    for (const SbChunk& c : plan.chunks) if (c.kind == 1 || c.kind == 2 || c.kind == 8 || c.kind == 9) return true;
    return false;
}
// `a` = the arguments of the forward's own sub-band launches (its rows / md_row belong to `plan`); out_elems = floats of the mask tensor
static int verify_pass(fsnp_handle* h, const SbPlan& plan, const Dims& d, int mode, int batch_offset, int global_batch, const LstmArgs& a,
                       const SubbandBuffers& sbuf, size_t out_elems, hipStream_t s) {
    const bool cumulative = h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAPLACE || h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAYER;
    // the exchange-free plan of exactly the column-split chunks' sequences; consecutive chunks (B = 8: three launches over rows
    // [0, 2056)) are re-run as ONE range - one round of half tiles instead of one per launch
    SbPlan vp;
    PlannerCtx pc = pctx(h);
    pc.lstm_coop = 0;
    pc.half_tiles_without_coop = true;
    pc.ih_bf16 = 0;                          // (the re-run is fp32 whatever fsnp_set_precision says: below)
    bool bad = false;
    int run0 = 0, run_n = 0;
    auto flush = [&]() {
        if (run_n == 0) return;
        const SbPlan one = plan_sb(pc, run_n);
        for (SbChunk k : one.chunks) {
            if (k.kind != 0 && k.kind != 4) { bad = true; return; }
            k.row0 += run0; k.slot0 = vp.total_slots; k.coop_tile0 = 0;
            vp.total_slots += k.num_tiles * k.rps;
            vp.chunks.push_back(k);
        }
        run_n = 0;
    };
    for (const SbChunk& c : plan.chunks) {
        const bool exch = c.kind == 1 || c.kind == 2 || c.kind == 8 || c.kind == 9;
        if (exch && run_n > 0 && c.row0 == run0 + run_n) { run_n += c.nrows; continue; }
        flush();
        if (exch) { run0 = c.row0; run_n = c.nrows; }
    }
    flush();
    if (bad) { set_error("fsnp_set_verify: no exchange-free kernel for this model"); return 2; }
    if (vp.chunks.empty()) return 0;
    const size_t out_b = align_up(out_elems * 4, 256), rows_b = align_up((size_t)vp.total_slots * sizeof(RowDesc), 256);
    const size_t md_b = cumulative ? align_up((size_t)vp.total_slots * d.Tp * sizeof(NormMD), 256) : 0;
    const size_t need = out_b + rows_b + md_b;
    if (need > h->verify_bytes) {
        if (h->verify_out) FSNP_HIP_CHECK(hipFreeAsync(h->verify_out, s));
        h->verify_out = nullptr; h->verify_bytes = 0;
        FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&h->verify_out), need, s));
        h->verify_bytes = need;
    }
    unsigned char* vb = reinterpret_cast<unsigned char*>(h->verify_out);
    RowDesc* vrows = reinterpret_cast<RowDesc*>(vb + out_b);
    NormMD* vmd = cumulative ? reinterpret_cast<NormMD*>(vb + out_b + rows_b) : nullptr;
    launch_build_rows(vp, vrows, h->F, d.T, mode, batch_offset, global_batch, 0, h->cfg.num_groups_in_drop_band, h->cfg.output_size, s);
    if (cumulative) {
        SubbandBuffers vs = sbuf;
        vs.md_row = vmd;
        launch_subband_stats(d, h->cfg.norm_type, vs, vrows, vp.total_slots, s);      // (cumulative norms: per-slot tables only)
    }
    LstmArgs va = a;
    va.rows = vrows; va.md_row = vmd; va.out = reinterpret_cast<float*>(vb);
    // the column-split kernels are fp32 in every precision mode, so their check is too: under fsnp_set_precision(h, 1) the row-tile /
    // half-tile kernels would otherwise run their bf16 ih-GEMM here and differ from a CORRECT exchange by ~1e-3 - a false alarm
    const int keep_bf16 = h->lw.ih_bf16;
    h->lw.ih_bf16 = 0;
    launch_sb_lstm(h, vp, va, nullptr, nullptr, nullptr, s);
    h->lw.ih_bf16 = keep_bf16;
    const long n = (long)vp.total_slots * h->cfg.output_size * d.T;
    if (!h->verify_key) FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->verify_key), 8));
    FSNP_HIP_CHECK(hipMemsetAsync(h->verify_key, 0xFF, 8, s));
    hipLaunchKernelGGL(verify_compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.out, reinterpret_cast<const float*>(vb), vrows,
                       vp.total_slots, d.T, h->cfg.output_size, a.out_stride_o, h->d_err, h->verify_key);
    FSNP_HIP_CHECK(hipGetLastError());
    h->verify_runs += 1;
    return 0;
}

// ---- fsnp_set_verify_sample: ONE row tile of a column-split launch, recomputed off the critical path.
// The sampled tile's normalised input and the mask values its launch wrote are SNAPSHOT on the launch's own stream (two tiny kernels);
// the recomputation - the exchange-free half-tile kernel on the snapshot, two workgroups, as long as a whole round: ~13 ms at 2 s clips -
// and the comparison run on a stream of their own and touch nothing but private buffers, so the caller may free / overwrite the mask and
// later forwards may rebuild the workspace meanwhile.  A mismatch flags the handle (code 7) whenever it is found: the report names the
// sampled forward's (utterance, bin, frame), the call that notices it is a later one.
__global__ __launch_bounds__(256) void vs_copy_kernel(const float* __restrict__ out, const RowDesc* __restrict__ rows, RowDesc* __restrict__ rows_copy,
                                                      float* __restrict__ snap, int num_slots, int T, int OC, long stride_o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < num_slots) rows_copy[i] = rows[i];
    if (i >= num_slots * OC * T) return;
    const int t = i % T, o = (i / T) % OC, slot = i / (OC * T);
    const RowDesc rd = rows[slot];
    snap[i] = rd.valid ? out[(size_t)rd.out_off + (size_t)o * stride_o + t] : 0.0f;
}
__global__ __launch_bounds__(256) void vs_compare_kernel(const float* __restrict__ snap, const float* __restrict__ ref, const RowDesc* __restrict__ rows,
                                                         int num_slots, int T, int OC, unsigned* err_host, unsigned long long* first_key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_slots * OC * T) return;
    const int t = i % T, slot = i / (OC * T);
    const RowDesc rd = rows[slot];
    if (!rd.valid) return;
    const float a = snap[i], b = ref[i];                        // ref: the dense run's [slot][o][t]
    if (!(fabsf(a - b) <= 1e-4f + 1e-3f * fabsf(b))) {
        const unsigned long long key = ((unsigned long long)(unsigned)rd.b << 44) | ((unsigned long long)((unsigned)rd.f & 0xFFFFFu) << 24) |
                                       (unsigned long long)((unsigned)t & 0xFFFFFFu);
        __hip_atomic_fetch_min(first_key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(err_host, kErrVerify, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// a plan that can be sampled: column-split launches only, each leaving a CU free ON EVERY XCD.  The recomputation's two workgroups own
// their CUs for ~13 ms (one each on two XCDs), and the dispatcher deals a launch's workgroups to the XCDs by index, not by room: B = 8's
// 252-workgroup launches put 32 on some XCDs - the 32nd waited for the sample to end and the launch (all of whose workgroups must be
// resident) with it: +7.7 ms per sampled forward (profiles/r06_verify_sample.md).  Such plans (B = 3, B = 8) are not sampled.
static bool plan_can_be_sampled(const fsnp_handle* h, const SbPlan& plan) {
    if (plan.chunks.empty() || !h->lstm16_ok || h->num_cus_real % 8 != 0) return false;
    const int cpx = h->num_cus_real / 8;
    for (const SbChunk& c : plan.chunks) {
        const bool exch = c.kind == 1 || c.kind == 2 || c.kind == 8 || c.kind == 9;
        if (!exch || c.rps != 32) return false;
        const int S = c.kind == 8 ? h->H / 16 : (c.kind == 1 || c.kind == 9) ? h->H / c.units : h->H / 128;
        const int T = c.kind == 2 ? c.groups : c.num_tiles;
        const int local = xcd_local_blocks_per_xcd(S, T, cpx);
        const int per_xcd = local <= cpx ? local : cdiv(T * S, 8);       // (launch_sb_lstm: XCD-local placement where it fits, else round robin)
        if (per_xcd > cpx - 1) return false;
    }
    return true;
}
// `st` = the stream the plan's launches were enqueued on (the caller's, or the side stream in the pipelined loop)
static int verify_sample(fsnp_handle* h, const SbPlan& plan, const Dims& d, const LstmArgs& a, const SubbandBuffers& sbuf, hipStream_t st) {
    if (h->vs_busy) {
        const hipError_t q = hipEventQuery(h->ev_vs_done);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); h->vs_skipped += 1; return 0; }      // the previous sample is still being recomputed
        h->vs_busy = false;
    }
    const long long n = h->vs_runs;
    const SbChunk& c = plan.chunks[(size_t)(n % (long long)plan.chunks.size())];
    const int tile = (int)((n / (long long)plan.chunks.size()) % c.num_tiles), nslots = 32;
    const int OC = h->cfg.output_size, T = d.T, Tp = d.Tp, NIN = h->NIN;
    const size_t x_b = align_up((size_t)nslots * Tp * NIN * 4, 256), o_b = align_up((size_t)nslots * OC * T * 4, 256);
    const size_t r_b = align_up((size_t)nslots * sizeof(RowDesc), 256);
    const size_t need = x_b + 2 * o_b + 2 * r_b + 256;
    if (!h->vs_stream) {
        FSNP_HIP_CHECK(hipStreamCreateWithFlags(&h->vs_stream, hipStreamNonBlocking));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_vs_snap, hipEventDisableTiming));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_vs_done, hipEventDisableTiming));
    }
    if (need > h->vs_bytes) {                 // (no sample in flight here: the buffer is idle)
        if (h->vs_buf) FSNP_HIP_CHECK(hipFreeAsync(h->vs_buf, st));
        h->vs_buf = nullptr; h->vs_bytes = 0;
        FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&h->vs_buf), need, st));
        h->vs_bytes = need;
    }
    float* vx = reinterpret_cast<float*>(h->vs_buf);
    float* vsnap = reinterpret_cast<float*>(h->vs_buf + x_b);
    float* vref = reinterpret_cast<float*>(h->vs_buf + x_b + o_b);
    RowDesc* vrows_src = reinterpret_cast<RowDesc*>(h->vs_buf + x_b + 2 * o_b);
    RowDesc* vrows = reinterpret_cast<RowDesc*>(h->vs_buf + x_b + 2 * o_b + r_b);
    unsigned long long* vkey = reinterpret_cast<unsigned long long*>(h->vs_buf + x_b + 2 * o_b + 2 * r_b);
    const size_t slot0 = (size_t)c.slot0 + (size_t)tile * c.rps;
    // ---- snapshot, behind the launches it samples
    SbGatherArgs ga{};
    ga.att_mag = a.att_mag; ga.fb_rel = a.fb_rel; ga.fb_branch_stride = a.fb_branch_stride;
    ga.rows = a.rows + slot0; ga.md_utt = a.md_utt; ga.md_row = a.md_row ? a.md_row + slot0 * Tp : nullptr;
    ga.x = vx; ga.xstride = NIN;
    ga.num_slots = nslots; ga.Tp = Tp; ga.FP = d.FP; ga.F = d.F; ga.NSBN = h->cfg.sb_num_neighbors; ga.NIN = NIN; ga.NFBN = h->cfg.fb_num_neighbors;
    launch_sb_gather(ga, st);
    const int nout = nslots * OC * T;
    hipLaunchKernelGGL(vs_copy_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, st, a.out, a.rows + slot0, vrows_src, vsnap, nslots, T, OC, a.out_stride_o);
    FSNP_HIP_CHECK(hipMemsetAsync(vkey, 0xFF, 8, st));
    FSNP_HIP_CHECK(hipEventRecord(h->ev_vs_snap, st));
    // ---- recomputation and comparison, on their own stream
    hipStream_t vs = h->vs_stream;
    FSNP_HIP_CHECK(hipStreamWaitEvent(vs, h->ev_vs_snap, 0));
    hipLaunchKernelGGL(build_rows_kernel, dim3(1), dim3(256), 0, vs, vrows, nslots, 2, 16, 1, T, 0, 0, 1, 1, 0, 2, OC);
    LstmArgs va{};
    va.rows = vrows; va.dense = vx; va.dense_stride = NIN; va.out = vref; va.out_stride_o = T;
    va.num_rows = nslots; va.num_tiles = 2; va.Tp = Tp; va.LA = d.LA; va.FP = 0; va.F = 1; va.NSBN = 0; va.act = h->cfg.sb_act;
    va.coop_own_cu = 1;                      // its two workgroups share the chip with the following forwards: they own their CUs (lstm16.hip: OWN)
    const int keep_bf16 = h->lw.ih_bf16;     // the column-split kernels are fp32 in every precision mode: so is their check
    h->lw.ih_bf16 = 0;
    launch_lstm16(h->lw, va, vs);
    h->lw.ih_bf16 = keep_bf16;
    hipLaunchKernelGGL(vs_compare_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, vs, vsnap, vref, vrows_src, nslots, T, OC, h->d_err, vkey);
    FSNP_HIP_CHECK(hipGetLastError());
    FSNP_HIP_CHECK(hipEventRecord(h->ev_vs_done, vs));
    h->vs_busy = true;
    h->vs_runs += 1;
    h->verify_key_sampled = vkey;
    return 0;
}

}  // namespace fsnp

extern "C" {

const char* fsnp_last_error(void) { return g_last_error.c_str(); }
const char* fsnp_version(void) { return "fsnp-hip 0.2 (gfx950)"; }
int32_t fsnp_abi_version(void) { return FSNP_ABI_VERSION; }
int32_t fsnp_config_size(void) { return (int32_t)sizeof(fsnp_config); }

int fsnp_create(const fsnp_config* cfg, fsnp_handle** out) {
    if (!cfg || !out) { set_error("fsnp_create: null argument"); return 1; }
    *out = nullptr;
    if (cfg->num_freqs < 2) { set_error("num_freqs must be >= 2"); return 2; }
    if (cfg->look_ahead < 0) { set_error("look_ahead must be >= 0"); return 2; }
    if (cfg->sb_num_neighbors < 0 || cfg->fb_num_neighbors < 0) { set_error("sb_num_neighbors / fb_num_neighbors must be >= 0"); return 2; }
    if (cfg->num_groups_in_drop_band < 1) { set_error("num_groups_in_drop_band must be >= 1"); return 2; }
    // output_size is a constructor argument of the reference (fullsubnet_plus.py:30,104,206: the sub-band model's Linear(H, output_size)
    // and the final reshape); every configuration file uses 2 (the cIRM) and the tuned kernels fuse exactly that Linear(H, 2).  Other
    // values run the sub-band recurrence on the runtime-sized kernel (lstm_generic.hip: OUT is a run-time argument there).
    if (cfg->output_size < 1 || cfg->output_size > 64) { set_error("output_size must be in [1, 64]"); return 2; }
    if (cfg->model == FSNP_MODEL_FULLSUBNET && cfg->output_size != 2) { set_error("FullSubNet has no output_size argument (fullsubnet.py:13-26): 2"); return 2; }
    if (cfg->sb_hidden < 1 && cfg->sequence_model != FSNP_SEQ_TCN) { set_error("sb_model_hidden_size must be >= 1"); return 2; }
    if (cfg->num_tcn_blocks < 0 || cfg->num_tcn_blocks > 8) { set_error("num_tcn_blocks must be in [0,8]"); return 2; }
    if (cfg->model != FSNP_MODEL_FULLSUBNET && cfg->tcn_hidden % 64 != 0) { set_error("tcn_hidden must be a multiple of 64"); return 2; }      // (TCN GEMM tiles; the reference hard-codes 512)
    if (cfg->norm_type < 0 || cfg->norm_type > 3) { set_error("unknown norm_type %d", cfg->norm_type); return 2; }
    if (cfg->attention < 0 || cfg->attention > 3) { set_error("unknown attention model %d", cfg->attention); return 2; }
    if (cfg->model != FSNP_MODEL_FULLSUBNET_PLUS && cfg->model != FSNP_MODEL_FULLSUBNET) { set_error("unknown model %d", cfg->model); return 2; }
    if (cfg->sequence_model < FSNP_SEQ_LSTM || cfg->sequence_model > FSNP_SEQ_TCN) { set_error("unknown sequence_model %d", cfg->sequence_model); return 2; }
    const bool fsn = cfg->model == FSNP_MODEL_FULLSUBNET;
    const int subband_num = cfg->subband_num > 0 ? cfg->subband_num : 1;
    if (cfg->subband_num < 0) { set_error("subband_num must be >= 1"); return 2; }
    if (subband_num > 1 && (fsn || cfg->attention != FSNP_ATT_ECA)) {
        set_error("subband_num > 1 needs channel_attention_model = ECA (the reference's other attention layers fail on it: "
                  "fullsubnet_plus.py:47-50,155-163)");
        return 2;
    }
    if (subband_num > 1 && subband_num - cfg->num_freqs % subband_num >= cfg->num_freqs) { set_error("subband_num too large for num_freqs (reflect pad)"); return 2; }
    if (fsn && cfg->sequence_model == FSNP_SEQ_TCN) { set_error("FullSubNet only supports GRU and LSTM"); return 2; }
    if (fsn && cfg->tcn_hidden < 1) { set_error("fb_model_hidden_size must be >= 1"); return 2; }
    const int nin = 2 * cfg->sb_num_neighbors + 1 + (fsn ? 1 : 3) * (2 * cfg->fb_num_neighbors + 1);
    // sizes without a tuned (MFMA) instantiation run on the runtime-sized kernel (lstm_generic.hip) - as long as one sequence's
    // state fits a CU's LDS
    const bool generic_sb = cfg->sequence_model != FSNP_SEQ_TCN && ((cfg->sb_hidden != 384 && cfg->sb_hidden != 256 && cfg->sb_hidden != 512) || nin > 64 || cfg->output_size != 2);
    if (cfg->sequence_model == FSNP_SEQ_TCN && cfg->output_size > nin) {      // the sub-band TCN's final Linear runs as an nin-column GEMM
        set_error("sequence_model=TCN: output_size %d exceeds the sub-band input width %d (not supported)", cfg->output_size, nin);
        return 2;
    }
    const bool generic_fb = fsn && (cfg->tcn_hidden != 512 || cfg->num_freqs > 264);
    if (generic_sb && lstm_generic_rows_per_group(cfg->sb_hidden, nin, 1, 1) == 0) { set_error("sb_model_hidden_size %d is too large for the runtime-sized kernel (LDS)", cfg->sb_hidden); return 2; }
    if (generic_fb && lstm_generic_rows_per_group(cfg->tcn_hidden, cfg->num_freqs, 1, 1) == 0) { set_error("fb_model_hidden_size %d is too large for the runtime-sized kernel (LDS)", cfg->tcn_hidden); return 2; }
    if (cfg->num_freqs <= cfg->sb_num_neighbors || cfg->num_freqs <= cfg->fb_num_neighbors) { set_error("num_freqs must exceed the neighbour counts (reflect pad)"); return 2; }
    for (int c = 0; c < 3; ++c)
        if (cfg->kersize[c] < 1 || cfg->kersize[c] > 16) { set_error("kersize must be in [1,16]"); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device visible: libfsnp_hip needs an MI355X (gfx950); there is no CPU fallback");
        return 3;
    }
    int dev = 0;
    FSNP_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FSNP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libfsnp_hip is built for gfx950 only", dev, prop.gcnArchName);
        return 3;
    }
    fsnp_handle* h = new fsnp_handle();
    h->cfg = *cfg;
    h->device = dev;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->num_cus_real = h->num_cus;
    h->generic_sb = generic_sb;
    h->generic_fb = generic_fb;
    if (cfg->sequence_model != FSNP_SEQ_TCN && !generic_sb && h->num_cus_real < cfg->sb_hidden / 8) {
        // the column-split kernels need at least one group of workgroups resident (GRU has no other kernel)
        set_error("device %d exposes %d compute units; the sub-band recurrent kernels need at least %d", dev, h->num_cus_real, cfg->sb_hidden / 8);
        delete h;
        return 3;
    }
    h->model = cfg->model;
    h->gru = cfg->sequence_model == FSNP_SEQ_GRU;
    // one-tile-per-CU kernels: LSTM lstm.hip (H = 384, and 256 without VALU rows), GRU lstm_gru.hip (384); other sizes run on
    // the column-split kernels only
    h->rowtile_ok = !generic_sb && (cfg->sb_hidden == 384 || (cfg->sb_hidden == 256 && !h->gru));
    {
        const char* e16 = getenv("FSNP_LSTM16");           // 0 = never plan the half-tile kernel
        h->lstm16_ok = !(e16 && e16[0] == '0') && !generic_sb && !h->gru && cfg->sequence_model == FSNP_SEQ_LSTM && cfg->sb_hidden == 384 && nin <= 40;
    }
    h->cost = initial_costs(cfg->sb_hidden, h->gru, cfg->sequence_model == FSNP_SEQ_TCN);
    const char* ce = getenv("FSNP_CALIBRATE");
    if (ce && ce[0] == '1') h->calibrate = 1;
    const char* oe = getenv("FSNP_COOP_OCC");          // 1 = never plan two column-split workgroups per CU
    h->coop_occ = oe && oe[0] == '1' ? 1 : 2;          // (2 is confirmed against the kernels' occupancy at commit time)
    h->sb_tcn = cfg->sequence_model == FSNP_SEQ_TCN;
    h->XS = (int)align_up(nin, 4);
    h->NG = h->gru ? 3 : 4;
    h->NFB = fsn ? 1 : 3;
    h->F = cfg->num_freqs;
    h->FP = (int)align_up(cfg->num_freqs, 4);
    h->CH = cfg->tcn_hidden;
    h->H = cfg->sb_hidden;
    h->NSB = 2 * cfg->sb_num_neighbors + 1;
    h->NIN = nin;
    h->KX = nin <= 40 ? 40 : 64;       // input width the recurrent kernels are instantiated for (zero-padded K)
    h->NB = fsn ? 0 : cfg->num_tcn_blocks;
    h->Fr = cfg->num_freqs / 2;
    build_specs(h);
    const char* cp = getenv("FSNP_LSTM_COOP");
    if (cp && cp[0] == '0') h->lstm_coop = 0;
    {
        // (The round-3 ping-pong K-split kernel lstm_pp.hip - opt-in, ahead of the other kernels at exactly 10 row tiles - was removed in
        // round 4; its measurements stay in profiles/r03_column_split.md and profiles/r03_pp_*.txt.)
        // The half-tile ping-pong kernel (lstm_hp.hip) is planned wherever the cost table says it pays (6 ... 10 row tiles: B = 1);
        // FSNP_COOP_HP=0: never
        const char* he = getenv("FSNP_COOP_HP");
        h->coop_hp = he && he[0] == '0' ? 0 : 1;
        h->coop_hp_cfg = h->coop_hp;
        const char* hw = getenv("FSNP_HP_WAVE");           // 0 = kind-8 launches on lstm_hp.hip (round 3) instead of lstm_hpw.hip (round 6)
        h->hp_wave = hw && hw[0] == '0' ? 0 : 1;
        h->hp_ok = !generic_sb && cfg->sequence_model == FSNP_SEQ_LSTM && (cfg->sb_hidden == 384 || cfg->sb_hidden == 256);
    }
    {
        const char* ve = getenv("FSNP_VERIFY_EVERY");      // N > 0: fsnp_set_verify(h, N) from the start (models with an exchange-free kernel)
        if (ve && atoi(ve) > 0 && h->rowtile_ok && !generic_sb && cfg->sequence_model != FSNP_SEQ_TCN) h->verify_every = atoi(ve);
        const char* we = getenv("FSNP_COOP_W");           // 0 = never plan the wave-owned column split (lstm_coopw.hip)
        h->coop_w = we && we[0] == '0' ? 0 : 1;
        h->coopw_ok = !generic_sb && cfg->sequence_model == FSNP_SEQ_LSTM && cfg->sb_hidden == 384;
    }
    const char* sk = getenv("FSNP_COOP_SKEW");
    if (sk && sk[0] == '0') h->coop_skew = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->d_err), 256, hipHostMallocMapped) != hipSuccess) {
        set_error("hipHostMalloc of the error word failed");
        delete h;
        return 4;
    }
    memset(h->d_err, 0, 256);
    if (hipMalloc(reinterpret_cast<void**>(&h->d_clk), 64) != hipSuccess || hipMemset(h->d_clk, 0, 64) != hipSuccess) {
        set_error("hipMalloc of the clock stamps failed");
        (void)hipHostFree(h->d_err);
        delete h;
        return 4;
    }
    const char* dbg = getenv("FSNP_DEBUG_STAGES");
    h->debug = dbg && dbg[0] == '1';
    *out = h;
    return 0;
}

void fsnp_destroy(fsnp_handle* h) {
    if (!h) return;
    fsnp::DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    if (h->ws) (void)hipFreeAsync(h->ws, nullptr);        // (allocated from the stream-ordered pool; the device is idle here)
    if (h->io) (void)hipFreeAsync(h->io, nullptr);
    (void)hipDeviceSynchronize();
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->d_stft) (void)hipFree(h->d_stft);
    if (h->d_weights) (void)hipFree(h->d_weights);
    drop_weight_watch(h);
    if (h->verify_out) (void)hipFree(h->verify_out);
    if (h->verify_key) (void)hipFree(h->verify_key);
    if (h->d_clk) (void)hipFree(h->d_clk);
    if (h->vs_buf) (void)hipFree(h->vs_buf);
    if (h->vs_stream) (void)hipStreamDestroy(h->vs_stream);
    if (h->ev_vs_snap) (void)hipEventDestroy(h->ev_vs_snap);
    if (h->ev_vs_done) (void)hipEventDestroy(h->ev_vs_done);
    if (h->d_err) (void)hipHostFree(h->d_err);
    for (auto& r : h->timing_recs)
        for (auto& e : r.e) (void)hipEventDestroy(e);
    for (auto& e : h->event_pool) (void)hipEventDestroy(e);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    for (auto& e : h->ev_side) if (e) (void)hipEventDestroy(e);
    delete h;
}

size_t fsnp_workspace_bytes(const fsnp_handle* h, int32_t batch, int32_t frames, int32_t mode) {
    if (!h || batch <= 0 || frames <= 0) return 0;
    return plan_workspace(h, batch, frames, mode).total;
}

static int forward_impl(fsnp_handle* h, const float* mag, const float* real, const float* imag, bool is_complex,
                        const int64_t strides[3][3], float* out, int32_t batch, int32_t frames,
                        int32_t mode, int32_t batch_offset, int32_t global_batch, void* hip_stream) {
    if (batch <= 0 || frames <= 0) { set_error("fsnp_forward: empty input (B=%d, T=%d)", batch, frames); return 2; }
    if (!h || !mag || !out || !strides) { set_error("fsnp_forward: null argument"); return 1; }
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    if (!fsn && !is_complex && (!real || !imag)) { set_error("fsnp_forward: null argument (FullSubNet+ takes mag, real and imag)"); return 1; }
    if (!h->committed) { set_error("fsnp_forward: weights not committed (call fsnp_commit_weights)"); return 2; }
    if (const int ec = take_device_errors(h, "an earlier forward on this handle failed")) return ec;     // set by a launch that has finished since
    if (mode != FSNP_MODE_FULL && mode != FSNP_MODE_PARITY) { set_error("unknown mode %d", mode); return 2; }
    if (mode == FSNP_MODE_PARITY && (h->cfg.num_groups_in_drop_band < 2 || global_batch <= h->cfg.num_groups_in_drop_band)) {
        set_error("PARITY mode needs num_groups_in_drop_band >= 2 and a global batch larger than it (feature.py:263)");
        return 2;
    }
    if (batch_offset < 0 || global_batch < batch_offset + batch) { set_error("bad batch_offset/global_batch"); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    Dims d;
    d.B = batch; d.T = frames; d.LA = h->cfg.look_ahead; d.Tp = frames + d.LA; d.F = h->F; d.FP = h->FP;
    d.CH = h->CH; d.H = h->H; d.NSB = h->NSB; d.NIN = h->NIN;
    int kmax = 1;
    for (int c = 0; c < 3; ++c) kmax = kmax > h->cfg.kersize[c] ? kmax : h->cfg.kersize[c];
    if (!fsn && h->cfg.attention == FSNP_ATT_TSSE && d.Tp < kmax) { set_error("too few frames: T + look_ahead = %d < largest TSSE kernel %d", d.Tp, kmax); return 2; }
    if ((double)3 * d.B * d.Tp * d.FP * 2 > 2.0e9) { set_error("batch too large for 32-bit gather offsets; split the batch"); return 2; }
    const int fb_units = (fsn && !h->generic_fb) ? fb_coop_units(h, batch) : 0;
    if (fsn && !h->generic_fb && fb_units == 0) { set_error("FullSubNet: at most %d utterances per call (full-band LSTM residency); split the batch", 32 * (h->num_cus_real / 16)); return 2; }

    FSNP_ON_DEVICE(h);
    if (calibrate_costs(h)) return 4;            // first planning call of the process for this kind of handle only (~0.1 s)
    const int num_rows = batch * rows_per_utt(h, mode);
    const SbPlan plan = plan_sb(h, num_rows, forward_gather_bytes(h, batch, frames));
    if (plan.chunks.empty()) {
        set_error("no kernel plan for %d sub-band sequences on this device (%d CUs): the %s sub-band model needs at least %d",
                  num_rows, h->num_cus_real, h->gru ? "GRU" : "LSTM", h->H / 128);
        return 2;
    }
    const Workspace w = plan_workspace(h, batch, frames, mode);
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_workspace(h, w.total, s)) return 4;
    // pipelined mode: alternate between the two workspace halves; the half about to be rebuilt was last read by the
    // deferred remainder chunks of the forward before the previous one
    const int slot = h->pipeline ? h->ws_slot : 0;
    if (h->pipeline) {
        h->ws_slot ^= 1;
        if (h->side_used[slot]) FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_side[slot], 0));
    }
    unsigned char* base = h->ws + (size_t)slot * h->ws_bytes;
    auto fptr = [&](size_t off) { return reinterpret_cast<float*>(base + off); };

    TimingRec rec{};
    if (h->timing) {
        if (take_timing_rec(h, rec)) return 4;
        h->timing_recs.push_back(rec);          // owned by the handle from here on (no leak on an early return)
        FSNP_HIP_CHECK(hipEventRecord(rec.e[0], s));
    }
    const int num_slots = plan.total_slots;
    RowDesc* rows = reinterpret_cast<RowDesc*>(base + w.rows);
    const bool cumulative = h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAPLACE || h->cfg.norm_type == FSNP_NORM_CUMULATIVE_LAYER;
    SubbandBuffers sbuf;
    sbuf.att_mag = fptr(w.att); sbuf.fb = fptr(w.fb); sbuf.refl_w = h->d_refl_w; sbuf.refl_wfb = h->d_refl_wfb;
    sbuf.NFBN = h->cfg.fb_num_neighbors;
    sbuf.acc = reinterpret_cast<double*>(base + w.sb_acc);
    sbuf.md_utt = reinterpret_cast<NormMD*>(base + w.md_utt);
    sbuf.md_row = cumulative ? reinterpret_cast<NormMD*>(base + w.md_row) : nullptr;
    // the prologue of both models, one launch: zero the accumulators / exchange images / counters, describe the sub-band rows, and
    // - fsnp_watch_weights - fingerprint the caller's source tensors (a .data edit since the pack flags the handle)
    auto prologue = [&](hipStream_t st) -> int {
        const bool watch = h->watch_nseg > 0 && (h->watch_calls++ % h->watch_every) == 0;
        if (plan.chunks.size() > 8) {                     // (more chunks than the argument block holds: the separate kernels)
            launch_zero_region(base + w.zero_begin, w.zero_end - w.zero_begin, st);
            launch_build_rows(plan, rows, h->F, frames, mode, batch_offset, global_batch, 0, h->cfg.num_groups_in_drop_band, h->cfg.output_size, st);
            return watch ? launch_weight_watch(h, st, false) : 0;
        }
        PrologueArgs pa{};
        pa.zero = reinterpret_cast<uint4*>(base + w.zero_begin);
        pa.n16 = (w.zero_end - w.zero_begin) / 16;
        pa.zero_blocks = (int)std::min<size_t>((pa.n16 + 255) / 256, 2048);
        pa.rows = rows;
        for (const SbChunk& c : plan.chunks) {
            PrologueChunk& k = pa.chunk[pa.nchunks++];
            k.slot0 = c.slot0; k.nrows = c.nrows; k.tiles = c.num_tiles; k.rt = c.rps; k.row0 = c.row0; k.blocks = cdiv(c.num_tiles * c.rps, 256);
            pa.rows_blocks += k.blocks;
        }
        pa.F = h->F; pa.T = frames; pa.mode = mode; pa.batch_offset = batch_offset; pa.global_batch = global_batch;
        pa.groups = h->cfg.num_groups_in_drop_band; pa.OC = h->cfg.output_size;
        pa.segs = static_cast<const WatchSeg*>(h->watch_segs); pa.nseg = watch ? h->watch_nseg : 0;
        pa.watch_blocks = watch ? std::min(h->watch_nseg, kWatchBlocks) : 0;
        pa.watch_acc = h->watch_acc; pa.err_host = h->d_err;
        hipLaunchKernelGGL(prologue_kernel, dim3(pa.zero_blocks + pa.rows_blocks + pa.watch_blocks), dim3(256), 0, st, pa);
        return 0;
    };
    if (prologue(s)) return 4;

    if (!fsn) {
        FrontendBuffers fbuf;
        fbuf.raw = fptr(w.raw); fbuf.frame = reinterpret_cast<double*>(base + w.frame);
        fbuf.md = reinterpret_cast<NormMD*>(base + w.md); fbuf.fsum = reinterpret_cast<double*>(base + w.fsum);
        fbuf.tot = reinterpret_cast<double*>(base + w.fe_tot);
        fbuf.gate = fptr(w.gate); fbuf.att = fptr(w.att);
        const float* in[3] = {mag, real, imag};
        TcnBuffers tbuf;
        tbuf.att = fptr(w.att); tbuf.x = fptr(w.x); tbuf.y1 = fptr(w.y1); tbuf.y2 = fptr(w.y2);
        tbuf.gn = reinterpret_cast<double*>(base + w.gn); tbuf.fb = fptr(w.fb);
        tbuf.dbg_tcn0 = h->debug ? fptr(w.dbg_tcn0) : nullptr;
        // the caller's tensors are read by the repack kernel only; everything up to the LSTM then stays in the workspace
        launch_frontend(d, h->cfg.norm_type, in, strides, is_complex, h->fw, fbuf, s);
        launch_tcn(d, h->cfg.fb_act, h->tw, tbuf, s);
        launch_subband_stats(d, h->cfg.norm_type, sbuf, rows, num_slots, s);
    } else {
        // fullsubnet.py:82-90: pad, norm(noisy_mag), 2-layer LSTM(F -> CH), Linear(CH, F) + fb_act
        FrontendBuffers fbuf{};
        fbuf.raw = fptr(w.att); fbuf.frame = reinterpret_cast<double*>(base + w.frame);
        fbuf.md = reinterpret_cast<NormMD*>(base + w.md);
        launch_frontend_mag(d, h->cfg.norm_type, mag, strides[0], is_complex, fbuf, s);
        // (runtime-sized kernel for a full-band model no K-split instantiation exists for: workgroups of fb_rg sequences)
        const int fb_rg = h->generic_fb ? lstm_generic_rows_per_group(h->CH, h->F, batch, h->num_cus_real) : 32;
        const int fb_tiles = h->generic_fb ? cdiv(batch, fb_rg) : fb_row_tiles(batch);
        RowDesc* fb_rows = reinterpret_cast<RowDesc*>(base + w.fb_rows);
        hipLaunchKernelGGL(build_rows_kernel, dim3(cdiv(fb_tiles * fb_rg, 256)), dim3(256), 0, s, fb_rows, batch, fb_tiles, fb_rg,
                           1, frames, 0, 0, 1, 1, 0, 2, 2);
        LstmArgs fa{};
        fa.rows = fb_rows; fa.dense = fptr(w.att); fa.dense_stride = d.FP; fa.md_seq = fbuf.md;
        fa.seq_out = fptr(w.y1);
        fa.num_rows = batch; fa.num_tiles = fb_tiles; fa.Tp = d.Tp; fa.LA = 0; fa.FP = d.FP; fa.F = d.F;
        fa.coop_hx = fptr(w.fb_hx); fa.coop_bar = reinterpret_cast<unsigned*>(base + w.fb_bar); fa.coop_err = h->d_err;
        fa.coop_abort = reinterpret_cast<unsigned*>(base + w.coop_abort) + 16;
        fa.coop_units = fb_units; fa.coop_chaos = h->coop_chaos;
        const int chp = (int)align_up(d.CH, 4);           // row stride of the h1 sequence (a float4 multiple; pad columns written as zeros)
        fa.seq_stride = chp;
        if (h->generic_fb) { fa.coop_rows_per_group = fb_rg; launch_lstm_generic(h->fbw, fa, true, s); }
        else if (h->fb_valu && lstm_fbv_available(h->fbw, batch, h->num_cus_real)) launch_coop_chained(h->device, s, [&] { launch_lstm_fbv(h->fbw, fa, s); });   // B <= 4: VALU
        else launch_coop_chained(h->device, s, [&] { launch_lstm_coop_seq(h->fbw, fa, s); }, h, fa.coop_units > 0 ? fb_tiles * (h->CH / fa.coop_units) : 0, h->num_cus_real);
        launch_linear_act(fptr(w.y1), chp, h->fsn_wf, h->fsn_kp, h->fsn_bf, fptr(w.fb), d.FP, d.CH, d.F, d.B, d.Tp,
                          h->cfg.fb_act, h->num_cus, s);
        launch_subband_stats(d, h->cfg.norm_type, sbuf, rows, num_slots, s);
    }
    if (h->timing) FSNP_HIP_CHECK(hipEventRecord(rec.e[1], s));

    if (h->sb_tcn) {
        // sequence_model="TCN": materialise the normalised sub-band input, run the TCN stack over the sub-band sequences
        // (one "utterance" per sequence: GroupNorm(1, 512) statistics are per sequence), Linear(34, 2), scatter
        SbGatherArgs ga{};
        ga.att_mag = fptr(w.att); ga.fb_rel = (int)((w.fb - w.att) / 4); ga.fb_branch_stride = d.B * d.Tp * d.FP;
        ga.rows = rows; ga.md_utt = sbuf.md_utt; ga.md_row = sbuf.md_row;
        ga.x = fptr(w.sbt_x0); ga.xstride = h->XS;
        ga.num_slots = num_slots; ga.Tp = d.Tp; ga.FP = d.FP; ga.F = d.F; ga.NSBN = h->cfg.sb_num_neighbors; ga.NIN = h->NIN;
        ga.NFBN = h->cfg.fb_num_neighbors;
        launch_sb_gather(ga, s);
        Dims ds = d;
        ds.B = num_slots; ds.F = h->NIN; ds.FP = h->XS;
        TcnBuffers tb{};
        tb.att = fptr(w.sbt_x0); tb.x = fptr(w.sbt_x); tb.y1 = fptr(w.sbt_y1); tb.y2 = fptr(w.sbt_y2);
        tb.gn = reinterpret_cast<double*>(base + w.sbt_gn); tb.fb = fptr(w.sbt_fb);
        h->sbt.num_cus = h->num_cus;
        launch_tcn(ds, h->cfg.sb_act, h->sbt, tb, s, 1);
        launch_sb_scatter(fptr(w.sbt_fb), h->XS, rows, out, (long)rows_per_utt(h, mode) * frames, num_slots, d.Tp, d.LA, h->cfg.output_size, s);
        if (h->timing) {
            FSNP_HIP_CHECK(hipEventRecord(rec.e[2], s));
            FSNP_HIP_CHECK(hipEventRecord(rec.e[3], s));
        }
        FSNP_HIP_CHECK(hipGetLastError());
        h->last_ws = w; h->last_dims = d; h->have_last = true; h->last_base = base;
        return mark_forward_done(h, s);
    }
    LstmArgs a{};
    a.att_mag = fptr(w.att); a.fb = fptr(w.fb);
    a.fb_rel = (int)((w.fb - w.att) / 4);
    a.fb_branch_stride = d.B * d.Tp * d.FP;
    a.rows = rows; a.md_utt = sbuf.md_utt; a.md_row = sbuf.md_row; a.dense = nullptr;
    a.out = out;
    a.out_stride_o = (long)rows_per_utt(h, mode) * frames;
    a.num_rows = num_rows; a.Tp = d.Tp; a.LA = d.LA; a.FP = d.FP; a.F = d.F; a.NSBN = h->cfg.sb_num_neighbors;
    a.NFBN = h->cfg.fb_num_neighbors;
    a.act = h->cfg.sb_act;
    unsigned* bar = reinterpret_cast<unsigned*>(base + w.coop_bar);
    unsigned* abort_word = reinterpret_cast<unsigned*>(base + w.coop_abort);
    // pipelined mode: column-split remainder chunks behind a row-tile chunk go to the side stream (after the row-tile
    // chunk: next to it they would only fight for its CUs), where they overlap the next forward's full-band stages
    // A plan that STARTS with a column-split launch (small batches) goes to the side stream whole: those launches leave CUs
    // idle too (B = 1: 216 of 256 busy, latency-bound), and the next forward's full-band stages fit beside them.
    int ndefer = 0;
    bool defer_all = false;
    // fsnp_set_verify: every Nth forward with a column-split launch is checked against the exchange-free kernel (nothing deferred then)
    const bool verify_now = h->verify_every > 0 && plan_has_exchange(plan) && (h->verify_calls++ % h->verify_every) == 0;
    // fsnp_set_verify_sample: every Nth forward of a plan made of column-split launches only, one row tile is recomputed off the critical path
    const bool sample_now = !verify_now && h->vs_every > 0 && plan_can_be_sampled(h, plan) && (h->vs_calls++ % h->vs_every) == 0;
    a.coop_corrupt = h->corrupt_exchange; h->corrupt_exchange = 0;
    if (verify_now) {
        launch_sb_lstm(h, plan, a, fptr(w.coop_hx), bar, abort_word, s, h->timing ? rec.e[3] : nullptr);
        if (h->timing) FSNP_HIP_CHECK(hipEventRecord(rec.e[2], s));
        const size_t out_elems = (size_t)(mode == FSNP_MODE_PARITY ? global_batch : batch) * h->cfg.output_size * rows_per_utt(h, mode) * frames;
        if (const int vr = verify_pass(h, plan, d, mode, batch_offset, global_batch, a, sbuf, out_elems, s)) return vr;
    } else if (h->pipeline) {
        const int first = plan_first_deferred(h, plan);
        if (first == 0) defer_all = true;
        else if (first < (int)plan.chunks.size()) ndefer = first;
    }
    if (verify_now) {
        // (launched above)
    } else if (defer_all) {
        FSNP_HIP_CHECK(hipEventRecord(h->ev_main, s));
        FSNP_HIP_CHECK(hipStreamWaitEvent(h->side_stream, h->ev_main, 0));
        launch_sb_lstm(h, plan, a, fptr(w.coop_hx), bar, abort_word, h->side_stream, h->timing ? rec.e[3] : nullptr);
        if (h->timing) FSNP_HIP_CHECK(hipEventRecord(rec.e[2], h->side_stream));
        FSNP_HIP_CHECK(hipEventRecord(h->ev_side[slot], h->side_stream));
        h->side_used[slot] = true;
    } else if (ndefer == 0) {
        launch_sb_lstm(h, plan, a, fptr(w.coop_hx), bar, abort_word, s, h->timing ? rec.e[3] : nullptr);
        if (h->timing) FSNP_HIP_CHECK(hipEventRecord(rec.e[2], s));
    } else {
        launch_sb_lstm(h, plan, a, fptr(w.coop_hx), bar, abort_word, s, (h->timing && ndefer > 1) ? rec.e[3] : nullptr, 0, ndefer);
        if (h->timing && ndefer == 1) FSNP_HIP_CHECK(hipEventRecord(rec.e[3], s));      // "after the first chunk"
        FSNP_HIP_CHECK(hipEventRecord(h->ev_main, s));
        FSNP_HIP_CHECK(hipStreamWaitEvent(h->side_stream, h->ev_main, 0));
        launch_sb_lstm(h, plan, a, fptr(w.coop_hx), bar, abort_word, h->side_stream, nullptr, ndefer, -1);
        if (h->timing) FSNP_HIP_CHECK(hipEventRecord(rec.e[2], h->side_stream));
        FSNP_HIP_CHECK(hipEventRecord(h->ev_side[slot], h->side_stream));
        h->side_used[slot] = true;
    }
    if (sample_now) {        // (a plan of column-split launches only runs on ONE stream: the side stream when the pipelined loop defers it, else s)
        if (const int vr = verify_sample(h, plan, d, a, sbuf, defer_all ? h->side_stream : s)) return vr;
        if (defer_all) FSNP_HIP_CHECK(hipEventRecord(h->ev_side[slot], h->side_stream));       // the snapshot reads this workspace half too
    }
    FSNP_HIP_CHECK(hipGetLastError());
    h->last_ws = w; h->last_dims = d; h->have_last = true; h->last_base = base;
    return mark_forward_done(h, s);
}

int fsnp_forward(fsnp_handle* h, const float* mag, const float* real, const float* imag,
                 const int64_t strides[3][3], float* out, int32_t batch, int32_t frames,
                 int32_t mode, int32_t batch_offset, int32_t global_batch, void* hip_stream) {
    return forward_impl(h, mag, real, imag, false, strides, out, batch, frames, mode, batch_offset, global_batch, hip_stream);
}

int fsnp_forward_complex(fsnp_handle* h, const float* noisy, const int64_t strides[3], float* out, int32_t batch,
                         int32_t frames, int32_t mode, int32_t batch_offset, int32_t global_batch, void* hip_stream) {
    if (!strides) { set_error("fsnp_forward_complex: null argument"); return 1; }
    int64_t st[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) st[i][j] = strides[j];
    return forward_impl(h, noisy, nullptr, nullptr, true, st, out, batch, frames, mode, batch_offset, global_batch, hip_stream);
}

int fsnp_reserve(fsnp_handle* h, int32_t max_batch, int32_t max_frames, int32_t mode, int32_t max_samples, void* hip_stream) {
    if (!h || max_batch <= 0 || max_frames <= 0 || max_samples < 0) { set_error("fsnp_reserve: bad argument"); return 1; }
    if (mode != FSNP_MODE_FULL && mode != FSNP_MODE_PARITY) { set_error("unknown mode %d", mode); return 2; }
    if (!h->committed) { set_error("fsnp_reserve: weights not committed (the plan depends on the kernels' occupancy)"); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    if (order_after_last_forward(h, s)) return 4;
    if (calibrate_costs(h)) return 4;            // FSNP_CALIBRATE=1: size the workspace for the plans the ADOPTED cost table will make
    // every batch size up to max_batch plans its own launches (the exchange region of a column-split plan can exceed the one of
    // the chip-filling batch): take the largest workspace over all of them - host arithmetic only
    size_t need = 0;
    for (int b = 1; b <= max_batch; ++b) {
        if (mode == FSNP_MODE_PARITY && b <= h->cfg.num_groups_in_drop_band) continue;
        need = std::max(need, plan_workspace(h, b, max_frames, mode).total);
    }
    if (ensure_workspace(h, need, s)) return 4;
    if (max_samples > 0) {
        if (h->cfg.output_size != 2) { set_error("fsnp_reserve: max_samples > 0 sizes the waveform path, which needs output_size = 2 (this handle: %d)", h->cfg.output_size); return 2; }
        if (ensure_stft(h)) return 2;
        const StftPlan p = stft_plan(h);
        const int T = 1 + max_samples / p.hop;
        const long xs = (long)align_up((size_t)max_samples + p.n_fft, 4);
        const size_t xp_b = align_up((size_t)max_batch * xs * 4, 256), spec_b = align_up((size_t)max_batch * T * p.sp * 4 + 256, 256);
        const size_t mask_b = align_up((size_t)max_batch * 2 * p.F * T * 4, 256), fr_b = (size_t)max_batch * T * p.n_fft * 4;
        if (ensure_io(h, xp_b + 2 * spec_b + mask_b + fr_b, s)) return 4;
    }
    return mark_forward_done(h, s);
}

int fsnp_apply_cirm(const float* mask, const float* noisy, const int64_t strides[3], float* out,
                    const int64_t out_strides[3], int32_t batch, int32_t freqs, int32_t frames, void* hip_stream) {
    if (!mask || !noisy || !out || !strides || !out_strides) { set_error("fsnp_apply_cirm: null argument"); return 1; }
    if (batch <= 0 || freqs <= 0 || frames <= 0) { set_error("fsnp_apply_cirm: empty input"); return 2; }
    launch_apply_cirm(mask, noisy, strides, out, out_strides, batch, freqs, frames, static_cast<hipStream_t>(hip_stream));
    FSNP_HIP_CHECK(hipGetLastError());
    return 0;
}

int fsnp_lstm2_fc(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps, void* hip_stream) {
    if (!h || !x || !out) { set_error("fsnp_lstm2_fc: null argument"); return 1; }
    if (!h->committed) { set_error("fsnp_lstm2_fc: weights not committed"); return 2; }
    if (h->sb_tcn) { set_error("fsnp_lstm2_fc: the sub-band model of this handle is a TCN (no recurrent kernel)"); return 2; }
    if (num_seq <= 0 || steps <= 0) { set_error("fsnp_lstm2_fc: empty input"); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    if ((double)num_seq * steps * h->NIN > 2.0e9) { set_error("fsnp_lstm2_fc: input too large for 32-bit offsets"); return 2; }
    if (calibrate_costs(h)) return 4;
    const SbPlan plan = plan_sb(h, num_seq, (double)num_seq * steps * h->NIN * 4.0);
    if (plan.chunks.empty()) { set_error("fsnp_lstm2_fc: no kernel plan for %d sequences on this device", num_seq); return 2; }
    return run_dense_plan(h, plan, x, out, num_seq, steps, s);
}

// ---- the forward's submodules as stages (include/fsnp.h: "third kind")
static int stage_args_ok(const char* fn, const fsnp_handle* h, int32_t branch, const float* in, const int64_t* strides, const float* out, int32_t B, int32_t T) {
    if (!h || !in || !strides || !out) { set_error("%s: null argument", fn); return 1; }
    if (!h->committed) { set_error("%s: weights not committed", fn); return 4; }
    if (h->model != FSNP_MODEL_FULLSUBNET_PLUS) { set_error("%s: FullSubNet+ handles only (the original FullSubNet has no attention layer and a recurrent full-band model)", fn); return 2; }
    if (branch < 0 || branch > 2) { set_error("%s: branch must be 0 (magnitude), 1 (real) or 2 (imaginary)", fn); return 2; }
    if (B <= 0 || T <= 0) { set_error("%s: empty input", fn); return 2; }
    if ((double)B * T * std::max(h->FP, h->CH) * 4.0 > 2.0e9) { set_error("%s: input too large for 32-bit offsets", fn); return 2; }
    return 0;
}
static Dims stage_dims(const fsnp_handle* h, int B, int T) {
    Dims d{};
    d.B = B; d.T = T; d.Tp = T; d.F = h->F; d.FP = h->FP; d.CH = h->CH; d.H = h->H; d.NIN = h->NIN; d.LA = 0;
    return d;
}

int fsnp_channel_attention(fsnp_handle* h, int32_t branch, const float* in, const int64_t strides[3], float* out, int32_t batch,
                           int32_t frames, void* hip_stream) {
    if (const int rc = stage_args_ok("fsnp_channel_attention", h, branch, in, strides, out, batch, frames)) return rc;
    if (branch == 0 && h->fw.subband_num > 1) {
        set_error("fsnp_channel_attention: with subband_num > 1 the magnitude branch's layer runs on the tensor the forward regroups around it (fullsubnet_plus.py:146-153): not a stage");
        return 2;
    }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const Dims d = stage_dims(h, batch, frames);
    const size_t plane = align_up((size_t)d.B * d.Tp * d.FP * 4, 256), md_b = align_up((size_t)d.B * d.Tp * sizeof(NormMD), 256);
    const size_t fsum_b = align_up((size_t)d.B * d.FP * 8, 256), tot_b = align_up((size_t)d.B * 2 * 8, 256), gate_b = align_up((size_t)d.B * d.FP * 4, 256);
    unsigned char* work = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&work), 2 * plane + md_b + fsum_b + tot_b + gate_b, s));
    FrontendBuffers buf{};
    buf.raw = reinterpret_cast<float*>(work); buf.att = reinterpret_cast<float*>(work + plane);
    buf.md = reinterpret_cast<NormMD*>(work + 2 * plane);
    buf.fsum = reinterpret_cast<double*>(work + 2 * plane + md_b);
    buf.tot = reinterpret_cast<double*>(work + 2 * plane + md_b + fsum_b);
    buf.gate = reinterpret_cast<float*>(work + 2 * plane + md_b + fsum_b + tot_b);
    FSNP_HIP_CHECK(hipMemsetAsync(buf.fsum, 0, fsum_b + tot_b, s));
    FrontendWeights w = h->fw;                      // the branch's weights in slot 0 (the kernels index by grid.y)
    for (int k = 0; k < 3; ++k) { w.conv_w[0][k] = h->fw.conv_w[branch][k]; w.conv_b[0][k] = h->fw.conv_b[branch][k]; }
    w.cat_w[0] = h->fw.cat_w[branch]; w.cat_b[0] = h->fw.cat_b[branch];
    w.fc1_wT[0] = h->fw.fc1_wT[branch]; w.fc1_b[0] = h->fw.fc1_b[branch];
    w.fc2_wT[0] = h->fw.fc2_wT[branch]; w.fc2_b[0] = h->fw.fc2_b[branch];
    launch_attention_stage(d, w, in, strides, buf, s);
    launch_tm_to_bft(buf.att, out, d.B, d.T, d.Tp, d.F, d.FP, s);
    FSNP_HIP_CHECK(hipGetLastError());
    FSNP_HIP_CHECK(hipFreeAsync(work, s));
    return 0;
}

int fsnp_fullband_model(fsnp_handle* h, int32_t branch, const float* in, const int64_t strides[3], float* out, int32_t batch,
                        int32_t frames, void* hip_stream) {
    if (const int rc = stage_args_ok("fsnp_fullband_model", h, branch, in, strides, out, batch, frames)) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const Dims d = stage_dims(h, batch, frames);
    const TcnWeights& t = h->tw;
    const size_t plane = align_up((size_t)d.B * d.Tp * d.FP * 4, 256), yplane = align_up((size_t)d.B * d.Tp * d.CH * 4, 256);
    const size_t gn_b = align_up((size_t)std::max(t.NB, 1) * 2 * d.B * kGnStride * 8, 256);
    unsigned char* work = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&work), 3 * plane + 2 * yplane + gn_b, s));
    TcnBuffers tb{};
    float* att = reinterpret_cast<float*>(work);
    tb.att = att; tb.x = reinterpret_cast<float*>(work + plane); tb.fb = reinterpret_cast<float*>(work + 2 * plane);
    tb.y1 = reinterpret_cast<float*>(work + 3 * plane); tb.y2 = reinterpret_cast<float*>(work + 3 * plane + yplane);
    tb.gn = reinterpret_cast<double*>(work + 3 * plane + 2 * yplane);
    FSNP_HIP_CHECK(hipMemsetAsync(tb.gn, 0, gn_b, s));
    // (the running activation's pad columns [F, FP) meet zero weights in every GEMM but must be finite: fresh pool memory is not)
    FSNP_HIP_CHECK(hipMemsetAsync(tb.x, 0, 2 * plane, s));
    TcnWeights wb = t;                              // this branch's slice of every [3][...] array (strides: launch_tcn)
    const long nb = t.NB, br = branch;
    wb.w1 += br * nb * t.N1P * t.K1P; wb.b1 += br * nb * t.N1P; wb.a1 += br * nb;
    wb.g1w += br * nb * d.CH; wb.g1b += br * nb * d.CH; wb.dw += br * nb * 3 * d.CH; wb.db += br * nb * d.CH; wb.a2 += br * nb;
    wb.g2w += br * nb * d.CH; wb.g2b += br * nb * d.CH;
    wb.w2 += br * nb * t.N2P * t.K2P; wb.b2 += br * nb * t.N2P;
    if (wb.w2g) { wb.w2g += br * nb * t.N2P * t.K2P; wb.c1 += br * nb * t.N2P; wb.c2 += br * nb * t.N2P; }
    wb.wf += br * (long)t.N2P * t.K1P; wb.bf += br * t.N2P;
    launch_repack_plane(d, in, strides, att, s);
    launch_tcn(d, h->cfg.fb_act, wb, tb, s, 1);
    launch_tm_to_bft(tb.fb, out, d.B, d.T, d.Tp, d.F, d.FP, s);
    FSNP_HIP_CHECK(hipGetLastError());
    FSNP_HIP_CHECK(hipFreeAsync(work, s));
    return 0;
}

int fsnp_read_stage(fsnp_handle* h, const char* name, float* host_out, int64_t numel) {
    if (!h || !name || !host_out) { set_error("fsnp_read_stage: null argument"); return 1; }
    if (!h->have_last) { set_error("fsnp_read_stage: no forward has run"); return 2; }
    const Dims& d = h->last_dims;
    const Workspace& w = h->last_ws;
    const size_t plane = (size_t)d.B * d.Tp * d.FP;       // one branch, padded rows
    const std::string n = name;
    const float* src = nullptr;
    bool is_gate = false;
    static const char* tags[3] = {"mag", "real", "imag"};
    for (int b = 0; b < h->NFB; ++b) {
        if (n == std::string("att_") + tags[b]) src = reinterpret_cast<float*>(h->last_base + w.att) + b * plane;
        if (n == std::string("fb_") + tags[b]) src = reinterpret_cast<float*>(h->last_base + w.fb) + b * plane;
        if (h->NFB == 3 && n == std::string("gate_") + tags[b]) { src = reinterpret_cast<float*>(h->last_base + w.gate) + (size_t)b * d.B * d.FP; is_gate = true; }
    }
    if (n == "tcn0_mag") {
        if (!h->debug) { set_error("tcn0_mag needs FSNP_DEBUG_STAGES=1 at fsnp_create time"); return 2; }
        src = reinterpret_cast<float*>(h->last_base + w.dbg_tcn0);
    }
    if (!src) { set_error("unknown stage %s", name); return 2; }
    const int64_t rows = is_gate ? d.B : (int64_t)d.B * d.Tp;
    if (numel != rows * d.F) { set_error("stage %s has %lld elements, caller asked %lld", name, (long long)(rows * d.F), (long long)numel); return 2; }
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    FSNP_HIP_CHECK(hipMemcpy2D(host_out, (size_t)d.F * 4, src, (size_t)d.FP * 4, (size_t)d.F * 4, (size_t)rows, hipMemcpyDeviceToHost));
    return 0;
}

int fsnp_set_timing(fsnp_handle* h, int32_t enable) {
    if (!h) { set_error("null handle"); return 1; }
    h->timing = enable != 0;
    return 0;
}

int fsnp_get_timing(fsnp_handle* h, double ms[4], int64_t count[4], int32_t reset) {
    if (!h || !ms || !count) { set_error("fsnp_get_timing: null argument"); return 1; }
    FSNP_ON_DEVICE(h);
    if (drain_timing(h)) return 1;
    for (int i = 0; i < 4; ++i) { ms[i] = h->acc_ms[i]; count[i] = h->acc_cnt[i]; }
    if (reset) for (int i = 0; i < 4; ++i) { h->acc_ms[i] = 0; h->acc_cnt[i] = 0; }
    return 0;
}

int fsnp_debug_plan_rows(int32_t num_rows, int32_t num_cus, int32_t hidden, int32_t gru, int32_t coop, double composite_gain,
                         int32_t* out, int32_t max_chunks) {
    return fsnp_debug_plan_rows2(num_rows, num_cus, hidden, gru, coop, composite_gain, 1, nullptr, out, max_chunks);
}

int fsnp_debug_plan_rows2(int32_t num_rows, int32_t num_cus, int32_t hidden, int32_t gru, int32_t coop, double composite_gain,
                          int32_t workgroups_per_cu, const double* costs, int32_t* out, int32_t max_chunks) {
    if (!out || num_rows <= 0 || num_cus <= 0 || hidden < 128 || hidden % 128 != 0 || max_chunks <= 0) { set_error("fsnp_debug_plan_rows: bad argument"); return -1; }
    PlannerCtx h;                       // host-only: the planner never touches the device
    h.H = hidden; h.num_cus = num_cus; h.num_cus_real = num_cus; h.gru = gru; h.lstm_coop = coop; h.composite_gain = composite_gain;
    h.cost = default_costs(); h.coop_occ = workgroups_per_cu >= 2 ? 2 : 1;
    for (int i = 0; i < 4; ++i) h.occ_ksplit[i] = h.coop_occ;
    for (int i = 0; i < 2; ++i) h.occ_coopn[i] = h.coop_occ;
    if (costs) costs_from_array(h.cost, costs);
    h.hp_ok = gru == 0 && (hidden == 384 || hidden == 256);
    h.coop_hp = 1;
    h.coopw_ok = gru == 0 && hidden == 384;
    h.coop_w = 1;
    h.lstm16_ok = gru == 0 && hidden == 384;
    h.rowtile_ok = gru == 0 || gru == 2;      // gru = 1: plan as if there were no one-tile-per-CU GRU kernel (round-1 shape)
    h.gru = gru != 0;
    if (gru == 2) h.cost.rowtile *= 0.75;
    const SbPlan plan = plan_sb(h, num_rows);
    int n = 0;
    for (const SbChunk& c : plan.chunks) {
        if (n >= max_chunks) break;
        int32_t* o = out + 8 * n;
        o[0] = c.kind; o[1] = c.row0; o[2] = c.nrows; o[3] = c.num_tiles; o[4] = c.ex; o[5] = (c.kind == 1 || c.kind == 9) ? c.units : c.groups;
        o[6] = c.rpg; o[7] = c.slot0;
        ++n;
    }
    return n;
}

int fsnp_get_costs(const fsnp_handle* h, double out[FSNP_NUM_COSTS], int32_t* calibrated, int32_t* occ) {
    if (!h || !out) { set_error("fsnp_get_costs: null argument"); return 1; }
    static_assert(kNumCosts == FSNP_NUM_COSTS, "planner.h and fsnp.h agree on the flat table");
    costs_to_array(h->cost, out);
    if (calibrated) *calibrated = h->cost.calibrated;
    if (occ) *occ = h->coop_occ;
    return 0;
}

int fsnp_measure_costs(fsnp_handle* h, double out[FSNP_NUM_COSTS]) {
    if (!h || !out) { set_error("fsnp_measure_costs: null argument"); return 1; }
    if (!h->committed) { set_error("fsnp_measure_costs: weights not committed"); return 2; }
    if (h->sb_tcn) { set_error("fsnp_measure_costs: the sub-band model of this handle is a TCN (no recurrent kernels)"); return 2; }
    FSNP_ON_DEVICE(h);
    CostTable t = h->cost;
    if (calibrate_costs(h, false, &t)) return 4;
    costs_to_array(t, out);
    return 0;
}

int fsnp_debug_set_costs(fsnp_handle* h, const double* costs, int32_t workgroups_per_cu) {
    if (!h || (workgroups_per_cu != 1 && workgroups_per_cu != 2)) { set_error("fsnp_debug_set_costs: bad argument"); return 1; }
    if (!h->committed) { set_error("fsnp_debug_set_costs: commit the weights first (the kernels' occupancy is checked then)"); return 2; }
    h->cost = initial_costs(h->H, h->gru != 0, h->sb_tcn != 0);
    if (costs) costs_from_array(h->cost, costs);
    h->cost.calibrated = 1;          // pinned: the lazy calibration will not replace it
    h->coop_occ = workgroups_per_cu;
    return 0;
}

int fsnp_describe_plan(const fsnp_handle* h, int32_t batch, int32_t mode, int32_t* out, int32_t max_chunks) {
    if (!h || !out || batch <= 0 || max_chunks <= 0) { set_error("fsnp_describe_plan: bad argument"); return -1; }
    const SbPlan plan = plan_sb(h, batch * rows_per_utt(h, mode));
    int n = 0;
    for (const SbChunk& c : plan.chunks) {
        if (n >= max_chunks) break;
        // kind 4 = half-tile kernel, 11 = runtime-sized kernel, 12 = half-tile ping-pong (lstm_hp.hip), 13 = wave-owned column split (lstm_coopw.hip),
        // 14 = half-tile ping-pong on its wave-owned kernel (lstm_hpw.hip: where launch_sb_lstm runs planner kind 8 by default)
        const bool hpw = h->lw.hp_wave && lstm_hpw_available(h->lw);
        out[4 * n + 0] = h->sb_tcn ? 3 : (c.kind == 7 ? 11 : c.kind == 8 ? (hpw ? 14 : 12) : c.kind == 9 ? 13 : c.kind);
        out[4 * n + 1] = c.nrows; out[4 * n + 2] = c.num_tiles; out[4 * n + 3] = c.ex;
        ++n;
    }
    return n;
}

int fsnp_describe_plan_ex(const fsnp_handle* h, int32_t batch, int32_t mode, int32_t* out, int32_t max_chunks) {
    if (!h || !out || batch <= 0 || max_chunks <= 0) { set_error("fsnp_describe_plan_ex: bad argument"); return -1; }
    int32_t base[4 * 64];
    const int n = fsnp_describe_plan(h, batch, mode, base, max_chunks < 64 ? max_chunks : 64);
    if (n < 0) return n;
    const SbPlan plan = plan_sb(h, batch * rows_per_utt(h, mode));
    const int first_deferred = plan_first_deferred(h, plan);
    for (int i = 0; i < n; ++i) {
        const SbChunk& c = plan.chunks[i];
        for (int k = 0; k < 4; ++k) out[7 * i + k] = base[4 * i + k];
        // arithmetic of THIS chunk: the bf16 variants exist for the one-tile-per-CU LSTM kernel only (lstm.hip) and the half-tile kernel (lstm16.hip);
        // sequences that the plan hands to any other kernel run in fp32 whatever fsnp_set_precision says
        int prec = 0;
        if (!h->sb_tcn && !h->gru && c.kind == 0) prec = h->ih_bf16 == 1 ? 1 : 0;
        if (!h->sb_tcn && !h->gru && c.kind == 4 && h->ih_bf16 == 1 && h->lw.wpack16_bf) prec = 1;       // half-tile kernel: bf16 ih-GEMM too (round 4)
        out[7 * i + 4] = prec;
        out[7 * i + 5] = h->sb_tcn ? 0 : chunk_workgroups(h, c);
        out[7 * i + 6] = i >= first_deferred ? 1 : 0;        // the pipelined loop runs this launch on the side stream
    }
    return n;
}

int fsnp_debug_lstm_profile(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps,
                            uint64_t* host_stamps, int64_t num_stamps) {
    if (!h || !x || !out || !host_stamps) { set_error("fsnp_debug_lstm_profile: null argument"); return 1; }
    if (!h->committed) { set_error("fsnp_debug_lstm_profile: weights not committed"); return 2; }
    if (num_stamps != (int64_t)steps * 8) { set_error("fsnp_debug_lstm_profile: need steps*8 stamps"); return 2; }
    if (h->gru || h->sb_tcn || h->H != 384) { set_error("fsnp_debug_lstm_profile: row-tile kernel only (LSTM, hidden 384)"); return 2; }
    FSNP_ON_DEVICE(h);
    const LstmPlan lp = plan_lstm_tiles(num_seq, h->num_cus);
    const int num_slots = lp.num_tiles * lp.rows_per_slot_tile;
    const size_t stamp_off = align_up((size_t)num_slots * sizeof(RowDesc), 256);
    if (ensure_workspace(h, stamp_off + ((size_t)num_stamps + (size_t)lp.num_tiles * 256) * 8, nullptr)) return 4;
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    RowDesc* rows = reinterpret_cast<RowDesc*>(h->ws);
    unsigned long long* dprof = reinterpret_cast<unsigned long long*>(h->ws + stamp_off);
    h->have_last = false;
    hipLaunchKernelGGL(build_rows_kernel, dim3(cdiv(num_slots, 256)), dim3(256), 0, 0, rows, num_seq, lp.num_tiles,
                       lp.rows_per_slot_tile, 1, steps, 0, 0, 1, 1, 0, 2, 2);
    LstmArgs a{};
    a.rows = rows; a.dense = x; a.out = out; a.out_stride_o = steps;
    a.num_rows = num_seq; a.num_tiles = lp.num_tiles; a.ex = lp.ex; a.Tp = steps; a.LA = 0; a.F = 1;
    a.act = h->cfg.sb_act; a.prof = dprof;
    launch_lstm(h->lw, a, 0);
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    FSNP_HIP_CHECK(hipMemcpy(host_stamps, dprof, (size_t)num_stamps * 8, hipMemcpyDeviceToHost));
    return 0;
}

int fsnp_debug_pp_profile(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps, int32_t tiles_per_group,
                          uint64_t* host_stamps, int64_t num_stamps) {
    if (!h || !x || !out || !host_stamps) { set_error("fsnp_debug_pp_profile: null argument"); return 1; }
    if (!h->committed || (!h->hp_ok && tiles_per_group == 0)) { set_error("fsnp_debug_pp_profile: no half-tile ping-pong kernel for this handle"); return 2; }
    // tiles_per_group: 0 = the half-tile ping-pong kernel (lstm_hp.hip: 2 halves x 16 stamps per step); 32 / 64 = the wave-owned column
    // split (lstm_coopw.hip) at that many units per workgroup (16 stamps per step: 8 per layer phase)
    const bool hp = tiles_per_group == 0, cw = tiles_per_group == 32 || tiles_per_group == 64 || tiles_per_group == 96;
    if ((!hp && !cw) || num_stamps != (int64_t)steps * (hp ? 32 : 16)) { set_error("fsnp_debug_pp_profile: tiles_per_group must be 0 (half-tile ping-pong kernel, steps * 32 stamps) or 32 / 64 (wave-owned column split, steps * 16 stamps)"); return 2; }
    if (cw && !h->coopw_ok) { set_error("fsnp_debug_pp_profile: no wave-owned column split for this handle"); return 2; }
    const int tiles = cdiv(num_seq, 32), groups = tiles, S = hp ? h->H / 16 : h->H / tiles_per_group;
    if (num_seq <= 0 || groups * S > h->num_cus_real) { set_error("fsnp_debug_pp_profile: the launch must fit the chip"); return 2; }
    FSNP_ON_DEVICE(h);
    SbPlan plan;
    plan.chunks = {hp ? SbChunk{8, 0, num_seq, tiles, 0, 32, 16, 0, 0, 0, 0} : SbChunk{9, 0, num_seq, tiles, 0, 32, tiles_per_group, 0, 0, 0, 0}};
    plan.total_slots = tiles * 32; plan.coop_tiles = tiles;
    const size_t rows_b = align_up((size_t)plan.total_slots * sizeof(RowDesc), 256), hx_b = align_up(lstm_coop_exchange_bytes(h->H, tiles), 256);
    const size_t bar_b = align_up(coop_counter_bytes(tiles), 256), st_b = (size_t)num_stamps * 8;
    if (order_after_last_forward(h, nullptr)) return 4;
    if (ensure_workspace(h, rows_b + hx_b + bar_b + 256 + st_b, nullptr)) return 4;
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    h->have_last = false;
    RowDesc* rows = reinterpret_cast<RowDesc*>(h->ws);
    FSNP_HIP_CHECK(hipMemsetAsync(h->ws + rows_b, 0, hx_b + bar_b + 256 + st_b, nullptr));
    launch_build_rows(plan, rows, 1, steps, 0, 0, 1, 1, 2, h->cfg.output_size, nullptr);
    LstmArgs a{};
    a.rows = rows; a.dense = x; a.dense_stride = h->NIN; a.out = out; a.out_stride_o = steps;
    a.num_rows = num_seq; a.Tp = steps; a.LA = 0; a.FP = 0; a.F = 1; a.NSBN = 0; a.act = h->cfg.sb_act;
    a.prof = reinterpret_cast<unsigned long long*>(h->ws + rows_b + hx_b + bar_b + 256);
    launch_sb_lstm(h, plan, a, reinterpret_cast<float*>(h->ws + rows_b), reinterpret_cast<unsigned*>(h->ws + rows_b + hx_b),
                   reinterpret_cast<unsigned*>(h->ws + rows_b + hx_b + bar_b), nullptr);
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    FSNP_HIP_CHECK(hipMemcpy(host_stamps, a.prof, st_b, hipMemcpyDeviceToHost));
    return fsnp_check_errors(h);
}

int fsnp_debug_launch_clock(fsnp_handle* h, double out[FSNP_LAUNCH_CLOCK_VALUES]) {
    if (!h || !out) { set_error("fsnp_debug_launch_clock: null argument"); return 1; }
    unsigned long long c[8] = {};
    if (!h->d_clk) { set_error("fsnp_debug_launch_clock: no completed launch of the one-tile-per-CU LSTM kernel on this handle"); return 2; }
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipMemcpy(c, h->d_clk, sizeof(c), hipMemcpyDeviceToHost));
    const unsigned long long t0 = c[0], r0 = c[1], t1 = c[2], r1 = c[3];
    if (r0 == 0 || r1 <= r0 || t1 <= t0) { set_error("fsnp_debug_launch_clock: no completed launch of the one-tile-per-CU LSTM kernel on this handle"); return 2; }
    out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0);
    out[2] = (double)(r1 - r0) * 1e-5;                          // 100 MHz ticks -> ms
    out[3] = (double)(t1 - t0) / (double)(r1 - r0) * 100.0;     // s_memtime ticks per microsecond
    out[4] = (double)c[4] * 1e-5;                               // the slowest workgroup of the launch, ms
    out[5] = c[6] == ~0ull ? 0.0 : (double)c[6] * 1e-5;         // the fastest
    out[6] = (double)c[5];                                      // the largest s_memtime tick count of a workgroup
    return 0;
}

int fsnp_set_precision(fsnp_handle* h, int32_t ih_bf16) {
    if (!h || ih_bf16 < 0 || ih_bf16 > 1) { set_error("fsnp_set_precision: 0 (fp32) or 1 (bf16 ih-GEMM, BASELINE.json configs[4])"); return 1; }
    if (ih_bf16 && (h->gru || h->sb_tcn)) { set_error("fsnp_set_precision: the bf16 ih-GEMM variant exists for the LSTM sub-band model only"); return 2; }
    if (ih_bf16 && h->H != 384) { set_error("fsnp_set_precision: the bf16 variants exist for sb_model_hidden_size = 384 only"); return 2; }
    if (ih_bf16 && (h->KX != 40 || h->NIN >= h->KX)) { set_error("fsnp_set_precision: the bf16 ih-GEMM variant exists for sub-band inputs of <= 39 features only (its layer-0 bias rides in a spare input column)"); return 2; }
    h->ih_bf16 = ih_bf16;
    h->lw.ih_bf16 = ih_bf16 == 1 ? 1 : 0;       // (launch_lstm's own switch: the bf16-ih variant of lstm.hip)
    return 0;
}

int fsnp_poll_errors(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    return take_device_errors(h, "fsnp_poll_errors");
}

int fsnp_set_verify(fsnp_handle* h, int32_t every) {
    if (!h || every < 0) { set_error("fsnp_set_verify: every must be >= 0 (0 = off)"); return 1; }
    if (every > 0 && (!h->rowtile_ok || h->sb_tcn || h->generic_sb)) {
        set_error("fsnp_set_verify: this model has no exchange-free (one-tile-per-CU) kernel to verify against");
        return 2;
    }
    h->verify_every = every; h->verify_calls = 0;
    return 0;
}

int fsnp_set_verify_sample(fsnp_handle* h, int32_t every) {
    if (!h || every < 0) { set_error("fsnp_set_verify_sample: every must be >= 0 (0 = off)"); return 1; }
    if (every == h->vs_every) return 0;            // (idempotent: a binding may state its setting in front of every forward)
    h->vs_every = every; h->vs_calls = 0;
    return 0;
}

int64_t fsnp_verify_count(const fsnp_handle* h) { return h ? (int64_t)h->verify_runs : -1; }

int fsnp_debug_verify_sample_stats(const fsnp_handle* h, int64_t out[3]) {
    if (!h || !out) { set_error("fsnp_debug_verify_sample_stats: null argument"); return 1; }
    out[0] = h->vs_runs; out[1] = h->vs_skipped; out[2] = h->vs_calls;
    return 0;
}

int fsnp_debug_corrupt_exchange(fsnp_handle* h, int32_t step) {
    if (!h || step < 0) { set_error("fsnp_debug_corrupt_exchange: step must be >= 0 (0 = off)"); return 1; }
    h->corrupt_exchange = step;
    return 0;
}

int fsnp_set_pipeline(fsnp_handle* h, int32_t enable) {
    if (!h || (enable != 0 && enable != 1)) { set_error("fsnp_set_pipeline: 0 or 1"); return 1; }
    if (enable == h->pipeline) return 0;
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipDeviceSynchronize());              // nothing of either mode is in flight while the workspace is re-shaped
    if (enable && !h->side_stream) {
        // the deferred remainder chunk is a latency-bound chain of inter-workgroup hand-offs: at the highest stream priority
        // its waves win the arbitration against the full-band GEMMs it shares CUs with
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            FSNP_HIP_CHECK(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, hi));
        else
            FSNP_HIP_CHECK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
        for (auto& e : h->ev_side) FSNP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (h->ws) { FSNP_HIP_CHECK(hipFreeAsync(h->ws, nullptr)); FSNP_HIP_CHECK(hipDeviceSynchronize()); h->ws = nullptr; h->ws_bytes = 0; }
    h->have_last = false;
    h->pipeline = enable;
    h->ws_slots = enable ? 2 : 1;
    h->ws_slot = 0;
    h->side_used[0] = h->side_used[1] = false;
    return 0;
}

int fsnp_flush(fsnp_handle* h, void* hip_stream) {
    if (!h) { set_error("null handle"); return 1; }
    if (!h->pipeline) return 0;
    FSNP_ON_DEVICE(h);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    for (int k = 0; k < 2; ++k)
        if (h->side_used[k]) FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_side[k], 0));
    return 0;
}

int fsnp_check_errors(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    return take_device_errors(h, "fsnp_check_errors");
}

int64_t fsnp_dump_config(const fsnp_handle* h, char* buf, int64_t cap) {
    if (!h) { set_error("null handle"); return -1; }
    auto env = [](const char* n) { const char* e = getenv(n); return e ? e : "(unset)"; };
    std::string o;
    char line[512];
    auto add = [&](const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(line, sizeof(line), fmt, ap);
        va_end(ap);
        o += line;
    };
    add("%s, ABI %d, device %d (%d CUs; planner sees %d)\n", fsnp_version(), FSNP_ABI_VERSION, h->device, h->num_cus_real, h->num_cus);
    add("model=%s num_freqs=%d look_ahead=%d sb_hidden=%d tcn_hidden=%d sub-band inputs=%d (kernels instantiated for K=%d) sequence_model=%s norm_type=%d attention=%d subband_num=%d\n",
        h->model == FSNP_MODEL_FULLSUBNET ? "FullSubNet" : "FullSubNet+", h->F, h->cfg.look_ahead, h->H, h->CH, h->NIN, h->KX,
        h->sb_tcn ? "TCN" : h->gru ? "GRU" : "LSTM", h->cfg.norm_type, h->cfg.attention, h->cfg.subband_num > 0 ? h->cfg.subband_num : 1);
    add("weights committed=%d precision=%d (0 fp32, 1 bf16 ih-GEMM) pipeline=%d timing=%d workspace=%zu bytes x %d\n",
        (int)h->committed, h->ih_bf16, h->pipeline, (int)h->timing, h->ws_bytes, h->ws_slots);
    add("effective settings (environment variable as read at fsnp_create = value in force):\n");
    add("  FSNP_LSTM_COOP=%s -> column-split kernels %s\n", env("FSNP_LSTM_COOP"), h->lstm_coop ? "planned (auto)" : "never");
    add("  FSNP_COOP_HP=%s -> half-tile ping-pong kernel (lstm_hp.hip / lstm_hpw.hip) %s\n", env("FSNP_COOP_HP"), !h->hp_ok ? "not built for this model" : h->coop_hp ? "planned" : "never");
    add("  FSNP_HP_WAVE=%s -> its launches run on %s\n", env("FSNP_HP_WAVE"), h->hp_wave && lstm_hpw_available(h->lw) ? "lstm_hpw.hip (wave-owned units, round 6)" : "lstm_hp.hip (gate-split waves, round 3)");
    add("  FSNP_COOP_W=%s -> wave-owned column split (lstm_coopw.hip) %s\n", env("FSNP_COOP_W"), !h->coopw_ok ? "not built for this model" : h->coop_w ? "planned" : "never");
    add("  FSNP_COOP_SKEW=%s -> K-split schedule %s\n", env("FSNP_COOP_SKEW"), h->coop_skew ? "layer-skewed" : "serial");
    add("  FSNP_COOP_OCC=%s -> column-split workgroups per CU the planner may use: %d\n", env("FSNP_COOP_OCC"), h->coop_occ);
    add("  FSNP_LSTM16=%s -> half-tile kernel %s\n", env("FSNP_LSTM16"), h->lstm16_ok ? "planned" : "not used");
    add("  FSNP_CALIBRATE=%s -> cost table %s\n", env("FSNP_CALIBRATE"), h->cost.calibrated ? "measured / pinned" : h->calibrate ? "to be measured at the first plan" : "built-in");
    add("  FSNP_GEMM_DMA=%s -> %d (0 = the general GEMM kernel everywhere)\n", env("FSNP_GEMM_DMA"), h->tw.gemm_dma);
    add("  FSNP_VERIFY_EVERY=%s -> exchange verification every %d forwards (0 = off)\n", env("FSNP_VERIFY_EVERY"), h->verify_every);
    add("  FSNP_DEBUG_STAGES=%s -> %d\n", env("FSNP_DEBUG_STAGES"), (int)h->debug);
    add("cost table (us per step): K split full %.1f / %.1f / %.1f / %.1f, one tile %.1f / %.1f / %.1f / %.1f, three-way %.1f / %.1f, one tile per CU %.1f (+%.2f per VALU row), half tile %.1f, half-tile ping-pong %.1f / %.1f, wave-owned split full %.1f / %.1f / %.1f, one tile %.1f / %.1f / %.1f\n",
        h->cost.ksplit[0][0], h->cost.ksplit[1][0], h->cost.ksplit[2][0], h->cost.ksplit[3][0], h->cost.ksplit1[0], h->cost.ksplit1[1],
        h->cost.ksplit1[2], h->cost.ksplit1[3], h->cost.coopn[0][0], h->cost.coopn[1][0], h->cost.rowtile, h->cost.rowtile_ex, h->cost.rowtile16,
        h->cost.hp[0], h->cost.hp[1], h->cost.coopw[0][1], h->cost.coopw[1][1], h->cost.coopw[2][1], h->cost.coopw[0][0], h->cost.coopw[1][0], h->cost.coopw[2][0]);
    if (buf && cap > 0) {
        const size_t n = o.size() < (size_t)cap - 1 ? o.size() : (size_t)cap - 1;
        memcpy(buf, o.data(), n);
        buf[n] = 0;
    }
    return (int64_t)o.size() + 1;
}

int fsnp_debug_inject_error(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    *reinterpret_cast<volatile unsigned*>(h->d_err) |= kErrTimeout;
    return 0;
}

int fsnp_debug_set_chaos(fsnp_handle* h, int32_t seed) {
    if (!h) { set_error("null handle"); return 1; }
    h->coop_chaos = seed;
    return 0;
}

int fsnp_debug_set_gemm_dma(fsnp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 3) { set_error("fsnp_debug_set_gemm_dma: mode must be 0 (general GEMM kernel), 1 (DMA kernels where they apply), 2 (as 1, never the small-batch split-K kernel) or 3 (the 128-row DMA kernel only)"); return 1; }
    h->tw.gemm_dma = mode;
    return 0;
}

int fsnp_debug_set_lstm_coop(fsnp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 4 || mode == 3) { set_error("fsnp_debug_set_lstm_coop: mode must be 0 (off), 1 (auto), 2 (auto, serial K-split schedule, no half-tile ping-pong kernel) or 4 (auto + the half-tile ping-pong kernel even where FSNP_COOP_HP=0)"); return 1; }
    h->lstm_coop = mode != 0;
    h->coop_skew = mode != 2;
    h->coop_hp = mode == 4 ? 1 : mode == 1 ? h->coop_hp_cfg : 0;
    h->fb_valu = mode == 1 || mode == 4;   // (FullSubNet: modes 0 and 2 keep the full-band LSTM on the K-split kernel, whatever the batch - mode 0 is
                                           // what the sync error policy retries with after a time-out, so it must not come back to the same exchange)
    h->cost.calibrated = h->calibrate ? 0 : h->cost.calibrated;    // the K-split costs depend on the schedule: measure again
    if (!h->cost.calibrated) h->cost = initial_costs(h->H, h->gru != 0, h->sb_tcn != 0);
    return 0;
}

int fsnp_debug_set_lstm_waves(fsnp_handle* h, int32_t waves) {
    if (!h || (waves != 0 && waves != 4 && waves != 12)) { set_error("fsnp_debug_set_lstm_waves: waves must be 0 (auto), 4 or 12"); return 1; }
    h->lstm_waves = waves;
    h->lw.waves = waves;
    return 0;
}

int fsnp_debug_set_num_cus(fsnp_handle* h, int32_t num_cus) {
    if (!h || num_cus <= 0) { set_error("fsnp_debug_set_num_cus: bad argument"); return 1; }
    h->num_cus = num_cus;
    h->tw.num_cus = num_cus;
    return 0;
}

int fsnp_debug_lstm_pack(int32_t hidden, int32_t input_size, int32_t kx, int32_t waves, const float* wih0, const float* whh0,
                         const float* wih1, const float* whh1, float* out, int64_t out_floats) {
    if (!wih0 || !whh0 || !wih1 || !whh1 || !out) { set_error("fsnp_debug_lstm_pack: null argument"); return 1; }
    if (waves <= 0 || hidden % (32 * waves) != 0 || kx % 8 != 0 || input_size > kx) { set_error("fsnp_debug_lstm_pack: bad sizes"); return 2; }
    if ((int64_t)lstm_pack_floats(hidden, kx, waves) != out_floats) {
        set_error("fsnp_debug_lstm_pack: need %lld floats", (long long)lstm_pack_floats(hidden, kx, waves));
        return 2;
    }
    lstm_pack_weights(hidden, input_size, kx, waves, wih0, whh0, wih1, whh1, out);
    return 0;
}

int fsnp_debug_lstm_coop_pack(int32_t hidden, int32_t input_size, int32_t kx, int32_t units, const float* wih0, const float* whh0,
                              const float* wih1, const float* whh1, float* out, int64_t out_floats) {
    if (!wih0 || !whh0 || !wih1 || !whh1 || !out) { set_error("fsnp_debug_lstm_coop_pack: null argument"); return 1; }
    if ((units != 8 && units != 16 && units != 32 && units != 64) || hidden % 64 != 0 || kx % 8 != 0 || input_size > kx) {
        set_error("fsnp_debug_lstm_coop_pack: bad sizes");
        return 2;
    }
    if ((int64_t)lstm_coop_pack_floats(hidden, kx, units) != out_floats) {
        set_error("fsnp_debug_lstm_coop_pack: need %lld floats", (long long)lstm_coop_pack_floats(hidden, kx, units));
        return 2;
    }
    lstm_coop_pack_weights(hidden, input_size, kx, units, wih0, whh0, wih1, whh1, out);
    return 0;
}

int fsnp_debug_lstm_hpw_pack(int32_t hidden, int32_t input_size, int32_t kx, const float* wih0, const float* whh0, const float* wih1,
                             const float* whh1, float* out, int64_t out_floats) {
    if (!wih0 || !whh0 || !wih1 || !whh1 || !out) { set_error("fsnp_debug_lstm_hpw_pack: null argument"); return 1; }
    if (hidden % 16 != 0 || kx % 4 != 0 || kx > 64 || input_size > kx) { set_error("fsnp_debug_lstm_hpw_pack: bad sizes"); return 2; }
    if (out_floats != (int64_t)lstm_hpw_pack_floats(hidden, kx)) {
        set_error("fsnp_debug_lstm_hpw_pack: need %lld floats", (long long)lstm_hpw_pack_floats(hidden, kx));
        return 2;
    }
    lstm_hpw_pack_weights(hidden, input_size, kx, wih0, whh0, wih1, whh1, out);
    return 0;
}

int fsnp_debug_lstm_coopw_pack(int32_t hidden, int32_t input_size, int32_t kx, const float* wih0, const float* whh0, const float* wih1,
                               const float* whh1, float* out, int64_t out_floats) {
    if (!wih0 || !whh0 || !wih1 || !whh1 || !out) { set_error("fsnp_debug_lstm_coopw_pack: null argument"); return 1; }
    if (hidden % 32 != 0 || kx % 8 != 0 || input_size > kx) { set_error("fsnp_debug_lstm_coopw_pack: bad sizes"); return 2; }
    if ((int64_t)lstm_coopw_pack_floats(hidden, kx) != out_floats) {
        set_error("fsnp_debug_lstm_coopw_pack: need %lld floats", (long long)lstm_coopw_pack_floats(hidden, kx));
        return 2;
    }
    lstm_coopw_pack_weights(hidden, input_size, kx, wih0, whh0, wih1, whh1, out);
    return 0;
}

int fsnp_debug_lstm_fbv_pack(int32_t hidden, int32_t input_size, const float* wih0, const float* whh0, const float* wih1, const float* whh1,
                             float* out, int64_t out_floats) {
    if (!wih0 || !whh0 || !wih1 || !whh1 || !out) { set_error("fsnp_debug_lstm_fbv_pack: null argument"); return 1; }
    if (hidden != 512 || input_size < 1 || input_size > 288) { set_error("fsnp_debug_lstm_fbv_pack: hidden 512, <= 288 inputs"); return 2; }
    if ((int64_t)lstm_fbv_pack_floats(hidden) != out_floats) {
        set_error("fsnp_debug_lstm_fbv_pack: need %lld floats", (long long)lstm_fbv_pack_floats(hidden));
        return 2;
    }
    lstm_fbv_pack_weights(hidden, input_size, wih0, whh0, wih1, whh1, out);
    return 0;
}

double fsnp_lstm_flops(const fsnp_handle* h, int64_t num_seq, int32_t steps) {
    if (!h) return 0;
    return (double)num_seq * steps * lstm_flops_per_step(h);
}

double fsnp_forward_flops(const fsnp_handle* h, int32_t batch, int32_t frames, int32_t mode) {
    if (!h) return 0;
    const double Tp = frames + h->cfg.look_ahead;
    const double full_band = h->model == FSNP_MODEL_FULLSUBNET ? fb_lstm_flops_per_frame(h) : 3.0 * tcn_flops_per_frame(h);
    return batch * Tp * (rows_per_utt(h, mode) * lstm_flops_per_step(h) + full_band);
}

}  // extern "C"
