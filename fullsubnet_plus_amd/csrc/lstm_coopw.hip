// lstm_coopw.hip - "wave-owned" column-split two-layer LSTM + Linear for 11 ... 64 row tiles per launch on gfx950 (round 5; round 6: the
// transposed product, the 96-unit instantiation, a rotating Linear owner).
//
// Same arithmetic as lstm.hip / lstm_coop.hip / lstm_coopn.hip (SequenceModel.forward's LSTM branch,
// speech_enhance/audio_zen/model/module/sequence_model.py:113-123: nn.LSTM(input, hidden, 2) + Linear(hidden, 2)).
// What it is for: B = 2 ... 12 utterances (17 ... 97 row tiles of 32 sequences).  Round 4 ran them on
//   * the K-split kernel (lstm_coop.hip) at 32 / 64 units per workgroup: the four waves of a workgroup split K, so every layer
//     pass ends in an LDS reduction of four partial tiles, a cell phase spread over 256 threads and two workgroup barriers, and
//   * the three-way split (lstm_coopn.hip: 128 units per workgroup): 65 row tiles x 3 = 195 of 256 CUs, serial schedule.
// Here a WAVE owns 8 NT hidden units (NT = 1, 2, 3 gate-interleaved 32-column accumulator tiles) over the FULL K, a workgroup 32 NT
// units, and S = H / (32 NT) = 12 / 6 / 4 workgroups share a row tile: 17 row tiles x 12 = 204 CUs (B = 2), 41 x 6 = 246 (B = 5),
// 64 x 4 = 256 (B = 8).  Nothing is shared between the waves of a workgroup, so the time loop has NO workgroup barrier:
//   * the product is TRANSPOSED (round 6, w_mfma below): the weight fragment is the A operand, h / x the B operand.  Column c of a
//     tile = gate (c & 3) of unit (c >> 2), so accumulator register q of lane (sequence = lane & 31, hi = lane >> 5) is gate q & 3 of
//     unit hi + 2 (q >> 2): the lane holds (i, f, g, o) of ITS four cells of every tile in registers - the cell update reads the
//     accumulators directly (round 5 took the tile through a wave-private LDS slice and back: that staging and its 2 x 32 NT LDS
//     instructions per phase are gone, and with them the registers that made NT = 3 spill).  The four results are exactly the four
//     floats of the lane's slot of the exchange image, so h leaves as ONE 16-byte write-through store per tile.  Same products in the
//     same order as the round-5 kernel: bit-identical results;
//   * the same lane / slot identity holds for the input: lane (row, hi) gathers features hi + 2 i of its row, which are the
//     components of its own A fragments of the x k-groups - x never touches LDS.  The x k-groups are multiplied LAST in layer 0, the
//     raw x_{t+1} is gathered into the same registers right behind them and normalised a whole phase pair later;
//   * biases ride in the accumulator initialisation; the Linear(H, 2) partial of a wave is 4 NT lane-local FMAs + one
//     cross-half add, two coalesced 128-byte stores per step; ONE participant sums the P partials of a step two phases later (loads
//     issued in front of a cell phase, summed behind it) - participant t mod P for step t (round 6: a fixed owner was the slowest wave
//     of every phase);
//   * layer-skewed schedule of lstm2_coop_skew_kernel with the WAVES as participants (P = 4 S per row tile): A_t = layer 0 of
//     step t, C_t = layer 1; every wave runs A_0, [A_1, C_0], [A_2, C_1], ...; counter b0 counts finished A phases, b1 finished C
//     phases; every wait is for an arrival that happened a whole phase earlier; h0 cycles through three images, h1 and the Linear
//     partials through two (lstm_coop.hip, the comment above lstm2_coop_skew_kernel).  The poll of a counter is issued in front of
//     a phase's LAST k-groups (so that the counted waits of the k-loop never wait for it) and looked at behind them;
//   * the operands of the next phase's first k-groups are fetched right behind a phase's MFMAs - under its cell phase - so no
//     phase starts with an exposed L2 round trip, and the store drain in front of the arrival only waits for the h stores.
// What bounds it (profiles/r05_column_split.md): the k-loops run at 0.83-0.87 of their MFMA time - every buffer load a wave issues
// costs its SIMD ~22 matrix-pipe cycles (profiles/r01_ubench_mfma_issue.txt) and a k-group is NT + 1 loads for 4 NT MFMAs; cell
// phases, staging, drain and arrival are ~1.1 us per phase at NT = 1.
// Exchange region, arrival counters, write-through hand-off and abort protocol: lstm_common.h (those of lstm_coop.hip).
// K is summed bias first, then h0 | x (layer 0) and h1 | h0 (layer 1): bit-identical to no sibling kernel, same oracle tolerance
// (tests/test_gpu_parity.py::test_wave_owned_column_split_kernel_vs_oracle), bitwise repeatable.
#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
};
__device__ __forceinline__ float4 wld(const WStream& s, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff, soff, 0));
}
__device__ __forceinline__ float4 wld_sc1(const WStream& s, int soff) {      // exchange images: bypass L1 (lstm_common.h)
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff, soff, kSc1));
}

template <int NT>
__device__ __forceinline__ void w_mfma(f32x16 (&acc)[NT], const float4& a, const float4 (&b)[NT]) {
#pragma unroll
    // TRANSPOSED product (round 6): the weight fragment is the A operand (M = the tile's 32 gate-interleaved columns), h / x the B operand
    // (N = 32 sequences).  Same lane -> (column, k) / (sequence, k) maps on both sides, so neither the weight pack nor the exchange images
    // change, and every output element sums the same products in the same order - but accumulator register q of lane (sequence, hi) is now
    // gate q & 3 of unit hi + 2 (q >> 2): the four gates of the lane's four cells, with no trip through LDS
    for (int n = 0; n < NT; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[n].x, a.x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[n].y, a.y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[n].z, a.z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[n].w, a.w, acc[n], 0, 0, 0);
    }
}

// Weight stream of the whole model: [k-group g (layer 0: x | h0, then layer 1: h1 | h0)][8-unit block ub][lane][4] - ONE array for
// every NT: participant `part` multiplies blocks part * NT + n, n < NT, of every k-group.
constexpr int kUb(int HID) { return HID / 8; }

// One K segment of G k-groups (G % D == 0, G >= 2 D) through a register pipeline D groups deep that the CALLER has filled with
// groups 0 .. D - 1 (w_prefill).  The last D groups refill the pipeline with the first NN (<= D) groups of the NEXT segment, so a
// pass over several segments - and, through the caller, over several phases - never starts with an empty pipeline.
// D: 4 groups of NT x 256 matrix-pipe cycles; 8 at NT = 1 (a 4-deep pipeline is 1024 cycles there: less than an L2 round trip
// under load - measured: the k-loops of the 32-unit kernel ran at 0.77-0.85 of their MFMA time).
template <int NT> constexpr int coopw_depth() { return NT == 1 ? 8 : 4; }
template <int NT, int HID, int D, typename ALoad>
__device__ __forceinline__ void w_prefill(float4 (&a)[D], float4 (&b)[D][NT], const WStream& ws, int part, int wbase, ALoad aload) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
        a[k] = aload(k);
#pragma unroll
        for (int n = 0; n < NT; ++n) b[k][n] = wld(ws, ((wbase + k) * kUb(HID) + part * NT + n) * 1024);
    }
}
template <int NT, int HID, int D, int G, int NN, typename ALoad, typename NLoad, typename Hook>
__device__ __forceinline__ void w_segment(f32x16 (&acc)[NT], float4 (&a)[D], float4 (&b)[D][NT], const WStream& ws, int part,
                                          int wbase, ALoad aload, int nwbase, NLoad nload, Hook pre_tail) {
    static_assert(G % D == 0 && G >= 2 * D && NN <= D, "pipeline shape");
    for (int g0 = 0; g0 < G - D; g0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            w_mfma<NT>(acc, a[k], b[k]);
            a[k] = aload(g0 + k + D);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[k][n] = wld(ws, ((wbase + g0 + k + D) * kUb(HID) + part * NT + n) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pre_tail();            // (a load issued here is the NEWEST of the queue: the tail's counted waits never wait for it)
#pragma unroll
    for (int k = 0; k < D; ++k) {
        w_mfma<NT>(acc, a[k], b[k]);
        if constexpr (NN > 0) {
            if (k < NN) {
                a[k] = nload(k);
#pragma unroll
                for (int n = 0; n < NT; ++n) b[k][n] = wld(ws, ((nwbase + k) * kUb(HID) + part * NT + n) * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int coopw_stride(int NT) { return 32 * NT + 4; }       // floats per row of the round-5 staging tile
// a wave's LDS slice: [32][stride] floats that were the staging tile of round 5 and now hold the lanes' bias quadruples ([2 layers][NT][2][4]
// float4: the accumulator initialisation of the transposed product), the lane-private gather offsets [KX / 2][64] and Linear weights [2 NT][64][4]
constexpr int coopw_slice_words(int NT, int KX) { return 32 * coopw_stride(NT) + (KX / 2) * 64 + 2 * NT * 256; }
constexpr size_t coopw_smem_bytes(int NT, int KX) { return (size_t)4 * coopw_slice_words(NT, KX) * 4 + 32 * sizeof(RowDesc); }

}  // namespace

template <int HID, int KX, int NT>
__global__ __launch_bounds__(256) void lstm2_coopw_kernel(LstmWeights w, LstmArgs a) {
    constexpr int UW = 8 * NT;                     // hidden units per wave
    constexpr int S = HID / (32 * NT);             // workgroups per row tile
    constexpr int P = 4 * S;                       // participants (waves) per row tile
    constexpr int KGX = KX / 8, KGH = HID / 8, KG0 = KGX + KGH;
    constexpr int NXL = KX / 2;                    // input features per lane
    constexpr int HIMG = KGH * 64;                 // float4 per exchange image (32 rows x HID)
    constexpr int FCP4 = 2 * (HID / 8) * 16;       // float4 of the Linear-partial slot of the exchange region (lstm_common.h)
    constexpr int STRIDE = coopw_stride(NT);
    constexpr int D = coopw_depth<NT>();            // depth of the k-loops' register pipeline, in k-groups
    constexpr int NXN = KGX < D ? KGX : D;         // x k-groups whose weights the h0 segment's tail fetches
    static_assert(HID % (32 * NT) == 0 && KX % 8 == 0 && KX <= 64, "shape");
    static_assert(2 * P * 64 <= FCP4 * 4, "Linear partials of every participant fit their slot of the exchange region");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SLICE = coopw_slice_words(NT, KX);
    float* stg_all = reinterpret_cast<float*>(smem_raw);                        // [4 waves][SLICE]: staging tile, gather offsets, Linear weights
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(stg_all + 4 * SLICE);          // [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rt = blockIdx.x / S, cs = blockIdx.x % S;          // XCD-local placement: lstm_common.h
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, a.num_tiles, a.coop_xcd, rt, cs)) return;
    const int part = cs * 4 + wave;                         // participant = block of UW hidden units [part * UW, part * UW + UW)
    const int Tp = a.Tp;
    if (tid < 32) rows_s[tid] = a.rows[rt * 32 + tid];
    __syncthreads();                                        // the only workgroup barrier of the kernel
    float* stg = stg_all + wave * SLICE;
    int* gofft = reinterpret_cast<int*>(stg + 32 * STRIDE) + lane;              // [NXL][64]: this lane's gather offsets (-1 = no source)
    float4* wfct = reinterpret_cast<float4*>(stg + 32 * STRIDE + NXL * 64) + lane;   // [NT][2 outputs][64]: Linear weights of this lane's cells

    // ---- exchange region of this row tile (lstm_common.h: coop_tile_f4)
    float4* hx = reinterpret_cast<float4*>(a.coop_hx) + (size_t)rt * coop_tile_f4(HID);
    auto h0off = [](int m3) -> int { return m3 < 2 ? m3 * HIMG : 4 * HIMG + FCP4; };       // float4 offsets inside the region
    auto h1off = [](int par) -> int { return (2 + par) * HIMG; };
    float* fcp = reinterpret_cast<float*>(hx + 4 * HIMG);                                   // [2][P][2 outputs][32 rows]
    unsigned* bar0 = FSNP_COOP_BAR(a, rt, 0);
    unsigned* bar1 = FSNP_COOP_BAR(a, rt, 1);
    WStream hs;
    hs.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(hx), 0, coop_tile_f4(HID) * 16, 0x00020000);
    hs.voff = lane * 16;
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w.wpack_coopw), 0, (KG0 + 2 * KGH) * kUb(HID) * 1024, 0x00020000);
    ws.voff = lane * 16;

    // ---- input plan: lane (row, hi) owns features hi + 2 i of its row = component i & 3 of ITS A fragment of x k-group i >> 2
    const int row = lane & 31, hi = lane >> 5;
    const RowDesc rd = rows_s[row];
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    NormMD md = {0.0f, 1.0f};
    const NormMD* md_t = nullptr;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int j = hi + 2 * i;
        int off = -1;
        if (rd.valid && j < w.NIN) {
            if (dense) off = rd.b * Tp * gstep + j;
            else off = sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
        }
        gofft[i * 64] = off;
    }
    if (rd.valid) {
        if (a.md_seq != nullptr) md_t = a.md_seq + (size_t)rd.b * Tp;
        else if (!dense && a.md_row != nullptr) md_t = a.md_row + (size_t)(rt * 32 + row) * Tp;
        else if (!dense) md = a.md_utt[rd.b];
    }
    float xv[NXL];                                 // the lane's A fragments of x_t: k-group g = xv[4 g .. 4 g + 3]
    // raw values of step t (issued as soon as x_{t-1}'s MFMAs are done; lanes without a source read element 0 and discard it: no
    // per-element branches) ...
    auto x_fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int g = gofft[i * 64];
            xv[i] = gbase[g >= 0 ? g + t * gstep : 0];
        }
    };
    // ... normalised in place behind the cell phase: (x - m) * (1 / d) - ONE division per step and row (20 true divisions cost 1.4 us of
    // a 22 us step: measured); the product is within 1.5 ulp of the other kernels' (x - m) / d
    auto x_commit = [&](float mm, float dd) {
        const float rdd = 1.0f / dd;
#pragma unroll
        for (int i = 0; i < NXL; ++i) xv[i] = gofft[i * 64] >= 0 ? (xv[i] - mm) * rdd : 0.0f;
    };
    auto xfrag = [&](int g) -> float4 { return make_float4(xv[4 * g], xv[4 * g + 1], xv[4 * g + 2], xv[4 * g + 3]); };
    // (m_t, d_t) of step t for this lane's row: one 8-byte load (cumulative norms) or the utterance's pair
    auto md_at = [&](int t, float& mm, float& dd) {
        mm = md.m; dd = md.d;
        if (md_t) { const NormMD v = md_t[t]; mm = v.m; dd = v.d; }
    };
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        float mm, dd;
        md_at(0, mm, dd);
        x_fetch(0);
        asm volatile("" : "+v"(mm), "+v"(dd));
        x_commit(mm, dd);
    }

    // ---- per-lane constants: the bias of the lane's column of every tile (accumulator initialisation), Linear weights of its cells
    // (accumulator register q of lane (sequence, hi) = gate q & 3 of unit hi + 2 (q >> 2): 16 NT bias values per lane and layer, kept in the
    //  wave's LDS slice - where the staging tile of the round-5 kernel used to be - as [layer][n][hi][16], read as four float4 per tile)
    float4* biast = reinterpret_cast<float4*>(stg);                           // [2 layers][NT][2][4]
    for (int i = lane; i < 2 * NT * 2 * 16; i += 64) {
        const int q = i & 15, bh = (i >> 4) & 1, n = (i >> 5) % NT, layer = i / (32 * NT);
        stg[i] = w.bias[layer * 4 * HID + (q & 3) * HID + part * UW + n * 8 + bh + 2 * (q >> 2)];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int u = part * UW + n * 8 + hi;
        wfct[(2 * n) * 64] = make_float4(w.wfc[u], w.wfc[u + 2], w.wfc[u + 4], w.wfc[u + 6]);
        wfct[(2 * n + 1) * 64] = make_float4(w.wfc[HID + u], w.wfc[HID + u + 2], w.wfc[HID + u + 4], w.wfc[HID + u + 6]);
    }
    float c0[NT][4], c1[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0[n][j] = 0.f; c1[n][j] = 0.f; }

    // accumulator tiles -> the wave's LDS slice -> the lane's 4 NT cells; emit(n, h) receives the float4 of the lane's slot of
    // k-group part * NT + n of the h image (components = units hi, hi + 2, hi + 4, hi + 6 of 8-unit block part * NT + n)
    auto cells = [&](f32x16 (&acc)[NT], float (&c)[NT][4], auto emit) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x2 ca{c[n][0], c[n][1]}, cb{c[n][2], c[n][3]};
            const f32x2 ha = lstm_cell_pair(f32x2{acc[n][0], acc[n][4]}, f32x2{acc[n][1], acc[n][5]}, f32x2{acc[n][2], acc[n][6]}, f32x2{acc[n][3], acc[n][7]}, ca);
            const f32x2 hb = lstm_cell_pair(f32x2{acc[n][8], acc[n][12]}, f32x2{acc[n][9], acc[n][13]}, f32x2{acc[n][10], acc[n][14]}, f32x2{acc[n][11], acc[n][15]}, cb);
            c[n][0] = ca.x; c[n][1] = ca.y; c[n][2] = cb.x; c[n][3] = cb.y;
            emit(n, make_float4(ha.x, ha.y, hb.x, hb.y));
        }
    };
    // accumulator initialisation = the biases of the lane's 16 (gate, unit) rows of tile n
    auto acc_init = [&](f32x16 (&acc)[NT], int layer) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 b4 = biast[((layer * NT + n) * 2 + hi) * 4 + u];
                acc[n][4 * u] = b4.x; acc[n][4 * u + 1] = b4.y; acc[n][4 * u + 2] = b4.z; acc[n][4 * u + 3] = b4.w;
            }
    };
    auto hstore = [&](int img_f4, int n, const float4& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), hs.rsrc, hs.voff,
                                               (img_f4 + (part * NT + n) * 64) * 16, kSc1);
    };

    // ---- hand-off, per wave (lstm_common.h): drained write-through stores, one relaxed arrival; waits polled ahead
    auto arrive = [&](unsigned* bar) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto poll = [&](unsigned* bar) -> unsigned { return __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto wait_for = [&](unsigned* bar, unsigned target, unsigned early) -> bool {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)early) >= target) return true;
        int ok = 1;
        if (lane == 0) ok = xchg_wait(bar, target, a.coop_abort, a.coop_err) ? 1 : 0;
        return __builtin_amdgcn_readfirstlane(ok) != 0;
    };
    // ONE participant sums the P partials of a finished step in a fixed order: the loads are issued where the step's partials are known
    // to be complete (fc_issue) and summed a cell phase later (fc_finish), so that their round trip is not on that wave's path.  Round 6:
    // the owner ROTATES (step t_done belongs to participant t_done mod P).  With participant 0 as the fixed owner its P extra loads and adds
    // per step made it the slowest wave of every step - and a launch runs at the pace of its slowest participant (the same finding as in
    // lstm_hpw.hip: profiles/r06_b1_kernel.md); rotating, every wave pays once in P steps, inside the slack in front of its counters.
    float fcv[P];
    const float bfc_lane = w.bfc[hi];                    // (lane (row, hi) finishes output hi of its row)
    auto fc_owner = [&](int t_done) -> bool { return part == t_done % P; };
    auto fc_issue = [&](int t_done) {
        if (fc_owner(t_done)) {
            const float* src = fcp + (size_t)(t_done & 1) * P * 64 + lane;
#pragma unroll
            for (int p = 0; p < P; ++p) fcv[p] = xchg_load(src + p * 64);
        }
    };
    auto fc_finish = [&](int t_done) {
        if (fc_owner(t_done)) {
            const int o = lane >> 5;
            float sum = bfc_lane;
#pragma unroll
            for (int p = 0; p < P; ++p) sum += fcv[p];
            if (rd.valid && t_done >= a.LA)
                a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t_done - a.LA)] = apply_act(sum, a.act);
        }
    };
    auto chaos = [&](int t, int phase) { chaos_delay(a.coop_chaos ? a.coop_chaos + 7919 * wave : 0, t, phase); };

    // ---- register pipeline of the k-loops
    float4 pa[D], pb[D][NT];

    // optional phase profile (fsnp_debug_pp_profile with tiles_per_group = 32 NT): lane 0 of wave 0 of workgroup 0 stamps the 100 MHz
    // wall clock: prof[t * 16 + k], k = 0 .. 7 in A_t, 8 .. 15 in C_t
    unsigned long long* prof = (a.prof != nullptr && blockIdx.x == 0 && tid == 0) ? a.prof : nullptr;
#define FSNP_W_STAMP(t, k) do { if (prof && (t) < Tp) prof[(t) * 16 + (k)] = (unsigned long long)wall_clock64(); } while (0)

    // A_t: layer 0 of step t over [h0_{t-1} | x_t]; the pipeline holds the first D h0 groups.
    // with_c: C_{t-1} follows - its b1 wait and the prefill of its first groups sit right behind the MFMAs, so that those loads land
    // under the cell phase and the drain only waits for the h0_t stores.
    float xm = 0.0f, xd = 1.0f;                    // (m, d) of the raw x values xv holds (x_{t+1}, fetched in A_t, normalised in A_{t+1})
    auto phase_a = [&](int t, int m3, int pm3, bool with_c) -> bool {
        chaos(t, 0);
        FSNP_W_STAMP(t, 0);
        const bool have_next = t + 1 < Tp;
        f32x16 acc[NT];
        acc_init(acc, 0);
        const int hprev = h0off(pm3) * 16;
        // (the last D h0 groups refill the pipeline's WEIGHT slots with the first x k-groups; their A fragments are registers)
        w_segment<NT, HID, D, KGH, NXN>(acc, pa, pb, ws, part, KGX, [&](int g) -> float4 { return wld_sc1(hs, hprev + g * 1024); },
                                        0, [&](int g) -> float4 { return xfrag(g); }, [] {});
        FSNP_W_STAMP(t, 1);
        if (t > 0) x_commit(xm, xd);                     // x_t: fetched a whole phase pair ago (behind A_{t-1}'s x k-groups)
        // the b1 poll for the C phase behind this one: issued here, looked at behind the x k-groups (arrivals are a phase old)
        const unsigned early1 = with_c ? poll(bar1) : 0u;
#pragma unroll
        for (int g = 0; g < KGX; ++g) {
            w_mfma<NT>(acc, xfrag(g), pb[g % D]);
            if (g + D < KGX) {
#pragma unroll
                for (int n = 0; n < NT; ++n) pb[g % D][n] = wld(ws, ((g + D) * kUb(HID) + part * NT + n) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        FSNP_W_STAMP(t, 2);
        if (have_next) {
            md_at(t + 1, xm, xd);
            x_fetch(t + 1);
        }
        const int h1p = h1off(t & 1) * 16;                                                   // C_{t-1} starts on h1_{t-2} (parity t & 1)
        if (with_c) {
            if (!wait_for(bar1, (unsigned)P * (unsigned)(t - 1), early1)) return false;      // h1_{t-2}, Linear partials of step t - 2
            w_prefill<NT, HID, D>(pa, pb, ws, part, KG0, [&](int g) -> float4 { return wld_sc1(hs, h1p + g * 1024); });
            if (t >= 2) fc_issue(t - 2);
        }
        FSNP_W_STAMP(t, 3);
        const int himg = h0off(m3);
        const bool corrupt = a.coop_corrupt != 0 && rt == 0 && part == 0 && lane == 0 && t + 1 == a.coop_corrupt;     // test hook (LstmArgs)
        cells(acc, c0, [&](int n, const float4& h) {
            float4 hp = h;
            if (n == 0 && corrupt) hp.x += 1.0f;
            hstore(himg, n, hp);
        });
        FSNP_W_STAMP(t, 4);
        FSNP_W_STAMP(t, 5);
        chaos(t, 1);
        arrive(bar0);
        if (with_c && t >= 2) fc_finish(t - 2);
        FSNP_W_STAMP(t, 6);
        return true;
    };
    // C_t: layer 1 of step t over [h1_{t-1} | h0_t]; the pipeline holds the first D h1 groups.  next_a: A_{t+2} follows - the b0
    // wait for h0_{t+1} and the prefill of its first groups sit right behind the MFMAs.
    auto phase_c = [&](int t, int m3, bool next_a, int nm3) -> bool {
        const int cur = t & 1, prv = cur ^ 1;
        chaos(t, 2);
        FSNP_W_STAMP(t + 1, 8);
        unsigned early0 = 0u;
        f32x16 acc[NT];
        acc_init(acc, 1);
        const int h1p = h1off(prv) * 16, h0c = h0off(m3) * 16;
        w_segment<NT, HID, D, KGH, D>(acc, pa, pb, ws, part, KG0, [&](int g) -> float4 { return wld_sc1(hs, h1p + g * 1024); },
                                      KG0 + KGH, [&](int g) -> float4 { return wld_sc1(hs, h0c + g * 1024); }, [] {});
        FSNP_W_STAMP(t + 1, 9);
        // (the b0 poll for the A phase behind this one is issued in front of the last D k-groups and looked at behind them)
        w_segment<NT, HID, D, KGH, 0>(acc, pa, pb, ws, part, KG0 + KGH, [&](int g) -> float4 { return wld_sc1(hs, h0c + g * 1024); },
                                      0, [&](int) -> float4 { return make_float4(0.f, 0.f, 0.f, 0.f); }, [&] { if (next_a) early0 = poll(bar0); });
        FSNP_W_STAMP(t + 1, 10);
        const int hn = h0off(nm3) * 16;
        if (next_a) {
            if (!wait_for(bar0, (unsigned)P * (unsigned)(t + 2), early0)) return false;       // h0_{t+1} published by every participant
            w_prefill<NT, HID, D>(pa, pb, ws, part, KGX, [&](int g) -> float4 { return wld_sc1(hs, hn + g * 1024); });
        }
        FSNP_W_STAMP(t + 1, 11);
        const int himg = h1off(cur);
        float p0 = 0.0f, p1 = 0.0f;
        cells(acc, c1, [&](int n, const float4& h) {
            hstore(himg, n, h);
            const float4 w0 = wfct[(2 * n) * 64], w1 = wfct[(2 * n + 1) * 64];
            p0 += h.x * w0.x + h.y * w0.y + h.z * w0.z + h.w * w0.w;
            p1 += h.x * w1.x + h.y * w1.y + h.z * w1.z + h.w * w1.w;
        });
        FSNP_W_STAMP(t + 1, 12);
        p0 += __shfl_xor(p0, 32);
        p1 += __shfl_xor(p1, 32);
        xchg_store(fcp + ((size_t)cur * P + part) * 64 + hi * 32 + row, hi == 0 ? p0 : p1);      // [output hi][row]
        FSNP_W_STAMP(t + 1, 13);
        chaos(t, 3);
        arrive(bar1);
        FSNP_W_STAMP(t + 1, 14);
        return true;
    };

    // ---- A_0 (h0_{-1} = the zeroed third image), then [A_t, C_{t-1}] for t = 1 .. Tp - 1, then C_{Tp-1}
    {
        const int hz = h0off(2) * 16;
        w_prefill<NT, HID, D>(pa, pb, ws, part, KGX, [&](int g) -> float4 { return wld_sc1(hs, hz + g * 1024); });
    }
    if (Tp == 1) {
        // (one step only: A_0 -> C_0 with blocking waits)
        if (!phase_a(0, 0, 2, false)) return;
        if (!wait_for(bar0, (unsigned)P, 0u)) return;
        w_prefill<NT, HID, D>(pa, pb, ws, part, KG0, [&](int g) -> float4 { return wld_sc1(hs, h1off(1) * 16 + g * 1024); });
        if (!phase_c(0, 0, false, 0)) return;
        if (!wait_for(bar1, (unsigned)P, 0u)) return;
        fc_issue(0); fc_finish(0);
        return;
    }
    if (!phase_a(0, 0, 2, false)) return;
    // A_1 needs h0_0 of every participant; its pipeline is filled here (no C phase in between yet)
    if (!wait_for(bar0, (unsigned)P, 0u)) return;
    w_prefill<NT, HID, D>(pa, pb, ws, part, KGX, [&](int g) -> float4 { return wld_sc1(hs, h0off(0) * 16 + g * 1024); });
    int m3 = 1, pm3 = 0;                                // t % 3, (t - 1) % 3 for t = 1
    for (int t = 1; t < Tp; ++t) {
        if (!phase_a(t, m3, pm3, true)) return;         // (waits for b1 >= P (t - 1) inside, prefills C_{t-1}, Linear of step t - 2)
        const int nm3 = m3;                             // A_{t+1} reads h0_t = image m3
        if (!phase_c(t - 1, pm3, t + 1 < Tp, nm3)) return;   // (waits for b0 >= P (t + 1) inside, then prefills A_{t+1})
        pm3 = m3;
        m3 = m3 == 2 ? 0 : m3 + 1;
    }
    // C_{Tp-1}: h0_{Tp-1} (b0 >= P Tp) and h1_{Tp-2} (b1 >= P (Tp - 1))
    if (!wait_for(bar0, (unsigned)P * (unsigned)Tp, 0u)) return;
    if (!wait_for(bar1, (unsigned)P * (unsigned)(Tp - 1), 0u)) return;
    fc_issue(Tp - 2); fc_finish(Tp - 2);
    w_prefill<NT, HID, D>(pa, pb, ws, part, KG0, [&](int g) -> float4 { return wld_sc1(hs, h1off(Tp & 1) * 16 + g * 1024); });
    if (!phase_c(Tp - 1, pm3, false, 0)) return;
    if (!wait_for(bar1, (unsigned)P * (unsigned)Tp, 0u)) return;
    fc_issue(Tp - 1); fc_finish(Tp - 1);
#undef FSNP_W_STAMP
}

// ------------------------------------------------------------------------------------------------
size_t lstm_coopw_pack_floats(int H, int KX) { return (size_t)(KX / 8 + 3 * (H / 8)) * (H / 8) * 256; }

// [k-group g (layer 0: x | h0, then layer 1: h1 | h0)][8-unit block ub][lane][k-pair p]: the B operand of MFMA p of k-group g for
// column c = lane & 31 of block ub is W[gate (c & 3)][unit 8 ub + (c >> 2)][k = 8 g' + 2 p + (lane >> 5)]
void lstm_coopw_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1,
                             float* wpack) {
    const int KGX = KX / 8, KGH = H / 8, KG0 = KGX + KGH, KGT = KG0 + 2 * KGH, NUB = H / 8;
    for (int g = 0; g < KGT; ++g)
        for (int ub = 0; ub < NUB; ++ub)
            for (int lane = 0; lane < 64; ++lane)
                for (int p = 0; p < 4; ++p) {
                    const int c = lane & 31;
                    const size_t wrow = (size_t)(c & 3) * H + ub * 8 + (c >> 2);
                    float v = 0.0f;
                    if (g < KG0) {
                        const int k = 8 * g + 2 * p + (lane >> 5);
                        if (k < KX) { if (k < NIN) v = wih0[wrow * NIN + k]; }
                        else v = whh0[wrow * H + (k - KX)];
                    } else {
                        const int k = 8 * (g - KG0) + 2 * p + (lane >> 5);
                        if (k < H) v = whh1[wrow * H + k];
                        else v = wih1[wrow * H + (k - H)];
                    }
                    wpack[(((size_t)g * NUB + ub) * 64 + lane) * 4 + p] = v;
                }
}

bool lstm_coopw_available(const LstmWeights& w, int units) {
    return !w.gru && w.H == 384 && (w.KX == 40 || w.KX == 64) && (units == 32 || units == 64 || units == 96) && w.wpack_coopw != nullptr;
}

template <int HID, int KX, int NT>
static void launch_coopw_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    constexpr int S = HID / (32 * NT);
    const size_t smem_need = coopw_smem_bytes(NT, KX);
    auto kern = lstm2_coopw_kernel<HID, KX, NT>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    if (occ) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(kern), 256, smem_need) != hipSuccess) *occ = 0;
        return;
    }
    const size_t smem = a.coop_own_cu > 0 && (size_t)a.coop_own_cu > smem_need ? (size_t)a.coop_own_cu : smem_need;
    const int grid = a.coop_xcd ? 8 * xcd_local_blocks_per_xcd(S, a.num_tiles, a.coop_xcd) : a.num_tiles * S;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, w, a);
}

template <int KX>
static void launch_coopw_kx(const LstmWeights& w, const LstmArgs& a, hipStream_t s, int* occ) {
    if (a.coop_units == 32) launch_coopw_inst<384, KX, 1>(w, a, s, occ);
    else if (a.coop_units == 96) launch_coopw_inst<384, KX, 3>(w, a, s, occ);
    else launch_coopw_inst<384, KX, 2>(w, a, s, occ);
}

// a.num_tiles row tiles x H / a.coop_units workgroups (a.coop_units = 32 or 64 hidden units per workgroup), all co-resident
void launch_lstm_coopw(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.KX == 64) launch_coopw_kx<64>(w, a, s, nullptr); else launch_coopw_kx<40>(w, a, s, nullptr);
}
int lstm_coopw_occupancy(const LstmWeights& w, int units) {
    LstmArgs a{};
    a.coop_units = units;
    int occ = 0;
    if (w.KX == 64) launch_coopw_kx<64>(w, a, nullptr, &occ); else launch_coopw_kx<40>(w, a, nullptr, &occ);
    return occ;
}

}  // namespace fsnp
