// lstm_hpw.hip - the half-tile ping-pong column split with WAVE-OWNED hidden units (round 6): B = 1, the reference CLI's batch size.
//
// Same arithmetic as every sibling (SequenceModel.forward's LSTM branch, speech_enhance/audio_zen/model/module/sequence_model.py:113-123:
// nn.LSTM(input, hidden, 2) + Linear(hidden, 2)), same launch shape, exchange region and counters as lstm_hp.hip (S = H / 16 workgroups
// per 32-row tile on one XCD, every row tile as two half tiles of 16 sequences in turn, one fused pass [layer 1 of step t | layer 0 of
// step t + 1] per half-phase) - but decomposed the way lstm_coopw.hip is: nothing is shared between the waves of a workgroup.
//
// lstm_hp.hip spends 13.85 us per step for 8.4 us of MFMAs (profiles/r05_hp_phase_profile.txt: 6.6 us per half-phase of which barrier +
// drain 0.23, slowest-wave wait 0.46, cells 0.52, in-pass events ~0.8).  Its waves split the GATES of the workgroup's 16 units, so the
// pre-activations cross LDS, the cells sit behind a workgroup barrier, the h slices are staged in LDS for wave 0 to publish, and the
// operands come by LDS DMA that three of the four waves issue from inside their MFMA streams (16 issues of 100-185 matrix-pipe cycles
// each: those waves are the slowest, and everybody waits for them at the next barrier).  Here:
//   * the product is TRANSPOSED: the weights are the A operand of v_mfma_f32_16x16x4_f32 (M = 16 gate columns of the wave), h / x the B
//     operand (N = 16 sequences).  Wave w of workgroup cs owns units u_j = 16 cs + w + 4 j, j < 4, and M row m = 4 j + gate: the
//     accumulator of lane (sequence = lane & 15, j = lane >> 4) then holds exactly (i, f, g, o) of ITS cell (sequence, u_j) in its four
//     registers - the cell update is lane-local with no transposition, no LDS and no barrier, both layers as one packed two-cell update;
//   * the four units of a wave are the four components of ONE float4 slot row of the exchange image (hp_a16: k & 3 = w, (k >> 2) & 3 =
//     j), so the 64 results of a wave are 256 contiguous bytes: h leaves as ONE 4-byte write-through store per lane and layer, straight
//     from the cell's register;
//   * the wave's weights (16 rows x 1200 k = 300 registers) stay resident as in lstm_hp.hip; the operands are streamed by the wave
//     itself, L2 -> registers, through a pipeline 8 k-groups deep (one 16-byte sc1 load per lane and k-group of an image: 22 matrix-pipe
//     cycles per issue against 100-185 for an LDS-DMA piece, and no ds_read in the loop at all);
//   * x is gathered by the lane that multiplies it (lane (sequence, k & 3) owns features k = (lane >> 4) + 4 i), normalised with one
//     reciprocal per step;
//   * the WAVES are the participants of the hand-off (P = 4 S per half tile): stores -> the other half's whole pass in between ->
//     s_waitcnt vmcnt(0) behind that pass's x k-groups -> one relaxed arrival per wave; every wave polls for itself (the load is issued
//     in front of a pass's last 8 k-groups and looked at behind them), then prefills its pipeline for the other half before its cells.
// The time loop has no workgroup barrier, no LDS traffic except the lane-private gather offsets, and no staging.
// Linear(H, 2): a wave's partial over its 4 units (two cross-lane adds), 128 bytes per participant and half-phase; the 32 sums of a
// half-phase (16 sequences x 2 outputs, each over the P partials in a fixed order: bitwise repeatable) are 32 jobs that ROTATE over the
// participants a half-phase later - a fixed set of summing waves was the slowest of every pass and paced the launch (+0.39 us per half-phase).
// 16x16x4 MFMAs in this operand order sum K like lstm_hp.hip does per accumulator (two chains per layer, added at the end): same oracle
// tolerance, bit-identical to no sibling.
#include <type_traits>
#include <utility>

#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

using f32x4w = __attribute__((ext_vector_type(4))) float;

template <typename F, int... I>
__device__ __forceinline__ void hpw_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void hpw_static_for(F&& f) { hpw_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int hpw_gx(int KX) { return (KX + 15) / 16; }
// per wave: gather offsets [2 halves][KX / 4][64] + the lanes' bias quadruples [2 layers][64] float4; + the row descriptors
constexpr size_t hpw_smem_bytes(int KX) { return (size_t)4 * (2 * (KX / 4) * 64 * 4 + 2 * 64 * 16) + 32 * sizeof(RowDesc); }

template <int N>
__device__ __forceinline__ float hpw_ror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false)); }
// sum over the 64 lanes of a wave in a fixed order, in every lane
__device__ __forceinline__ float hpw_wave_sum(float v) {
    v += hpw_ror<8>(v); v += hpw_ror<4>(v); v += hpw_ror<2>(v); v += hpw_ror<1>(v);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

// weights = A operand (M = the wave's 16 gate columns), h / x = B operand (N = 16 sequences)
#define HPW_MF(acc, wv, hv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, hv, acc, 0, 0, 0)

}  // namespace

// {0, 1}: what rows without a normalisation table multiply by (a valid address keeps the per-step (m, d) load unconditional)
__device__ const NormMD kHpwIdentityMD = {0.0f, 1.0f};

template <int HID, int KX, bool PROF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lstm2_coop_hpw_kernel(LstmWeights w, LstmArgs a) {
    constexpr int S = HID / 16;                      // workgroups per row tile
    constexpr int P = 4 * S;                         // participants (waves) per half tile
    constexpr int GX = hpw_gx(KX), GH = HID / 16;    // k-groups of 16: x (zero padded), one h image
    constexpr int NX = KX / 4;                       // input features per lane (k = kq + 4 i < KX; the zero-padded tail of the last k-group is skipped)
    constexpr int NW0 = GX + GH, NW1 = 2 * GH;       // the wave's weight fragments (float4): layer 0 [x | h0], layer 1 [h1 | h0]
    constexpr int HALF_B = GH * 1024;                // bytes of one half image (16 sequences x HID)
    constexpr int IMG_B = (HID / 8) * 1024;          // bytes of one 32-row image slot of the exchange region (lstm_common.h)
    constexpr int TILE_BYTES = coop_tile_f4(HID) * 16;
    constexpr int H0OFF = 0, H1OFF = 2 * IMG_B, FCOFF = 4 * IMG_B;     // (the Linear partials also use the third h0 image behind their slot)
    constexpr int D = GH % 6 == 0 ? 6 : 8;           // depth of the operand pipeline, in k-groups of 384 matrix-pipe cycles (H = 384: 6, the registers allow no more; H = 256: 8)
    constexpr int NPL = (P + 63) / 64;               // Linear partials per lane of the summing wave
    constexpr int XLOAD_G = 2;                       // k-group of the pass behind which the raw x of step t + 2 is requested
    constexpr int POLL_AHEAD = 2;                    // k-groups between the poll of the next half's counter and the tail that acts on it
    static_assert(KX <= 64 && GX <= 4 && KX % 4 == 0, "gathered sub-band input");
    static_assert(GH % D == 0 && GH >= 2 * D, "pipeline shape");
    static_assert(P % 32 == 0 && P >= 32, "the 32 Linear sums of a half-phase rotate over the participants");
    static_assert((NW0 + NW1) * 4 <= 320 && NW1 * 4 <= 192, "the wave's weights must fit the register file (layer 1: AGPRs)");
    static_assert(FCOFF + 4 * P * 128 <= TILE_BYTES, "Linear partials [parity][half][participant][16 sequences][2] fit the exchange region");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* goff_all = reinterpret_cast<int*>(smem_raw);                                   // [4 waves][2 halves][NX][64]: lane-private gather offsets
    float4* bias_all = reinterpret_cast<float4*>(goff_all + 4 * 2 * NX * 64);           // [4 waves][2 layers][64]: (i, f, g, o) biases of the lane's cell
    RowDesc* rows_s = reinterpret_cast<RowDesc*>(bias_all + 4 * 2 * 64);                // [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rt = blockIdx.x / S, cs = blockIdx.x % S;
    if (a.coop_xcd && !xcd_local_decode(blockIdx.x, S, a.num_tiles, a.coop_xcd, rt, cs)) return;
    const int part = cs * 4 + wave;                  // participant index inside the row tile
    const int Tp = a.Tp;
    if (tid < 32) rows_s[tid] = a.rows[rt * 32 + tid];
    __syncthreads();                                 // the only workgroup barrier of the kernel
    // half tiles of this row tile that hold sequences (a launch's sequences are spread evenly over its tiles, from slot 0 up)
    const int nh = rows_s[16].valid ? 2 : 1;
    const int seq = lane & 15, kq = lane >> 4;       // the lane's sequence; kq = k & 3 of its operand fragments = unit index j of its cells
    const int unit = cs * 16 + wave + 4 * kq;        // hidden unit of the lane's cells

    // ---- input plan: lane (seq, kq) owns features k = 16 g + 4 j + kq = component j of ITS fragment of x k-group g
    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    int* gofft = goff_all + (wave * 2 * NX) * 64 + lane;                    // [(hf * NX + i) * 64]: BYTE offsets of the lane's features
    // (m_t, d_t) of the lane's sequence: ONE unconditional 8-byte load per step from md_p[t * md_s] - a per-step table (cumulative norms,
    // md_s = 1), the utterance's pair (md_s = 0) or the identity (dense input).  Global address space spelled out: a FLAT load would
    // make hipcc drain vmcnt(0) in front of the k-loop (flat loads may overtake buffer loads)
    using gmd_ptr = const __attribute__((address_space(1))) NormMD*;
    gmd_ptr md_p[2];
    int md_s[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const RowDesc rd = rows_s[hf * 16 + seq];
        md_p[hf] = (gmd_ptr)&kHpwIdentityMD;
        md_s[hf] = 0;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            // features beyond NIN meet zero weights and rows without a sequence are never written out: both read element 0 of the
            // input (a finite value) - no per-element select anywhere
            const int k = 4 * i + kq;               // = 16 g + 4 j + kq with i = 4 g + j
            int off = 0;
            if (rd.valid && k < w.NIN) {
                if (dense) off = rd.b * Tp * gstep + k;
                else off = sb_feature_offset(k, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
            }
            gofft[(hf * NX + i) * 64] = off * 4;
        }
        if (rd.valid) {
            if (a.md_seq != nullptr) { md_p[hf] = (gmd_ptr)(a.md_seq + (size_t)rd.b * Tp); md_s[hf] = 1; }
            else if (!dense && a.md_row != nullptr) { md_p[hf] = (gmd_ptr)(a.md_row + (size_t)(rt * 32 + hf * 16 + seq) * Tp); md_s[hf] = 1; }
            else if (!dense) md_p[hf] = (gmd_ptr)(a.md_utt + rd.b);
        }
    }
    // (one buffer load per element and step: the per-lane byte offset is loop invariant, the step goes into the scalar offset)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbase), 0, 0x7FFFFFFC, 0x00020000);
    float xq[2][NX];                                 // the lane's fragments of x of each half: k-group g = xq[hf][4 g .. 4 g + 3]
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- the wave's weights, resident: [participant][fragment][lane][4]
    float4 w0[NW0], w1[NW1];
    {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(w.wpack_hpw) + (size_t)part * (NW0 + NW1) * 256, 0, (NW0 + NW1) * 1024, 0x00020000);
#pragma unroll
        for (int i = 0; i < NW0; ++i) w0[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, i * 1024, 0));
#pragma unroll
        for (int i = 0; i < NW1; ++i) w1[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, (NW0 + i) * 1024, 0));
        // layer 1 in AGPRs (the MFMA reads its operands from either file); layer 0, the operand pipeline and everything else share the VGPRs
#pragma unroll
        for (int i = 0; i < NW1; ++i) asm volatile("" : "+a"(w1[i].x), "+a"(w1[i].y), "+a"(w1[i].z), "+a"(w1[i].w));
    }
    // accumulator register i = gate i of the lane's cell: the biases ride in the initialisation (kept in the wave's LDS slice, two
    // ds_read_b128 per half-phase: the VGPRs are needed for the operand pipeline)
    float4* biast = bias_all + wave * 2 * 64 + lane;
    biast[0] = make_float4(w.bias[0 * HID + unit], w.bias[1 * HID + unit], w.bias[2 * HID + unit], w.bias[3 * HID + unit]);
    biast[64] = make_float4(w.bias[4 * HID + unit], w.bias[5 * HID + unit], w.bias[6 * HID + unit], w.bias[7 * HID + unit]);
    auto bias_of = [&](int layer) -> f32x4w { const float4 v = biast[layer * 64]; return f32x4w{v.x, v.y, v.z, v.w}; };
    const float wfc0 = w.wfc[unit], wfc1 = w.wfc[HID + unit];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- exchange region of this row tile
    const __amdgpu_buffer_rsrc_t hr =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.coop_hx) + (size_t)rt * (TILE_BYTES / 4), 0, TILE_BYTES, 0x00020000);
    unsigned* const bars[2] = {FSNP_COOP_BAR(a, rt, 0), FSNP_COOP_BAR(a, rt, 1)};     // one 128-byte line each
    const int hst = part * 256 + seq * 16 + kq * 4;  // byte offset of the lane's cell inside a half image (hp_a16(seq, unit))
    auto hload = [&](int soff) -> float4 { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hr, lane * 16, soff, kSc1)); };
    auto hstore = [&](float v, int soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), hr, hst, soff, kSc1); };

    auto x_fetch = [&](auto HF, int t) {
        constexpr int hf = decltype(HF)::value;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xq[hf][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, gofft[(hf * NX + i) * 64], t * gstep * 4, 0));
    };
    auto x_commit = [&](auto HF, NormMD m) {          // (x - m) * (1 / d): one division per step and sequence (within 1.5 ulp of (x - m) / d)
        constexpr int hf = decltype(HF)::value;
        const float rdd = 1.0f / m.d;
#pragma unroll
        for (int i = 0; i < NX; ++i) xq[hf][i] = (xq[hf][i] - m.m) * rdd;
    };
    auto md_at = [&](int hf, int t) -> NormMD { const gmd_ptr p = md_p[hf] + t * md_s[hf]; return NormMD{p->m, p->d}; };

    f32x2 cst[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // {c1, c0} of the lane's cell of each half

    // ---- Linear(H, 2): the 32 sums (16 sequences x 2 outputs) of a half-phase are 32 JOBS that rotate over the participants - job
    // j = (participant + 32 k) mod P < 32 of half-phase k = 2 t + half: every wave sums one (sequence j >> 1, output j & 1) every P / 32
    // half-phases.  (With fixed owners those 16 waves were 0.39 us per half-phase slower than everybody else and set the pace of the
    // whole launch; spread out, the cost disappears in the slack every wave has in front of its counter - profiles/r06_b1_kernel.md.)
    constexpr int JOBS = 32;
    float fcv[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) fcv[j] = 0.0f;
    int fc_job = -1, fc_hf = 0, fc_t = 0;            // fc_job >= 0: the partials of (sequence, output) fc_job of step fc_t of half fc_hf are in flight / in fcv
    auto fc_issue = [&](int hn, int t_done) {
        const int j = (part + JOBS * ((2 * t_done + hn) % (P / JOBS))) % P;
        if (j >= JOBS) return;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int p = q * 64 + lane;
            fcv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr, ((p < P ? p : 0) * 32 + j) * 4, FCOFF + (((t_done & 1) * 2 + hn) * P) * 128, kSc1));
        }
        fc_job = j; fc_hf = hn; fc_t = t_done;
    };
    const float bfc0 = w.bfc[0], bfc1 = w.bfc[1];       // (uniform: scalar loads, once - a per-job global load would sit on the job's critical path)
    auto fc_finish = [&]() {
        if (fc_job < 0) return;
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < NPL; ++q)
            if (q * 64 + lane < P) sum += fcv[q];
        sum = hpw_wave_sum(sum);
        const int o = fc_job & 1;
        const RowDesc rd = rows_s[fc_hf * 16 + (fc_job >> 1)];
        if (lane == 0 && rd.valid && fc_t >= a.LA)
            a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (fc_t - a.LA)] = apply_act(sum + (o ? bfc1 : bfc0), a.act);
        fc_job = -1;
    };

    auto arrive = [&](unsigned* bar) {
        if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // blocking wait of a wave for a counter; a wait that gives up (time-out / launch-wide abort: lstm_common.h) ENDS the wave - nothing
    // in this kernel waits for a wave of its own workgroup, and a plain exit keeps the hot loop's control flow free of merge points
    auto wave_wait = [&](unsigned* bar, unsigned target) {
        int ok = 1;
        if (lane == 0) ok = xchg_wait(bar, target, a.coop_abort, a.coop_err) ? 1 : 0;
        if (__builtin_amdgcn_readfirstlane(ok) == 0) __builtin_amdgcn_endpgm();
    };
    auto chaos = [&](int t, int phase) { chaos_delay(a.coop_chaos ? a.coop_chaos + 7919 * wave : 0, t, phase); };

    // ================= phase -1 of every half: h0_0 = cell(W_ih0 x_0) -> h0img[0] =================
    hpw_static_for<2>([&](auto HC) {
        constexpr int hf = decltype(HC)::value;
        if (hf >= nh) return;
        x_fetch(HC, 0);
        x_commit(HC, md_at(hf, 0));
        f32x4w a0a = bias_of(0), a0b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NX; ++i) {           // component i & 3 of weight fragment i >> 2; two chains alternate
            const float wv = (i & 3) == 0 ? w0[i >> 2].x : (i & 3) == 1 ? w0[i >> 2].y : (i & 3) == 2 ? w0[i >> 2].z : w0[i >> 2].w;
            if (i & 1) HPW_MF(a0b, wv, xq[hf][i]); else HPW_MF(a0a, wv, xq[hf][i]);
        }
        a0a += a0b;
        f32x2 c = {0.f, cst[hf].y};
        const f32x2 hh = lstm_cell_pair(f32x2{0.f, a0a[0]}, f32x2{0.f, a0a[1]}, f32x2{0.f, a0a[2]}, f32x2{0.f, a0a[3]}, c);
        cst[hf].y = c.y;
        hstore(hh.y, H0OFF + hf * HALF_B);
        const int t1 = Tp > 1 ? 1 : 0;
        x_fetch(HC, t1);
        x_commit(HC, md_at(hf, t1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrive(bars[hf]);
    });

    // ================= phases 0 .. Tp - 1, the two halves in turn =================
    float4 pp[D], pq[D];                   // operand pipeline: k-groups of h1_{t-1} and h0_t of the half-phase at hand
    auto prefill = [&](int hn, int tn) {
        const int s1 = H1OFF + ((tn & 1) ^ 1) * IMG_B + hn * HALF_B, s0 = H0OFF + (tn & 1) * IMG_B + hn * HALF_B;
#pragma unroll
        for (int k = 0; k < D; ++k) { pp[k] = hload(s1 + k * 1024); pq[k] = hload(s0 + k * 1024); }
    };
    // optional phase profile (fsnp_debug_pp_profile): thread 0 of workgroup 0 stamps the 100 MHz wall clock: prof[(t * 2 + hf) * 16 + k]
    // (compile-time variant: the stamp branches would otherwise sit in the hot loop's control flow and blur hipcc's wait counts)
    unsigned long long* prof = (PROF && a.prof != nullptr && blockIdx.x == 0 && tid == 0) ? a.prof : nullptr;
#define FSNP_HPW_STAMP(k) do { if constexpr (PROF) { if (prof) prof[(t * 2 + hf) * 16 + (k)] = (unsigned long long)wall_clock64(); } } while (0)

    // one half-phase (hf, t); TWO = the row tile has two half tiles (the next half-phase is the other half's)
    auto half_phase = [&](auto HC, auto TWOC, int t, unsigned*& pend_bar) {
        constexpr int hf = decltype(HC)::value;
        constexpr int ho = hf ^ 1;
        constexpr bool two = decltype(TWOC)::value;
        const int nt = (two && hf == 0) ? t : t + 1;              // the next half-phase is (two ? ho : 0, nt)
        const bool has_next = nt < Tp;
        unsigned* nbar = bars[two ? ho : 0];
        const unsigned ntarget = (unsigned)P * (unsigned)(nt + 1);
        chaos(t, hf);
        FSNP_HPW_STAMP(0);
        // ---- layer 0 over x_{t+1}: registers only - runs while the stores of the previous half-phase and this one's first operands land
        f32x4w a0a = bias_of(0), a0b = {0.f, 0.f, 0.f, 0.f};
        f32x4w a1a = bias_of(1), a1b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NX; ++i) {           // component i & 3 of weight fragment i >> 2; two chains alternate
            const float wv = (i & 3) == 0 ? w0[i >> 2].x : (i & 3) == 1 ? w0[i >> 2].y : (i & 3) == 2 ? w0[i >> 2].z : w0[i >> 2].w;
            if (i & 1) HPW_MF(a0b, wv, xq[hf][i]); else HPW_MF(a0a, wv, xq[hf][i]);
        }
        // ---- the previous half-phase's h stores are complete once this wave's queue is empty: arrive for it
        __builtin_amdgcn_sched_barrier(0);       // (behind the x k-groups, which need nothing from memory)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pend_bar) { arrive(pend_bar); pend_bar = nullptr; }
        FSNP_HPW_STAMP(1);
        fc_finish();
        FSNP_HPW_STAMP(2);
        FSNP_HPW_STAMP(3);
        // ---- one pass over h: acc1 = W_hh1 h1_{t-1} + W_ih1 h0_t, acc0 += W_hh0 h0_t
        // raw x_{t+2} goes into the registers the x k-groups have just been read from, issued from inside the pass (behind its first
        // k-groups: at the top it would sit in front of every operand in the in-order queue) and normalised behind the cells.
        // Unconditional (the last two steps re-read step Tp - 1 and never use it): straight-line code lets hipcc count its waits exactly
        const int tx = t + 2 < Tp ? t + 2 : Tp - 1;
        NormMD mdn = {0.f, 1.f};
        unsigned seen = 0;
        const int s1 = H1OFF + ((t & 1) ^ 1) * IMG_B + hf * HALF_B, s0 = H0OFF + (t & 1) * IMG_B + hf * HALF_B;
        // operands of the NEXT half-phase (two halves: the other half's; its step nt)
        const int ntc = has_next ? nt : Tp - 1;                    // (behind the last half-phase of all: a harmless re-read)
        const int n1 = H1OFF + ((ntc & 1) ^ 1) * IMG_B + (two ? ho : 0) * HALF_B, n0 = H0OFF + (ntc & 1) * IMG_B + (two ? ho : 0) * HALF_B;
        auto group = [&](auto G, auto REFILL_NEXT) {
            constexpr int g = decltype(G)::value;
            constexpr bool refill_next = decltype(REFILL_NEXT)::value;
            const float4 p = pp[g % D], q = pq[g % D];
            // (the three accumulator chains are kept interleaved: left alone hipcc issues the four a1a MFMAs of a k-group back to back,
            //  and a dependent 16x16x4 chain costs 15 % - profiles/r01_ubench_mfma_issue.txt, pattern K against L / M)
            HPW_MF(a1a, w1[g].x, p.x); HPW_MF(a0a, w0[GX + g].x, q.x); HPW_MF(a1b, w1[GH + g].x, q.x);
            __builtin_amdgcn_sched_barrier(0);
            HPW_MF(a1a, w1[g].y, p.y); HPW_MF(a0a, w0[GX + g].y, q.y); HPW_MF(a1b, w1[GH + g].y, q.y);
            __builtin_amdgcn_sched_barrier(0);
            HPW_MF(a1a, w1[g].z, p.z); HPW_MF(a0a, w0[GX + g].z, q.z); HPW_MF(a1b, w1[GH + g].z, q.z);
            __builtin_amdgcn_sched_barrier(0);
            HPW_MF(a1a, w1[g].w, p.w); HPW_MF(a0a, w0[GX + g].w, q.w); HPW_MF(a1b, w1[GH + g].w, q.w);
            if constexpr (g + D < GH) {
                pp[g % D] = hload(s1 + (g + D) * 1024);
                pq[g % D] = hload(s0 + (g + D) * 1024);
            } else if constexpr (refill_next) {                   // the tail refills the pipeline with the next half-phase's first k-groups
                pp[g % D] = hload(n1 + (g + D - GH) * 1024);
                pq[g % D] = hload(n0 + (g + D - GH) * 1024);
            }
            if constexpr (g == XLOAD_G) {
                mdn = md_at(hf, tx);
                x_fetch(HC, tx);
            }
            // the next half's counter: its arrivals are a whole pass old.  The load is issued POLL_AHEAD k-groups in front of the tail and
            // looked at where the tail starts: complete (the rule) -> the tail itself prefetches the next half-phase's operands
            if constexpr (two && g == GH - D - POLL_AHEAD) seen = __hip_atomic_load(nbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        };
        hpw_static_for<GH - D>([&](auto G) { group(G, std::false_type{}); });
        bool early = true;
        if constexpr (two) {
            early = !has_next || (unsigned)__builtin_amdgcn_readfirstlane((int)seen) >= ntarget;
            if (early) {
                hpw_static_for<D>([&](auto K) { group(std::integral_constant<int, GH - D + decltype(K)::value>{}, std::true_type{}); });
            } else {                       // not complete yet (drift): finish the pass, wait for real, fetch behind it
                hpw_static_for<D>([&](auto K) { group(std::integral_constant<int, GH - D + decltype(K)::value>{}, std::false_type{}); });
                wave_wait(nbar, ntarget);
                prefill(ho, ntc);
            }
            if (has_next && nt >= 1) fc_issue(ho, nt - 1);      // (the arrivals that completed the counter also published the partials of step nt - 1)
        } else {
            hpw_static_for<D>([&](auto K) { group(std::integral_constant<int, GH - D + decltype(K)::value>{}, std::false_type{}); });
        }
        FSNP_HPW_STAMP(4);
        FSNP_HPW_STAMP(5);
        // ---- cells: {layer 1 (h1_t), layer 0 (h0_{t+1})} of (sequence seq, unit) as ONE packed two-cell update, from the accumulators
        a0a += a0b; a1a += a1b;
        const f32x2 hh = lstm_cell_pair(f32x2{a1a[0], a0a[0]}, f32x2{a1a[1], a0a[1]}, f32x2{a1a[2], a0a[2]}, f32x2{a1a[3], a0a[3]}, cst[hf]);
        // (test hook, LstmArgs::coop_corrupt: h0 of step t + 1, sequence 0, unit 0 of row tile 0 is published with 1.0 added)
        const bool corrupt = a.coop_corrupt != 0 && rt == 0 && part == 0 && hf == 0 && lane == 0 && t + 2 == a.coop_corrupt;
        hstore(corrupt ? hh.y + 1.0f : hh.y, H0OFF + ((t + 1) & 1) * IMG_B + hf * HALF_B);
        hstore(hh.x, H1OFF + (t & 1) * IMG_B + hf * HALF_B);
        {
            float p0 = hh.x * wfc0, p1 = hh.x * wfc1;              // Linear partial over the wave's 4 units
            p0 += __shfl_xor(p0, 16); p1 += __shfl_xor(p1, 16);
            p0 += __shfl_xor(p0, 32); p1 += __shfl_xor(p1, 32);
            if (lane < 16) {
                const float2 pv = make_float2(p0, p1);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, pv), hr, (part * 32 + seq * 2) * 4,
                                                      FCOFF + (((t & 1) * 2 + hf) * P) * 128, kSc1);
            }
        }
        FSNP_HPW_STAMP(6);
        x_commit(HC, mdn);
        chaos(t, 2 + hf);
        FSNP_HPW_STAMP(7);
        if constexpr (PROF) { if (prof) prof[(t * 2 + hf) * 16 + 15] = early ? 1ull : 0ull; }
        if constexpr (two) {
            pend_bar = bars[hf];           // arrived for behind the next half-phase's x k-groups
        } else {                           // one half only: the next wait is for this very half - drain, arrive, wait, fetch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            arrive(bars[hf]);
            if (has_next) {
                wave_wait(nbar, ntarget);
                prefill(0, nt);
                if (nt >= 1) fc_issue(0, nt - 1);
            }
        }
        FSNP_HPW_STAMP(8);
    };

    unsigned* pend_bar = nullptr;          // the arrival of the previous half-phase has not been issued yet (its stores are in flight)
    wave_wait(bars[0], (unsigned)P);
    prefill(0, 0);
    if (nh == 2) {
        for (int t = 0; t < Tp; ++t) {
            half_phase(std::integral_constant<int, 0>{}, std::true_type{}, t, pend_bar);
            half_phase(std::integral_constant<int, 1>{}, std::true_type{}, t, pend_bar);
        }
    } else {
        for (int t = 0; t < Tp; ++t) half_phase(std::integral_constant<int, 0>{}, std::false_type{}, t, pend_bar);
    }
#undef FSNP_HPW_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pend_bar) arrive(pend_bar);
    fc_finish();
    // ================= the Linear of the last step =================
    for (int hf = 0; hf < nh; ++hf) {
        fc_issue(hf, Tp - 1);                    // (decides whether this wave holds a job of that half-phase)
        if (fc_job < 0) continue;
        wave_wait(hf ? bars[1] : bars[0], (unsigned)P * (unsigned)(Tp + 1));
        fc_issue(hf, Tp - 1);                    // ... and reads the partials once they are all published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fc_finish();
    }
}

// ------------------------------------------------------------------------------------------------
size_t lstm_hpw_pack_floats(int H, int KX) { return (size_t)(H / 16) * 4 * (hpw_gx(KX) + 3 * (H / 16)) * 256; }

// [participant = 4 cs + w][fragment: x k-groups | W_hh0 | W_hh1 | W_ih1][lane][4]: the A operand of MFMA j of a k-group is
// W[gate * H + unit][k = 16 g + 4 j + (lane >> 4)] with M row m = lane & 15 = 4 jj + gate and unit = 16 cs + w + 4 jj
void lstm_hpw_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out) {
    const int S = H / 16, GX = hpw_gx(KX), GH = H / 16, NF = GX + 3 * GH;
    for (int part = 0; part < 4 * S; ++part)
        for (int f = 0; f < NF; ++f)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int m = lane & 15, gate = m & 3, jj = m >> 2;
                    const size_t wrow = (size_t)gate * H + (part >> 2) * 16 + (part & 3) + 4 * jj;
                    float v = 0.0f;
                    if (f < GX) { const int k = 16 * f + 4 * j + (lane >> 4); if (k < NIN) v = wih0[wrow * NIN + k]; }
                    else if (f < GX + GH) v = whh0[wrow * H + 16 * (f - GX) + 4 * j + (lane >> 4)];
                    else if (f < GX + 2 * GH) v = whh1[wrow * H + 16 * (f - GX - GH) + 4 * j + (lane >> 4)];
                    else v = wih1[wrow * H + 16 * (f - GX - 2 * GH) + 4 * j + (lane >> 4)];
                    out[(((size_t)part * NF + f) * 64 + lane) * 4 + j] = v;
                }
}

template <int HID, int KX>
static void launch_hpw_inst(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    constexpr int S = HID / 16;
    const size_t smem_need = hpw_smem_bytes(KX);
    const size_t smem = a.coop_own_cu > 0 && (size_t)a.coop_own_cu > smem_need ? (size_t)a.coop_own_cu : smem_need;
    const int grid = a.coop_xcd ? 8 * xcd_local_blocks_per_xcd(S, a.num_tiles, a.coop_xcd) : a.num_tiles * S;
    if constexpr (HID == 384 && KX == 40) {          // the phase-profile variant exists for the default sizes only
        if (a.prof != nullptr) {
            auto kp = lstm2_coop_hpw_kernel<HID, KX, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            hipLaunchKernelGGL(kp, dim3(grid), dim3(256), smem, s, w, a);
            return;
        }
    }
    auto kern = lstm2_coop_hpw_kernel<HID, KX, false>;
    static PerDeviceOnce attr_once;
    attr_once.run([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); });
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, w, a);
}

// (H = 384 with 41 ... 64 input features - fb_num_neighbors >= 2 - stays on lstm_hp.hip: 16 x fragments per lane next to the 300 weight
//  registers and the operand pipeline do not fit the register file without spills)
bool lstm_hpw_available(const LstmWeights& w) { return !w.gru && ((w.H == 384 && w.KX == 40) || (w.H == 256 && (w.KX == 40 || w.KX == 64))) && w.wpack_hpw != nullptr; }

// a.num_tiles row tiles x H / 16 workgroups, all co-resident (the launch shape of lstm_hp.hip)
void launch_lstm_hpw(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    if (w.H == 256) { if (w.KX == 64) launch_hpw_inst<256, 64>(w, a, s); else launch_hpw_inst<256, 40>(w, a, s); return; }
    launch_hpw_inst<384, 40>(w, a, s);
}

}  // namespace fsnp
