// lstm_fbv.hip - the full-band 2-layer LSTM of the original FullSubNet for 1 ... 4 utterances, on the VALU (round 5).
//
// speech_enhance/fullsubnet/model/fullsubnet.py:39-47, 86-90: SequenceModel(input_size = num_freqs = 257, hidden_size = 512,
// num_layers = 2, LSTM) over ONE sequence per utterance - the only recurrence whose timings the reference itself records
// (fullsubnet.py:141-144).  The reference CLI runs it at B = 1 (audio_zen/inferencer/base_inferencer.py:65-69).
// Round 4 ran it on lstm2_coop_kernel<512, 264, 8, SEQ> (lstm_coop.hip): 64 workgroups x 8 units, K split over the waves, weights
// resident - but on 32-row MFMA tiles: 57 k-groups x 4 v_mfma_f32_32x32x2_f32 = 6.1 us of matrix-pipe time per step for ONE live
// row, 12.6 us per step with the two LDS reductions and the hand-off (profiles/r04_bench_configs.md: 1.6 ms of a 3.3 ms forward).
// With one to four rows the product is a matrix-VECTOR product: the same 64 x 8 units and resident weights, plain v_fma:
//   * thread (c = tid & 31, ks = tid >> 5) owns gate column c (gate c & 3 of unit 8 cs + (c >> 2)) over k-slice ks of both layers:
//     100 + 128 weights in registers (layer 0: [x (288, zero padded) | h0 (512)] = 8 x 100, layer 1: [h0 | h1] = 8 x 128);
//   * the activation vectors sit in LDS as [x | h0 | h1] per row (h0 is shared by the two layers' k ranges); the 32 lanes of a
//     k-slice read the same float4 (broadcast), 57 ds_read_b128 per row and step;
//   * the eight k-slices of a column are added through one lane swap and a 4-wave LDS reduction; 8 NB threads run the cells;
//   * serial schedule of lstm2_coop_kernel: ONE inter-workgroup barrier per step (after h0_t is published; h1_{t-1} was published
//     before its writer arrived), write-through hand-off of lstm_common.h; the exchange images are plain [parity][row][512] floats.
// Per step: ~0.4 us of FMAs + reductions per layer and one hand-off: measured in profiles/r05_fullsubnet.md.
// h1_t also goes row-major to seq_out[seq][t][H] for the Linear(512, 257) GEMM (tcn.hip), as from the SEQ K-split kernel.  Same cell
// arithmetic (fast_sigmoid / fast_tanh) as that kernel; K is summed in another order: same oracle tolerance
// (tests/test_gpu_fullsubnet.py), bitwise repeatable.  LSTM only (a GRU full-band model keeps the K-split kernel).
#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {
constexpr int kFbvXP = 288;                 // x part of layer 0's k range (num_freqs <= 288, zero padded)
constexpr int kFbvK0 = 100, kFbvK1 = 128;   // weights per thread: layer 0 (8 slices x 100 = 288 + 512), layer 1 (8 x 128 = 512 + 512)
}  // namespace

template <int HID, int NB>
__global__ __launch_bounds__(256) void lstm2_fbv_kernel(LstmWeights w, LstmArgs a) {
    static_assert(HID == 512 && kFbvXP + HID == 8 * kFbvK0 && 2 * HID == 8 * kFbvK1, "k slices");
    constexpr int S = HID / 8;                      // workgroups (8 units each)
    constexpr int ROW = kFbvXP + 2 * HID;           // floats per row of the activation vectors: [x | h0 | h1]
    __shared__ __attribute__((aligned(16))) float V[NB * ROW];
    __shared__ float red[2][4][32][NB];             // [layer parity][wave][column][row]
    __shared__ RowDesc rows_s[NB];
    __shared__ int abort_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = blockIdx.x;                      // column slice: units [8 cs, 8 cs + 8)
    const int c = tid & 31, ks = tid >> 5;
    const int Tp = a.Tp;
    if (tid == 0) abort_s = 0;
    if (tid < NB) rows_s[tid] = tid < a.num_rows ? a.rows[tid] : RowDesc{0, 0, 0, 0};
    for (int i = tid; i < NB * ROW; i += 256) V[i] = 0.0f;

    // ---- the thread's weights: [cs][fragment j4 < 57][tid][4]
    float w0[kFbvK0], w1[kFbvK1];
    {
        const float4* __restrict__ wp = reinterpret_cast<const float4*>(w.wpack_fbv) + (size_t)cs * ((kFbvK0 + kFbvK1) / 4) * 256 + tid;
#pragma unroll
        for (int j = 0; j < kFbvK0 / 4; ++j) { const float4 v = wp[j * 256]; w0[4 * j] = v.x; w0[4 * j + 1] = v.y; w0[4 * j + 2] = v.z; w0[4 * j + 3] = v.w; }
#pragma unroll
        for (int j = 0; j < kFbvK1 / 4; ++j) { const float4 v = wp[(kFbvK0 / 4 + j) * 256]; w1[4 * j] = v.x; w1[4 * j + 1] = v.y; w1[4 * j + 2] = v.z; w1[4 * j + 3] = v.w; }
    }
    // ---- exchange images (zeroed per forward): h0 [2][NB][HID], h1 [2][NB][HID]
    float* hx = a.coop_hx;
    auto h0img = [&](int par) -> float* { return hx + (size_t)par * NB * HID; };
    auto h1img = [&](int par) -> float* { return hx + (size_t)(2 + par) * NB * HID; };
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 4 * NB * HID * 4, 0x00020000);
    unsigned* bar = FSNP_COOP_BAR(a, 0, 0);
    __syncthreads();

    // ---- cells: thread (u = tid & 7, row = tid >> 3) of the first 8 NB threads
    const bool cell_thread = tid < 8 * NB;
    const int cu = tid & 7, crow = (tid >> 3) < NB ? (tid >> 3) : 0;
    const int unit = cs * 8 + cu;
    float bias0[4], bias1[4], c0 = 0.0f, c1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { bias0[g] = w.bias[g * HID + unit]; bias1[g] = w.bias[4 * HID + g * HID + unit]; }
    const RowDesc crd = rows_s[crow];

    // x_t of row r: (dense[b][t][j] - m) / d, j < NIN
    auto stage_x = [&](int t) {
        for (int i = tid; i < NB * kFbvXP; i += 256) {
            const int r = i / kFbvXP, j = i % kFbvXP;
            const RowDesc rd = rows_s[r];
            float v = 0.0f;
            if (rd.valid && j < w.NIN) {
                NormMD m{0.0f, 1.0f};
                if (a.md_seq != nullptr) m = a.md_seq[(size_t)rd.b * Tp + t];
                v = (a.dense[((size_t)rd.b * Tp + t) * a.dense_stride + j] - m.m) / m.d;
            }
            V[r * ROW + j] = v;
        }
    };
    // one layer's matrix-vector products of this thread's k-slice (NK weights over V[row][vbase + ks NK ...]), the two k-slices of a wave
    // added by a lane swap, the waves' partials left in red[layer][wave][column][row].  Rows go in pairs through a REAL loop: unrolled
    // over four rows hipcc hoisted every ds_read of the phase and spilled 137 registers.
    auto gemv = [&](int layer, const auto& wk, int vbase) {
        constexpr int NK = sizeof(wk) / sizeof(float);
        constexpr int RP = NB < 2 ? 1 : 2;
#pragma unroll 1
        for (int r0 = 0; r0 < NB; r0 += RP) {
            float acc[RP];
#pragma unroll
            for (int r = 0; r < RP; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < NK / 4; ++j) {
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    const float4 v = *reinterpret_cast<const float4*>(V + (r0 + r) * ROW + vbase + ks * NK + 4 * j);
                    acc[r] = fmaf(wk[4 * j], v.x, acc[r]); acc[r] = fmaf(wk[4 * j + 1], v.y, acc[r]);
                    acc[r] = fmaf(wk[4 * j + 2], v.z, acc[r]); acc[r] = fmaf(wk[4 * j + 3], v.w, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                acc[r] += __shfl_xor(acc[r], 32);
                if (lane < 32) red[layer][wave][c][r0 + r] = acc[r];
            }
        }
    };
    auto inter_wg_barrier = [&](unsigned target) -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!xchg_wait(bar, target, a.coop_abort, a.coop_err)) abort_s = 1;
        }
        __syncthreads();
        return abort_s == 0;
    };
    // sum of the 4 waves' partials of column 4 cu + g, row crow (+ bias) -> gate pre-activation
    auto pre = [&](int par, int g, const float (&b)[4]) -> float {
        const int col = 4 * cu + g;
        return ((red[par][0][col][crow] + red[par][1][col][crow]) + (red[par][2][col][crow] + red[par][3][col][crow])) + b[g];
    };

    stage_x(0);
    __syncthreads();
    for (int t = 0; t < Tp; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        chaos_delay(a.coop_chaos, t, 0);
        // ---------------- layer 0: [x_t | h0_{t-1}] ----------------
        gemv(0, w0, 0);
        __syncthreads();
        if (cell_thread) {
            const float ig = fast_sigmoid(pre(0, 0, bias0)), fg = fast_sigmoid(pre(0, 1, bias0));
            const float gg = fast_tanh(pre(0, 2, bias0)), og = fast_sigmoid(pre(0, 3, bias0));
            c0 = fg * c0 + ig * gg;
            const float h = og * fast_tanh(c0);
            if ((tid >> 3) < NB) xchg_store(h0img(cur) + crow * HID + unit, h);
        }
        if (t + 1 < Tp) stage_x(t + 1);                      // (x_t was last read by the layer-0 products above, in front of the barrier)
        chaos_delay(a.coop_chaos, t, 1);
        if (!inter_wg_barrier((unsigned)S * (unsigned)(t + 1))) return;      // h0_t and h1_{t-1} of every slice are now visible
        chaos_delay(a.coop_chaos, t, 2);
        // h0_t -> V[.][x | H0 | .], h1_{t-1} -> V[.][. | . | H1]: 2 NB HID floats as 16-byte write-through-coherent loads
        for (int i = tid; i < 2 * NB * (HID / 4); i += 256) {
            const int which = i / (NB * (HID / 4)), rem = i % (NB * (HID / 4)), r = rem / (HID / 4), q = rem % (HID / 4);
            const int img = which == 0 ? cur : 2 + prv;
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, ((img * NB + r) * HID + 4 * q) * 4, 0, kSc1));
            *reinterpret_cast<float4*>(V + r * ROW + kFbvXP + which * HID + 4 * q) = v;
        }
        __syncthreads();
        // ---------------- layer 1: [h0_t | h1_{t-1}] ----------------
        gemv(1, w1, kFbvXP);
        __syncthreads();
        if (cell_thread) {
            const float ig = fast_sigmoid(pre(1, 0, bias1)), fg = fast_sigmoid(pre(1, 1, bias1));
            const float gg = fast_tanh(pre(1, 2, bias1)), og = fast_sigmoid(pre(1, 3, bias1));
            c1 = fg * c1 + ig * gg;
            const float h = og * fast_tanh(c1);
            if ((tid >> 3) < NB) {
                xchg_store(h1img(cur) + crow * HID + unit, h);
                if (crd.valid) a.seq_out[((size_t)crd.b * Tp + t) * (a.seq_stride ? a.seq_stride : HID) + unit] = h;
            }
        }
        chaos_delay(a.coop_chaos, t, 3);
        // (the next step's layer-0 products read V's x and h0 parts: x_{t+1} was staged in front of the barrier, h0_t behind it; red[0] is
        //  rewritten only behind the next __syncthreads of that step - all ordered by the two barriers above)
    }
}

// ------------------------------------------------------------------------------------------------
size_t lstm_fbv_pack_floats(int H) { return (size_t)(H / 8) * (kFbvK0 + kFbvK1) * 256; }

// [column slice cs][fragment j4][thread tid][4]: thread (c = tid & 31, ks = tid >> 5) holds, for gate c & 3 of unit 8 cs + (c >> 2),
// layer-0 weights of k = 100 ks + j over [x (288, zero padded beyond NIN) | h0] and layer-1 weights of k = 128 ks + j over [h0 | h1]
void lstm_fbv_pack_weights(int H, int NIN, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out) {
    const int NF = (kFbvK0 + kFbvK1) / 4;
    for (int cs = 0; cs < H / 8; ++cs)
        for (int tid = 0; tid < 256; ++tid) {
            const int c = tid & 31, ks = tid >> 5;
            const size_t wrow = (size_t)(c & 3) * H + cs * 8 + (c >> 2);
            for (int j = 0; j < kFbvK0 + kFbvK1; ++j) {
                float v = 0.0f;
                if (j < kFbvK0) {
                    const int k = kFbvK0 * ks + j;
                    if (k < kFbvXP) { if (k < NIN) v = wih0[wrow * NIN + k]; }
                    else v = whh0[wrow * H + (k - kFbvXP)];
                } else {
                    const int k = kFbvK1 * ks + (j - kFbvK0);
                    v = k < H ? wih1[wrow * H + k] : whh1[wrow * H + (k - H)];
                }
                out[(((size_t)cs * NF + j / 4) * 256 + tid) * 4 + (j & 3)] = v;
            }
        }
}

// The H / 8 = 64 workgroups hand h over to each other every step, so all of them must be resident at once; with ~230 weight registers
// per thread ONE workgroup fits a CU: a device (or partition) with fewer CUs than workgroups would sit in the exchange until its 2 s
// time-out - the caller then keeps the K-split kernel, whose unit count is chosen from the CU count.
bool lstm_fbv_available(const LstmWeights& w, int batch, int num_cus) {
    return !w.gru && w.H == 512 && w.NIN <= kFbvXP && batch >= 1 && batch <= 4 && w.wpack_fbv != nullptr && num_cus >= w.H / 8;
}

// H / 8 workgroups, all co-resident; a.num_rows <= 4 sequences; a.coop_hx: >= 4 x 4 x H floats, zeroed; a.coop_bar: one counter
void launch_lstm_fbv(const LstmWeights& w, const LstmArgs& a, hipStream_t s) {
    const int grid = w.H / 8;
    if (a.num_rows <= 1) hipLaunchKernelGGL((lstm2_fbv_kernel<512, 1>), dim3(grid), dim3(256), 0, s, w, a);
    else if (a.num_rows == 2) hipLaunchKernelGGL((lstm2_fbv_kernel<512, 2>), dim3(grid), dim3(256), 0, s, w, a);
    else hipLaunchKernelGGL((lstm2_fbv_kernel<512, 4>), dim3(grid), dim3(256), 0, s, w, a);
}

}  // namespace fsnp
