// lstm_generic.hip - runtime-sized two-layer LSTM / GRU (+ Linear): the "correct first" kernel for the sizes no tuned kernel is
// instantiated for (round 3).
//
// The reference builds `torch.nn.LSTM / GRU(input_size, hidden_size, num_layers=2)` for ANY sizes
// (speech_enhance/audio_zen/model/module/sequence_model.py:31-46; fullsubnet_plus.py:102-110, fullsubnet/model/fullsubnet.py:39-56).
// The MFMA kernels of lstm*.hip are instantiated for sb_model_hidden_size 256 / 384 / 512, sub-band inputs of <= 64 features and, for
// the original FullSubNet's full-band model, hidden 512 / <= 264 bins.  Everything else used to be rejected at fsnp_create;
// now it runs here: plain fp32 FMAs (v_fma_f32), every size a run-time argument.
//   * a workgroup owns RG sequences (1, 2, 4 or 8: fewer when there are few sequences, so that more CUs work) and keeps their
//     x_t, h0, h1, c0, c1 and the pre-activations of one layer in LDS;
//   * thread c of a 1024-column chunk owns gate column c of all RG rows: per 4 k it issues 4 coalesced weight loads
//     (wT[k][column], a transposed copy packed at fsnp_commit_weights) and RG broadcast ds_read_b128 of the operands;
//   * cell update, Linear(H, OUT) epilogue / h1 sequence output, input gather and normalisation as in the tuned kernels
//     (same LstmArgs, same four-slot gate layout: LSTM i, f, g, o; GRU r, z, n_x, n_h).
// Speed is not the point (hidden 320, 128 steps: 9.8 ms at B = 1, 76 ms at B = 32 against 2.2 / 27.3 ms on the tuned kernels at 384);
// results meet the same oracle tolerance (tests/test_gpu_parity.py::test_generic_recurrent_kernel_*).
#include "fsnp_common.h"
#include "lstm_common.h"

namespace fsnp {

namespace {

__device__ __forceinline__ float gen_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gen_tanh(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

constexpr int gen_pad4(int v) { return (v + 3) / 4 * 4; }
// dynamic LDS in floats: x [RG][NINP], h0, h1, c0, c1 [RG][HP], pre [RG][4 HP]
inline size_t gen_smem_floats(int RG, int H, int NIN) { return (size_t)RG * (gen_pad4(NIN) + 8 * (size_t)gen_pad4(H)); }

}  // namespace

// SEQ = true: the full-band model of the original FullSubNet (dense rows of NIN features, per-frame (m, d) table, h1 sequence
// out, no Linear); SEQ = false: the sub-band model (gathered or dense input, fused Linear + activation + look-ahead slice).
// 1024 threads: the kernel is bound by the latency of its weight stream (every column reads K weights from L2 per step) and LDS
// allows one workgroup per CU, so the workgroup itself brings the 16 waves that hide it.
constexpr int kGenThreads = 1024;
template <int RG, bool SEQ>
__global__ __launch_bounds__(kGenThreads) void lstm2_generic_kernel(LstmWeights w, LstmArgs a) {
    constexpr int NTHR = kGenThreads;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int H = w.H, NIN = w.NIN, OUT = w.OUT, G4 = 4 * H;
    const int HP = gen_pad4(H), NINP = gen_pad4(NIN);
    float* xs = gsm;                         // [RG][NINP]
    float* h0 = xs + RG * NINP;              // [RG][HP]
    float* h1 = h0 + RG * HP;
    float* c0 = h1 + RG * HP;
    float* c1 = c0 + RG * HP;
    float* pre = c1 + RG * HP;               // [RG][4 HP]   (column c of row r at pre[r * 4 HP + c])
    __shared__ RowDesc rows_s[RG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot0 = blockIdx.x * RG;
    const int Tp = a.Tp;
    const bool gru = w.gru != 0;
    if (tid < RG) rows_s[tid] = a.rows[slot0 + tid];
    for (int i = tid; i < RG * (NINP + 8 * HP); i += NTHR) gsm[i] = 0.0f;
    __syncthreads();

    const bool dense = a.dense != nullptr;
    const float* __restrict__ gbase = dense ? a.dense : a.att_mag;
    const int gstep = dense ? a.dense_stride : a.FP;
    const float* wT0 = w.wgen;                                   // [NIN + H][4H]
    const float* wT1 = w.wgen + (size_t)(NIN + H) * G4;          // [2H][4H]

    // acc[r] += sum_k wT[k][col] * op[r][k]   (op rows of stride `ld` floats in LDS, 16-byte aligned, zero padded to 4).
    // Blocks of 16 k, software-pipelined: the 16 coalesced weight loads of block b + 1 are in flight while block b is multiplied
    // (without the pipeline every block waited out an L2 round trip: 400 us per step at B = 1 instead of ~45).
    auto accumulate = [&](float (&acc)[RG], const float* wT, int col, const float* op, int ld, int K) {
        constexpr int KB = 16;                   // k per block: 16 weight loads in flight per thread
        int k = 0;
        if (K >= KB) {
            float wn[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) wn[j] = wT[(size_t)j * G4 + col];
            for (; k + KB <= K; k += KB) {
                float wc[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) wc[j] = wn[j];
                if (k + 2 * KB <= K) {
#pragma unroll
                    for (int j = 0; j < KB; ++j) wn[j] = wT[(size_t)(k + KB + j) * G4 + col];
                }
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    float t = acc[r];
#pragma unroll
                    for (int q = 0; q < KB / 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(op + r * ld + k + 4 * q);
                        t = fmaf(wc[4 * q], v.x, t); t = fmaf(wc[4 * q + 1], v.y, t); t = fmaf(wc[4 * q + 2], v.z, t); t = fmaf(wc[4 * q + 3], v.w, t);
                    }
                    acc[r] = t;
                }
            }
        }
        for (; k < K; ++k) {
            const float wv = wT[(size_t)k * G4 + col];
#pragma unroll
            for (int r = 0; r < RG; ++r) acc[r] = fmaf(wv, op[r * ld + k], acc[r]);
        }
    };
    // one layer: pre = bias + W [opA | opB], then the cell update of every (row, unit) into (c, h)
    auto layer = [&](const float* wT, const float* bias, const float* opA, int ldA, int KA, const float* opB, int KB, float* c, float* h) {
        for (int col = tid; col < G4; col += NTHR) {
            float acc[RG];
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < RG; ++r) acc[r] = b;
            accumulate(acc, wT, col, opA, ldA, KA);
            accumulate(acc, wT + (size_t)KA * G4, col, opB, HP, KB);
#pragma unroll
            for (int r = 0; r < RG; ++r) pre[r * 4 * HP + col] = acc[r];
        }
        __syncthreads();
        for (int i = tid; i < RG * H; i += NTHR) {
            const int r = i / H, u = i % H;
            const float* p = pre + r * 4 * HP;
            float hv;
            if (gru) {           // r = s(a_r), z = s(a_z), n = tanh(a_nx + r a_nh), h' = (1 - z) n + z h      (torch.nn.GRU)
                const float rg = gen_sigmoid(p[u]), zg = gen_sigmoid(p[H + u]);
                const float ng = gen_tanh(p[2 * H + u] + rg * p[3 * H + u]);
                hv = ng + zg * (h[r * HP + u] - ng);
            } else {
                const float ig = gen_sigmoid(p[u]), fg = gen_sigmoid(p[H + u]), gg = gen_tanh(p[2 * H + u]), og = gen_sigmoid(p[3 * H + u]);
                const float cn = fg * c[r * HP + u] + ig * gg;
                c[r * HP + u] = cn;
                hv = og * gen_tanh(cn);
            }
            h[r * HP + u] = hv;      // (its readers - this layer's accumulate - finished before the barrier above)
        }
        __syncthreads();
    };

    for (int t = 0; t < Tp; ++t) {
        // ---- x_t of the RG rows, normalised
        for (int i = tid; i < RG * NIN; i += NTHR) {
            const int r = i / NIN, j = i % NIN;
            const RowDesc rd = rows_s[r];
            float v = 0.0f;
            if (rd.valid) {
                NormMD md = {0.0f, 1.0f};
                if (a.md_seq != nullptr) md = a.md_seq[(size_t)rd.b * Tp + t];
                else if (!dense && a.md_row != nullptr) md = a.md_row[(size_t)(slot0 + r) * Tp + t];
                else if (!dense) md = a.md_utt[rd.b];
                const int off = dense ? rd.b * Tp * gstep + j
                                      : sb_feature_offset(j, rd.f, rd.b * Tp * a.FP, a.F, a.NSBN, a.NFBN, a.fb_rel, a.fb_branch_stride);
                v = (gbase[off + t * gstep] - md.m) / md.d;
            }
            xs[r * NINP + j] = v;
        }
        __syncthreads();
        layer(wT0, w.bias, xs, NINP, NIN, h0, H, c0, h0);              // layer 0 over [x_t | h0_{t-1}]
        layer(wT1, w.bias + G4, h0, HP, H, h1, H, c1, h1);             // layer 1 over [h0_t | h1_{t-1}]
        if constexpr (SEQ) {
            const int ss = a.seq_stride > 0 ? a.seq_stride : H;        // (pad columns [H, ss) are written as zeros: a GEMM operand)
            for (int i = tid; i < RG * ss; i += NTHR) {
                const int r = i / ss, u = i % ss;
                const RowDesc rd = rows_s[r];
                if (rd.valid) a.seq_out[((size_t)rd.b * Tp + t) * ss + u] = u < H ? h1[r * HP + u] : 0.0f;
            }
        } else {
            for (int item = wave; item < RG * OUT; item += NTHR / 64) {        // Linear(H, OUT): a wave per (row, output), K across the lanes
                const int r = item / OUT, o = item % OUT;
                float s = 0.0f;
                for (int u = lane; u < H; u += 64) s = fmaf(w.wfc[(size_t)o * H + u], h1[r * HP + u], s);
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
                const RowDesc rd = rows_s[r];
                if (lane == 0 && rd.valid && t >= a.LA)
                    a.out[(size_t)rd.out_off + (size_t)o * a.out_stride_o + (t - a.LA)] = apply_act(s + w.bfc[o], a.act);
            }
        }
        // (the next step's first write to h1 / xs is behind the barriers of its own layers; xs is rewritten right away, but its
        //  last readers - layer 0 - are two barriers back)
    }
}

// ------------------------------------------------------------------------------------------------
size_t lstm_generic_pack_floats(int H, int NIN) { return (size_t)(NIN + H) * 4 * H + (size_t)2 * H * 4 * H; }

// wih0 [4H][NIN], whh0 [4H][H], wih1 [4H][H], whh1 [4H][H] (four-slot matrices) -> [layer][k][4H]: layer 0 k = [x | h0], layer 1 k = [h0 | h1]
void lstm_generic_pack_weights(int H, int NIN, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out) {
    const size_t G4 = (size_t)4 * H;
    float* l0 = out;
    float* l1 = out + (size_t)(NIN + H) * G4;
    for (size_t c = 0; c < G4; ++c) {
        for (int k = 0; k < NIN; ++k) l0[(size_t)k * G4 + c] = wih0[c * NIN + k];
        for (int k = 0; k < H; ++k) {
            l0[(size_t)(NIN + k) * G4 + c] = whh0[c * H + k];
            l1[(size_t)k * G4 + c] = wih1[c * H + k];
            l1[(size_t)(H + k) * G4 + c] = whh1[c * H + k];
        }
    }
}

// sequences per workgroup: as many as LDS allows, fewer when there are few sequences (more CUs take part); 0 = does not fit at all
int lstm_generic_rows_per_group(int H, int NIN, int num_seq, int num_cus) {
    // every workgroup streams ALL the weights from L2 once per step (5 MB at hidden 320), so few sequences per workgroup means
    // weight traffic, many means idle CUs: 4 from 128 sequences up (B = 1: 65 workgroups), 8 once 8 x the CUs are filled
    (void)num_cus;
    int rg = num_seq >= 2048 ? 8 : num_seq >= 128 ? 4 : num_seq >= 32 ? 2 : 1;
    while (rg > 1 && gen_smem_floats(rg, H, NIN) * 4 > (size_t)150 * 1024) rg /= 2;
    return gen_smem_floats(rg, H, NIN) * 4 <= (size_t)150 * 1024 ? rg : 0;
}

// LDS opt-in of one instantiation (per device) + a residency check: 1024 threads with `smem` bytes of dynamic LDS must fit a CU.
// Returns hipSuccess, or the error of the attribute call / hipErrorLaunchOutOfResources when no workgroup fits.
template <int RG, bool SEQ>
static hipError_t generic_prepare(size_t smem) {
    auto k = lstm2_generic_kernel<RG, SEQ>;
    static PerDeviceOnce once;
    static hipError_t attr_err[64] = {};                       // per device: a failure on one GPU is that GPU's alone
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = dev >= 0 && dev < 64 ? dev : 0;
    once.run([&] { attr_err[di] = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); });
    if (attr_err[di] != hipSuccess) return attr_err[di];
    if (smem == 0) return hipSuccess;
    int blocks = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(k), kGenThreads, smem);
    if (e != hipSuccess) return e;
    return blocks >= 1 ? hipSuccess : hipErrorLaunchOutOfResources;
}

template <int RG>
static void launch_generic_rg(const LstmWeights& w, const LstmArgs& a, bool seq, hipStream_t s) {
    const size_t smem = gen_smem_floats(RG, w.H, w.NIN) * 4;
    // (a failed LDS opt-in makes the launch itself fail: forward_impl's hipGetLastError reports it - never a silent no-op)
    if (seq) {
        if (generic_prepare<RG, true>(0) != hipSuccess) set_error("lstm_generic: LDS opt-in of the runtime-sized kernel failed on this device");
        hipLaunchKernelGGL((lstm2_generic_kernel<RG, true>), dim3(a.num_tiles), dim3(kGenThreads), smem, s, w, a);
    } else {
        if (generic_prepare<RG, false>(0) != hipSuccess) set_error("lstm_generic: LDS opt-in of the runtime-sized kernel failed on this device");
        hipLaunchKernelGGL((lstm2_generic_kernel<RG, false>), dim3(a.num_tiles), dim3(kGenThreads), smem, s, w, a);
    }
}

// Commit-time check (fsnp_commit_weights): every instantiation a plan may launch for these sizes gets its LDS opt-in and must be
// resident with its real LDS size; a failure is reported THERE with a clear message instead of surfacing as an asynchronous launch error.
int lstm_generic_check(int H, int NIN, bool seq) {
    hipError_t e = hipSuccess;
    auto one = [&](int rg) -> hipError_t {
        const size_t smem = gen_smem_floats(rg, H, NIN) * 4;
        if (smem > (size_t)150 * 1024) return hipSuccess;           // never planned (lstm_generic_rows_per_group)
        switch (rg) {
            case 8: return seq ? generic_prepare<8, true>(smem) : generic_prepare<8, false>(smem);
            case 4: return seq ? generic_prepare<4, true>(smem) : generic_prepare<4, false>(smem);
            case 2: return seq ? generic_prepare<2, true>(smem) : generic_prepare<2, false>(smem);
            default: return seq ? generic_prepare<1, true>(smem) : generic_prepare<1, false>(smem);
        }
    };
    for (int rg = 1; rg <= 8 && e == hipSuccess; rg *= 2) e = one(rg);
    if (e != hipSuccess) {
        set_error("runtime-sized recurrent kernel (hidden %d, %d inputs): a 1024-thread workgroup with its LDS does not fit a CU of this device: %s",
                  H, NIN, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

// a.num_tiles workgroups of a.coop_rows_per_group (1, 2, 4 or 8) sequences each; a.rows holds num_tiles * rows-per-group slots
void launch_lstm_generic(const LstmWeights& w, const LstmArgs& a, bool seq, hipStream_t s) {
    switch (a.coop_rows_per_group) {
        case 8: launch_generic_rg<8>(w, a, seq, s); break;
        case 4: launch_generic_rg<4>(w, a, seq, s); break;
        case 2: launch_generic_rg<2>(w, a, seq, s); break;
        default: launch_generic_rg<1>(w, a, seq, s); break;
    }
}

}  // namespace fsnp
