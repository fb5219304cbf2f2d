// fsnp_common.h - internal declarations shared by the HIP translation units of libfsnp_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/fsnp.h"
#include "../../include/fsnp_debug.h"

namespace fsnp {

// ---------------------------------------------------------------------------------------------
// Problem dimensions of one forward call (device layouts are time-major: [utt][t][freq]).
struct Dims {
    int B;    // utterances in this call
    int T;    // un-padded frames
    int Tp;   // T + look_ahead
    int F;    // num_freqs (257)
    int FP;   // padded row stride of [.,.,F] buffers (multiple of 4)
    int CH;   // TCN hidden channels (512)
    int H;    // sub-band LSTM hidden (384)
    int NSB;  // 2*sb_num_neighbors+1 (31)
    int NIN;  // LSTM input size (34)
    int LA;   // look_ahead
};

// One (m_t, d_t) pair: normalised = (x - m) / d.
struct NormMD { float m, d; };

// EPSILON of audio_zen/constant.py:8 == np.finfo(np.float32).eps
#define FSNP_EPS 1.1920928955078125e-07f

// Turn running (sum, sumsq, count) into (m, d) for the four norm types (base_model.py:210-316).
__host__ __device__ __forceinline__ NormMD norm_md(int norm_type, double sum, double sq, double count) {
    NormMD r;
    const double mean = sum / count;
    if (norm_type == FSNP_NORM_OFFLINE_LAPLACE) {
        r.m = 0.0f;
        r.d = (float)mean + 1e-5f;
    } else if (norm_type == FSNP_NORM_CUMULATIVE_LAPLACE) {
        r.m = 0.0f;
        r.d = (float)mean + FSNP_EPS;
    } else if (norm_type == FSNP_NORM_OFFLINE_GAUSSIAN) {
        double var = (sq - count * mean * mean) / (count - 1.0);   // torch.std: unbiased
        if (var < 0) var = 0;
        r.m = (float)mean;
        r.d = (float)sqrt(var) + 1e-5f;
    } else {  // cumulative layer norm: var = (pow - 2*mean*sum)/count + mean^2 ; std = sqrt(var + EPS)
        double var = (sq - 2.0 * mean * sum) / count + mean * mean;
        r.m = (float)mean;
        r.d = (float)sqrt(var + (double)FSNP_EPS);
    }
    return r;
}


// ---------------------------------------------------------------------------------------------
// frontend.hip : strided input -> raw, norm statistics, TSSE gate, att = norm(x) * gate
struct FrontendWeights {
    // per branch (mag, real, imag): depthwise conv weights [F][k], biases [F], for 3 kernel sizes
    const float* conv_w[3][3];
    const float* conv_b[3][3];
    const float* cat_w[3];   // [3]
    const float* cat_b[3];   // [1]
    const float* fc1_wT[3];  // [F][F/2]   fc1.weight transposed (thread-per-output, coalesced)
    const float* fc1_b[3];   // [F/2]
    const float* fc2_wT[3];  // [F/2][F]   fc2.weight transposed
    const float* fc2_b[3];   // [F]
    int ksize[3];
    int attention;           // FSNP_ATT_*; for ECA cat_w holds the 3 conv taps
    int subband_num;         // fullsubnet_plus.py:146-153 (ECA only): > 1 groups the magnitude branch's channels
};

struct FrontendBuffers {
    float* raw;       // [3][B][Tp][FP]
    double* frame;    // [3][B][Tp][2]   per-frame (sum, sum of squares) over F
    NormMD* md;       // [3][B][Tp]
    double* fsum;     // [3][B][FP]      per-frequency sum over t: of the RAW input (offline norms: accumulated by the repack kernel) or of
                      //                 the normalised input (cumulative norms: fe_fsum_kernel)
    double* tot;      // [3][B][2]       (sum, sum of squares) of the raw input per branch and utterance (repack kernel), zeroed per forward
    float* gate;      // [3][B][FP]
    float* att;       // [3][B][Tp][FP]
};

// stage-level entry points (fsnp_channel_attention, fsnp_fullband_model): one branch on a caller's [B, F, T] tensor
void launch_attention_stage(const Dims& d, const FrontendWeights& w_slot0, const float* in, const int64_t strides[3],
                            const FrontendBuffers& buf, hipStream_t s);
void launch_repack_plane(const Dims& d, const float* in, const int64_t strides[3], float* raw, hipStream_t s);
void launch_tm_to_bft(const float* tm, float* out, int B, int T, int Tp, int F, int FP, hipStream_t s);       // stages.hip
void launch_apply_cirm(const float* mask, const float* noisy, const int64_t strides[3], float* out,
                       const int64_t out_strides[3], int B, int F, int T, hipStream_t s);
// is_complex: in[0] is the interleaved complex64 STFT buffer (strides[0] in complex elements); mag / real / imag are
// derived inside the repack kernel
// (buf.fsum and buf.tot must be zero when this is called: the repack kernel accumulates the offline norms' statistics into them)
void launch_frontend(const Dims& d, int norm_type, const float* const in[3], const int64_t strides[3][3], bool is_complex,
                     const FrontendWeights& w, const FrontendBuffers& buf, hipStream_t s);
// original FullSubNet: magnitude only - repack into buf.raw [B][Tp][FP] and the norm's (m_t, d_t) table into buf.md
void launch_frontend_mag(const Dims& d, int norm_type, const float* mag, const int64_t strides[3], bool is_complex,
                         const FrontendBuffers& buf, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// tcn.hip : 8 x TCNBlock + ReLU + Linear + activation for the three full-band branches at once
struct TcnWeights {
    // packed per branch/block; see pack_tcn() in fsnp_abi.hip for layouts
    const float* w1;     // [3][NB][N1P][K1P]  conv1x1  (N1P = CH padded to 64, K1P = F padded to 16)
    const float* b1;     // [3][NB][N1P]
    const float* a1;     // [3][NB]            PReLU slope
    const float* g1w;    // [3][NB][CH]        GroupNorm gamma
    const float* g1b;    // [3][NB][CH]
    const float* dw;     // [3][NB][3][CH]     depthwise taps, tap-major
    const float* db;     // [3][NB][CH]
    const float* a2;     // [3][NB]
    const float* g2w;    // [3][NB][CH]
    const float* g2b;    // [3][NB][CH]
    const float* w2;     // [3][NB][N2P][K2P]  sconv  (N2P = F padded to 64, K2P = CH padded to 16)
    const float* b2;     // [3][NB][N2P]
    const float* w2g;    // [3][NB][N2P][K2P]  sconv weights times norm2.weight[k]  (GroupNorm folded: tcn_gemm_dma_kernel)
    const float* c1;     // [3][NB][N2P]       sconv.bias[n] + sum_k norm2.bias[k] W2[n][k]
    const float* c2;     // [3][NB][N2P]       sum_k norm2.weight[k] W2[n][k]
    const float* wf;     // [3][N2P][K1P]      fc_output_layer
    const float* bf;     // [3][N2P]
    int NB, N1P, K1P, N2P, K2P;
    int num_cus;         // for the per-launch column-tile choice (N1P, N2P are multiples of 384: any BN fits)
    int dilation[16];
    int gemm_dma;        // 1 = the full-band GEMMs run on tcn_gemm_dma_kernel / (small batches) tcn_gemm_sk_kernel where their requirements hold,
                         // 2 = never the small-batch kernel, 3 = the 128-row DMA kernel only (neither the small-batch nor the 64-row sconv
                         // kernel), 0 = the general kernel (fsnp_debug_set_gemm_dma)
};

struct TcnBuffers {
    const float* att;  // [3][B][Tp][FP]  input (kept intact)
    float* x;          // [3][B][Tp][FP]  running activation
    float* y1;         // [3][B][Tp][CH]
    float* y2;         // [3][B][Tp][CH]
    double* gn;        // [NB][2][3][B][kGnStride] GroupNorm (sum, sumsq) accumulators (one 128-byte line per pair), zeroed per forward
    float* fb;         // [3][B][Tp][FP]  output
    float* dbg_tcn0;   // optional [B][Tp][FP]: mag branch after block 0
};

void launch_tcn(const Dims& d, int fb_act, const TcnWeights& w, const TcnBuffers& buf, hipStream_t s, int branches = 3);
// C[utt][t][0..N) = act(A[utt][t][0..K) * W^T + bias), W [N pad 384][ldw] zero padded.  a_utt_stride / a_cols: optional
// utterance stride of A and readable floats per row when rows overlap (lda < K: the STFT's hop-strided frames).
void launch_linear_act(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int K,
                       int N, int B, int Tp, int act, int num_cus, hipStream_t s, long a_utt_stride = 0, int a_cols = 0);
// stft.hip : SURVEY.md 8(f-3) second half - torch.stft / torch.istft of audio_zen/acoustics/feature.py:10-56 as
// reflect-pad + DFT GEMM and inverse-DFT GEMM + windowed overlap-add (n_fft = win_length, hop = n_fft / 2, hann)
void launch_stft_pad(const float* wav, long wav_stride, float* xp, long xp_stride, int B, int L, int n_fft, hipStream_t s);
void launch_istft_ola(const float* frames, const float* window, float* wav, long wav_stride, int B, int T, int L, int n_fft,
                      hipStream_t s);
void stft_build_matrices(int n_fft, float* fwd /*[N2 pad 384][n_fft]*/, float* inv /*[n_fft pad 384][K pad 16]*/,
                         float* window /*[n_fft]*/);

// ---------------------------------------------------------------------------------------------
// stages.hip : BaseModel's public helpers as stage-level entry points (fsnp_norm, fsnp_unfold)
int launch_norm_stage(int norm_type, const float* in, const int64_t strides[4], float* out, int B, int C, int F, int T, hipStream_t s);
void launch_unfold_stage(const float* in, const int64_t strides[4], float* out, int B, int C, int F, int T, int num_neighbor, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// subband.hip : statistics of the (never materialised) sub-band input tensor
struct SubbandBuffers {
    const float* att_mag;  // [B][Tp][FP]
    const float* fb;       // [NFB][B][Tp][FP]  NFB = NIN - NSB full-band branches (3: FullSubNet+, 1: FullSubNet)
    const float* refl_w;   // [F] multiplicity of each frequency row inside the unfold of the magnitude (sb_num_neighbors)
    const float* refl_wfb; // [F] the same for the unfold of the full-band outputs (fb_num_neighbors; all ones for 0)
    int NFBN;
    double* acc;           // [B][2]  (sum, sumsq) over the whole [F,NIN,Tp] tensor, zeroed per forward
    NormMD* md_utt;        // [B]
    NormMD* md_row;        // [Nrows][Tp] (cumulative norms only)
};
// one sub-band sequence slot.  b = utterance (gather mode) or sequence index (dense mode)
struct RowDesc { int b, f, out_off, valid; };
// GroupNorm(1, C) statistics of the TCN stacks: one {sum, sum of squares} fp64 pair per (block, norm, branch, utterance), fed by one
// atomic pair per workgroup; doubles between consecutive pairs (a 128-byte line each: the atomics of different planes do not serialise on one line)
constexpr int kGnStride = 16;

void launch_subband_stats(const Dims& d, int norm_type, const SubbandBuffers& buf, const RowDesc* rows,
                          int num_slots, hipStream_t s);
// sequence_model="TCN" only: materialise the normalised sub-band input x[slot][t][xstride] (the recurrent kernels gather
// it on the fly instead), and scatter y[slot][t][0..1] into the caller's mask tensor (look-ahead slice included)
struct SbGatherArgs {
    const float* att_mag; int fb_rel, fb_branch_stride;
    const RowDesc* rows; const NormMD* md_utt; const NormMD* md_row;
    float* x; int xstride;
    int num_slots, Tp, FP, F, NSBN, NFBN, NIN;
};
void launch_sb_gather(const SbGatherArgs& a, hipStream_t s);
void launch_sb_scatter(const float* y, int ystride, const RowDesc* rows, float* out, long out_stride_o, int num_slots,
                       int Tp, int LA, int out_channels, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// lstm.hip : fused 2-layer LSTM + Linear over 32-sequence tiles, one workgroup per tile
struct LstmWeights {
    const float* wpack;  // MFMA-fragment-ordered [wave][layer-0 stream | layer-1 stream], KX = 40
    const float* wpack12; // same for the 12-wave kernel (32 hidden units per wave)
    const float* wpack_bf[2];   // bf16-ih streams (4-wave, 12-wave): layer-1 W_ih as bf16 k-steps (configs[4])
    int ih_bf16;                // 1 = use them
    const float* wpack_coop[4]; // column-split kernel, 8 << i hidden units per workgroup: [split][k-group][tile][lane][4]
    const float* wpack_coopn;   // three-way column-split kernel (lstm_coopn.hip): [32-unit block][k-group][gate][lane][4]
    const float* wpack16;       // half-tile kernel (lstm16.hip): [wave][k-group of 16][24 tiles of 16 columns][lane][4]
    const float* wpack16_bf;    // its bf16-ih stream: layer-1 W_ih as bf16 k-steps of 32 (configs[4]); nullptr = not packed
    const float* wpack_gru;     // one-tile-per-CU GRU kernel (lstm_gru.hip): [wave][k-group][3 live tiles x ST][lane][4]
    const float* wpack_hp;      // half-tile ping-pong kernel (lstm_hp.hip): [column slice of 16 units][gate][k-group of 16][lane][4]
    const float* wpack_hpw;     // the same launch shape with wave-owned units (lstm_hpw.hip, round 6): [participant = 4 cs + wave][fragment][lane][4]; nullptr = not packed
    int hp_wave;                // 1 = planner kind 8 launches run on lstm_hpw.hip (default), 0 = on lstm_hp.hip (FSNP_HP_WAVE=0)
    const float* wpack_fbv;     // full-band LSTM of FullSubNet for <= 4 utterances on the VALU (lstm_fbv.hip): [slice][fragment][thread][4]; nullptr = not packed
    const float* wpack_coopw;   // wave-owned column split (lstm_coopw.hip): [k-group][8-unit block, gate-interleaved columns][lane][4]; nullptr = not packed
    const float* wgen;          // runtime-sized kernel (lstm_generic.hip): transposed [layer][k][4H], layer 0 k = [x | h0], layer 1 = [h0 | h1]
    int gru;             // 1 = nn.GRU cell (column-split kernels only); weights / biases are packed as 4 slots r, z, n_x, n_h
    int waves;           // 4 or 12 waves per workgroup
    const float* bias;   // [2][4H]  b_ih + b_hh, reference gate order i,f,g,o
    const float* wfc;    // [OUT][H]
    const float* bfc;    // [OUT]
    int H, NIN, KX, OUT;
};
struct LstmArgs {
    // gather mode (dense == nullptr): x_j(t) built from att_mag / fb (fullsubnet_plus.py:167-189)
    const float* att_mag;  // [B][Tp][FP]
    const float* fb;       // [3][B][Tp][FP]
    int fb_rel;            // fb - att_mag in floats (same workspace allocation)
    int fb_branch_stride;  // floats between fb branches
    const RowDesc* rows;   // [num_tiles*32]
    const NormMD* md_utt;  // [B]   (offline norms)
    const NormMD* md_row;  // [rows][Tp] or nullptr
    // dense mode: x[row.b][t][dense_stride] (first NIN entries of each row)
    const float* dense;
    int dense_stride;          // cooperative kernel only (the row-tile kernel assumes NIN)
    const NormMD* md_seq;      // cooperative kernel only: optional [sequence = row.b][Tp] table, overrides md_utt / md_row
    float* seq_out;            // cooperative SEQ kernel: h1 of the second layer, [row.b][t][H]
    int seq_stride;            // floats between its rows (0 = H); >= H
    float* out;            // out[row.out_off + o*out_stride_o + (t-LA)]
    long out_stride_o;
    int num_rows;          // valid rows
    int num_tiles;         // workgroups; rows[] holds num_tiles * (32 + ex) slots
    int ex;                // VALU rows per tile: 0, 1, 2 or 4
    int Tp, LA, FP, F, NSBN;  // NSBN = sb_num_neighbors
    int NFBN;                 // fb_num_neighbors
    int act;               // FSNP_ACT_* on the Linear output
    unsigned long long* prof;  // optional [Tp][8] s_memtime stamps of workgroup 0 (debug)
    // column-split (cooperative) kernel only
    float* coop_hx;            // per row tile: h0/h1 exchange images (double buffered) + Linear partials, zeroed per launch
    unsigned* coop_bar;        // per row tile arrival counter, zeroed per launch
    unsigned* coop_bar2;       // second counter per row tile (layer-skewed K-split kernel: finished layer-1 phases)
    int coop_skew;             // lstm_coop.hip: 1 = layer-skewed schedule (lstm2_coop_skew_kernel)
    unsigned* coop_err;        // host-mapped: set to 1 if a barrier wait timed out
    unsigned* coop_abort;      // device word (zeroed per forward): raised by the first waiter that gives up, polled by all
    int coop_units;            // hidden units per workgroup: 8, 16, 32 or 64
    int coop_bar_stride;       // words between the counters of consecutive row tiles (lstm_common.h: 64, second counter at + 32); 0 = packed: coop_bar[tile], coop_bar2[tile]
    int coop_xcd;              // > 0 = CUs per XCD: place the workgroups that share a row tile on one XCD (lstm_common.h)
    int coop_own_cu;           // lstm_coop.hip: > 0 = claim this many bytes of dynamic LDS (the whole CU's) so that no workgroup of a
                               // concurrent kernel that needs LDS shares the CU (deferred remainder chunk in the pipelined loop)
    int coop_groups;           // lstm_coopn.hip: groups of 3 workgroups; group g owns row tiles g, g + groups
    int coop_rows_per_group;   // lstm_coopn.hip: 1 or 2 row tiles per group; lstm_generic.hip: sequences per workgroup
    int coop_corrupt;          // test hook (fsnp_debug_corrupt_exchange): > 0 = the workgroup / wave that owns hidden unit 0 of row tile 0 publishes
                               // h0 of row 0 at step coop_corrupt - 1 with 1.0 added (its own state stays right): a corrupted exchange image
    int coop_chaos;            // test hook (fsnp_debug_set_chaos): != 0 = seed of pseudo-random, workgroup-uniform delays at the phase boundaries
                               // of the column-split kernels, so that the workgroups of a launch drift apart instead of running in lockstep
    unsigned long long* clk;   // one-tile-per-CU LSTM kernel (lstm.hip) only, optional (device memory, 8 words): workgroup 0 writes {s_memtime,
                               // s_memrealtime} when it starts into clk[0..1] and when it ends into clk[2..3]; every workgroup folds its duration into
                               // clk[4] (max, 100 MHz ticks), clk[5] (max, s_memtime ticks), clk[6] (min, 100 MHz ticks): fsnp_debug_launch_clock
};

struct LstmPlan { int num_tiles, ex, rows_per_slot_tile; };
LstmPlan plan_lstm_tiles(int num_rows, int num_cus);
void launch_lstm(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
// lstm16.hip: the same decomposition on 16-row tiles (v_mfma_f32_16x16x4_f32): 4096 sequences per round of 256 workgroups
void launch_lstm16(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
size_t lstm16_pack_floats(int H, int KX);
size_t lstm16_pack_floats_bf16ih(int H, int KX);
void lstm16_pack_weights_bf16ih(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* wpack);
void lstm16_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* wpack);
// lstm_gru.hip: the same decomposition for nn.GRU (three live gate tiles per k-group, no VALU rows)
void launch_gru(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
size_t gru_pack_floats(int H, int KX, int NW);
void gru_pack_weights(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1,
                      const float* whh1, float* wpack);   // inputs: the four-slot [4H][cols] matrices (r, z, n_x | n_h)
// lstm_coop.hip: column-split kernel for small batches (row_tiles * H/32 workgroups, all co-resident)
void launch_lstm_coop(const LstmWeights& w, const LstmArgs& a, hipStream_t s);       // H = 384, KX = 40, Linear(H, 2) fused
void launch_lstm_coop_seq(const LstmWeights& w, const LstmArgs& a, hipStream_t s);   // H = 512, KX = 264, h1 sequence out
size_t lstm_coop_pack_floats(int H, int KX, int units);
void lstm_coop_pack_weights(int H, int NIN, int KX, int units, const float* wih0, const float* whh0, const float* wih1,
                            const float* whh1, float* wpack);
size_t lstm_coop_exchange_bytes(int H, int row_tiles);
// lstm_hp.hip: 16 units per workgroup (S = H / 16 workgroups per row tile, one XCD), waves split the gates, weights resident,
// every row tile worked on as two half tiles of 16 sequences in turn (the hand-off of one half hidden behind the other)
void launch_lstm_hp(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
bool lstm_hp_available(const LstmWeights& w);
// lstm_hpw.hip: the same launch shape, exchange region and counters; a WAVE owns 4 units x 4 gates over the whole K (transposed MFMA:
// lane-local cells), operands streamed L2 -> registers, no workgroup barrier in the time loop
void launch_lstm_hpw(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
bool lstm_hpw_available(const LstmWeights& w);
size_t lstm_hpw_pack_floats(int H, int KX);
void lstm_hpw_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out);
size_t lstm_hp_pack_floats(int H, int KX);
void lstm_hp_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out);
// lstm_coopw.hip: a wave owns 8 NT hidden units over the full K, S = H / (32 NT) workgroups per row tile (a.coop_units = 32 NT in
// {32, 64}), layer-skewed schedule without workgroup barriers: 11 ... 42 row tiles per launch (B = 2 ... 8)
void launch_lstm_coopw(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
bool lstm_coopw_available(const LstmWeights& w, int units);
int lstm_coopw_occupancy(const LstmWeights& w, int units);
size_t lstm_coopw_pack_floats(int H, int KX);
void lstm_coopw_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out);
// lstm_fbv.hip: the full-band LSTM(num_freqs -> 512 x 2) of the original FullSubNet for 1 ... 4 utterances as matrix-VECTOR products
// on the VALU (weights resident, H / 8 workgroups, serial schedule with one hand-off per step); h1 sequence out
void launch_lstm_fbv(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
bool lstm_fbv_available(const LstmWeights& w, int batch, int num_cus);
size_t lstm_fbv_pack_floats(int H);
void lstm_fbv_pack_weights(int H, int NIN, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out);
// lstm_generic.hip: runtime-sized fp32-FMA kernel for the sizes no tuned kernel is instantiated for (any hidden size / input width);
// a.num_tiles workgroups of a.coop_rows_per_group (1, 2, 4, 8) sequences; seq = the full-band model of the original FullSubNet
void launch_lstm_generic(const LstmWeights& w, const LstmArgs& a, bool seq, hipStream_t s);
size_t lstm_generic_pack_floats(int H, int NIN);
void lstm_generic_pack_weights(int H, int NIN, const float* wih0, const float* whh0, const float* wih1, const float* whh1, float* out);
int lstm_generic_rows_per_group(int H, int NIN, int num_seq, int num_cus);   // 0 = the sizes do not fit a CU's LDS
int lstm_generic_check(int H, int NIN, bool seq);   // commit time: LDS opt-in + residency of every instantiation; != 0 (error set) on failure
// lstm_coopn.hip: 3 workgroups x 128 hidden units share 1-2 row tiles (43..170 row tiles)
void launch_lstm_coopn(const LstmWeights& w, const LstmArgs& a, hipStream_t s);
size_t lstm_coopn_pack_floats(int H, int KX);
void lstm_coopn_pack_weights(int H, int NIN, int KX, const float* wih0, const float* whh0, const float* wih1,
                             const float* whh1, float* wpack);
int lstm_coopn_plan(int H, int row_tiles, int num_cus, int* groups);   // rows tiles per group (0 = not applicable)
int lstm_coop_pick_units(int H, int row_tiles, int num_cus, int min_units);   // 0 = not applicable
// workgroups of one instantiation that fit a CU at once (hipOccupancyMaxActiveBlocksPerMultiprocessor; 0 = unknown)
int lstm_coop_occupancy(const LstmWeights& w, int units);
int lstm_coopn_occupancy(const LstmWeights& w, int rows_per_group);
size_t lstm_pack_floats(int H, int KX, int NW);  // size of wpack in floats
// host-side packer: W_ih0 [4H][NIN], W_hh0 [4H][H], W_ih1 [4H][H], W_hh1 [4H][H] -> wpack
size_t lstm_pack_floats_bf16ih(int H, int KX, int NW);
void lstm_pack_weights_bf16ih(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1,
                              const float* whh1, const float* bias0, float* wpack);
void lstm_pack_weights(int H, int NIN, int KX, int NW, const float* wih0, const float* whh0, const float* wih1,
                       const float* whh1, float* wpack);

// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define FSNP_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            fsnp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Runs `f` once per device (hipFuncSetAttribute is per device; one process may drive several GPUs from several threads).
// A race can run it twice, never zero times before the launch that follows.
struct PerDeviceOnce {
    unsigned long long done = 0;
    template <typename F>
    void run(F f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
        if (bit && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) return;
        f();
        if (bit) __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
    }
};

// reflect index used by BaseModel.unfold's reflect padding (base_model.py:38): refl(-k)=k, refl(F-1+k)=F-1-k
__host__ __device__ inline int reflect_index(int i, int F) {
    if (i < 0) i = -i;
    if (i >= F) i = 2 * (F - 1) - i;
    return i;
}

// Float offset (from att_mag) of feature j of the sub-band input of frequency f, frame 0 of utterance base
// (fullsubnet_plus.py:167-188 / fullsubnet.py:91-100): the 2 nsbn + 1 reflect-padded neighbours of the (attention)
// magnitude, then for every full-band branch its 2 nfbn + 1 reflect-padded neighbours (BaseModel.unfold, base_model.py:15-47).
__host__ __device__ inline int sb_feature_offset(int j, int f, int base, int F, int nsbn, int nfbn, int fb_rel, int fb_branch_stride) {
    const int nsb = 2 * nsbn + 1;
    if (j < nsb) return base + reflect_index(f - nsbn + j, F);
    const int nfb = 2 * nfbn + 1, jj = j - nsb;
    return fb_rel + (jj / nfb) * fb_branch_stride + base + reflect_index(f - nfbn + jj % nfb, F);
}

}  // namespace fsnp
