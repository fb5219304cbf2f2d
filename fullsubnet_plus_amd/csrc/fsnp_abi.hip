// fsnp_abi.hip - C ABI of libfsnp_hip.so (include/fsnp.h): handle, strict weight loading + packing,
// workspace management and the forward orchestration.  Host code only; kernels live in
// frontend.hip / tcn.hip / subband.hip / lstm.hip.
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

struct WeightSpec {
    std::string name;
    int64_t numel;
};

struct Workspace {
    // offsets in bytes from the workspace base
    size_t att, fb, raw, x, y1, y2, gate, md, md_utt, md_row, rows, fb_rows, frame, sbt_x0, sbt_x, sbt_fb, sbt_y1, sbt_y2, zero_begin,
        fsum, gn, sb_acc, coop_hx, coop_bar, coop_abort, fb_hx, fb_bar, sbt_gn, zero_end, dbg_tcn0, total;
};

// The full-band part of a FullSubNet+ forward is ~75 tiny launches that only touch the workspace (0.9 ms at B = 1); it
// can be captured once per (shape, mode, plan) into a hipGraph and replayed (opt-in: FSNP_GRAPH=1).  Measured: no gain -
// the launches are asynchronous and the host runs ahead of the GPU, so the chain is bound by the kernels' own latency.
struct GraphKey {
    int B, T, mode, boff, gb, num_cus, coop, bf16, debug;
    const void* ws;
    const void* weights;
    bool operator==(const GraphKey& o) const {
        return B == o.B && T == o.T && mode == o.mode && boff == o.boff && gb == o.gb && num_cus == o.num_cus &&
               coop == o.coop && bf16 == o.bf16 && debug == o.debug && ws == o.ws && weights == o.weights;
    }
};
struct GraphEntry { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; };

struct TimingRec {
    hipEvent_t e[4];  // start, after full-band stages, after the sub-band model (= end), after its FIRST chunk
};

// Every entry point that touches the device runs on the handle's device and puts the caller's current device back.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define FSNP_ON_DEVICE(h)                                                              \
    fsnp::DeviceGuard _dev_guard((h)->device);                                         \
    if (!_dev_guard.ok) { fsnp::set_error("hipSetDevice(%d) failed", (h)->device); return 1; }

}  // namespace fsnp

using namespace fsnp;

struct fsnp_handle {
    fsnp_config cfg{};
    int device = 0;
    int F = 0, FP = 0, CH = 0, H = 0, NSB = 0, NIN = 0, KX = 0, NB = 0, Fr = 0;
    std::vector<WeightSpec> specs;
    std::map<std::string, std::vector<float>> host_w;
    bool committed = false;

    float* d_weights = nullptr;
    FrontendWeights fw{};
    TcnWeights tw{};
    LstmWeights lw{};
    // original FullSubNet only: full-band 2-layer LSTM(F -> CH) (cooperative kernel) + Linear(CH, F) (GEMM)
    int model = FSNP_MODEL_FULLSUBNET_PLUS;
    int NFB = 3;                 // full-band features per sub-band frame: 3 (FullSubNet+) or 1 (FullSubNet)
    int gru = 0;                 // 1 = nn.GRU cells (sub-band model; FullSubNet: also the full-band model)
    bool lstm16_ok = false;      // the half-tile kernel (lstm16.hip) exists for this handle (LSTM, H = 384, K = 40) and is enabled
    bool rowtile_ok = true;      // a one-tile-per-CU kernel (lstm.hip / lstm_gru.hip) exists for this handle's sub-band model
    CostTable cost{};            // per-step costs the planner minimises (defaults, then measured on the device)
    int coop_occ = 1;            // workgroups per CU the column-split kernels may be planned with (FSNP_COOP_OCC; 1 or 2) ...
    int occ_ksplit[4] = {1, 1, 1, 1}, occ_coopn[2] = {1, 1};   // ... and what each instantiation really fits (measured at commit)
    int calibrate = 0;           // FSNP_CALIBRATE=1: replace the built-in table by one measured on this device at the first planning call
    int sb_tcn = 0;              // 1 = the sub-band model is a TCN stack (FullSubNet+ with sequence_model="TCN")
    TcnWeights sbt{};            //     its weights (one branch, NIN input channels)
    int XS = 0;                  //     row stride of its [slot][t][NIN] activations
    int NG = 4;                  // gate blocks per weight matrix: 4 (LSTM) or 3 (GRU)
    LstmWeights fbw{};
    const float* fsn_wf = nullptr;   // [F pad 384][CH pad 16]
    const float* fsn_bf = nullptr;   // [F pad 384]
    int fsn_kp = 0;
    const float* d_refl_w = nullptr;
    const float* d_refl_wfb = nullptr;

    unsigned char* ws = nullptr;
    size_t ws_bytes = 0;
    Workspace last_ws{};
    Dims last_dims{};
    bool have_last = false;
    bool debug = false;
    int num_cus = 256;
    int num_cus_real = 256;   // never overridden: residency of the cooperative kernel depends on the real chip
    int ih_bf16 = 0;             // 1 = BASELINE.json configs[4]: layer-1 ih-GEMM of the sub-band LSTM in bf16
    int lstm_coop = 1;           // 0 = never, 1 = automatic (small batches)
    int coop_skew = 1;           // K-split kernel: 1 = layer-skewed schedule (lstm2_coop_skew_kernel), 0 = the serial one (FSNP_COOP_SKEW=0)
    int coop_split_cfg = 1;      // (coop_split as configured at fsnp_create: fsnp_debug_set_lstm_coop(h, 2) turns it off, 1 restores it)
    bool generic_sb = false;     // the sub-band recurrent model runs on the runtime-sized kernel (lstm_generic.hip): a hidden size or an
                                 // input width no tuned kernel is instantiated for
    bool generic_fb = false;     // FullSubNet: the same for the full-band recurrent model (fb_model_hidden_size != 512 or > 264 bins)
    bool pp_ok = false;          // the ping-pong K-split kernel (lstm_pp.hip) exists for this handle's sub-band model
    int coop_pp = 0, coop_pp_cfg = 0;   // ... and the planner may use it (opt-in: FSNP_COOP_PP=1 / fsnp_debug_set_lstm_coop(h, 3))
    int coop_split = 1;          // K-split kernel: 1 = the planner may use the role-split schedule (lstm2_coop_split_kernel: 2 S workgroups
                                 // per row tile), 0 = never, 2 = wherever it fits (FSNP_COOP_SPLIT, tuning)
    unsigned* d_err = nullptr;   // [0] = an inter-workgroup wait timed out in a column-split LSTM kernel.  Host-mapped,
                                 // so the NEXT call on the handle can fail loudly without a device synchronisation
    int lstm_waves = 0;   // 0 = auto: 12 waves when the tile plan uses VALU rows, else 4

    // STFT / iSTFT around the model (stft.hip): DFT GEMM operands, built on first use, and an I/O workspace
    float* d_stft = nullptr;     // [fwd (2F pad 384) x n_fft][inv (n_fft pad 384) x (2F pad 16)][window n_fft][zero bias 768]
    unsigned char* io = nullptr;
    size_t io_bytes = 0;

    double composite_gain = 0.97;   // a row-tile + remainder plan must be estimated this much cheaper to be chosen
    int use_graph = 0;           // 0 = plain launches (default: measured no gain), 1 = replay on the
                                 // private stream, 2 = replay straight into the caller's stream (FSNP_GRAPH=1|2)
    hipStream_t cap_stream = nullptr;   // private stream: graph capture and replay
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::vector<GraphEntry> graphs;

    bool timing = false;
    std::vector<TimingRec> timing_recs;   // recorded, not yet read back (drained by fsnp_get_timing, or when 256 pile up)
    std::vector<hipEvent_t> event_pool;   // events are re-used: a forward with timing on allocates nothing in steady state
    double acc_ms[4] = {0, 0, 0, 0};
    int64_t acc_cnt[4] = {0, 0, 0, 0};

    // pipelined serving mode (fsnp_set_pipeline): the column-split remainder chunks that follow a row-tile chunk run on
    // `side_stream`, so that they overlap the full-band stages of the NEXT forward (which leave most CUs idle); the
    // workspace is double buffered because forward i+1 rebuilds att / fb while the remainder of forward i still reads them
    int pipeline = 0;
    int defer_small = 1;         // pipelined mode: plans that start with a column-split launch run on the side stream whole (FSNP_DEFER_SMALL=0: off)
    int ws_slots = 1, ws_slot = 0;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_side[2] = {nullptr, nullptr};
    bool side_used[2] = {false, false};
    unsigned char* last_base = nullptr;   // workspace half of the last forward (fsnp_read_stage)
    hipEvent_t ev_done = nullptr;         // end of the last forward on its stream: a forward on another stream waits for it
    hipStream_t done_stream = nullptr;
    bool done_valid = false;
};

namespace fsnp {

// what the planner needs to know of a handle
static PlannerCtx pctx(const fsnp_handle* h) {
    PlannerCtx c;
    c.H = h->H; c.NIN = h->NIN; c.num_cus = h->num_cus; c.num_cus_real = h->num_cus_real;
    c.gru = h->gru != 0; c.sb_tcn = h->sb_tcn != 0; c.generic_sb = h->generic_sb; c.rowtile_ok = h->rowtile_ok; c.lstm16_ok = h->lstm16_ok;
    c.pp_ok = h->pp_ok; c.ih_bf16 = h->ih_bf16; c.lstm_coop = h->lstm_coop; c.coop_occ = h->coop_occ;
    for (int i = 0; i < 4; ++i) c.occ_ksplit[i] = h->occ_ksplit[i];
    for (int i = 0; i < 2; ++i) c.occ_coopn[i] = h->occ_coopn[i];
    c.coop_split = h->coop_split; c.coop_pp = h->coop_pp; c.pipeline = h->pipeline; c.composite_gain = h->composite_gain;
    c.cost = h->cost;
    return c;
}

static const int kDilations[8] = {1, 2, 5, 9, 1, 2, 5, 9};  // sequence_model.py:48-57
static const char* kAtt[3] = {"channel_attention", "channel_attention_real", "channel_attention_imag"};
static const char* kFb[3] = {"fb_model", "fb_model_real", "fb_model_imag"};
static const char* kConvNames[3] = {"smallConv1d", "middleConv1d", "largeConv1d"};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void drop_graphs(fsnp_handle* h) {
    for (auto& e : h->graphs) { (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph); }
    h->graphs.clear();
}

// Launches a cached graph on the handle's private stream, ordered after / before the caller's stream with events
// (the caller's stream is usually torch's legacy default stream, which cannot be captured; keeping the replay on the
// capture stream makes the ordering explicit instead of relying on default-stream semantics).
static int launch_graph_between(fsnp_handle* h, hipGraphExec_t exec, hipStream_t s) {
    if (h->use_graph == 2) { FSNP_HIP_CHECK(hipGraphLaunch(exec, s)); return 0; }     // straight into the caller's stream
    FSNP_HIP_CHECK(hipEventRecord(h->ev_in, s));
    FSNP_HIP_CHECK(hipStreamWaitEvent(h->cap_stream, h->ev_in, 0));
    FSNP_HIP_CHECK(hipGraphLaunch(exec, h->cap_stream));
    FSNP_HIP_CHECK(hipEventRecord(h->ev_out, h->cap_stream));
    FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_out, 0));
    return 0;
}

// Replays `middle` (workspace-only launches) from a cached hipGraph; captures it on the private stream the first time.
template <typename F>
static int run_graphed(fsnp_handle* h, const GraphKey& key, hipStream_t s, F middle) {
    if (!h->cap_stream) {
        FSNP_HIP_CHECK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    }
    for (auto& e : h->graphs)
        if (e.key == key) return launch_graph_between(h, e.exec, s);
    FSNP_HIP_CHECK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    middle(h->cap_stream);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(h->cap_stream, &g);
    if (ec != hipSuccess || g == nullptr) {
        set_error("hipGraph capture of the full-band stages failed: %s (set FSNP_GRAPH=0 to launch kernel by kernel)", hipGetErrorString(ec));
        return 4;
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        set_error("hipGraphInstantiate failed: %s (set FSNP_GRAPH=0 to launch kernel by kernel)", hipGetErrorString(ei));
        return 4;
    }
    if (h->graphs.size() >= 8) {
        (void)hipGraphExecDestroy(h->graphs.front().exec);
        (void)hipGraphDestroy(h->graphs.front().graph);
        h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back({key, g, ex});
    return launch_graph_between(h, ex, s);
}

static void build_specs(fsnp_handle* h) {
    auto add = [&](const std::string& n, int64_t numel) { h->specs.push_back({n, numel}); };
    const int F = h->F, CH = h->CH, H = h->H, Fr = h->Fr;
    const int att = h->cfg.attention;
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    if (fsn) {   // fullsubnet.py:39-47: SequenceModel(257 -> 512 x 2 -> 257), same key layout as nn.LSTM
        const std::string f = "fb_model.sequence_model.";
        const int64_t G = h->NG;
        add(f + "weight_ih_l0", G * CH * F);
        add(f + "weight_hh_l0", G * CH * CH);
        add(f + "bias_ih_l0", G * CH);
        add(f + "bias_hh_l0", G * CH);
        add(f + "weight_ih_l1", G * CH * CH);
        add(f + "weight_hh_l1", G * CH * CH);
        add(f + "bias_ih_l1", G * CH);
        add(f + "bias_hh_l1", G * CH);
        add("fb_model.fc_output_layer.weight", (int64_t)F * CH);
        add("fb_model.fc_output_layer.bias", F);
    }
    for (int a = 0; a < 3 && !fsn; ++a) {
        const std::string p = kAtt[a];
        if (att == FSNP_ATT_TSSE) {
            for (int c = 0; c < 3; ++c) {
                add(p + "." + kConvNames[c] + ".0.weight", (int64_t)F * h->cfg.kersize[c]);
                add(p + "." + kConvNames[c] + ".0.bias", F);
            }
            add(p + ".feature_concate_fc.weight", 3);
            add(p + ".feature_concate_fc.bias", 1);
        }
        if (att == FSNP_ATT_ECA) {
            add(p + ".conv.weight", 3);
        } else {
            add(p + ".fc1.weight", (int64_t)Fr * F);
            add(p + ".fc1.bias", Fr);
            add(p + ".fc2.weight", (int64_t)F * Fr);
            add(p + ".fc2.bias", F);
        }
    }
    for (int b = 0; b < 3 && !fsn; ++b) {
        for (int i = 0; i < h->NB; ++i) {
            const std::string p = std::string(kFb[b]) + ".sequence_model." + std::to_string(i);
            add(p + ".conv1x1.weight", (int64_t)CH * F);
            add(p + ".conv1x1.bias", CH);
            add(p + ".prelu1.weight", 1);
            add(p + ".norm1.weight", CH);
            add(p + ".norm1.bias", CH);
            add(p + ".depthwise_conv.weight", (int64_t)CH * 3);
            add(p + ".depthwise_conv.bias", CH);
            add(p + ".prelu2.weight", 1);
            add(p + ".norm2.weight", CH);
            add(p + ".norm2.bias", CH);
            add(p + ".sconv.weight", (int64_t)F * CH);
            add(p + ".sconv.bias", F);
        }
        add(std::string(kFb[b]) + ".fc_output_layer.weight", (int64_t)F * F);
        add(std::string(kFb[b]) + ".fc_output_layer.bias", F);
    }
    if (h->sb_tcn) {
        for (int i = 0; i < 8; ++i) {
            const std::string p = "sb_model.sequence_model." + std::to_string(i);
            add(p + ".conv1x1.weight", (int64_t)CH * h->NIN);
            add(p + ".conv1x1.bias", CH);
            add(p + ".prelu1.weight", 1);
            add(p + ".norm1.weight", CH);
            add(p + ".norm1.bias", CH);
            add(p + ".depthwise_conv.weight", (int64_t)CH * 3);
            add(p + ".depthwise_conv.bias", CH);
            add(p + ".prelu2.weight", 1);
            add(p + ".norm2.weight", CH);
            add(p + ".norm2.bias", CH);
            add(p + ".sconv.weight", (int64_t)h->NIN * CH);
            add(p + ".sconv.bias", h->NIN);
        }
        add("sb_model.fc_output_layer.weight", (int64_t)h->cfg.output_size * h->NIN);
        add("sb_model.fc_output_layer.bias", h->cfg.output_size);
        return;
    }
    const std::string s = "sb_model.sequence_model.";
    const int64_t G = h->NG;
    add(s + "weight_ih_l0", G * H * h->NIN);
    add(s + "weight_hh_l0", G * H * H);
    add(s + "bias_ih_l0", G * H);
    add(s + "bias_hh_l0", G * H);
    add(s + "weight_ih_l1", G * H * H);
    add(s + "weight_hh_l1", G * H * H);
    add(s + "bias_ih_l1", G * H);
    add(s + "bias_hh_l1", G * H);
    add("sb_model.fc_output_layer.weight", (int64_t)h->cfg.output_size * H);
    add("sb_model.fc_output_layer.bias", h->cfg.output_size);
}

// Row slots of the sub-band problem.  Tile i owns `rt` slots (32 MFMA rows + ex VALU rows) and gets
// base (+1 for the first rem tiles) consecutive sequences; slot -> (utterance, frequency, output offset).
__global__ void build_rows_kernel(RowDesc* rows, int num_rows, int num_tiles, int rt, int F, int T, int mode,
                                  int batch_offset, int global_batch, int dense_out, int n_base, int groups) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= num_tiles * rt) return;
    const int tile = slot / rt, sl = slot % rt;
    const int base = num_rows / num_tiles, rem = num_rows % num_tiles;
    const int cnt = base + (tile < rem ? 1 : 0);
    const int n = n_base + tile * base + (tile < rem ? tile : rem) + sl;      // n_base: first sequence of this chunk
    RowDesc r{0, 0, 0, 0};
    if (sl < cnt) {
        r.valid = 1;
        if (dense_out) {               // fsnp_lstm2_fc: x[n][t][:] -> out[n][o][t]
            r.b = n; r.f = 0; r.out_off = n * 2 * T;
        } else if (mode == FSNP_MODE_FULL) {
            r.b = n / F; r.f = n % F;
            r.out_off = ((r.b * 2) * F + r.f) * T;
        } else {                       // drop_band (feature.py:254-285) with G = num_groups_in_drop_band groups:
            // global sample s keeps bins p + G i (p = s % G, i < (F - F % G) / G); output rows = group 0's samples, group 1's, ...
            const int G = groups, Fh = F / G;
            r.b = n / Fh;
            const int i = n % Fh;
            const int s = batch_offset + r.b, p = s % G;
            int orow = s / G;
            for (int q = 0; q < p; ++q) orow += (global_batch - q + G - 1) / G;     // samples of the groups in front
            r.f = p + G * i;
            r.out_off = ((orow * 2) * Fh + i) * T;
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

// Column-split launches need all their workgroups co-resident.  Two of them running at once (two handles / two streams
// of one process) could each hold part of the chip and wait for peers that cannot be scheduled, so within a process
// they are chained per device: each one waits for the previous one's completion event.  Other kernels always finish,
// so they cannot close a cycle; a foreign PROCESS still can - that case ends in the kernels' wall-clock timeout.
static std::mutex g_coop_mu;
static hipEvent_t g_coop_ev[64] = {};
static bool g_coop_used[64] = {};
template <typename F>
static void launch_coop_chained(int dev, hipStream_t s, F launch) {
    if (dev < 0 || dev >= 64) { launch(); return; }
    std::lock_guard<std::mutex> lk(g_coop_mu);
    if (g_coop_used[dev]) (void)hipStreamWaitEvent(s, g_coop_ev[dev], 0);
    launch();
    if (!g_coop_ev[dev] && hipEventCreateWithFlags(&g_coop_ev[dev], hipEventDisableTiming) != hipSuccess) { g_coop_ev[dev] = nullptr; return; }
    g_coop_used[dev] = hipEventRecord(g_coop_ev[dev], s) == hipSuccess;
}

// The planner itself (cost table, launch shapes, shortest path over tile counts) is host-only code: planner.h / planner.cpp.
static PlannerCtx pctx(const fsnp_handle* h);
static SbPlan plan_sb(const fsnp_handle* h, int num_rows) { return plan_sb(pctx(h), num_rows); }
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
            else if (h->ih_bf16 == 2 && c.ex == 0) launch_lstm_bf3(h->lw, ca, s);      // (VALU-row tiles exist in fp32 / bf16-ih only)
            else launch_lstm(h->lw, ca, s);
            continue;
        }
        if (c.kind == 4) { launch_lstm16(h->lw, ca, s); continue; }
        if (c.kind == 7) { ca.coop_rows_per_group = c.rpg; launch_lstm_generic(h->lw, ca, false, s); continue; }
        ca.coop_hx = hx + (size_t)c.coop_tile0 * hx_floats_per_tile;
        ca.coop_bar = bar + c.coop_tile0;
        ca.coop_bar2 = bar + plan.coop_tiles + c.coop_tile0;      // second half of the counter array
        ca.coop_skew = h->coop_skew;
        ca.coop_split = c.kind == 1 && c.rpg ? 1 : 0;
        // pipelined loop: a deferred K-split chunk shares the chip with the next forward's full-band GEMMs; a GEMM workgroup that
        // lands on one of its CUs runs at ~0.6x (and each GEMM launch lasts as long as its slowest workgroup: stage 1.25 -> 1.85 ms),
        // so the chunk claims its CUs' whole LDS and the GEMM workgroups go to the other CUs (FSNP_OWN_CU=0: off)
        static const int own_cu = [] { const char* e = getenv("FSNP_OWN_CU"); return e && e[0] == '0' ? 0 : 1; }();
        // (never for a launch planned with two workgroups per CU: they would no longer be co-resident)
        ca.coop_own_cu = (own_cu && h->side_stream && s == h->side_stream && chunk_workgroups(h, c) <= h->num_cus_real) ? 160 * 1024 - 256 : 0;
        ca.coop_err = h->d_err;
        ca.coop_abort = abort_word;
        ca.coop_units = c.units; ca.coop_groups = c.groups; ca.coop_rows_per_group = c.rpg;
        // XCD-local workgroup placement (lstm_common.h), unless FSNP_COOP_XCD=0 or a launch planned with two workgroups per CU
        static const int xcd_local = [] { const char* e = getenv("FSNP_COOP_XCD"); return e && e[0] == '0' ? 0 : 1; }();
        {
            const int S = c.kind == 1 ? (h->H / c.units) * (c.rpg ? 2 : 1) : h->H / 128, T = c.kind == 1 ? c.num_tiles : c.groups, cpx = h->num_cus_real / 8;
            ca.coop_xcd = c.kind != 6 && xcd_local && h->num_cus_real % 8 == 0 && xcd_local_blocks_per_xcd(S, T, cpx) <= cpx ? cpx : 0;
        }
        launch_coop_chained(h->device, s, [&] {
            if (c.kind == 6) launch_lstm_pp(h->lw, ca, s);
            else if (c.kind == 1) launch_lstm_coop(h->lw, ca, s);
            else launch_lstm_coopn(h->lw, ca, s);
        });
    }
    if (after_first && last_chunk == nchunks) (void)hipEventRecord(after_first, s);
}
static void launch_build_rows(const SbPlan& plan, RowDesc* rows, int F, int T, int mode, int batch_offset, int global_batch,
                              int dense_out, int groups, hipStream_t s) {
    for (const SbChunk& c : plan.chunks) {
        const int slots = c.num_tiles * c.rps;
        hipLaunchKernelGGL(build_rows_kernel, dim3(cdiv(slots, 256)), dim3(256), 0, s, rows + c.slot0, c.nrows, c.num_tiles, c.rps,
                           F, T, mode, batch_offset, global_batch, dense_out, c.row0, groups);
    }
}
// full-band LSTM of the original FullSubNet: B sequences, always the cooperative kernel (units in {8, 16, 32})
static int fb_row_tiles(int B) { return cdiv(B, 32); }
static int fb_coop_units(const fsnp_handle* h, int B) {
    const int u = lstm_coop_pick_units(h->CH, fb_row_tiles(B), h->num_cus_real, 8);
    return u > 32 ? 0 : u;
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
    const SbPlan plan = plan_sb(h, B * rows_per_utt(h, mode));
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
    w.gn = take(fsn ? 0 : (size_t)h->NB * 2 * 3 * B * 2 * 8);
    w.sb_acc = take((size_t)B * 2 * 8);
    w.coop_hx = take(lstm_coop_exchange_bytes(h->H, plan.coop_tiles));
    w.coop_bar = take((size_t)plan.coop_tiles * 2 * 4);      // two arrival counters per row tile
    w.coop_abort = take(256);             // [0] sub-band launches, [16] full-band LSTM (FullSubNet)
    w.fb_hx = take(fsn ? lstm_coop_exchange_bytes(h->CH, fb_row_tiles(B)) : 0);
    w.fb_bar = take(fsn ? (size_t)fb_row_tiles(B) * 4 : 0);
    w.sbt_gn = take(h->sb_tcn ? (size_t)8 * 2 * nrows_pad * 2 * 8 : 0);
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
    drop_graphs(h);
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
static int order_after_last_forward(fsnp_handle* h, hipStream_t s) {
    if (h->done_valid && h->done_stream != s) FSNP_HIP_CHECK(hipStreamWaitEvent(s, h->ev_done, 0));
    return 0;
}
static int mark_forward_done(fsnp_handle* h, hipStream_t s) {
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
    const size_t coop_bar_bytes = align_up((size_t)plan.coop_tiles * 2 * 4, 256);
    const size_t coop_bytes = coop ? coop_hx_bytes + coop_bar_bytes + 256 : 0;       // images, counters, abort word
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_workspace(h, coop_off + coop_bytes, s)) return 4;
    if (h->pipeline && h->side_stream) FSNP_HIP_CHECK(hipStreamSynchronize(h->side_stream));   // slot 0 may still be read
    RowDesc* rows = reinterpret_cast<RowDesc*>(h->ws);
    h->have_last = false;   // the workspace no longer holds a forward's stages
    if (coop) FSNP_HIP_CHECK(hipMemsetAsync(h->ws + coop_off, 0, coop_bytes, s));
    launch_build_rows(plan, rows, 1, steps, 0, 0, 1, 1, 2, s);
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
            if (adopt) { h->cost = it->second; drop_graphs(h); }
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
        plan.chunks = {c}; plan.total_slots = c.num_tiles * c.rps; plan.coop_tiles = (c.kind == 1 || c.kind == 2 || c.kind == 6) ? c.num_tiles : 0;
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
    for (int rpg = 1; rpg <= 4 && rc == 0 && h->pp_ok; ++rpg) {
        const int groups = h->num_cus_real / (h->H / 8);
        if (groups <= 0) break;
        const double us = time_shape(SbChunk{6, 0, 0, groups * rpg, 0, 32, 8, groups, rpg, 0, 0});
        if (us < 0) rc = 4; else t.pp[rpg - 1] = us;
    }
    if (rc == 0 && h->lstm16_ok) {
        const double us16 = time_shape(SbChunk{4, 0, 0, h->num_cus_real, 0, 16, 0, 0, 0, 0, 0});
        if (us16 < 0) rc = 4; else t.rowtile16 = us16;
    }
    cleanup();
#undef FSNP_CAL_CHECK
    if (rc) { if (g_last_error.empty()) set_error("calibration of the sub-band planner failed"); return rc; }
    if (*reinterpret_cast<volatile unsigned*>(h->d_err) != 0) {
        // a calibration launch gave up (its workgroups were not all resident): never plan two workgroups per CU
        *reinterpret_cast<volatile unsigned*>(h->d_err) = 0;
        t = default_costs();
        h->coop_occ = 1;
    }
    t.calibrated = 1;
    if (measured) *measured = t;
    if (adopt) { h->cost = t; drop_graphs(h); }
    std::lock_guard<std::mutex> lk(g_cal_mu);
    g_cal_cache[key] = t;
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
    if (cfg->output_size != 2) { set_error("output_size must be 2"); return 2; }
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
    const bool generic_sb = cfg->sequence_model != FSNP_SEQ_TCN && ((cfg->sb_hidden != 384 && cfg->sb_hidden != 256 && cfg->sb_hidden != 512) || nin > 64);
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
    const char* csp = getenv("FSNP_COOP_SPLIT");
    if (csp && csp[0] >= '0' && csp[0] <= '3') h->coop_split = csp[0] - '0';
    h->coop_split_cfg = h->coop_split;
    {
        // The ping-pong K-split kernel (lstm_pp.hip) is OPT-IN (FSNP_COOP_PP=1): measured (profiles/r03_column_split.md) it only
        // beats the round-2 kernels at exactly 10 row tiles (15.7 vs 16.7 us per step); everywhere else its fixed cost per
        // tile-phase (cell phase 1.0 us, operand fetch issue 1.1 us, barriers 0.5 us on top of 4.4 us of MFMAs) loses
        const char* pe = getenv("FSNP_COOP_PP");
        h->coop_pp = pe && pe[0] == '1' ? 1 : 0;
        h->coop_pp_cfg = h->coop_pp;
        h->pp_ok = !generic_sb && cfg->sequence_model == FSNP_SEQ_LSTM && (cfg->sb_hidden == 384 || cfg->sb_hidden == 256);
    }
    const char* dsm = getenv("FSNP_DEFER_SMALL");
    if (dsm && dsm[0] == '0') h->defer_small = 0;
    const char* sk = getenv("FSNP_COOP_SKEW");
    if (sk && sk[0] == '0') h->coop_skew = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->d_err), 256, hipHostMallocMapped) != hipSuccess) {
        set_error("hipHostMalloc of the error word failed");
        delete h;
        return 4;
    }
    memset(h->d_err, 0, 256);
    const char* cg = getenv("FSNP_COMPOSITE_GAIN");      // tuning: 0 = never split a batch into row-tile rounds + remainder
    if (cg) h->composite_gain = atof(cg);
    const char* gp = getenv("FSNP_GRAPH");
    if (gp && gp[0] == '1') h->use_graph = 1;
    if (gp && gp[0] == '2') h->use_graph = 2;
    const char* nw = getenv("FSNP_LSTM_WAVES");
    if (nw && atoi(nw) == 4) h->lstm_waves = 4;
    if (nw && atoi(nw) == 12) h->lstm_waves = 12;
    const char* dbg = getenv("FSNP_DEBUG_STAGES");
    h->debug = dbg && dbg[0] == '1';
    *out = h;
    return 0;
}

void fsnp_destroy(fsnp_handle* h) {
    if (!h) return;
    fsnp::DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    drop_graphs(h);
    if (h->cap_stream) { (void)hipStreamDestroy(h->cap_stream); (void)hipEventDestroy(h->ev_in); (void)hipEventDestroy(h->ev_out); }
    if (h->ws) (void)hipFreeAsync(h->ws, nullptr);        // (allocated from the stream-ordered pool; the device is idle here)
    if (h->io) (void)hipFreeAsync(h->io, nullptr);
    (void)hipDeviceSynchronize();
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->d_stft) (void)hipFree(h->d_stft);
    if (h->d_weights) (void)hipFree(h->d_weights);
    if (h->d_err) (void)hipHostFree(h->d_err);
    for (auto& r : h->timing_recs)
        for (auto& e : r.e) (void)hipEventDestroy(e);
    for (auto& e : h->event_pool) (void)hipEventDestroy(e);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    for (auto& e : h->ev_side) if (e) (void)hipEventDestroy(e);
    delete h;
}

int fsnp_num_weights(const fsnp_handle* h) { return h ? (int)h->specs.size() : 0; }

int fsnp_weight_info(const fsnp_handle* h, int index, const char** name, int64_t* numel) {
    if (!h || index < 0 || index >= (int)h->specs.size()) { set_error("fsnp_weight_info: bad index"); return 1; }
    if (name) *name = h->specs[index].name.c_str();
    if (numel) *numel = h->specs[index].numel;
    return 0;
}

int fsnp_set_weight(fsnp_handle* h, const char* name, const float* host_data, int64_t numel) {
    if (!h || !name || !host_data) { set_error("fsnp_set_weight: null argument"); return 1; }
    for (const auto& s : h->specs) {
        if (s.name == name) {
            if (s.numel != numel) {
                set_error("size mismatch for %s: expected %lld elements, got %lld", name, (long long)s.numel, (long long)numel);
                return 2;
            }
            h->host_w[s.name].assign(host_data, host_data + numel);
            h->committed = false;
            return 0;
        }
    }
    set_error("unexpected key in state_dict: %s", name);
    return 2;
}

int fsnp_commit_weights(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    for (const auto& s : h->specs)
        if (!h->host_w.count(s.name)) { set_error("missing key in state_dict: %s", s.name.c_str()); return 2; }
    const int F = h->F, CH = h->CH, H = h->H, NB = h->NB, Fr = h->Fr;
    std::vector<float> blob;
    auto alloc = [&](size_t n) { size_t o = blob.size(); blob.resize(align_up(o + n, 64), 0.0f); return o; };
    auto W = [&](const std::string& n) -> const std::vector<float>& { return h->host_w.at(n); };
    auto put = [&](const std::string& n) { const auto& v = W(n); size_t o = alloc(v.size()); std::copy(v.begin(), v.end(), blob.begin() + o); return o; };

    // ---- frontend (TSSE) : reference layouts are already what the kernels want
    size_t o_conv_w[3][3] = {}, o_conv_b[3][3] = {}, o_cat_w[3] = {}, o_cat_b[3] = {}, o_fc1w[3] = {}, o_fc1b[3] = {},
           o_fc2w[3] = {}, o_fc2b[3] = {};
    const int att = h->cfg.attention;
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    const int nbr_w = fsn ? 0 : 3;          // the original FullSubNet has neither attention nor TCN branches
    for (int a = 0; a < nbr_w; ++a) {
        const std::string p = kAtt[a];
        if (att == FSNP_ATT_TSSE) {
            for (int c = 0; c < 3; ++c) {
                o_conv_w[a][c] = put(p + "." + kConvNames[c] + ".0.weight");
                o_conv_b[a][c] = put(p + "." + kConvNames[c] + ".0.bias");
            }
            o_cat_w[a] = put(p + ".feature_concate_fc.weight");
            o_cat_b[a] = put(p + ".feature_concate_fc.bias");
        }
        if (att == FSNP_ATT_ECA) {
            o_cat_w[a] = put(p + ".conv.weight");            // the 3 taps of Conv1d(1,1,3) over the channel axis
            continue;
        }
        {   // transposed copies: fc1 [Fr][F] -> [F][Fr], fc2 [F][Fr] -> [Fr][F]
            const auto& w1 = W(p + ".fc1.weight");
            o_fc1w[a] = alloc((size_t)F * Fr);
            for (int o = 0; o < Fr; ++o)
                for (int f = 0; f < F; ++f) blob[o_fc1w[a] + (size_t)f * Fr + o] = w1[(size_t)o * F + f];
            const auto& w2 = W(p + ".fc2.weight");
            o_fc2w[a] = alloc((size_t)Fr * F);
            for (int o = 0; o < F; ++o)
                for (int f = 0; f < Fr; ++f) blob[o_fc2w[a] + (size_t)f * F + o] = w2[(size_t)o * Fr + f];
        }
        o_fc1b[a] = put(p + ".fc1.bias");
        o_fc2b[a] = put(p + ".fc2.bias");
    }
    // ---- TCN stacks (SequenceModel(sequence_model="TCN"), sequence_model.py:47-58,80-81): zero-padded row-major
    // [N pad 384][K pad 16] GEMM operands, [model][block] major.  `cin` channels in / out of every TCNBlock, `fc_out` rows of
    // the final Linear(cin, fc_out).  Used for the three full-band models and for a sub-band TCN.
    struct TcnOff { size_t w1, b1, a1, g1w, g1b, dw, db, a2, g2w, g2b, w2, b2, w2g, c1, c2, wf, bf; int NB, N1P, K1P, N2P, K2P; };
    auto pack_tcn = [&](const std::vector<std::string>& models, int nb, int cin, int fc_out) {
        TcnOff t{};
        const size_t nm = models.size() ? models.size() : 1;
        t.NB = nb; t.N1P = (int)align_up(CH, 384); t.K1P = (int)align_up(cin, 16); t.N2P = (int)align_up(cin, 384); t.K2P = (int)align_up(CH, 16);
        t.w1 = alloc(nm * nb * t.N1P * t.K1P); t.b1 = alloc(nm * nb * t.N1P); t.a1 = alloc(nm * nb + 1);
        t.g1w = alloc(nm * nb * CH); t.g1b = alloc(nm * nb * CH);
        t.dw = alloc(nm * nb * 3 * CH); t.db = alloc(nm * nb * CH); t.a2 = alloc(nm * nb + 1);
        t.g2w = alloc(nm * nb * CH); t.g2b = alloc(nm * nb * CH);
        t.w2 = alloc(nm * nb * t.N2P * t.K2P); t.b2 = alloc(nm * nb * t.N2P);
        t.w2g = alloc(nm * nb * t.N2P * t.K2P); t.c1 = alloc(nm * nb * t.N2P); t.c2 = alloc(nm * nb * t.N2P);
        t.wf = alloc(nm * t.N2P * t.K1P); t.bf = alloc(nm * t.N2P);
        for (size_t b = 0; b < models.size(); ++b) {
            for (int i = 0; i < nb; ++i) {
                const std::string p = models[b] + ".sequence_model." + std::to_string(i);
                const size_t bi = b * nb + i;
                const auto& w1 = W(p + ".conv1x1.weight");           // [CH][cin][1]
                for (int n = 0; n < CH; ++n)
                    for (int k = 0; k < cin; ++k) blob[t.w1 + (bi * t.N1P + n) * t.K1P + k] = w1[(size_t)n * cin + k];
                std::copy(W(p + ".conv1x1.bias").begin(), W(p + ".conv1x1.bias").end(), blob.begin() + t.b1 + bi * t.N1P);
                blob[t.a1 + bi] = W(p + ".prelu1.weight")[0];
                std::copy(W(p + ".norm1.weight").begin(), W(p + ".norm1.weight").end(), blob.begin() + t.g1w + bi * CH);
                std::copy(W(p + ".norm1.bias").begin(), W(p + ".norm1.bias").end(), blob.begin() + t.g1b + bi * CH);
                const auto& dw = W(p + ".depthwise_conv.weight");     // [CH][1][3] -> tap major
                for (int c = 0; c < CH; ++c)
                    for (int jj = 0; jj < 3; ++jj) blob[t.dw + (bi * 3 + jj) * CH + c] = dw[(size_t)c * 3 + jj];
                std::copy(W(p + ".depthwise_conv.bias").begin(), W(p + ".depthwise_conv.bias").end(), blob.begin() + t.db + bi * CH);
                blob[t.a2 + bi] = W(p + ".prelu2.weight")[0];
                std::copy(W(p + ".norm2.weight").begin(), W(p + ".norm2.weight").end(), blob.begin() + t.g2w + bi * CH);
                std::copy(W(p + ".norm2.bias").begin(), W(p + ".norm2.bias").end(), blob.begin() + t.g2b + bi * CH);
                const auto& w2 = W(p + ".sconv.weight");              // [cin][CH][1]
                for (int n = 0; n < cin; ++n)
                    for (int k = 0; k < CH; ++k) blob[t.w2 + (bi * t.N2P + n) * t.K2P + k] = w2[(size_t)n * CH + k];
                std::copy(W(p + ".sconv.bias").begin(), W(p + ".sconv.bias").end(), blob.begin() + t.b2 + bi * t.N2P);
                // GroupNorm 2 folded into the sconv GEMM (tcn.hip tcn_gemm_dma_kernel): weights times gamma, and the two
                // per-output constants of  sum_k ((a - m) r g_k + b_k) W[n][k] = r sum_k a g_k W[n][k] + c1[n] - r m c2[n]
                const auto& g2 = W(p + ".norm2.weight");
                const auto& be2 = W(p + ".norm2.bias");
                const auto& sb2 = W(p + ".sconv.bias");
                for (int n = 0; n < cin; ++n) {
                    double s1 = sb2[n], s2 = 0.0;
                    for (int k = 0; k < CH; ++k) {
                        const double wv = w2[(size_t)n * CH + k];
                        blob[t.w2g + (bi * t.N2P + n) * t.K2P + k] = (float)(wv * (double)g2[k]);
                        s1 += (double)be2[k] * wv;
                        s2 += (double)g2[k] * wv;
                    }
                    blob[t.c1 + bi * t.N2P + n] = (float)s1;
                    blob[t.c2 + bi * t.N2P + n] = (float)s2;
                }
            }
            const auto& wf = W(models[b] + ".fc_output_layer.weight");   // [fc_out][cin]: top rows of a zero-padded [N2P][K1P]
            for (int n = 0; n < fc_out; ++n)
                for (int k = 0; k < cin; ++k) blob[t.wf + (b * t.N2P + n) * t.K1P + k] = wf[(size_t)n * cin + k];
            const auto& bf = W(models[b] + ".fc_output_layer.bias");
            std::copy(bf.begin(), bf.end(), blob.begin() + t.bf + b * t.N2P);
        }
        return t;
    };
    auto bind_tcn = [&](TcnWeights& t, const TcnOff& o, const float* d) {
        t.w1 = d + o.w1; t.b1 = d + o.b1; t.a1 = d + o.a1; t.g1w = d + o.g1w; t.g1b = d + o.g1b;
        t.dw = d + o.dw; t.db = d + o.db; t.a2 = d + o.a2; t.g2w = d + o.g2w; t.g2b = d + o.g2b;
        t.w2 = d + o.w2; t.b2 = d + o.b2; t.wf = d + o.wf; t.bf = d + o.bf;
        t.w2g = d + o.w2g; t.c1 = d + o.c1; t.c2 = d + o.c2;
        t.num_cus = h->num_cus; t.NB = o.NB; t.N1P = o.N1P; t.K1P = o.K1P; t.N2P = o.N2P; t.K2P = o.K2P;
        for (int i = 0; i < o.NB; ++i) t.dilation[i] = kDilations[i];
        const char* de = getenv("FSNP_GEMM_DMA");          // 0 = the general GEMM kernel everywhere (tuning / A-B)
        t.gemm_dma = de && de[0] == '0' ? 0 : 1;
    };
    std::vector<std::string> fb_models;
    for (int b = 0; b < nbr_w; ++b) fb_models.push_back(kFb[b]);
    const TcnOff fb_off = pack_tcn(fb_models, NB, F, F);
    const TcnOff sb_off = h->sb_tcn ? pack_tcn({"sb_model"}, 8, h->NIN, h->cfg.output_size) : TcnOff{};
    // ---- recurrent models: MFMA B-fragment order + summed biases.  Every kernel sees FOUR column slots per hidden unit:
    // LSTM i, f, g, o (the reference's gate order); GRU r, z, n_x, n_h with W_in only in the input rows of K and W_hn only
    // in the hidden rows (zero blocks elsewhere), biases b_ir + b_hr, b_iz + b_hz, b_in, b_hn.
    struct Rnn4 { std::vector<float> wih0, whh0, wih1, whh1, bias; };
    auto expand = [&](const std::string& pre, int Hh, int nin) {
        Rnn4 r;
        const auto &a0 = W(pre + "weight_ih_l0"), &a1 = W(pre + "weight_hh_l0"), &a2 = W(pre + "weight_ih_l1"), &a3 = W(pre + "weight_hh_l1");
        r.bias.assign((size_t)2 * 4 * Hh, 0.0f);
        if (!h->gru) {
            r.wih0 = a0; r.whh0 = a1; r.wih1 = a2; r.whh1 = a3;
            for (int l = 0; l < 2; ++l) {
                const auto& bi = W(pre + "bias_ih_l" + std::to_string(l));
                const auto& bh = W(pre + "bias_hh_l" + std::to_string(l));
                for (int i = 0; i < 4 * Hh; ++i) r.bias[(size_t)l * 4 * Hh + i] = bi[i] + bh[i];
            }
            return r;
        }
        auto spread = [&](const std::vector<float>& src, int cols, bool hidden) {   // [3H][cols] -> [4H][cols]
            std::vector<float> dst((size_t)4 * Hh * cols, 0.0f);
            std::copy(src.begin(), src.begin() + (size_t)2 * Hh * cols, dst.begin());                        // r, z
            std::copy(src.begin() + (size_t)2 * Hh * cols, src.end(), dst.begin() + (size_t)(hidden ? 3 : 2) * Hh * cols);   // n
            return dst;
        };
        r.wih0 = spread(a0, nin, false); r.whh0 = spread(a1, Hh, true);
        r.wih1 = spread(a2, Hh, false); r.whh1 = spread(a3, Hh, true);
        for (int l = 0; l < 2; ++l) {
            const auto& bi = W(pre + "bias_ih_l" + std::to_string(l));
            const auto& bh = W(pre + "bias_hh_l" + std::to_string(l));
            float* b = r.bias.data() + (size_t)l * 4 * Hh;
            for (int i = 0; i < 2 * Hh; ++i) b[i] = bi[i] + bh[i];
            for (int i = 0; i < Hh; ++i) { b[2 * Hh + i] = bi[2 * Hh + i]; b[3 * Hh + i] = bh[2 * Hh + i]; }
        }
        return r;
    };
    const Rnn4 sbw = h->sb_tcn ? Rnn4{} : expand("sb_model.sequence_model.", H, h->NIN);
    const bool tuned = !h->sb_tcn && !h->generic_sb;            // MFMA kernels exist for this cell / hidden size / input width
    size_t o_wgen = 0;
    if (h->generic_sb) {                                        // runtime-sized kernel: transposed [layer][k][4H]
        o_wgen = alloc(lstm_generic_pack_floats(H, h->NIN));
        lstm_generic_pack_weights(H, h->NIN, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wgen);
    }
    size_t o_wpack = 0, o_wpack12 = 0, o_wpack_bf[2] = {0, 0};
    if (!h->gru && tuned && (H == 384 || H == 256)) {      // the row-tile kernel (and its bf16 variant) exists for LSTM only
        o_wpack = alloc(lstm_pack_floats(H, h->KX, 4));
        lstm_pack_weights(H, h->NIN, h->KX, 4, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack);
        if (H == 384) {
            o_wpack12 = alloc(lstm_pack_floats(H, h->KX, 12));
            lstm_pack_weights(H, h->NIN, h->KX, 12, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack12);
        }
        for (int i = 0; i < 2 && h->KX == 40 && H == 384; ++i) {      // the bf16-ih variant is built for the default input width only
            const int nw = i == 0 ? 4 : 12;
            o_wpack_bf[i] = alloc(lstm_pack_floats_bf16ih(H, h->KX, nw));
            lstm_pack_weights_bf16ih(H, h->NIN, h->KX, nw, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(),
                                     blob.data() + o_wpack_bf[i]);
        }
    }
    size_t o_wpack16 = 0;
    if (h->lstm16_ok) {
        o_wpack16 = alloc(lstm16_pack_floats(H, h->KX));
        lstm16_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack16);
    }
    size_t o_wpack_bf3 = 0;
    if (!h->gru && tuned && h->KX == 40 && H == 384) {      // optional split-bf16 variant of the one-tile-per-CU kernel
        o_wpack_bf3 = alloc(lstm_bf3_pack_floats(H, h->KX, 12));
        lstm_bf3_pack_weights(H, h->NIN, h->KX, 12, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_bf3);
    }
    size_t o_wpack_gru = 0;
    if (h->gru && tuned && H == 384) {
        o_wpack_gru = alloc(gru_pack_floats(H, h->KX, 4));
        gru_pack_weights(H, h->NIN, h->KX, 4, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_gru);
    }
    size_t o_wpack_coop[4] = {0, 0, 0, 0};
    for (int ui = 0; ui < 4 && tuned; ++ui) {
        const int units = 8 << ui;
        o_wpack_coop[ui] = alloc(lstm_coop_pack_floats(H, h->KX, units));
        lstm_coop_pack_weights(H, h->NIN, h->KX, units, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(),
                               blob.data() + o_wpack_coop[ui]);
    }
    const size_t o_wpack_coopn = alloc(tuned ? lstm_coopn_pack_floats(H, h->KX) : 0);
    if (tuned)
        lstm_coopn_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(),
                                blob.data() + o_wpack_coopn);
    // ---- original FullSubNet: full-band recurrent model (cooperative kernel, KX = 264) + Linear(CH, F) as a GEMM operand
    constexpr int KXF = 264;
    size_t o_fbpack[3] = {0, 0, 0}, o_fbbias = 0, o_fsn_wf = 0, o_fsn_bf = 0, o_fbgen = 0;
    const int fsn_kp = (int)align_up(CH, 16), fsn_np = (int)align_up(F, 384);
    if (fsn) {
        const Rnn4 fbw = expand("fb_model.sequence_model.", CH, F);
        if (h->generic_fb) {
            o_fbgen = alloc(lstm_generic_pack_floats(CH, F));
            lstm_generic_pack_weights(CH, F, fbw.wih0.data(), fbw.whh0.data(), fbw.wih1.data(), fbw.whh1.data(), blob.data() + o_fbgen);
        }
        for (int ui = 0; ui < 3 && !h->generic_fb; ++ui) {
            const int units = 8 << ui;
            o_fbpack[ui] = alloc(lstm_coop_pack_floats(CH, KXF, units));
            lstm_coop_pack_weights(CH, F, KXF, units, fbw.wih0.data(), fbw.whh0.data(), fbw.wih1.data(), fbw.whh1.data(),
                                   blob.data() + o_fbpack[ui]);
        }
        o_fbbias = alloc(fbw.bias.size());
        std::copy(fbw.bias.begin(), fbw.bias.end(), blob.begin() + o_fbbias);
        o_fsn_wf = alloc((size_t)fsn_np * fsn_kp);
        const auto& wf = W("fb_model.fc_output_layer.weight");          // [F][CH]
        for (int n = 0; n < F; ++n)
            for (int k = 0; k < CH; ++k) blob[o_fsn_wf + (size_t)n * fsn_kp + k] = wf[(size_t)n * CH + k];
        o_fsn_bf = alloc(fsn_np);
        const auto& bf = W("fb_model.fc_output_layer.bias");
        std::copy(bf.begin(), bf.end(), blob.begin() + o_fsn_bf);
    }
    const size_t o_lbias = alloc(sbw.bias.size());
    std::copy(sbw.bias.begin(), sbw.bias.end(), blob.begin() + o_lbias);
    const size_t o_wfc = h->sb_tcn ? 0 : put("sb_model.fc_output_layer.weight");
    const size_t o_bfc = h->sb_tcn ? 0 : put("sb_model.fc_output_layer.bias");
    // ---- unfold multiplicities w_r, by brute force over (f, j)
    const size_t o_refl = alloc(F), o_reflfb = alloc(F);
    for (int f = 0; f < F; ++f) {
        for (int j = 0; j < h->NSB; ++j) blob[o_refl + reflect_index(f - h->cfg.sb_num_neighbors + j, F)] += 1.0f;
        for (int j = 0; j < 2 * h->cfg.fb_num_neighbors + 1; ++j) blob[o_reflfb + reflect_index(f - h->cfg.fb_num_neighbors + j, F)] += 1.0f;
    }

    FSNP_ON_DEVICE(h);
    drop_graphs(h);
    if (h->d_weights) { FSNP_HIP_CHECK(hipDeviceSynchronize()); FSNP_HIP_CHECK(hipFree(h->d_weights)); h->d_weights = nullptr; }
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_weights), blob.size() * sizeof(float)));
    FSNP_HIP_CHECK(hipMemcpy(h->d_weights, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* d = h->d_weights;
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) { h->fw.conv_w[a][c] = d + o_conv_w[a][c]; h->fw.conv_b[a][c] = d + o_conv_b[a][c]; }
        h->fw.cat_w[a] = d + o_cat_w[a]; h->fw.cat_b[a] = d + o_cat_b[a];
        h->fw.fc1_wT[a] = d + o_fc1w[a]; h->fw.fc1_b[a] = d + o_fc1b[a];
        h->fw.fc2_wT[a] = d + o_fc2w[a]; h->fw.fc2_b[a] = d + o_fc2b[a];
    }
    for (int c = 0; c < 3; ++c) h->fw.ksize[c] = h->cfg.kersize[c];
    h->fw.attention = h->cfg.attention;
    h->fw.subband_num = h->cfg.subband_num > 0 ? h->cfg.subband_num : 1;
    bind_tcn(h->tw, fb_off, d);
    h->lw.wpack = d + o_wpack; h->lw.wpack12 = d + o_wpack12; for (int ui = 0; ui < 4; ++ui) h->lw.wpack_coop[ui] = d + o_wpack_coop[ui];
    h->lw.wpack_coopn = d + o_wpack_coopn;
    h->lw.wpack_gru = d + o_wpack_gru;
    h->lw.wpack_bf3 = d + o_wpack_bf3;
    h->lw.wpack16 = d + o_wpack16;
    h->lw.wpack_bf[0] = d + o_wpack_bf[0]; h->lw.wpack_bf[1] = d + o_wpack_bf[1]; h->lw.ih_bf16 = h->ih_bf16; h->lw.waves = h->lstm_waves; h->lw.bias = d + o_lbias; h->lw.wfc = d + o_wfc; h->lw.bfc = d + o_bfc;
    h->lw.H = H; h->lw.NIN = h->NIN; h->lw.KX = h->KX; h->lw.OUT = h->cfg.output_size; h->lw.gru = h->gru;
    h->lw.wgen = d + o_wgen;
    if (h->sb_tcn) bind_tcn(h->sbt, sb_off, d);
    if (fsn) {
        h->fbw = LstmWeights{};
        for (int ui = 0; ui < 3; ++ui) h->fbw.wpack_coop[ui] = d + o_fbpack[ui];
        h->fbw.bias = d + o_fbbias;
        h->fbw.H = CH; h->fbw.NIN = F; h->fbw.KX = KXF; h->fbw.OUT = 0; h->fbw.gru = h->gru;
        h->fbw.wgen = d + o_fbgen;
        h->fsn_wf = d + o_fsn_wf; h->fsn_bf = d + o_fsn_bf; h->fsn_kp = fsn_kp;
    }
    h->d_refl_w = d + o_refl;
    h->d_refl_wfb = d + o_reflfb;
    if (tuned) {                                // which column-split instantiations fit twice on a CU (registers, LDS)
        for (int ui = 0; ui < 4; ++ui) h->occ_ksplit[ui] = std::max(1, lstm_coop_occupancy(h->lw, 8 << ui));
        for (int rpg = 1; rpg <= 2; ++rpg) h->occ_coopn[rpg - 1] = std::max(1, lstm_coopn_occupancy(h->lw, rpg));
    }
    h->committed = true;
    (void)Fr;
    return 0;
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
    if (*reinterpret_cast<volatile unsigned*>(h->d_err) != 0) {     // set by an earlier launch that has finished since
        *reinterpret_cast<volatile unsigned*>(h->d_err) = 0;
        set_error("an earlier forward on this handle failed: an inter-workgroup wait timed out in a column-split LSTM kernel "
                  "(its workgroups were not co-resident - is the GPU shared?); that result was invalid - set FSNP_LSTM_COOP=0");
        return 5;
    }
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
    const SbPlan plan = plan_sb(h, num_rows);
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
    // workspace-only prologue shared by both models: zero the accumulators, describe the sub-band rows
    auto prologue = [&](hipStream_t st) {
        launch_zero_region(base + w.zero_begin, w.zero_end - w.zero_begin, st);
        launch_build_rows(plan, rows, h->F, frames, mode, batch_offset, global_batch, 0, h->cfg.num_groups_in_drop_band, st);
    };

    if (!fsn) {
        FrontendBuffers fbuf;
        fbuf.raw = fptr(w.raw); fbuf.frame = reinterpret_cast<double*>(base + w.frame);
        fbuf.md = reinterpret_cast<NormMD*>(base + w.md); fbuf.fsum = reinterpret_cast<double*>(base + w.fsum);
        fbuf.gate = fptr(w.gate); fbuf.att = fptr(w.att);
        const float* in[3] = {mag, real, imag};
        TcnBuffers tbuf;
        tbuf.att = fptr(w.att); tbuf.x = fptr(w.x); tbuf.y1 = fptr(w.y1); tbuf.y2 = fptr(w.y2);
        tbuf.gn = reinterpret_cast<double*>(base + w.gn); tbuf.fb = fptr(w.fb);
        tbuf.dbg_tcn0 = h->debug ? fptr(w.dbg_tcn0) : nullptr;
        // the caller's tensors are read by the repack kernel only; everything up to the LSTM then stays in the workspace
        launch_frontend(d, h->cfg.norm_type, in, strides, is_complex, h->fw, fbuf, s, FE_PHASE_REPACK);
        auto middle = [&](hipStream_t st) {
            prologue(st);
            launch_frontend(d, h->cfg.norm_type, in, strides, is_complex, h->fw, fbuf, st, FE_PHASE_REST);
            launch_tcn(d, h->cfg.fb_act, h->tw, tbuf, st);
            launch_subband_stats(d, h->cfg.norm_type, sbuf, rows, num_slots, st);
        };
        if (h->use_graph) {
            const GraphKey key{batch, frames, mode, batch_offset, global_batch, h->num_cus, h->lstm_coop, h->ih_bf16,
                               h->debug ? 1 : 0, base, h->d_weights};
            if (run_graphed(h, key, s, middle)) return 4;
        } else {
            middle(s);
        }
    } else {
        prologue(s);
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
                           1, frames, 0, 0, 1, 1, 0, 2);
        LstmArgs fa{};
        fa.rows = fb_rows; fa.dense = fptr(w.att); fa.dense_stride = d.FP; fa.md_seq = fbuf.md;
        fa.seq_out = fptr(w.y1);
        fa.num_rows = batch; fa.num_tiles = fb_tiles; fa.Tp = d.Tp; fa.LA = 0; fa.FP = d.FP; fa.F = d.F;
        fa.coop_hx = fptr(w.fb_hx); fa.coop_bar = reinterpret_cast<unsigned*>(base + w.fb_bar); fa.coop_err = h->d_err;
        fa.coop_abort = reinterpret_cast<unsigned*>(base + w.coop_abort) + 16;
        fa.coop_units = fb_units;
        const int chp = (int)align_up(d.CH, 4);           // row stride of the h1 sequence (a float4 multiple; pad columns written as zeros)
        fa.seq_stride = chp;
        if (h->generic_fb) { fa.coop_rows_per_group = fb_rg; launch_lstm_generic(h->fbw, fa, true, s); }
        else launch_coop_chained(h->device, s, [&] { launch_lstm_coop_seq(h->fbw, fa, s); });
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
        launch_sb_scatter(fptr(w.sbt_fb), h->XS, rows, out, (long)rows_per_utt(h, mode) * frames, num_slots, d.Tp, d.LA, s);
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
    auto fills_chip = [](const SbChunk& c) { return c.kind == 0 || c.kind == 4; };       // one (half) tile per CU, no exchange
    if (h->pipeline && plan.chunks.size() > 1 && fills_chip(plan.chunks[0])) {
        ndefer = 1;
        while (ndefer < (int)plan.chunks.size() && fills_chip(plan.chunks[ndefer])) ++ndefer;
        if (ndefer == (int)plan.chunks.size()) ndefer = 0;
    } else if (h->pipeline && h->defer_small && !fills_chip(plan.chunks[0])) {
        // ... if its launches leave room: they own their CUs, and the overlapped stages crawl on what is left (B = 5: 246 of 256
        // CUs taken, full-band stage 0.5 -> 5.4 ms: no gain)
        int busiest = 0;
        for (const SbChunk& c : plan.chunks) busiest = std::max(busiest, chunk_workgroups(h, c));
        defer_all = busiest <= h->num_cus_real - 32;
    }
    if (defer_all) {
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

// ---------------------------------------------------------------------------------------------- STFT / iSTFT (f-3)
namespace fsnp {
struct StftPlan {
    int n_fft, hop, F, N2, sp;          // sp = padded float stride of one internal spectrum row (multiple of 4)
    size_t o_fwd, o_inv, o_win, o_zero, total;   // float offsets inside d_stft
    int inv_ld;
};
static StftPlan stft_plan(const fsnp_handle* h) {
    StftPlan p{};
    p.F = h->F; p.n_fft = 2 * (h->F - 1); p.hop = p.n_fft / 2; p.N2 = 2 * p.F; p.sp = (int)align_up(p.N2, 4);
    p.inv_ld = (int)align_up(p.N2, 16);
    p.o_fwd = 0;
    p.o_inv = p.o_fwd + align_up(p.N2, 384) * (size_t)p.n_fft;
    p.o_win = p.o_inv + align_up(p.n_fft, 384) * (size_t)p.inv_ld;
    p.o_zero = p.o_win + align_up(p.n_fft, 64);
    p.total = p.o_zero + align_up(p.N2 > p.n_fft ? p.N2 : p.n_fft, 384);
    return p;
}
static int ensure_stft(fsnp_handle* h) {
    if (h->d_stft) return 0;
    if (h->F < 3 || ((h->F - 1) & (h->F - 2)) != 0) { set_error("STFT: num_freqs - 1 must be a power of two (n_fft = 2 (num_freqs - 1))"); return 2; }
    const StftPlan p = stft_plan(h);
    std::vector<float> host(p.total, 0.0f);
    stft_build_matrices(p.n_fft, host.data() + p.o_fwd, host.data() + p.o_inv, host.data() + p.o_win);
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_stft), p.total * sizeof(float)));
    FSNP_HIP_CHECK(hipMemcpy(h->d_stft, host.data(), p.total * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
static int ensure_io(fsnp_handle* h, size_t bytes, hipStream_t s) {      // stream-ordered, like ensure_workspace
    if (bytes <= h->io_bytes) return 0;
    if (order_after_last_forward(h, s)) return 4;
    unsigned char* nio = nullptr;
    FSNP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&nio), bytes, s));
    if (h->io) FSNP_HIP_CHECK(hipFreeAsync(h->io, s));
    h->io = nio;
    h->io_bytes = bytes;
    return 0;
}
__global__ void spec_repack_kernel(const float2* __restrict__ in, long sb, long sf, long st, float2* __restrict__ out, int B,
                                   int F, int T, int spc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (b, t, f) over the padded row of spc complex elements
    if (i >= (long)B * T * spc) return;
    const int f = (int)(i % spc), t = (int)((i / spc) % T), b = (int)(i / ((long)spc * T));
    out[i] = f < F ? in[b * sb + f * sf + t * st] : make_float2(0.f, 0.f);
}
// wav [B][L] -> spectrum rows [B][T][ldc] (interleaved complex), T = 1 + L / hop
static void stft_into(fsnp_handle* h, const StftPlan& p, const float* wav, long wav_stride, float* xp, long xs, float* spec,
                      int ldc, int B, int L, hipStream_t s) {
    const int T = 1 + L / p.hop;
    launch_stft_pad(wav, wav_stride, xp, xs, B, L, p.n_fft, s);
    launch_linear_act(xp, p.hop, h->d_stft + p.o_fwd, p.n_fft, h->d_stft + p.o_zero, spec, ldc, p.n_fft, p.N2, B, T,
                      FSNP_ACT_NONE, h->num_cus, s, xs, p.n_fft);
}
// spectrum rows [B][T][sp] -> wav [B][L]
static void istft_from(fsnp_handle* h, const StftPlan& p, const float* spec, float* frames, float* wav, long wav_stride, int B,
                       int T, int L, hipStream_t s) {
    launch_linear_act(spec, p.sp, h->d_stft + p.o_inv, p.inv_ld, h->d_stft + p.o_zero, frames, p.n_fft, p.N2, p.n_fft, B, T,
                      FSNP_ACT_NONE, h->num_cus, s, (long)T * p.sp, p.N2);
    launch_istft_ola(frames, h->d_stft + p.o_win, wav, wav_stride, B, T, L, p.n_fft, s);
}
}  // namespace fsnp

int fsnp_reserve(fsnp_handle* h, int32_t max_batch, int32_t max_frames, int32_t mode, int32_t max_samples, void* hip_stream) {
    if (!h || max_batch <= 0 || max_frames <= 0 || max_samples < 0) { set_error("fsnp_reserve: bad argument"); return 1; }
    if (mode != FSNP_MODE_FULL && mode != FSNP_MODE_PARITY) { set_error("unknown mode %d", mode); return 2; }
    if (!h->committed) { set_error("fsnp_reserve: weights not committed (the plan depends on the kernels' occupancy)"); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    if (order_after_last_forward(h, s)) return 4;
    // every batch size up to max_batch plans its own launches (the exchange region of a column-split plan can exceed the one of
    // the chip-filling batch): take the largest workspace over all of them - host arithmetic only
    size_t need = 0;
    for (int b = 1; b <= max_batch; ++b) {
        if (mode == FSNP_MODE_PARITY && b <= h->cfg.num_groups_in_drop_band) continue;
        need = std::max(need, plan_workspace(h, b, max_frames, mode).total);
    }
    if (ensure_workspace(h, need, s)) return 4;
    if (max_samples > 0) {
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

int fsnp_stft(fsnp_handle* h, const float* wav, int64_t wav_stride, float* spec, int32_t batch, int32_t samples, void* hip_stream) {
    if (!h || !wav || !spec) { set_error("fsnp_stft: null argument"); return 1; }
    if (batch <= 0) { set_error("fsnp_stft: empty input"); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if (samples <= p.hop) { set_error("fsnp_stft: need more than n_fft/2 = %d samples (reflect padding)", p.hop); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const long xs = (long)align_up((size_t)samples + p.n_fft, 4);
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, (size_t)batch * xs * 4, s)) return 4;
    stft_into(h, p, wav, wav_stride, reinterpret_cast<float*>(h->io), xs, spec, p.N2, batch, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

int fsnp_istft(fsnp_handle* h, const float* spec, const int64_t strides[3], float* wav, int64_t wav_stride, int32_t batch,
               int32_t frames, int32_t samples, void* hip_stream) {
    if (!h || !spec || !strides || !wav) { set_error("fsnp_istft: null argument"); return 1; }
    if (batch <= 0 || frames <= 0 || samples <= 0) { set_error("fsnp_istft: empty input"); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if ((long)(frames - 1) * p.hop + p.n_fft < (long)samples + p.hop) { set_error("fsnp_istft: %d frames do not cover %d samples", frames, samples); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const size_t spec_b = align_up((size_t)batch * frames * p.sp * 4 + 256, 256), fr_b = (size_t)batch * frames * p.n_fft * 4;
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, spec_b + fr_b, s)) return 4;
    float* sp = reinterpret_cast<float*>(h->io);
    float* fr = reinterpret_cast<float*>(h->io + spec_b);
    const long n = (long)batch * frames * (p.sp / 2);
    hipLaunchKernelGGL(spec_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2*>(spec),
                       (long)strides[0], (long)strides[1], (long)strides[2], reinterpret_cast<float2*>(sp), batch, p.F, frames, p.sp / 2);
    istft_from(h, p, sp, fr, wav, wav_stride, batch, frames, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
    return mark_forward_done(h, s);
}

int fsnp_enhance_wave(fsnp_handle* h, const float* wav, int64_t wav_stride, float* out, int64_t out_stride, int32_t batch,
                      int32_t samples, void* hip_stream) {
    if (!h || !wav || !out) { set_error("fsnp_enhance_wave: null argument"); return 1; }
    if (batch <= 0) { set_error("fsnp_enhance_wave: empty input"); return 2; }
    if (!h->committed) { set_error("fsnp_enhance_wave: weights not committed (call fsnp_commit_weights)"); return 2; }
    if (ensure_stft(h)) return 2;
    const StftPlan p = stft_plan(h);
    if (samples <= p.hop) { set_error("fsnp_enhance_wave: need more than n_fft/2 = %d samples (reflect padding)", p.hop); return 2; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    FSNP_ON_DEVICE(h);
    const int T = 1 + samples / p.hop;
    const long xs = (long)align_up((size_t)samples + p.n_fft, 4);
    const size_t xp_b = align_up((size_t)batch * xs * 4, 256), spec_b = align_up((size_t)batch * T * p.sp * 4 + 256, 256);
    const size_t mask_b = align_up((size_t)batch * 2 * p.F * T * 4, 256), fr_b = (size_t)batch * T * p.n_fft * 4;
    if (order_after_last_forward(h, s)) return 4;
    if (ensure_io(h, xp_b + 2 * spec_b + mask_b + fr_b, s)) return 4;
    float* xp = reinterpret_cast<float*>(h->io);
    float* noisy = reinterpret_cast<float*>(h->io + xp_b);
    float* enh = reinterpret_cast<float*>(h->io + xp_b + spec_b);
    float* mask = reinterpret_cast<float*>(h->io + xp_b + 2 * spec_b);
    float* fr = reinterpret_cast<float*>(h->io + xp_b + 2 * spec_b + mask_b);
    // inferencer.py:142-158: stft -> (mag, real, imag) -> model -> decompress_cIRM, complex multiply -> istft(length)
    stft_into(h, p, wav, wav_stride, xp, xs, noisy, p.sp, batch, samples, s);
    const int64_t cst[3] = {(int64_t)T * (p.sp / 2), 1, p.sp / 2};       // complex-element strides of [B][T][sp/2] as (b, f, t)
    const int rc = fsnp_forward_complex(h, noisy, cst, mask, batch, T, FSNP_MODE_FULL, 0, batch, hip_stream);
    if (rc) return rc;
    // pipelined mode: the forward left chunks of the sub-band plan on the side stream (at B = 1 the whole plan); `mask` is read
    // right here and lives in the single-buffered io area, so `s` waits for them now (fsnp_flush) - nothing of this call is deferred
    if (h->pipeline && fsnp_flush(h, hip_stream)) return 4;
    // the pad column of every row of `enh` is never written by apply_cirm and multiplies zero weights: clear it once
    FSNP_HIP_CHECK(hipMemsetAsync(enh, 0, spec_b, s));
    launch_apply_cirm(mask, noisy, cst, enh, cst, batch, p.F, T, s);
    istft_from(h, p, enh, fr, out, out_stride, batch, T, samples, s);
    FSNP_HIP_CHECK(hipGetLastError());
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
    const SbPlan plan = plan_sb(h, num_seq);
    if (plan.chunks.empty()) { set_error("fsnp_lstm2_fc: no kernel plan for %d sequences on this device", num_seq); return 2; }
    return run_dense_plan(h, plan, x, out, num_seq, steps, s);
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
    if (costs) {
        for (int i = 0; i < 4; ++i) { h.cost.ksplit[i][0] = costs[2 * i]; h.cost.ksplit[i][1] = costs[2 * i + 1]; }
        for (int i = 0; i < 2; ++i) { h.cost.coopn[i][0] = costs[8 + 2 * i]; h.cost.coopn[i][1] = costs[9 + 2 * i]; }
        h.cost.rowtile = costs[12]; h.cost.rowtile_ex = costs[13];
        for (int i = 0; i < 4; ++i) h.cost.ksplit1[i] = costs[14 + i];
        h.cost.rowtile16 = costs[18];
        for (int i = 0; i < 4; ++i) h.cost.pp[i] = costs[20 + i];
    }
    h.pp_ok = gru == 0 && (hidden == 384 || hidden == 256);
    h.coop_pp = costs != nullptr;          // (a caller's table prices the ping-pong launches in or out; the built-in plans do not use them)
    h.lstm16_ok = gru == 0 && hidden == 384;
    h.rowtile_ok = gru == 0 || gru == 2;      // gru = 1: plan as if there were no one-tile-per-CU GRU kernel (round-1 shape)
    h.gru = gru != 0;
    if (gru == 2) h.cost.rowtile *= 0.75;
    const SbPlan plan = plan_sb(h, num_rows);
    int n = 0;
    for (const SbChunk& c : plan.chunks) {
        if (n >= max_chunks) break;
        int32_t* o = out + 8 * n;
        o[0] = c.kind; o[1] = c.row0; o[2] = c.nrows; o[3] = c.num_tiles; o[4] = c.ex; o[5] = c.kind == 1 ? c.units : c.groups;      // (kind 6: groups)
        o[6] = c.rpg; o[7] = c.slot0;
        ++n;
    }
    return n;
}

int fsnp_get_costs(const fsnp_handle* h, double out[24], int32_t* calibrated, int32_t* occ) {
    if (!h || !out) { set_error("fsnp_get_costs: null argument"); return 1; }
    for (int i = 0; i < 4; ++i) { out[2 * i] = h->cost.ksplit[i][0]; out[2 * i + 1] = h->cost.ksplit[i][1]; }
    for (int i = 0; i < 2; ++i) { out[8 + 2 * i] = h->cost.coopn[i][0]; out[9 + 2 * i] = h->cost.coopn[i][1]; }
    out[12] = h->cost.rowtile; out[13] = h->cost.rowtile_ex;
    for (int i = 0; i < 4; ++i) out[14 + i] = h->cost.ksplit1[i];
    out[18] = h->cost.rowtile16; out[19] = 0.0;
    for (int i = 0; i < 4; ++i) out[20 + i] = h->cost.pp[i];
    if (calibrated) *calibrated = h->cost.calibrated;
    if (occ) *occ = h->coop_occ;
    return 0;
}

int fsnp_measure_costs(fsnp_handle* h, double out[24]) {
    if (!h || !out) { set_error("fsnp_measure_costs: null argument"); return 1; }
    if (!h->committed) { set_error("fsnp_measure_costs: weights not committed"); return 2; }
    if (h->sb_tcn) { set_error("fsnp_measure_costs: the sub-band model of this handle is a TCN (no recurrent kernels)"); return 2; }
    FSNP_ON_DEVICE(h);
    CostTable t = h->cost;
    if (calibrate_costs(h, false, &t)) return 4;
    for (int i = 0; i < 4; ++i) { out[2 * i] = t.ksplit[i][0]; out[2 * i + 1] = t.ksplit[i][1]; out[14 + i] = t.ksplit1[i]; }
    for (int i = 0; i < 2; ++i) { out[8 + 2 * i] = t.coopn[i][0]; out[9 + 2 * i] = t.coopn[i][1]; }
    out[12] = t.rowtile; out[13] = t.rowtile_ex; out[18] = t.rowtile16; out[19] = 0.0;
    for (int i = 0; i < 4; ++i) out[20 + i] = t.pp[i];
    return 0;
}

int fsnp_debug_set_costs(fsnp_handle* h, const double* costs, int32_t workgroups_per_cu) {
    if (!h || (workgroups_per_cu != 1 && workgroups_per_cu != 2)) { set_error("fsnp_debug_set_costs: bad argument"); return 1; }
    if (!h->committed) { set_error("fsnp_debug_set_costs: commit the weights first (the kernels' occupancy is checked then)"); return 2; }
    h->cost = initial_costs(h->H, h->gru != 0, h->sb_tcn != 0);
    if (costs) {
        for (int i = 0; i < 4; ++i) { h->cost.ksplit[i][0] = costs[2 * i]; h->cost.ksplit[i][1] = costs[2 * i + 1]; }
        for (int i = 0; i < 2; ++i) { h->cost.coopn[i][0] = costs[8 + 2 * i]; h->cost.coopn[i][1] = costs[9 + 2 * i]; }
        h->cost.rowtile = costs[12]; h->cost.rowtile_ex = costs[13];
        for (int i = 0; i < 4; ++i) h->cost.ksplit1[i] = costs[14 + i];
        h->cost.rowtile16 = costs[18];
        for (int i = 0; i < 4; ++i) h->cost.pp[i] = costs[20 + i];
    }
    h->cost.calibrated = 1;          // pinned: the lazy calibration will not replace it
    h->coop_occ = workgroups_per_cu;
    drop_graphs(h);
    return 0;
}

int fsnp_describe_plan(const fsnp_handle* h, int32_t batch, int32_t mode, int32_t* out, int32_t max_chunks) {
    if (!h || !out || batch <= 0 || max_chunks <= 0) { set_error("fsnp_describe_plan: bad argument"); return -1; }
    const SbPlan plan = plan_sb(h, batch * rows_per_utt(h, mode));
    int n = 0;
    for (const SbChunk& c : plan.chunks) {
        if (n >= max_chunks) break;
        // kind 4 = half-tile kernel, 5 = role-split K split, 7..10 = ping-pong K split with 1..4 row tiles per group, 11 = runtime-sized kernel
        out[4 * n + 0] = h->sb_tcn ? 3 : (c.kind == 1 && c.rpg ? 5 : c.kind == 6 ? 6 + c.rpg : c.kind == 7 ? 11 : c.kind);
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
    for (int i = 0; i < n; ++i) {
        const SbChunk& c = plan.chunks[i];
        for (int k = 0; k < 4; ++k) out[6 * i + k] = base[4 * i + k];
        // arithmetic of THIS chunk: the bf16 variants exist for the one-tile-per-CU LSTM kernel only (lstm.hip / lstm_bf3.hip);
        // sequences that the plan hands to any other kernel run in fp32 whatever fsnp_set_precision says
        int prec = 0;
        if (!h->sb_tcn && !h->gru && c.kind == 0) prec = h->ih_bf16 == 1 ? 1 : (h->ih_bf16 == 2 && c.ex == 0) ? 2 : 0;
        out[6 * i + 4] = prec;
        out[6 * i + 5] = h->sb_tcn ? 0 : chunk_workgroups(h, c);
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
                       lp.rows_per_slot_tile, 1, steps, 0, 0, 1, 1, 0, 2);
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
    if (!h->committed || !h->pp_ok) { set_error("fsnp_debug_pp_profile: no ping-pong K-split kernel for this handle"); return 2; }
    if (tiles_per_group < 1 || tiles_per_group > 4 || num_stamps != (int64_t)steps * tiles_per_group * 8) { set_error("fsnp_debug_pp_profile: need steps * tiles_per_group * 8 stamps"); return 2; }
    const int tiles = cdiv(num_seq, 32), groups = cdiv(tiles, tiles_per_group);
    if (num_seq <= 0 || groups * (h->H / 8) > h->num_cus_real) { set_error("fsnp_debug_pp_profile: the launch must fit the chip"); return 2; }
    FSNP_ON_DEVICE(h);
    SbPlan plan;
    plan.chunks = {SbChunk{6, 0, num_seq, tiles, 0, 32, 8, groups, tiles_per_group, 0, 0}};
    plan.total_slots = tiles * 32; plan.coop_tiles = tiles;
    const size_t rows_b = align_up((size_t)plan.total_slots * sizeof(RowDesc), 256), hx_b = align_up(lstm_coop_exchange_bytes(h->H, tiles), 256);
    const size_t bar_b = align_up((size_t)tiles * 2 * 4, 256), st_b = (size_t)num_stamps * 8;
    if (order_after_last_forward(h, nullptr)) return 4;
    if (ensure_workspace(h, rows_b + hx_b + bar_b + 256 + st_b, nullptr)) return 4;
    FSNP_HIP_CHECK(hipDeviceSynchronize());
    h->have_last = false;
    RowDesc* rows = reinterpret_cast<RowDesc*>(h->ws);
    FSNP_HIP_CHECK(hipMemsetAsync(h->ws + rows_b, 0, hx_b + bar_b + 256 + st_b, nullptr));
    launch_build_rows(plan, rows, 1, steps, 0, 0, 1, 1, 2, nullptr);
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

int fsnp_set_precision(fsnp_handle* h, int32_t ih_bf16) {
    if (!h || ih_bf16 < 0 || ih_bf16 > 2) { set_error("fsnp_set_precision: 0 (fp32), 1 (bf16 ih-GEMM) or 2 (split-bf16 emulation of fp32)"); return 1; }
    if (ih_bf16 && (h->gru || h->sb_tcn)) { set_error("fsnp_set_precision: the bf16 ih-GEMM variant exists for the LSTM sub-band model only"); return 2; }
    if (ih_bf16 && h->H != 384) { set_error("fsnp_set_precision: the bf16 variants exist for sb_model_hidden_size = 384 only"); return 2; }
    if (ih_bf16 && h->KX != 40) { set_error("fsnp_set_precision: the bf16 ih-GEMM variant exists for sub-band inputs of <= 40 features only"); return 2; }
    h->ih_bf16 = ih_bf16;
    h->lw.ih_bf16 = ih_bf16 == 1 ? 1 : 0;       // (launch_lstm's own switch: the bf16-ih variant of lstm.hip)
    return 0;
}

int fsnp_poll_errors(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    const unsigned e = *reinterpret_cast<volatile unsigned*>(h->d_err);
    if (e != 0) {
        *reinterpret_cast<volatile unsigned*>(h->d_err) = 0;
        set_error("column-split LSTM kernel: an inter-workgroup wait timed out (its workgroups were not co-resident - is the "
                  "GPU shared with another process?); the result of that call is invalid.  FSNP_LSTM_COOP=0 avoids these kernels");
        return 5;
    }
    return 0;
}

int fsnp_set_pipeline(fsnp_handle* h, int32_t enable) {
    if (!h || (enable != 0 && enable != 1)) { set_error("fsnp_set_pipeline: 0 or 1"); return 1; }
    if (enable == h->pipeline) return 0;
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipDeviceSynchronize());              // nothing of either mode is in flight while the workspace is re-shaped
    if (enable && !h->side_stream) {
        // the deferred remainder chunk is a latency-bound chain of inter-workgroup hand-offs: at the highest stream priority
        // its waves win the arbitration against the full-band GEMMs it shares CUs with (FSNP_SIDE_PRIO=0: default priority)
        int lo = 0, hi = 0;
        const char* pe = getenv("FSNP_SIDE_PRIO");
        if (!(pe && pe[0] == '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            FSNP_HIP_CHECK(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, hi));
        else
            FSNP_HIP_CHECK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        FSNP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
        for (auto& e : h->ev_side) FSNP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    drop_graphs(h);
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
    const unsigned e = *reinterpret_cast<volatile unsigned*>(h->d_err);
    if (e != 0) {
        *reinterpret_cast<volatile unsigned*>(h->d_err) = 0;
        set_error("cooperative LSTM kernel: an inter-workgroup wait timed out (a workgroup was not resident); the "
                  "results of that forward are invalid - set FSNP_LSTM_COOP=0");
        return 5;
    }
    return 0;
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
    add("weights committed=%d precision=%d (0 fp32, 1 bf16 ih-GEMM, 2 split-bf16) pipeline=%d timing=%d workspace=%zu bytes x %d\n",
        (int)h->committed, h->ih_bf16, h->pipeline, (int)h->timing, h->ws_bytes, h->ws_slots);
    add("effective settings (environment variable as read at fsnp_create = value in force):\n");
    add("  FSNP_LSTM_COOP=%s -> column-split kernels %s\n", env("FSNP_LSTM_COOP"), h->lstm_coop ? "planned (auto)" : "never");
    add("  FSNP_COOP_PP=%s -> ping-pong K split (lstm_pp.hip) %s\n", env("FSNP_COOP_PP"), !h->pp_ok ? "not built for this model" : h->coop_pp ? "planned" : "never");
    add("  FSNP_COOP_SKEW=%s -> K-split schedule %s\n", env("FSNP_COOP_SKEW"), h->coop_skew ? "layer-skewed from 16 units up" : "serial");
    add("  FSNP_COOP_SPLIT=%s -> role-split K split mode %d (0 never, 1 auto outside the pipelined loop, 2 wherever it fits, 3 auto also pipelined)\n", env("FSNP_COOP_SPLIT"), h->coop_split);
    add("  FSNP_COOP_OCC=%s -> column-split workgroups per CU the planner may use: %d\n", env("FSNP_COOP_OCC"), h->coop_occ);
    add("  FSNP_COOP_XCD=%s (0 = no XCD-local workgroup placement)  FSNP_OWN_CU=%s (0 = deferred chunks do not claim their CUs' LDS)\n", env("FSNP_COOP_XCD"), env("FSNP_OWN_CU"));
    add("  FSNP_LSTM16=%s -> half-tile kernel %s\n", env("FSNP_LSTM16"), h->lstm16_ok ? "planned" : "not used");
    add("  FSNP_LSTM_WAVES=%s -> %d (0 = auto)\n", env("FSNP_LSTM_WAVES"), h->lstm_waves);
    add("  FSNP_CALIBRATE=%s -> cost table %s\n", env("FSNP_CALIBRATE"), h->cost.calibrated ? "measured / pinned" : h->calibrate ? "to be measured at the first plan" : "built-in");
    add("  FSNP_COMPOSITE_GAIN=%s -> %.3f\n", env("FSNP_COMPOSITE_GAIN"), h->composite_gain);
    add("  FSNP_DEFER_SMALL=%s -> %d  FSNP_SIDE_PRIO=%s\n", env("FSNP_DEFER_SMALL"), h->defer_small, env("FSNP_SIDE_PRIO"));
    add("  FSNP_GRAPH=%s -> %d (0 plain launches, 1 / 2 hipGraph replay of the full-band stages)\n", env("FSNP_GRAPH"), h->use_graph);
    add("  FSNP_GEMM_DMA=%s -> %d  FSNP_GEMM_BN=%s FSNP_GEMM_PF=%s (tuning of the general GEMM kernel)\n", env("FSNP_GEMM_DMA"), h->tw.gemm_dma, env("FSNP_GEMM_BN"), env("FSNP_GEMM_PF"));
    add("  FSNP_DEBUG_STAGES=%s -> %d\n", env("FSNP_DEBUG_STAGES"), (int)h->debug);
    add("cost table (us per step): K split full %.1f / %.1f / %.1f / %.1f, one tile %.1f / %.1f / %.1f / %.1f, three-way %.1f / %.1f, one tile per CU %.1f (+%.2f per VALU row), half tile %.1f, ping-pong %.1f / %.1f / %.1f / %.1f\n",
        h->cost.ksplit[0][0], h->cost.ksplit[1][0], h->cost.ksplit[2][0], h->cost.ksplit[3][0], h->cost.ksplit1[0], h->cost.ksplit1[1],
        h->cost.ksplit1[2], h->cost.ksplit1[3], h->cost.coopn[0][0], h->cost.coopn[1][0], h->cost.rowtile, h->cost.rowtile_ex, h->cost.rowtile16,
        h->cost.pp[0], h->cost.pp[1], h->cost.pp[2], h->cost.pp[3]);
    if (buf && cap > 0) {
        const size_t n = o.size() < (size_t)cap - 1 ? o.size() : (size_t)cap - 1;
        memcpy(buf, o.data(), n);
        buf[n] = 0;
    }
    return (int64_t)o.size() + 1;
}

int fsnp_debug_inject_error(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    *reinterpret_cast<volatile unsigned*>(h->d_err) = 1u;
    return 0;
}

int fsnp_debug_set_graph(fsnp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 2) { set_error("fsnp_debug_set_graph: mode must be 0 (plain launches), 1 or 2 (hipGraph replay)"); return 1; }
    h->use_graph = mode;
    return 0;
}

int fsnp_debug_set_gemm_dma(fsnp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) { set_error("fsnp_debug_set_gemm_dma: mode must be 0 (general GEMM kernel) or 1 (DMA kernel where it applies)"); return 1; }
    h->tw.gemm_dma = mode;
    drop_graphs(h);          // a captured full-band chain holds the other kernels
    return 0;
}

int fsnp_debug_set_lstm_coop(fsnp_handle* h, int32_t mode) {
    if (!h || mode < 0 || mode > 3) { set_error("fsnp_debug_set_lstm_coop: mode must be 0 (off), 1 (auto), 2 (auto, serial K-split schedule) or 3 (auto + the ping-pong K split)"); return 1; }
    h->lstm_coop = mode != 0;
    h->coop_skew = mode == 1 || mode == 3;
    h->coop_split = mode == 1 || mode == 3 ? h->coop_split_cfg : 0;
    h->coop_pp = mode == 3 ? 1 : mode == 1 ? h->coop_pp_cfg : 0;
    h->cost.calibrated = h->calibrate ? 0 : h->cost.calibrated;    // the K-split costs depend on the schedule: measure again
    if (!h->cost.calibrated) h->cost = initial_costs(h->H, h->gru != 0, h->sb_tcn != 0);
    drop_graphs(h);            // a captured chain holds row descriptors / a zero region laid out for the old plan
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
