// fsnp_handle.h - the handle behind the C ABI (include/fsnp.h) and the helpers its translation units share:
// fsnp_abi.hip (create / forward orchestration / workspace / tuning hooks), fsnp_weights.hip (strict weight loading + packing),
// fsnp_stft_abi.hip (STFT / iSTFT / waveform entry points).  Host code only.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "fsnp_common.h"
#include "planner.h"

namespace fsnp {

struct WeightSpec {
    std::string name;
    int64_t numel;
};

struct Workspace {
    // offsets in bytes from the workspace base
    size_t att, fb, raw, x, y1, y2, gate, md, md_utt, md_row, rows, fb_rows, frame, sbt_x0, sbt_x, sbt_fb, sbt_y1, sbt_y2, zero_begin,
        fsum, fe_tot, gn, sb_acc, coop_hx, coop_bar, coop_abort, fb_hx, fb_bar, sbt_gn, zero_end, dbg_tcn0, total;
};

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

using namespace fsnp;      // (internal header of three host translation units)

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
    int coop_chaos = 0;          // fsnp_debug_set_chaos: drift injection seed for the column-split kernels (0 = off)
    int coop_skew = 1;           // K-split kernel: 1 = layer-skewed schedule (lstm2_coop_skew_kernel), 0 = the serial one (FSNP_COOP_SKEW=0)
    bool generic_sb = false;     // the sub-band recurrent model runs on the runtime-sized kernel (lstm_generic.hip): a hidden size or an
                                 // input width no tuned kernel is instantiated for
    bool generic_fb = false;     // FullSubNet: the same for the full-band recurrent model (fb_model_hidden_size != 512 or > 264 bins)
    bool hp_ok = false;          // the half-tile ping-pong kernel (lstm_hp.hip) exists for this handle's sub-band model
    int hp_wave = 1;             // kind-8 launches run on the wave-owned variant (lstm_hpw.hip); FSNP_HP_WAVE=0: lstm_hp.hip
    int coop_hp = 0, coop_hp_cfg = 0;   // ... and the planner may use it (FSNP_COOP_HP=0: never; fsnp_debug_set_lstm_coop(h, 4): even then)
    int fb_valu = 1;             // FullSubNet: the full-band LSTM of <= 4 utterances runs on the VALU kernel (lstm_fbv.hip); fsnp_debug_set_gemm_dma-like
                                 // test switch: fsnp_debug_set_lstm_coop(h, 2) turns it off together with the other round-3+ schedules
    bool coopw_ok = false;       // the wave-owned column split (lstm_coopw.hip) exists for this handle's sub-band model (LSTM, H = 384) ...
    int coop_w = 1;              // ... and the planner may use it (FSNP_COOP_W=0: never)
    unsigned* d_err = nullptr;   // [0] = error bits of finished launches (kErr*): an inter-workgroup wait timed out in a column-split LSTM
                                 // kernel / the watched source tensors no longer match the packed weights / a verification pass
                                 // disagreed.  Host-mapped, so the NEXT call on
                                 // the handle can fail loudly without a device synchronisation
    // fsnp_watch_weights: the caller's SOURCE tensors of the packed weights, fingerprinted on the device in front of every forward
    void* watch_segs = nullptr;              // device: WatchSeg[watch_nseg]
    unsigned long long* watch_acc = nullptr; // device: {-, finished blocks, baseline, ..., [8 + b] partial sum of block b}
    int watch_nseg = 0, watch_every = 1;
    long long watch_calls = 0;
    // fsnp_set_verify: every Nth forward whose plan holds a column-split launch is re-run on the one-tile-per-CU kernel and compared
    int verify_every = 0;
    long long verify_calls = 0, verify_runs = 0;
    float* verify_out = nullptr;             // scratch mask of the verification pass (stream-ordered allocation)
    unsigned long long* d_clk = nullptr;     // device: clock stamps of the last one-tile-per-CU LSTM launch (LstmArgs::clk, fsnp_debug_launch_clock)
    unsigned long long* verify_key_sampled = nullptr;   // the same key of the sampled check (inside vs_buf)
    unsigned long long* verify_key = nullptr; // device: smallest (utterance << 44 | bin << 24 | frame) at which a verification pass disagreed
    size_t verify_bytes = 0;
    // fsnp_set_verify_sample (round 6): every Nth forward whose plan is column-split launches only, ONE row tile of one of them (they come
    // up in turn) is recomputed on the exchange-free half-tile kernel from a snapshot, on a stream of its own, beside the forwards that follow
    int vs_every = 0;
    long long vs_calls = 0, vs_runs = 0, vs_skipped = 0;
    hipStream_t vs_stream = nullptr;
    hipEvent_t ev_vs_snap = nullptr, ev_vs_done = nullptr;
    bool vs_busy = false;                    // ev_vs_done has been recorded and not yet seen complete
    unsigned char* vs_buf = nullptr;         // snapshot + reference buffers (private: nothing a later forward or the caller touches)
    size_t vs_bytes = 0;
    int corrupt_exchange = 0;                // fsnp_debug_corrupt_exchange: flip one word of the next column-split launch's exchange (test hook)
    int lstm_waves = 0;   // 0 = auto: 12 waves when the tile plan uses VALU rows, else 4

    // STFT / iSTFT around the model (stft.hip): DFT GEMM operands, built on first use, and an I/O workspace
    float* d_stft = nullptr;     // [fwd (2F pad 384) x n_fft][inv (n_fft pad 384) x (2F pad 16)][window n_fft][zero bias 768]
    unsigned char* io = nullptr;
    size_t io_bytes = 0;

    double composite_gain = 0.97;   // a row-tile + remainder plan must be estimated this much cheaper to be chosen

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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// bits of the host-mapped error word (fsnp_handle::d_err[0])
constexpr unsigned kErrTimeout = 1u, kErrStaleWeights = 2u, kErrVerify = 4u;
// fsnp_weights.hip: fingerprint of the watched source tensors on stream s (no-op without a watch); the device function is shared with
// the forward's prologue kernel (fsnp_abi.hip), which runs the same blocks beside its zeroing / row-descriptor blocks
int launch_weight_watch(fsnp_handle* h, hipStream_t s, bool baseline);
struct WatchSeg { const unsigned* p; unsigned n; unsigned long long first; };      // (device function: weight_watch.h)
constexpr int kWatchSeg = 8192;              // elements per segment of the watched tensors (8 uint4 loads per thread of a 256-thread block)
constexpr int kWatchBlocks = 512;            // blocks of one fingerprint pass (each walks its share of the segments)
void drop_weight_watch(fsnp_handle* h);
// decodes and clears the error word: 0 = clean, else the return code of the call that notices (5 time-out, 6 stale weights, 7 verify) + message
int take_device_errors(fsnp_handle* h, const char* where);
// fsnp_weights.hip
void build_specs(fsnp_handle* h);
// cross-stream ordering of a handle's shared buffers (fsnp_abi.hip)
int order_after_last_forward(fsnp_handle* h, hipStream_t s);
int mark_forward_done(fsnp_handle* h, hipStream_t s);
// fsnp_stft_abi.hip
struct StftPlan {
    int n_fft, hop, F, N2, sp;          // sp = padded float stride of one internal spectrum row (multiple of 4)
    size_t o_fwd, o_inv, o_win, o_zero, total;   // float offsets inside d_stft
    int inv_ld;
};
StftPlan stft_plan(const fsnp_handle* h);
int ensure_stft(fsnp_handle* h);
int ensure_io(fsnp_handle* h, size_t bytes, hipStream_t s);

}  // namespace fsnp
