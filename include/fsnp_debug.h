/*
 * fsnp_debug.h - introspection, profiling and test hooks of libfsnp_hip.so.
 *
 * Nothing in here is needed to run the path: fsnp.h is the drop-in surface (create / load weights / forward / errors).  These entry
 * points are what the test-suite, bench.py and the tools under tools/ use to look inside a forward (which kernels a batch is cut
 * into, what each stage took, what a stage buffer holds), to pin or measure the planner's cost table, to calibrate a box and to
 * inject faults.  Same library, same FSNP_ABI_VERSION; the reference has no counterpart for any of them.
 */
#ifndef FSNP_DEBUG_H
#define FSNP_DEBUG_H

#include "fsnp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Copy an internal stage buffer of the LAST forward to host (tests / debugging).
 * names: "att_mag","att_real","att_imag" [B,T',F]; "fb_mag","fb_real","fb_imag" [B,T',F];
 *        "tcn0_mag" [B,T',F] (after the first TCN block); "gate_mag|real|imag" [B,F].
 * (time-major: element (b,t,f)).  Synchronises the device. */
int fsnp_read_stage(fsnp_handle* h, const char* name, float* host_out, int64_t numel);

/* Per-kernel timing with hipEvents recorded on the forward's own stream.
 * enable != 0 turns it on for subsequent forwards (adds event records only).
 * fsnp_get_timing synchronises, then returns the accumulated milliseconds and launch
 * counts since the last reset: index 0 = the sub-band model (all of its kernels), 1 = full-band
 * (frontend+TCN) kernels, 2 = whole forward, 3 = the FIRST chunk of the sub-band plan alone (the dominant kernel:
 * see fsnp_describe_plan). */
int fsnp_set_timing(fsnp_handle* h, int32_t enable);
int fsnp_get_timing(fsnp_handle* h, double ms[4], int64_t count[4], int32_t reset);
/* How a forward of `batch` utterances runs its sub-band sequences: up to max_chunks records of 4 ints
 * {kernel (0 = lstm2_fc row-tile, 1 = lstm2_coop K-split, 2 = lstm2_coopn three-way split, 3 = sub-band TCN, 4 = lstm2_fc16
 *  half-tile: 16-row tiles, csrc/lstm16.hip, 11 = lstm2_generic runtime-sized kernel, csrc/lstm_generic.hip; 12 = lstm2_coop_hp: 16 units per
 *  workgroup, gate-split waves, resident weights, every row tile as two half tiles in turn, csrc/lstm_hp.hip - planned for 6-10 row
 *  tiles, FSNP_COOP_HP=0 = never; 13 = lstm2_coopw: a wave owns 8 / 16 hidden units over the whole K, 12 / 6 workgroups per row
 *  tile, no workgroup barrier in the time loop, csrc/lstm_coopw.hip - planned from 11 row tiles up, FSNP_COOP_W=0 = never; 14 = the launch
 *  shape of 12 on its wave-owned kernel, csrc/lstm_hpw.hip: a wave owns 4 of the workgroup's 16 units over the whole K, no workgroup
 *  barrier - what runs where that kernel is instantiated, FSNP_HP_WAVE=0 = kind 12 instead),
 *  sequences, 32-row tiles, VALU rows per tile}, in launch order.  Returns the number of chunks (< 0 on error). */
int fsnp_describe_plan(const fsnp_handle* h, int32_t batch, int32_t mode, int32_t* out, int32_t max_chunks);
/* The same with 7 ints per record: {kernel, sequences, tiles, VALU rows, precision, workgroups, deferred}; deferred = 1: in the
 * pipelined serving loop (fsnp_set_pipeline) this launch runs on the side stream, beside the next forward's full-band stages - the
 * planner's decision: only launches that leave at least 32 CUs free are deferred (a remainder that fills the chip would only slow
 * the stages it shares it with: B = 40, B = 21); precision = the arithmetic of
 * THAT launch under the handle's fsnp_set_precision mode: 0 = fp32, 1 = fp32 with the layer-1 ih-GEMM in bf16 (BASELINE
 * configs[4]).  The bf16 variant exists for the one-tile-per-CU and the half-tile LSTM kernels only: the
 * sequences a plan hands to any other kernel (small batches, the remainder of a chip-filling batch) run in fp32. */
int fsnp_describe_plan_ex(const fsnp_handle* h, int32_t batch, int32_t mode, int32_t* out, int32_t max_chunks);
/* The planner's per-step cost table (microseconds), which it minimises when it cuts the sub-band sequences into launches:
 * out[0..7] = K-split kernel at 8 / 16 / 32 / 64 hidden units per workgroup x {at most one, two workgroups per CU} when the
 * launch is full, out[14..17] = the same four with ONE row tile (costs in between are interpolated in the tile count),
 * out[8..11] = three-way split with 1 / 2 row tiles per group x {one, two}, out[12] = one round of the one-tile-per-CU
 * kernel, out[13] = its relative surcharge per VALU row, out[18] = one round of the half-tile kernel, out[19..20] = half-tile ping-pong
 * kernel (csrc/lstm_hp.hip): one row tile, a full launch; out[21..22] = wave-owned column split (csrc/lstm_coopw.hip) at 32 / 64
 * units per workgroup, a full launch, out[23..24] = the same with ONE row tile, out[25..26] = (round 6) the 96-unit instantiation of that
 * kernel: a full launch (64 row tiles), one row tile.  The built-in table holds measurements
 * (profiles/r03_planner_costs.json, profiles/r05_planner_costs.json), so plans - and performance - are reproducible from run to run
 * and box to box.  fsnp_measure_costs MEASURES the same numbers on the device (every launch shape on zeros at two step counts, slope;
 * ~0.3 s, synchronises; cached per process) without touching the plan: tests/test_gpu_parity.py asserts that the built-in
 * table has not drifted from the kernels.  FSNP_CALIBRATE=1 makes a handle ADOPT the measured table at its first planning
 * call (*calibrated = 1 from then on) - short calibration launches run at other clocks than a forward, so measured tables
 * move near-ties between plans by up to 10 % either way, which is why adoption is opt-in.  *occ = workgroups per CU the
 * column-split kernels may be planned with (2 = allowed for the launch shapes whose kernel fits a CU twice; never chosen
 * with measured costs: two co-resident workgroups starve each other; FSNP_COOP_OCC=1 forces 1). */
#define FSNP_NUM_COSTS 27
int fsnp_get_costs(const fsnp_handle* h, double out[FSNP_NUM_COSTS], int32_t* calibrated, int32_t* occ);
int fsnp_measure_costs(fsnp_handle* h, double out[FSNP_NUM_COSTS]);
/* The planner alone (host only, no device, no handle): how `num_rows` sub-band sequences would be cut on a chip with
 * `num_cus` CUs.  Records of 8 ints {kernel, first sequence, sequences, tiles, VALU rows per tile, units per workgroup
 * (kernels 1, 9) or groups (kernel 2), row tiles per group, first slot}; kernel 8 = the half-tile ping-pong kernel, 9 = the
 * wave-owned column split.  Used by the CPU tests. */
int fsnp_debug_plan_rows(int32_t num_rows, int32_t num_cus, int32_t hidden, int32_t gru, int32_t coop, double composite_gain,
                         int32_t* out, int32_t max_chunks);

/* Test hook: pin the handle's cost table (fsnp_get_costs' layout; NULL = the built-in round-1 table) and the number of
 * column-split workgroups the planner may put on a CU (2 only ever applies to the launch shapes whose kernel fits a CU
 * twice - registers, LDS - as measured with hipOccupancyMaxActiveBlocksPerMultiprocessor at commit time); the lazy
 * calibration then leaves it alone.  Needs committed weights. */
int fsnp_debug_set_costs(fsnp_handle* h, const double* costs, int32_t workgroups_per_cu);
/* The same with `workgroups_per_cu` (1 or 2) column-split workgroups allowed per CU and, if costs != NULL, a cost table in
 * fsnp_get_costs' layout instead of the built-in one. */
int fsnp_debug_plan_rows2(int32_t num_rows, int32_t num_cus, int32_t hidden, int32_t gru, int32_t coop, double composite_gain,
                          int32_t workgroups_per_cu, const double* costs, int32_t* out, int32_t max_chunks);

/* Static facts for roofline accounting: algorithmic FLOPs of one forward. */
double fsnp_forward_flops(const fsnp_handle* h, int32_t batch, int32_t frames, int32_t mode);
double fsnp_lstm_flops(const fsnp_handle* h, int64_t num_seq, int32_t steps);

/* Profiling hook of the half-tile ping-pong kernel (csrc/lstm_hp.hip): num_seq sequences as ONE launch; workgroup 0 stamps the
 * 100 MHz wall clock.  tiles_per_group must be 0 (1..4 selected the round-3 ping-pong K-split kernel, removed in round 4).
 * host_stamps: steps * 2 * 16 values per (step, half): 0 phase start,
 * 1 operands in LDS, 2 / 3 before / after the deferred arrival inside the pass, 4 MFMA pass done, 5 pre-activations exchanged,
 * 6 cells done, 7 past the barrier, 8 published / next operands issued, [15] = 1 if the next half-phase was fetched early. */
int fsnp_debug_pp_profile(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps, int32_t tiles_per_group,
                          uint64_t* host_stamps, int64_t num_stamps);
/* Profiling hook: fsnp_lstm2_fc on the default stream + s_memtime stamps of workgroup 0 at 8 points of
 * every step (0 step start, 1 layer-0 MFMA done, 2 past barrier, 3 cell-0/x/FC done, 4 past barrier,
 * 5 layer-1 MFMA done, 6 past barrier, 7 cell-1 done).  host_stamps receives steps*8 values. Synchronises. */
int fsnp_debug_lstm_profile(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps,
                            uint64_t* host_stamps, int64_t num_stamps);
/* Test hook: the next forward's column-split launches publish ONE wrong h0 value (row 0, unit 0 of their row tile 0, step `step` - 1;
 * the publisher's own state stays right) - what a stale or corrupted exchange image looks like to its consumers.  0 = off. */
int fsnp_debug_corrupt_exchange(fsnp_handle* h, int32_t step);

/* Tuning hook: 1 (default) = the sub-band sequences are planned over all three kernels - the column-split kernels
 * (csrc/lstm_coop.hip <= 42 row tiles, csrc/lstm_coopn.hip 43..170; all their workgroups must be co-resident) for small
 * batches and for the remainder of larger ones, the one-tile-per-CU kernel for full rounds (fsnp_describe_plan shows the
 * cut); 0 = the one-tile-per-CU kernel only (also FSNP_LSTM_COOP=0 at fsnp_create time; use it when the GPU is shared
 * with other work).  Ignored by GRU models, which have no one-tile-per-CU kernel. */
int fsnp_debug_set_lstm_coop(fsnp_handle* h, int32_t mode);   /* 2 = as 1, but the K-split kernel runs its serial (round-1) step
                                                                  schedule instead of the layer-skewed one (also FSNP_COOP_SKEW=0);
                                                                  4 = as 1 + the planner may use the half-tile ping-pong kernel
                                                                  even where FSNP_COOP_HP=0 was set at fsnp_create time */
/* Tuning hook: 1 (default) = the conv1x1 / sconv GEMMs of the full-band TCN stacks run on tcn_gemm_dma_kernel (operands by
 * LDS DMA, GroupNorm folded into the sconv weights at fsnp_create; csrc/tcn.hip) where its layout requirements hold;
 * 0 = the general tcn_gemm_kernel everywhere (also FSNP_GEMM_DMA=0 at fsnp_create time).  Both meet the same tolerance;
 * they are not bit-identical (GroupNorm is applied after the k-sum instead of before it).  Small batches (at most 6 workgroups per CU
 * on 32-row tiles: B <= 16 at 2 s clips) run the same GEMMs on tcn_gemm_sk_kernel - 32 x 64 tiles whose four waves split K, no
 * workgroup barrier in the k-loop - and the sconv GEMMs of larger problems on the 64-row kernel; mode 2 = as 1 but never the
 * split-K kernel; mode 3 = the 128-row DMA kernel only (the environment switches that used to select these were removed in ABI 9). */
int fsnp_debug_set_gemm_dma(fsnp_handle* h, int32_t mode);
/* Test hook: sets the device error word as a timed-out inter-workgroup wait would (the next fsnp_forward /
 * fsnp_check_errors on the handle must then fail, once). */
int fsnp_debug_inject_error(fsnp_handle* h);

/* Test hook (round 4): drift injection for the column-split recurrent kernels.  seed != 0: every workgroup of such a launch
 * sleeps a pseudo-random, workgroup-uniform time (nothing on 7 of 8 phase boundaries, 3 ... 24 us otherwise, ~200 us once in
 * 1024) so that the workgroups that share a row tile drift apart by whole steps instead of running in the lockstep an idle
 * chip gives them; results must stay bit-identical (csrc/lstm_common.h: chaos_delay).  0 = off (default). */
int fsnp_debug_set_chaos(fsnp_handle* h, int32_t seed);

/* Tuning hook: waves per workgroup of the fused LSTM kernel: 12 (three per SIMD), 4 (one per SIMD) or
 * 0 = automatic (default: 12 when the tile plan carries VALU rows, else 4). */
int fsnp_debug_set_lstm_waves(fsnp_handle* h, int32_t waves);

/* Test hook: pretend the device has `num_cus` compute units when planning the LSTM tiles (a tile =
 * 32 MFMA rows + up to 4 VALU rows; see csrc/lstm.hip plan_lstm_tiles), so that small inputs exercise
 * the multi-round / extra-row tile shapes. */
int fsnp_debug_set_num_cus(fsnp_handle* h, int32_t num_cus);

/* Test hook (host only, no GPU needed): run the LSTM weight packer that fsnp_commit_weights uses.
 * out receives (kx/8 + 3*hidden/8) * (hidden/32) * 4 * 64 * 4 floats in MFMA B-fragment order
 * [wave][k-group][tile][lane][k-pair] for a `waves`-wave workgroup (layout documented in csrc/lstm.hip). */
int fsnp_debug_lstm_pack(int32_t hidden, int32_t input_size, int32_t kx, int32_t waves, const float* wih0, const float* whh0,
                         const float* wih1, const float* whh1, float* out, int64_t out_floats);
/* Same for the column-split cooperative kernel (csrc/lstm_coop.hip): `units` hidden units per workgroup (8, 16,
 * 32 or 64); layout [split][wave][local k-group][tile][lane][4]. */
int fsnp_debug_lstm_coop_pack(int32_t hidden, int32_t input_size, int32_t kx, int32_t units, const float* wih0,
                              const float* whh0, const float* wih1, const float* whh1, float* out, int64_t out_floats);

/* Same for the wave-owned column split (csrc/lstm_coopw.hip): ONE array for every split width, [k-group (layer 0: x | h0, then layer
 * 1: h1 | h0)][8-unit block, columns gate-interleaved: column c = gate c & 3 of unit c >> 2][lane][4]; (kx/8 + 3*hidden/8) * (hidden/8) *
 * 256 floats. */
int fsnp_debug_lstm_coopw_pack(int32_t hidden, int32_t input_size, int32_t kx, const float* wih0, const float* whh0, const float* wih1,
                               const float* whh1, float* out, int64_t out_floats);

/* Same for the wave-owned variant of the half-tile ping-pong kernel (csrc/lstm_hpw.hip): [participant = 4 cs + wave][fragment: x k-groups
 * of 16 | W_hh0 | W_hh1 | W_ih1][lane][4] - the A operand of v_mfma_f32_16x16x4_f32 number j of a k-group g is
 * W[gate * H + unit][k = 16 g + 4 j + (lane >> 4)], M row lane & 15 = 4 jj + gate, unit = 16 cs + wave + 4 jj;
 * (hidden / 16) * 4 * ((kx + 15) / 16 + 3 * hidden / 16) * 256 floats. */
int fsnp_debug_lstm_hpw_pack(int32_t hidden, int32_t input_size, int32_t kx, const float* wih0, const float* whh0, const float* wih1,
                             const float* whh1, float* out, int64_t out_floats);

/* Same for the matrix-vector full-band kernel of the original FullSubNet (csrc/lstm_fbv.hip; hidden 512, <= 288 inputs):
 * [column slice of 8 units][fragment j4 < 57][thread (c = tid & 31: gate c & 3 of unit c >> 2; ks = tid >> 5: k slice)][4] -
 * fragments 0 .. 24 = layer 0 over [x (288, zero padded) | h0], k = 100 ks + 4 j4 + e; fragments 25 .. 56 = layer 1 over [h0 | h1],
 * k = 128 ks + 4 (j4 - 25) + e; hidden / 8 * 57 * 1024 floats. */
int fsnp_debug_lstm_fbv_pack(int32_t hidden, int32_t input_size, const float* wih0, const float* whh0, const float* wih1, const float* whh1,
                             float* out, int64_t out_floats);

/* fsnp_set_verify_sample's counters: out[0] = samples recomputed so far, out[1] = samples skipped because the previous one was still in
 * flight, out[2] = eligible forwards seen (plans of column-split launches only) since the setting last changed. */
int fsnp_debug_verify_sample_stats(const fsnp_handle* h, int64_t out[3]);

/* Round 6 - calibration of the box a measurement ran on (bench.py's `box` object).
 *
 * fsnp_debug_box_probe: runs nothing but v_mfma_f32_32x32x2_f32 - the instruction the dominant kernel is bound by - on every SIMD of
 * the current device for about target_ms milliseconds (one wave per SIMD, twelve independent accumulators, random operands) and
 * reports what the matrix pipes of THIS box sustain: out[0] = fp32 MFMA TFLOP/s over the launch (hipEvents), out[1] / [2] / [3] = the
 * shader clock in MHz that the per-workgroup wall time (s_memrealtime, 100 MHz) implies at 64 cycles per MFMA: mean / slowest /
 * fastest workgroup, out[4] = rate of the s_memtime counter in MHz, out[5] = s_memtime ticks per MFMA, out[6] = launch milliseconds,
 * out[7] = compute units, out[8] = TFLOP/s from out[1] alone (65,536 FLOP per cycle on 256 CUs; no launch ramp / tail).  Needs no
 * handle; enqueues on hip_stream and waits for its own launch. */
#define FSNP_BOX_PROBE_VALUES 9
int fsnp_debug_box_probe(double target_ms, double out[FSNP_BOX_PROBE_VALUES], void* hip_stream);
/* The clocks of the LAST launch of the one-tile-per-CU LSTM kernel (csrc/lstm.hip) on this handle: its workgroup 0 stamps s_memtime
 * (= shader cycles) and s_memrealtime (100 MHz) when it starts and when it ends, and every workgroup folds its own duration into a
 * maximum / minimum (a handful of scalar instructions and three atomics per workgroup, outside the time loop).  out[0] = workgroup 0's
 * s_memtime ticks, out[1] = its s_memrealtime ticks (10 ns each), out[2] = its wall time in ms, out[3] = the shader clock it held in MHz,
 * out[4] / out[5] = wall time of the SLOWEST / fastest workgroup of the launch in ms (the launch lasts as long as its slowest workgroup;
 * the XCDs of a chip do not all hold the same clock), out[6] = the largest shader-cycle count of a workgroup.  The caller synchronises
 * first; returns 2 if no such launch has completed. */
#define FSNP_LAUNCH_CLOCK_VALUES 7
int fsnp_debug_launch_clock(fsnp_handle* h, double out[FSNP_LAUNCH_CLOCK_VALUES]);

#ifdef __cplusplus
}
#endif
#endif /* FSNP_DEBUG_H */
