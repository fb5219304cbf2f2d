/*
 * fsnp.h - C ABI of libfsnp_hip.so: the MI355X (gfx950) FullSubNet+ inference forward.
 *
 * The reference (RookieJunChen/FullSubNet-plus) is pure Python and has no FFI; its
 * plugin boundary for this path is
 *     initialize_module(config["model"]["path"], args=config["model"]["args"])
 *         speech_enhance/audio_zen/inferencer/base_inferencer.py:99
 *     model.load_state_dict(ckpt["model"])            base_inferencer.py:100-107
 *     pred_crm = self.model(noisy_mag, noisy_real, noisy_imag)
 *         speech_enhance/fullsubnet_plus/inferencer/inferencer.py:150
 * Each entry point below states which of those steps it replaces.  The Python class
 * fullsubnet_plus_amd.model.FullSubNet_Plus binds these with ctypes (see
 * INTEGRATION.md); nothing here takes a torch type.
 *
 * All functions return 0 on success, non-zero on error (message: fsnp_last_error()).  Codes: 1 bad argument / HIP call failed, 2 bad
 * value or state, 3 no usable device, 4 device-side set-up failed, 5 an inter-workgroup wait of an earlier forward timed out, 6 the
 * watched source tensors no longer match the packed weights (fsnp_watch_weights), 7 exchange verification failed (fsnp_set_verify).
 * No function aborts.  A handle is not thread-safe; distinct handles are independent.
 * All device work is enqueued on the caller's HIP stream; no device synchronisation
 * happens inside fsnp_forward.
 *
 * This header is the surface a maintainer of the reference binds (33 entry points).  Planner introspection, per-kernel timing,
 * stage read-back, calibration probes and every test / tuning hook live in fsnp_debug.h (same library, same ABI version).
 */
#ifndef FSNP_H
#define FSNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsnp_handle fsnp_handle;

/* norm_type of FullSubNet_Plus.__init__ (fullsubnet_plus.py:28, base_model.py:318-330) */
enum {
    FSNP_NORM_OFFLINE_LAPLACE = 0,    /* base_model.py:210-225 */
    FSNP_NORM_CUMULATIVE_LAPLACE = 1, /* base_model.py:227-258 */
    FSNP_NORM_OFFLINE_GAUSSIAN = 2,   /* base_model.py:260-275 */
    FSNP_NORM_CUMULATIVE_LAYER = 3    /* base_model.py:277-316 */
};

/* output activation of SequenceModel (sequence_model.py:85-96); 0 == TOML `false` */
enum { FSNP_ACT_NONE = 0, FSNP_ACT_RELU = 1, FSNP_ACT_RELU6 = 2, FSNP_ACT_TANH = 3 };

/* channel_attention_model of FullSubNet_Plus.__init__ (fullsubnet_plus.py:51-70) */
enum {
    FSNP_ATT_TSSE = 0, /* ChannelTimeSenseSELayer  attention_model.py:43-98 (config/inference.toml) */
    FSNP_ATT_SE = 1,   /* ChannelSELayer           attention_model.py:6-40   */
    FSNP_ATT_ECA = 2,  /* ChannelECAlayer          attention_model.py:335-359 */
    FSNP_ATT_CBAM = 3  /* ChannelCBAMLayer         attention_model.py:296-332 */
};

/* Which reference model the handle implements.
 * FSNP_MODEL_FULLSUBNET is the original FullSubNet (speech_enhance/fullsubnet/model/fullsubnet.py:12-118, the
 * commented alternative of config/inference.toml:11,28): ONE magnitude input, full-band model = 2-layer
 * LSTM(num_freqs -> tcn_hidden) + Linear(tcn_hidden, num_freqs) + fb_act, sub-band input = 31 neighbours of the RAW
 * magnitude + 1 full-band feature.  For it tcn_hidden means fb_model_hidden_size (must be 512), num_tcn_blocks /
 * kersize / attention are ignored, and fsnp_forward takes real = imag = NULL. */
enum { FSNP_MODEL_FULLSUBNET_PLUS = 0, FSNP_MODEL_FULLSUBNET = 1 };

/* `sequence_model` kwarg of both reference models (SequenceModel, audio_zen/model/module/sequence_model.py:31-46):
 * the recurrent cell of the sub-band model (and, for FSNP_MODEL_FULLSUBNET, of the full-band model).  GRU runs on the
 * column-split kernels only (csrc/lstm_coop.hip, csrc/lstm_coopn.hip) and has no bf16 variant. */
enum {
    FSNP_SEQ_LSTM = 0,
    FSNP_SEQ_GRU = 1,
    FSNP_SEQ_TCN = 2 /* FullSubNet+ only: the sub-band model is 8 TCNBlocks(34 -> tcn_hidden -> 34) + Linear(34, 2)
                        (sequence_model.py:47-58,106-112); sb_hidden is ignored */
};

/* B > 1 semantics */
enum {
    FSNP_MODE_FULL = 0,  /* every utterance keeps all num_freqs bins: out [B,OC,F,T]         */
    FSNP_MODE_PARITY = 1 /* reproduces drop_band (feature.py:254-285): out [B,OC,F/2,T],     */
                         /* rows re-ordered even samples first - the reference's literal B>1 */
};

/* Mirrors the constructor kwargs of FullSubNet_Plus (fullsubnet_plus.py:17-34). */
typedef struct fsnp_config {
    int32_t num_freqs;          /* 257 */
    int32_t look_ahead;         /* 2   */
    int32_t sb_num_neighbors;   /* 15  */
    int32_t fb_num_neighbors;   /* 0; > 0 as long as the sub-band input (2 sb + 1) + branches (2 fb + 1) has <= 64 features
                                   (recurrent kernels are instantiated for K = 40 and K = 64 input columns) */
    int32_t tcn_hidden;         /* 512: TCNBlock hidden_channel (causal_conv.py:68) */
    int32_t num_tcn_blocks;     /* 8, dilations 1,2,5,9,1,2,5,9 (sequence_model.py:48-57) */
    int32_t sb_hidden;          /* 384: sb_model_hidden_size.  256, 384 and 512 are built for the recurrent sub-band models (the MFMA
                                   tilings of csrc/lstm*.hip are instantiated for them: 384 on every kernel, 256 on the column-split
                                   kernels + the one-tile-per-CU LSTM kernel, 512 on the column-split kernels only; anything else
                                   is rejected by fsnp_create, never mis-computed).  Ignored for FSNP_SEQ_TCN. */
    int32_t output_size;        /* 2 */
    int32_t norm_type;          /* FSNP_NORM_* */
    int32_t fb_act;             /* FSNP_ACT_* : fb_output_activate_function */
    int32_t sb_act;             /* FSNP_ACT_* : sb_output_activate_function */
    int32_t kersize[3];         /* 3,5,10 : TSSE depthwise kernel sizes (attention_model.py:49) */
    int32_t num_groups_in_drop_band; /* 2; PARITY mode needs >= 2 and a global batch larger than it (feature.py:263) */
    int32_t attention;          /* FSNP_ATT_* : channel_attention_model */
    int32_t model;              /* FSNP_MODEL_* (0 = FullSubNet+) */
    int32_t sequence_model;     /* FSNP_SEQ_* : sequence_model kwarg (0 = LSTM) */
    int32_t subband_num;        /* 1.  > 1 (fullsubnet_plus.py:146-153) regroups the channels of the MAGNITUDE branch's attention
                                   layer; the reference itself only survives it with attention = FSNP_ATT_ECA (its TSSE / SE / CBAM
                                   layers are built for num_freqs / subband_num + 1 channels but the real / imag branches feed
                                   them num_freqs, fullsubnet_plus.py:47-50,155-163), so every other combination is rejected.
                                   0 is read as 1. */
} fsnp_config;

/* Replaces `FullSubNet_Plus(**model.args)` (base_inferencer.py:99).  Needs a visible
 * gfx950 device; returns an error (never falls back to the CPU) otherwise. */
int fsnp_create(const fsnp_config* cfg, fsnp_handle** out);
void fsnp_destroy(fsnp_handle* h);

/* Replaces `load_state_dict` (base_inferencer.py:107): hand over one tensor of the
 * reference state_dict by its reference name (e.g.
 * "sb_model.sequence_model.weight_hh_l0"), as contiguous fp32 in HOST memory with the
 * reference's shape.  Unknown names or wrong sizes are errors (strict loading). */
int fsnp_set_weight(fsnp_handle* h, const char* name, const float* host_data, int64_t numel);
/* Packs everything into the device layouts (MFMA fragment order etc.).  Fails if any
 * tensor of the parameter tree is missing.  Call again after changing weights. */
int fsnp_commit_weights(fsnp_handle* h);
/* Number of tensors / i-th tensor name+numel the handle expects (for strict loaders). */
int fsnp_num_weights(const fsnp_handle* h);
int fsnp_weight_info(const fsnp_handle* h, int index, const char** name, int64_t* numel);

/* Device workspace (owned by the handle, grown on demand) needed for a [B,T] forward. */
size_t fsnp_workspace_bytes(const fsnp_handle* h, int32_t batch, int32_t frames, int32_t mode);
/* Grows the workspace ONCE to what any forward of up to max_batch utterances x max_frames frames needs (and, if max_samples
 * > 0, the STFT / iSTFT area of fsnp_enhance_wave for that many samples per utterance), so that a serving loop with varying
 * clip lengths never re-allocates.  The reference has no counterpart (torch's caching allocator plays this role for
 * `self.model(...)`, inferencer.py:150).  Like every growth of the workspace inside fsnp_forward it is STREAM-ORDERED on
 * `hip_stream` (hipMallocAsync / hipFreeAsync): no device-wide synchronisation, other streams keep running.  Needs
 * committed weights. */
int fsnp_reserve(fsnp_handle* h, int32_t max_batch, int32_t max_frames, int32_t mode, int32_t max_samples, void* hip_stream);

/* Replaces `self.model(noisy_mag, noisy_real, noisy_imag)` (inferencer.py:150;
 * FullSubNet_Plus.forward fullsubnet_plus.py:122-209).
 *   mag/real/imag : DEVICE pointers to the [B,1,F,T] fp32 inputs; strides[i] = element
 *                   strides (batch, freq, time) of input i - torch.stft views are
 *                   non-contiguous and are consumed in place (no .contiguous()).
 *   out           : DEVICE pointer, contiguous fp32 [B,OC,F,T] (FULL) or [B,OC,F/2,T] (PARITY), OC = fsnp_config.output_size
 *                   (2 in every configuration file: the cIRM).
 *   batch_offset/global_batch : this call's utterances are samples
 *                   [batch_offset, batch_offset+batch) of a global batch (multi-GPU
 *                   sharding of PARITY mode needs the global sample parity and the global
 *                   row order; for single-GPU use 0 / batch).  In PARITY mode `out` is the
 *                   GLOBAL [global_batch,2,F/2,T] tensor base and only this shard's rows
 *                   are written.
 *   hip_stream    : hipStream_t to enqueue on (NULL = default stream). */
int fsnp_forward(fsnp_handle* h, const float* mag, const float* real, const float* imag,
                 const int64_t strides[3][3], float* out, int32_t batch, int32_t frames,
                 int32_t mode, int32_t batch_offset, int32_t global_batch, void* hip_stream);

/* the same forward fed with the interleaved complex64 STFT buffer itself (what torch.stft returns;
 * inferencer.py:142-147 derives mag / real / imag from it with three torch ops): element (b,f,t) is the float pair at
 * noisy + 2*(b*strides[0] + f*strides[1] + t*strides[2]) (strides in complex elements).  mag = |X| is computed by the
 * repack kernel; for FSNP_MODEL_FULLSUBNET only the magnitude is derived.  Other arguments as fsnp_forward. */
int fsnp_forward_complex(fsnp_handle* h, const float* noisy, const int64_t strides[3], float* out, int32_t batch,
                         int32_t frames, int32_t mode, int32_t batch_offset, int32_t global_batch, void* hip_stream);

/* SURVEY.md 8(f-3), second half: the transforms around the model, so that the reference inferencer's inner loop
 * (`mag_complex_full_band_crm_mask`, speech_enhance/fullsubnet_plus/inferencer/inferencer.py:142-158) is one call.
 * n_fft = win_length = 2 (num_freqs - 1), hop_length = n_fft / 2, periodic hann window (config/inference.toml:1-5).
 *   fsnp_stft   replaces audio_zen/acoustics/feature.py:10-31 (torch.stft, center / reflect, onesided):
 *               wav DEVICE fp32 [B][samples] (row stride wav_stride) -> spec DEVICE complex64 in torch.stft's memory
 *               order [B][T][num_freqs] (i.e. the [B,F,T] tensor with strides (T F, 1, F)), T = 1 + samples / hop.
 *   fsnp_istft  replaces feature.py:34-56 (torch.istft(..., length=samples)): spec element (b,f,t) at
 *               spec + 2 (b strides[0] + f strides[1] + t strides[2]) -> wav [B][samples].
 *   fsnp_enhance_wave = stft -> fsnp_forward_complex (all bins) -> fsnp_apply_cirm -> istft, noisy waveform in,
 *               enhanced waveform out (DEVICE fp32, row strides in floats). */
int fsnp_stft(fsnp_handle* h, const float* wav, int64_t wav_stride, float* spec, int32_t batch, int32_t samples, void* hip_stream);
int fsnp_istft(fsnp_handle* h, const float* spec, const int64_t strides[3], float* wav, int64_t wav_stride, int32_t batch,
               int32_t frames, int32_t samples, void* hip_stream);
int fsnp_enhance_wave(fsnp_handle* h, const float* wav, int64_t wav_stride, float* out, int64_t out_stride, int32_t batch,
                      int32_t samples, void* hip_stream);

/* the step right after the model in the reference inferencer
 * (`decompress_cIRM` speech_enhance/audio_zen/acoustics/mask.py:60-63 + complex multiply
 * speech_enhance/fullsubnet_plus/inferencer/inferencer.py:152-157) as one kernel.
 *   mask  : DEVICE fp32 [B,2,F,T] contiguous (what fsnp_forward wrote, FULL mode, for a handle with output_size = 2: the cIRM
 *           has exactly two planes; fsnp_enhance_wave refuses handles with another output_size)
 *   noisy : DEVICE interleaved complex64, element (b,f,t) at noisy + 2*(b*strides[0]+f*strides[1]+t*strides[2])
 *           (strides in complex elements - torch.stft's [B][T][F] layout is consumed in place)
 *   out   : DEVICE interleaved complex64, same logical shape, strides out_strides (complex elements). */
int fsnp_apply_cirm(const float* mask, const float* noisy, const int64_t strides[3], float* out,
                    const int64_t out_strides[3], int32_t batch, int32_t freqs, int32_t frames, void* hip_stream);

/* Stage-level entry point (unit tests, f-2 wiring): the fused two-layer LSTM + Linear
 * of SequenceModel.forward (sequence_model.py:113-123) on a dense input.
 *   x   : DEVICE fp32 [N, T, input_size]  (time-major rows, i.e. x.permute(0,2,1))
 *   out : DEVICE fp32 [N, output_size, T] */
int fsnp_lstm2_fc(fsnp_handle* h, const float* x, float* out, int32_t num_seq, int32_t steps,
                  void* hip_stream);

/* Stage-level entry points, second kind: the PUBLIC HELPERS every reference model object carries next to forward() - `self.norm`
 * (fullsubnet_plus.py:115, fullsubnet.py:61) = BaseModel.norm_wrapper(norm_type) (audio_zen/model/base_model.py:318-330) and
 * BaseModel.unfold (base_model.py:15-47) - on arbitrary device tensors.  No handle (they are static functions of the reference).
 *   fsnp_norm   : in DEVICE fp32 [B,C,F,T] with ELEMENT strides[4] (batch, channel, freq, time) -> out contiguous [B,C,F,T];
 *                 norm_type = FSNP_NORM_*: offline norms take the statistics over (C,F,T) of an utterance (base_model.py:211-226,
 *                 261-275), cumulative norms a prefix over the frames of every (b, c) row (base_model.py:228-258, 278-316).
 *   fsnp_unfold : in as above -> out contiguous [B, F, C, 2 num_neighbor + 1, T]: the sub-band units along the frequency axis,
 *                 reflect padded (functional.pad(mode="reflect") + functional.unfold); num_neighbor < 1: [B, F, C, 1, T].
 * Stream-ordered (scratch from hipMallocAsync on hip_stream), no synchronisation. */
int fsnp_norm(int32_t norm_type, const float* in, const int64_t strides[4], float* out, int32_t batch, int32_t channels, int32_t freqs,
              int32_t frames, void* hip_stream);
int fsnp_unfold(const float* in, const int64_t strides[4], float* out, int32_t batch, int32_t channels, int32_t freqs, int32_t frames,
                int32_t num_neighbor, void* hip_stream);

/* Stage-level entry points, third kind: the SUBMODULES the reference's forward calls, one branch at a time, on a caller's tensor -
 * what `model.channel_attention(x)` / `model.fb_model(x)` (and their `_real` / `_imag` siblings) are on the reference model object
 * (fullsubnet_plus.py:160-165, 171-173).  FullSubNet+ handles only; branch 0 = magnitude, 1 = real, 2 = imaginary.
 *   in  : DEVICE fp32 [B, F, T] with ELEMENT strides[3] (batch, freq, time);  out : DEVICE fp32 contiguous [B, F, T].
 *   fsnp_channel_attention : the configured attention layer (ChannelTimeSenseSELayer.forward attention_model.py:78-101; ChannelSELayer
 *                            :25-41, ChannelECAlayer :349-360, ChannelCBAMLayer :315-341) of that branch: in * gate.  With subband_num > 1
 *                            the magnitude branch's layer is built for the regrouped tensor the FORWARD makes around it
 *                            (fullsubnet_plus.py:146-153): not available as a stage (code 2).
 *   fsnp_fullband_model    : SequenceModel.forward of the branch's full-band TCN stack (sequence_model.py:106-112: 8 TCNBlocks,
 *                            ReLU, Linear, fb_output_activate_function).
 * Same kernels as the forward's stages (the TCN GEMMs on the general kernel: one branch is not the three-branch DMA launch); scratch is
 * stream-ordered (hipMallocAsync on hip_stream), nothing of the handle's forward workspace is touched, no synchronisation. */
int fsnp_channel_attention(fsnp_handle* h, int32_t branch, const float* in, const int64_t strides[3], float* out, int32_t batch,
                           int32_t frames, void* hip_stream);
int fsnp_fullband_model(fsnp_handle* h, int32_t branch, const float* in, const int64_t strides[3], float* out, int32_t batch,
                        int32_t frames, void* hip_stream);

/* For bug reports: a text dump of the handle's configuration and of EVERY effective FSNP_* setting (the environment variables
 * are read at fsnp_create; the value in force is printed next to each).  Writes at most cap bytes (NUL-terminated) into buf
 * and returns the size the full text needs (call with buf = NULL to ask); < 0 on error. */
int64_t fsnp_dump_config(const fsnp_handle* h, char* buf, int64_t cap);

/* BASELINE.json configs[4]: 0 = fp32 everywhere (default, the headline path); 1 = the layer-1 input-to-hidden GEMM of
 * the sub-band LSTM (W_ih_l1 x h0_t, 32 % of the LSTM FLOPs) runs on v_mfma_f32_32x32x16_bf16 with bf16 operands
 * and fp32 accumulation; the recurrent GEMMs, layer 0, the cell and everything else stay fp32.  Only the one-tile-per-CU
 * kernel has this variant: sequences scheduled on the column-split kernels (small batches, the remainder tile of a
 * composite plan) are computed in fp32.  h0_t enters the product as a bf16 hi + lo pair (round 6: two MFMAs per weight fragment), so
 * the weights' rounding is the only bf16 error left.  Tolerance of this mode vs the fp32 reference: 2.5e-3 rel on the recurrent model,
 * 4e-3 on the whole forward at B = 32 (measured 1.2e-3 / 2.75e-3 over four weight seeds and 2 s / 10 s clips, no growth with the clip
 * length: profiles/r06_bf16_error.md; tests/test_gpu_parity.py::test_bf16_ih_variant, ::test_bf16_ih_forward_b32,
 * ::test_bf16_ih_forward_seeds_and_long_clips).  Sub-band inputs of <= 39 features only (the layer-0 bias rides in a spare input column). */
int fsnp_set_precision(fsnp_handle* h, int32_t ih_bf16);
/* Synchronises the device and reports asynchronous kernel-side failures of earlier calls (today: a timed-out
 * inter-workgroup wait in a column-split LSTM kernel, whose workgroups must all be co-resident).  0 = none.
 * The error word is host-mapped: without calling this, the NEXT fsnp_forward on the handle fails instead (once) as soon
 * as the failed launch has completed - a wrong result is never silent for long. */
int fsnp_check_errors(fsnp_handle* h);
/* The same check WITHOUT a device synchronisation: reads (and clears) the host-mapped error word.  Call it once the
 * stream (or an event recorded after the forward) has been synchronised by other means - i.e. at the point where the
 * result is consumed; FullSubNet_Plus.forward does so in its default error_check="sync" mode.  0 = no failure so far. */
int fsnp_poll_errors(fsnp_handle* h);

/* Pipelined serving mode (off by default).  With 1, the column-split remainder chunks that follow a one-tile-per-CU
 * chunk in the sub-band plan (B = 32: the 32 sequences left over after 8192 fill the chip, 1.1 ms on 48 CUs) are
 * enqueued on a private stream after that chunk, so they overlap the full-band stages of the NEXT fsnp_forward on the
 * handle, which leave most CUs idle; the workspace is double buffered for it.  A plan that STARTS with a column-split
 * launch (small batches: B = 1 is one K-split launch on 216 CUs) is deferred whole - every row of `out` is then pending.
 * Deferred launches claim the whole LDS of their CUs, so no LDS-using workgroup of the overlapped kernels shares a CU with
 * them.  Contract: after fsnp_forward returns,
 * work enqueued on the caller's stream is NOT ordered after the deferred chunks (the rows of `out` they own are
 * still being written) until fsnp_flush(h, stream) has made `stream` wait for all deferred work - call it before
 * anything consumes `out`; a serving loop calls it once per batch it hands on, a benchmark once before its final
 * synchronisation.  Results are bit-identical to the non-pipelined call.  Switching the mode synchronises the device. */
int fsnp_set_pipeline(fsnp_handle* h, int32_t enable);
int fsnp_flush(fsnp_handle* h, void* hip_stream);

/* Round 5 - the two ways a forward could be silently wrong, made detectable.
 *
 * fsnp_watch_weights: the packed device weights are a COPY of the caller's parameters (fsnp_set_weight); a caller that edits its source
 * tensors in place afterwards (PyTorch: `p.data.add_()`, `init.normal_(m.weight.data)` as the reference's own BaseModel.weight_init does,
 * audio_zen/model/base_model.py:339-355, EMA / weight averaging) changes no pointer and no version counter.  Register the n SOURCE
 * tensors (device pointers of fp32 data, numels[i] elements each; they must stay valid until the next fsnp_watch_weights / fsnp_destroy)
 * right after fsnp_commit_weights: every `every`-th forward then fingerprints them (35 MB for the default model) inside its prologue
 * launch - 9 us more on that launch, which zeroes the accumulators and describes the sub-band rows anyway - and flags the handle when they no longer match the pack - fsnp_poll_errors / fsnp_check_errors / the next call return 6
 * with an explanatory message (the forwards since the edit ran on the old weights: re-pack and register again).  n = 0 unregisters.
 *
 * fsnp_set_verify: the column-split recurrent kernels (small batches, remainder tiles) exchange h between workgroups through global
 * memory; their hand-off can only detect a TIME-OUT (code 5).  With every = N > 0, every Nth forward whose plan holds such a launch runs
 * the same sequences AGAIN on a kernel without any exchange (up to 4096 sequences: one round of 16-row half tiles, ~103 us per step;
 * more: the one-tile-per-CU kernel; always fp32, like the kernels it checks) into a scratch mask and compares on the device
 * (|a - b| <= 1e-4 + 1e-3 |b|); a mismatch flags the handle with code 7 and the first differing (utterance, bin, frame).  Cost: one
 * such round per N forwards - at N = 64 and 2 s clips +0.21 ... +0.26 ms per forward.
 * fsnp_verify_count = verification passes run so far.  0 = off (default; FSNP_VERIFY_EVERY=N at fsnp_create time sets it). */
int fsnp_watch_weights(fsnp_handle* h, const void* const* dev_ptrs, const int64_t* numels, int32_t n, int32_t every, void* hip_stream);
int fsnp_set_verify(fsnp_handle* h, int32_t every);
/* The same check cheap enough to leave on (round 6): with every = N > 0, every Nth forward whose plan consists of column-split launches
 * ONLY (small batches - where those kernels do all the work; each launch must leave two CUs free) has ONE row tile of one of its launches
 * - they come up in turn, so every tile of every launch is visited - recomputed on the exchange-free half-tile kernel from a snapshot of
 * its input and compared with a snapshot of what the launch wrote.  Snapshot: two tiny kernels behind the launch; recomputation (as long
 * as a whole round of that kernel: ~13 ms at 2 s clips) and comparison: a stream of their own, private buffers, beside the forwards that
 * follow - nothing on the caller's critical path.  A sample is skipped while the previous one is still in flight.  A mismatch flags the
 * handle with code 7 when it is FOUND, i.e. a few calls after the forward it belongs to.  Plans with a chip-filling launch (their
 * column-split launches are remainders: < 1 % of the work; two more workgroups would cost the chip-filling launch a second round) are not
 * sampled - fsnp_set_verify covers them.  0 = off (the C default; the Python module's default
 * error policy "sync" sets 16). */
int fsnp_set_verify_sample(fsnp_handle* h, int32_t every);
int64_t fsnp_verify_count(const fsnp_handle* h);

const char* fsnp_last_error(void);
const char* fsnp_version(void);
/* Binding sanity: FSNP_ABI_VERSION of the header the library was built from and sizeof(fsnp_config) as it sees it; a
 * binding compares both with its own idea before the first real call (fullsubnet_plus_amd/_lib.py does). */
#define FSNP_ABI_VERSION 10
int32_t fsnp_abi_version(void);
int32_t fsnp_config_size(void);

#ifdef __cplusplus
}
#endif
#endif /* FSNP_H */
