/*
 * iamf_hip.h — C ABI of the MI355X-native IAMF post-decode renderer (libiamf_hip.so).
 *
 * Plain C: pointers, sizes and PODs only.  Every entry point names the reference interface
 * (Samsung/iac, paths relative to the reference tree) it stands in for.  The reference renders
 * one decoder handle, one frame at a time on the CPU; this ABI renders a BATCH of independent
 * streams x frames per call on one GPU, with the same per-stream arithmetic and stage order
 * (render -> element gain -> mix -> output gain -> loudness -> limiter -> PCM pack,
 * src/iamf_dec/IAMF_decoder.c:3335-3500).  There is no CPU fallback: every call fails with
 * IAMF_HIP_ERR_DEVICE if HIP is unusable.
 *
 * Pointers named d_* are DEVICE pointers owned by the caller (any allocator: hipMalloc, a
 * torch tensor's data_ptr ...).  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Calls are asynchronous on that stream unless stated otherwise.
 */
#ifndef IAMF_HIP_H
#define IAMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: same numeric values as IAMF_defines.h:181-190, plus one for the device */
enum {
  IAMF_HIP_OK = 0,
  IAMF_HIP_ERR_BAD_ARG = -1,
  IAMF_HIP_ERR_BUFFER_TOO_SMALL = -2,
  IAMF_HIP_ERR_INTERNAL = -3,
  IAMF_HIP_ERR_INVALID_STATE = -5,
  IAMF_HIP_ERR_UNIMPLEMENTED = -6,
  IAMF_HIP_ERR_ALLOC_FAIL = -7,
  IAMF_HIP_ERR_DEVICE = -100
};

/* rendering ids of loudspeaker layouts, same values as src/iamf_dec/ae_rdr.h:40-61 */
enum {
  IAMF_HIP_SS_A = 0x020, IAMF_HIP_SS_B = 0x050, IAMF_HIP_SS_C = 0x250, IAMF_HIP_SS_D = 0x450,
  IAMF_HIP_SS_E = 0x451, IAMF_HIP_SS_F = 0x370, IAMF_HIP_SS_G = 0x490, IAMF_HIP_SS_H = 0x9A3,
  IAMF_HIP_SS_I = 0x070, IAMF_HIP_SS_J = 0x470,
  IAMF_HIP_L_STEREO = 0x200, IAMF_HIP_L_51 = 0x510, IAMF_HIP_L_512 = 0x512,
  IAMF_HIP_L_514 = 0x514, IAMF_HIP_L_71 = 0x710, IAMF_HIP_L_714 = 0x714,
  IAMF_HIP_L_MONO = 0x100, IAMF_HIP_L_712 = 0x712, IAMF_HIP_L_312 = 0x312,
  IAMF_HIP_L_BINAURAL = 0x1020
};

/* element renderer kinds.  DMX = the parametric down-mixer (downmix_renderer.c): matrix.in_id /
 * out_id are IAChannelLayoutType values (IAMF_defines.h:196-209), matrix.mat is unused. */
enum { IAMF_HIP_KIND_H2M = 0, IAMF_HIP_KIND_M2M = 1, IAMF_HIP_KIND_DMX = 2, IAMF_HIP_KIND_FIR = 3 };
/* FIR = binaural HRTF convolution: of a scene-based element (the role of
 * IAMF_element_renderer_render_H2B, h2b_rdr.c:109-130; matrix.m = ambisonics channels 1 / 4 / 9 / 16) or of
 * a channel-based element (the role of IAMF_element_renderer_render_M2B, m2b_rdr.c:103-121, taken by the
 * reference when headphones_rendering_mode == 1, IAMF_decoder.c:2562-2570; matrix.m = channels of the
 * element's loudspeaker layout 2 / 6 / 8 / 10 / 12, playback order, one HRIR pair per loudspeaker — a
 * zero pair mutes a channel, e.g. the LFE).  matrix.n = 2, matrix.mat = HRIRs h[ear][channel][fir_taps]
 * (host), out_channels = 2, limiter on.
 *   y[ear][t] = sum_c sum_k h[ear][c][k] * x[c][t-k]     (MFMA; fir_taps <= 256)
 * PARITY UNPINNED: the reference's binauraliser arithmetic lives in Resonance Audio / BEAR, which
 * are not in the reference tree; this formula is this library's specification. */

/* Projection arithmetic for output layouts wider than stereo.
 *   EXACT: VALU, separate f32 multiply and add in the reference's order -> bit-identical PCM.
 *   MFMA : v_mfma_f32_16x16x4_f32 (render_wide4.hpp; v_mfma_f32_32x32x2_f32 in the fallback render_wide.hpp):
 *          exact f32 products, fused k-ordered accumulation -> PCM within
 *          +-1 LSB of the reference (rounding ties only).
 *   AUTO : MFMA for the HOA projection (h2m_rdr.c:1103-1112, a dense contraction), EXACT for
 *          channel-layout matrices (m2m_rdr.c:1826-1837) and always for mono/stereo/binaural. */
enum { IAMF_HIP_PROJ_AUTO = 0, IAMF_HIP_PROJ_EXACT = 1, IAMF_HIP_PROJ_MFMA = 2 };

/* output sample formats.  16/24/32 = interleaved little-endian integer PCM exactly as
 * iamf_decoder_plane2stride_out writes it (IAMF_decoder.c:121-167).  F32 = the limiter output
 * as interleaved float, unscaled: a stage tap for parity tests, not a reference format. */
enum { IAMF_HIP_FMT_S16 = 16, IAMF_HIP_FMT_S24 = 24, IAMF_HIP_FMT_S32 = 32, IAMF_HIP_FMT_F32 = -32 };

/* A static rendering matrix: what IAMF_element_renderer_get_H2M_matrix (h2m_rdr.c:1070-1081,
 * struct h2m_rdr_t ae_rdr.h:142-151) or _get_M2M_matrix (m2m_rdr.c:1786-1804, struct m2m_rdr_t
 * ae_rdr.h:134-140) return.  `mat` points into the library's table (host memory, m*n floats,
 * H2M: mat[n*m_size+m], M2M: mat[m*n_size+n]) and stays valid for the process lifetime. */
typedef struct {
  int32_t kind;     /* IAMF_HIP_KIND_* */
  int32_t in_id;    /* H2M: ambisonics order 0..3 ; M2M: rendering id of the input layout */
  int32_t out_id;   /* rendering id of the output layout */
  int32_t channels; /* H2M: table `channels` field ; M2M: n */
  int32_t lfe1, lfe2;
  int32_t m, n;
  const float *mat;
} iamf_hip_matrix;

/* replaces IAMF_element_renderer_get_H2M_matrix, src/iamf_dec/h2m_rdr.c:1070 (0 or -1) */
int iamf_hip_get_h2m_matrix(int order, int out_id, iamf_hip_matrix *out);
/* replaces IAMF_element_renderer_get_M2M_matrix, src/iamf_dec/m2m_rdr.c:1786 (0 or -1) */
int iamf_hip_get_m2m_matrix(int in_id, int out_id, iamf_hip_matrix *out);
/* The reference has two sets of layout->layout tables, chosen at build time: m2m_rdr.c:36-830 under
 * -DSAMSUNG_TV (the upstream default, CMakeLists.txt:14-19) and :831-1628 otherwise; 47 of the 140
 * matrices differ.  This library carries both; iamf_hip_get_m2m_matrix is VARIANT_DEFAULT (= the
 * -DSAMSUNG_TV=OFF build).  The HOA tables are the same in both builds. */
enum { IAMF_HIP_VARIANT_DEFAULT = 0, IAMF_HIP_VARIANT_SAMSUNG_TV = 1 };
int iamf_hip_get_m2m_matrix_variant(int variant, int in_id, int out_id, iamf_hip_matrix *out);
/* channels of an output layout by rendering id (IAMF_decoder.c:3998-4008); 0 if unknown */
int iamf_hip_layout_channels(int out_id);

/* ------------------------------------------------------------------------------------------
 * Batched renderer: N independent streams that share one topology (same element layout, same
 * output layout, same frame size, bit depth and limiter settings) and keep separate state.
 * ---------------------------------------------------------------------------------------- */
typedef struct iamf_hip_batch iamf_hip_batch;

typedef struct {
  int32_t n_streams;
  int32_t frame_size;     /* samples per frame of element PCM (e.g. 1024) */
  int32_t sample_rate;    /* Hz; limiter time constants depend on it */
  int32_t out_channels;   /* channels of the output layout = PCM stride (IAMF_decoder.c:3497-3499) */
  int32_t out_format;     /* IAMF_HIP_FMT_* (IAMF_decoder_set_bit_depth, IAMF_decoder.h:161) */
  iamf_hip_matrix matrix; /* element renderer; `mat` may also be a caller-owned host array */
  int32_t limiter_enable; /* IAMF_decoder_peak_limiter_enable, IAMF_decoder.h:170 */
  float limiter_threshold_db; /* IAMF_decoder_peak_limiter_set_threshold, IAMF_decoder.h:179;
                                 attack 1 ms, release 200 ms, look-ahead 240 are the reference's
                                 constants (common/audio_defines.h:38-41) */
  int32_t loudness_enable; /* normalization_loudness != 0 (IAMF_decoder.c:3480) */
  int32_t projection;      /* IAMF_HIP_PROJ_*: how layouts with more than 2 channels are projected */
  int32_t fir_taps;        /* kind FIR: taps per HRIR (1..256) */
  int32_t lfe_hoa;         /* kind H2M into a layout with an LFE slot: 1 = the HOA LFE generator of
                              h2m_rdr.c:1151-1239 fills it (2nd-order Butterworth low-pass at 120 Hz over
                              ambisonics channel 0, scaled by 1/sqrt(n)), as a reference built with its
                              switch -DDISABLE_LFE_HOA=0 does (call site IAMF_decoder.c:2625-2636);
                              0 = the default build's silence (ae_rdr.h:63-65) */
  int32_t pcm_stride_channels; /* 0 = out_channels.  Otherwise the PCM is laid out as
                              iamf_decoder_plane2stride_out does with that `stride` (IAMF_decoder.c:121-167): the
                              -DSAMSUNG_TV build always passes SAMSUNG_SPECIFIC_CHANNELS = 12 (:3492-3495,
                              IAMF_defines.h:211-213) — sample-frame i starts at element 12 i, slots beyond
                              out_channels are zero, and for layouts of more than 12 channels the surplus
                              channels overwrite the following sample-frame exactly as the reference's loop
                              does.  A stream then needs (n * stride + max(0, out_channels - stride)) samples
                              of room per call. */
  /* 0, or K < out_channels: the OUTPUT mix gain (constant or ramp) multiplies output channels 0 .. K-1 only.  What the
   * reference's -DSAMSUNG_TV build does after a run-time switch to a layout of more channels: its mixed frame keeps the
   * channel count of the layout the presentation was enabled with, and iamf_frame_gain loops over that
   * (IAMF_decoder.c:1383-1408,3462-3469).  Such a batch renders through the general kernel. */
  int32_t out_gain_channels;
  int32_t reserved[2];
} iamf_hip_batch_config;

/* Creates device state for cfg->n_streams streams on the CURRENT HIP device.  Synchronous.
 * The batch remembers that device: every later call on it (setters, render, flush, reset) returns
 * IAMF_HIP_ERR_INVALID_STATE unless the same device is current; destroy switches to it and back.
 * Replaces, per stream: iamf_stream_renderer_open (IAMF_decoder.c:2480),
 * audio_effect_peak_limiter_create/_init (audio_effect_peak_limiter.c:50,73). */
int iamf_hip_batch_create(const iamf_hip_batch_config *cfg, iamf_hip_batch **out);
void iamf_hip_batch_destroy(iamf_hip_batch *b);

/* Per-stream linear gains (host arrays of n_streams floats, NULL = leave unchanged):
 * element mix gain and output mix gain as iamf_frame_gain applies a constant gain
 * (IAMF_decoder.c:1392-1397: only if != 1 and > 0), and the loudness gain
 * db2lin(target - loudness) of iamf_loudness_process (IAMF_decoder.c:3206-3221).  Synchronous, and ordered after the renders already queued: it
 * first waits for the batch's last render / flush call (an event the batch itself records behind every call, so the
 * caller's stream need not outlive the call). */
int iamf_hip_batch_set_gains(iamf_hip_batch *b, const float *element_gain,
                             const float *output_gain, const float *loudness_gain);

/* Renders n_frames frames of every stream.
 *   d_in : planar f32 element PCM, sample i of channel c of frame f of stream s at
 *          d_in[s*in_stream_stride + f*in_frame_stride + c*frame_size + i]   (strides in floats)
 *   d_pcm: packed output; the k-th sample-frame this call emits for stream s starts at byte
 *          s*pcm_stream_stride_bytes + k*out_channels*bytes_per_sample
 * Returns the number of sample-frames emitted PER STREAM (>= 0; the limiter withholds the
 * first 240 of a stream's life exactly like audio_effect_peak_limiter_process_block :185-201)
 * or a negative IAMF_HIP_ERR_*.  This is the batched form of the render..pack section of
 * iamf_decoder_internal_decode (IAMF_decoder.c:3374-3500) i.e. of iamf_stream_render,
 * iamf_frame_gain, iamf_mixer_mix, iamf_loudness_process,
 * audio_effect_peak_limiter_process_block and iamf_decoder_plane2stride_out. */
int iamf_hip_batch_render(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride,
                          int64_t in_frame_stride, int32_t n_frames, void *d_pcm,
                          int64_t pcm_stream_stride_bytes, void *stream);

/* ---- parametric down-mixer control (host side of src/iamf_dec/downmix_renderer.c) ---- */
typedef struct {
  int32_t mode, w_idx;     /* -1 until set (DMRenderer_open, downmix_renderer.c:151-152) */
  int32_t w_idx_offset;
  int32_t reserved;
  float alpha, beta, gamma, delta; /* MixFactors of the mode (IAMF_utils.c:236-240) */
  float gamma_w;                   /* gamma * w(w_idx): the TL/TR weight (downmix_renderer.c:199-211) */
} iamf_hip_dmx_state;

/* one frame of one stream: samples [0, offset) use `prev`, the rest `cur`
 * ({alpha, beta, gamma, delta, gamma_w}); this is how iamf_stream_render applies a demixing mode
 * change (IAMF_decoder.c:2574-2583) */
typedef struct {
  int32_t offset;
  float prev[5];
  float cur[5];
} iamf_hip_dmx_frame;

/* 1 if DMRenderer_open(in, out) would succeed (downmix_renderer.c:77-91,131-139), else 0 */
int iamf_hip_dmx_valid(int in_layout, int out_layout);
/* channel count of an IAChannelLayoutType (IAMF_utils.c:111), 0 if invalid */
int iamf_hip_dmx_layout_channels(int layout);
void iamf_hip_dmx_state_init(iamf_hip_dmx_state *st);
/* replaces DMRenderer_set_mode_weight (downmix_renderer.c:180-216): 0 or IAMF_HIP_ERR_BAD_ARG */
int iamf_hip_dmx_set_mode_weight(iamf_hip_dmx_state *st, int mode, int w_idx);
/* copies the state's five coefficients into out[5] */
void iamf_hip_dmx_coefficients(const iamf_hip_dmx_state *st, float out[5]);

/* A second audio element for the mix (a sub-mix carries 1 or 2, IAMF_OBU.c:742-752): its renderer
 * matrix (H2M or M2M) and, per stream, its constant element mix gain.  Call before the first
 * render.  The mix is  0 + element0 + element1  in that order (iamf_mixer_mix,
 * IAMF_decoder.c:2719-2730). */
int iamf_hip_batch_set_second_element(iamf_hip_batch *b, const iamf_hip_matrix *mx,
                                      const float *element2_gain);

/* HOA LFE generator with TWO scene-based elements: the reference keeps the filter in the output layout
 * (IAMF_decoder.c:2629-2632, h2m_rdr.c:1151-1239), so both elements' W channels run through the same two histories in
 * turn, frame by frame, in presentation order.  Two batches created with lfe_hoa (same n_streams and sample rate, nothing
 * rendered yet) that render such a pair: `b` adopts `owner`'s filter state; the caller issues each frame's calls in
 * presentation order on ONE HIP stream, and destroys `b` before `owner`.  IAMF_HIP_OK / _BAD_ARG / _INVALID_STATE. */
int iamf_hip_batch_share_lfe_state(iamf_hip_batch *b, iamf_hip_batch *owner);
/* HOA LFE generator: a frame that is rendered and then trimmed away completely (start + end trim = its length, neither of
 * them the whole frame: iamf_stream_render precedes iamf_frame_trim, IAMF_decoder.c:3372-3406).  The generator's filter
 * runs over the frame's n_samples (d_in: the element's planar rows of the frame, as for a render call) for streams
 * [first, first + count); nothing is emitted.  IAMF_HIP_OK (also for a batch without a generator) / _BAD_ARG / _DEVICE. */
int iamf_hip_batch_lfe_advance(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride, int32_t n_samples, void *stream,
                               int32_t first, int32_t count);
/* Ambisonics projection de-mapping in front of element 0's renderer (projection-mode scene-based
 * elements): x[r] = sum over the l_in decoded channels l, ascending, of in[l] * matrix[l*m + r]
 * (iamf_core_decoder_convert_projection, src/iamf_dec/IAMF_core_decoder.c:116-130).  `matrix` is a
 * host array of l_in * m floats (the Q15 demixing matrix of the bitstream as floats); afterwards
 * element 0's input carries l_in channels per frame instead of m.  Call before the first render.
 * With IAMF_HIP_PROJ_EXACT the two stages run as the reference runs them (bit-exact); in tolerance
 * mode (IAMF_HIP_PROJ_MFMA, or _AUTO with an H2M matrix: +-1 LSB of the PCM) they are composed into
 * one matrix on the host and the call costs what a mono-mode element of l_in channels costs. */
int iamf_hip_batch_set_projection(iamf_hip_batch *b, const float *matrix, int l_in);

/* ------------------------------------------------------------------------------------------
 * Demixer of scalable channel audio in front of element 0's renderer (reference
 * src/iamf_dec/demixer.c, driven per frame by iamf_stream_scale_decoder_demix,
 * IAMF_decoder.c:2324-2349).  Element 0's input then carries the decoded channels of all layers up
 * to the chosen one in their bitstream order; the demixer applies the layers' output gains,
 * reconstructs the channels the target layout lacks (S1to2 ... S5to7, TF2toT2, T2toT4 with the
 * frame's demixing mode), smooths the recon gains and hands the renderer the target layout in
 * playback order.  Channel ids are IAChannel values (IAMF_types.h:61-90).
 * ------------------------------------------------------------------------------------------ */
typedef struct iamf_hip_demix_config {
  int32_t layout;         /* target IAChannelLayoutType 0..8 (demixer_set_channel_layout) */
  int32_t n_in;           /* decoded channels = channels of `layout` (demixer_set_channels_order) */
  int32_t chs_in[12];
  int32_t n_gain;         /* demixer_set_output_gain */
  int32_t gain_ch[12];
  float gain[12];
  uint32_t frame_offset;  /* demixer_set_frame_offset: the codec's delay (0 for LPCM) */
} iamf_hip_demix_config;

/* Call before the first render; the batch's matrix must take the target layout's channels. */
int iamf_hip_batch_set_demixer(iamf_hip_batch *b, const iamf_hip_demix_config *cfg);

/* Host-side state of one stream's demixer (demixer_set_demixing_info, demixer.c:592-618, and the
 * recon-gain smoothing of dmx_rms, :447-478). */
typedef struct iamf_hip_demix_state {
  int32_t mode, last_mode, w_idx, last_w_idx;
  float last_sfavg[24];
} iamf_hip_demix_state;

/* What the kernel needs for one frame of one stream (device array [n_streams][n_frames]). */
typedef struct iamf_hip_demix_frame {
  float prev[5], cur[5];            /* alpha, beta, gamma, delta, w of the previous / current mode:
                                       the first frame_offset % frame_size samples use prev */
  int32_t n_recon;                  /* channels of demixer_set_recon_gain valid for this frame: decoded or */
  int32_t recon_ch[12];             /* reconstructed channels of the target layout; distinct IAChannel ids */
  float recon_prev[12], recon_cur[12]; /* per recon channel: smoothed gain of the last frame / of this one */
} iamf_hip_demix_frame;

void iamf_hip_demix_state_init(iamf_hip_demix_state *st);
/* replaces demixer_set_demixing_info: 0 or IAMF_HIP_ERR_BAD_ARG (then nothing changes) */
int iamf_hip_demix_set_info(iamf_hip_demix_state *st, int mode, int w_idx);
/* fills one frame record from the state and the recon channels / gains valid for this frame
 * (recon_gain[i] belongs to recon_ch[i]), and advances the smoothing */
void iamf_hip_demix_frame_fill(iamf_hip_demix_state *st, int n_recon, const int32_t *recon_ch,
                               const float *recon_gain, iamf_hip_demix_frame *out);

/* Everything one render call can take.  Unused pointers are NULL. */
typedef struct {
  const float *d_in;            /* element 0 planar f32, layout as iamf_hip_batch_render */
  int64_t in_stream_stride, in_frame_stride;
  const float *d_in2;           /* element 1 (needs iamf_hip_batch_set_second_element) */
  int64_t in2_stream_stride, in2_frame_stride;
  /* per-sample mix gains for this call, [n_streams][n_frames*frame_size] f32 on the device,
   * built by the host like iamf_database_parameter_get_mix_gain_unit (IAMF_decoder.c:857-982);
   * when given they are multiplied in unconditionally (iamf_frame_gain, :1401-1405) and the
   * constant gain of that stage is not used */
  const float *d_element_ramp, *d_element2_ramp, *d_output_ramp;
  int64_t ramp_stream_stride;   /* floats */
  const iamf_hip_dmx_frame *d_dmx_frames; /* kind DMX: [n_streams][n_frames] on the device */
  int32_t n_frames;
  int32_t n_samples;            /* 0, or with n_frames == 1: render only the first n_samples of the frame
                                   (a frame shortened by iamf_frame_trim, IAMF_decoder.c:1361-1381) */
  void *d_pcm;
  int64_t pcm_stream_stride_bytes;
  void *stream;
  const iamf_hip_demix_frame *d_demix_frames; /* with a demixer: [n_streams][n_frames] on the device */
  int32_t demix_sample0;        /* with a demixer and n_frames == 1: position inside its frame of the call's
                                   first sample (a frame whose start was trimmed before the call) */
  /* HOA LFE generator and a TRIMMED frame (n_frames == 1): the reference renders the whole frame and trims the result
   * (iamf_stream_render then iamf_frame_trim, IAMF_decoder.c:3424-3430), so its low-pass filter also runs over the samples
   * that are cut.  d_in points at the first KEPT sample of every channel row; the generator additionally takes the
   * lfe_pre_samples in front of it and the lfe_post_samples behind the call's last sample from the same rows (W, or every
   * decoded channel in projection mode).  lfe_pre_samples + samples + lfe_post_samples <= frame_size.  0 otherwise. */
  int32_t lfe_pre_samples, lfe_post_samples, reserved0;
} iamf_hip_render_args;

/* Extended form of iamf_hip_batch_render; same return value. */
int iamf_hip_batch_render_ex(iamf_hip_batch *b, const iamf_hip_render_args *args);

/* The same for the streams [stream0, stream0 + n_streams) of the batch only; the others neither advance nor emit.
 * Buffers, strides and the per-stream arrays of `args` are still indexed by the stream's number in the batch.  The
 * streams of the range must stand at the same position (they have consumed the same number of samples) and must not
 * have been flushed, else IAMF_HIP_ERR_INVALID_STATE; after ranges have diverged, the whole-batch calls return that too
 * until the positions meet again.  IAMF_HIP_ERR_UNIMPLEMENTED for a proper sub-range of a batch of kind FIR or with
 * lfe_hoa (state that is kept per batch).  This is what lets a set of decoder handles that do NOT advance in step —
 * a frame trimmed in one stream, a temporal unit still incomplete in another, one stream ending before the rest — share
 * one batch (iamf_hip_decoder_group below); each reference handle is its own state machine, IAMF_decoder.c:3303-3525. */
int iamf_hip_batch_render_range(iamf_hip_batch *b, const iamf_hip_render_args *args, int32_t stream0, int32_t n_streams);
int iamf_hip_batch_flush_range(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stream_stride_bytes, void *stream,
                               int32_t stream0, int32_t n_streams);

/* End of stream: pushes 240 zero samples through each stream's limiter and emits the withheld
 * tail (iamf_delay_buffer_handle, IAMF_decoder.c:3250-3301).  Returns sample-frames per stream. */
int iamf_hip_batch_flush(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stream_stride_bytes,
                         void *stream);

/* ------------------------------------------------------------------------------------------
 * Sample-rate converter for a batch of streams (the decoder's speexdsp-derived resampler at
 * quality 4, src/iamf_dec/resample.c; the decoder only creates one when the stream rate differs
 * from the requested output rate, IAMF_decoder.c:3193-3199).  Buffers are interleaved f32
 * [sample-frame][channel] on the device: the output of a batch with IAMF_HIP_FMT_F32, the input
 * of a batch with frame_size 1.
 * ---------------------------------------------------------------------------------------- */
typedef struct iamf_hip_resampler iamf_hip_resampler;
/* replaces speex_resampler_init(channels, in, out, 4) + speex_resampler_skip_zeros as
 * iamf_stream_resampler_open calls them (IAMF_decoder.c:1892-1909); one state per stream */
int iamf_hip_resampler_create(int n_streams, int channels, int in_rate, int out_rate,
                              iamf_hip_resampler **out);
void iamf_hip_resampler_destroy(iamf_hip_resampler *r);
/* sample-frames the output buffer must hold for ns input frames: ns * (out/in + 1), integer
 * division, as iamf_resample asks (IAMF_decoder.c:3225-3226) */
int iamf_hip_resampler_out_capacity(const iamf_hip_resampler *r, int ns);
int iamf_hip_resampler_flush_capacity(const iamf_hip_resampler *r);
/* replaces iamf_resample / speex_resampler_process_interleaved_float (IAMF_decoder.c:3223-3248,
 * resample.c:974-997).  Stream strides in floats.  Returns sample-frames produced per stream. */
int iamf_hip_resampler_process(iamf_hip_resampler *r, const float *d_in, int64_t in_stream_stride,
                               int ns, float *d_out, int64_t out_stream_stride, void *stream);
/* end of stream (rest_flag 2, IAMF_decoder.c:3227-3232): drains the filter latency */
int iamf_hip_resampler_flush(iamf_hip_resampler *r, float *d_out, int64_t out_stream_stride, void *stream);
/* The same for the streams [stream0, stream0 + n_streams) only (buffers and strides are indexed by the stream's number in
 * the resampler): every stream keeps its own phase and history, so streams need not advance in step — a range must consist
 * of streams in ONE state (iamf_hip_resampler_same_state; IAMF_HIP_ERR_INVALID_STATE otherwise): streams that have
 * consumed the same sequence of call lengths always are. */
int iamf_hip_resampler_process_range(iamf_hip_resampler *r, const float *d_in, int64_t in_stream_stride, int ns, float *d_out,
                                     int64_t out_stream_stride, void *stream, int32_t stream0, int32_t n_streams);
int iamf_hip_resampler_flush_range(iamf_hip_resampler *r, float *d_out, int64_t out_stream_stride, void *stream, int32_t stream0,
                                   int32_t n_streams);
int iamf_hip_resampler_same_state(const iamf_hip_resampler *r, int32_t stream_a, int32_t stream_b);

/* Forgets all stream state (new IA sequence: limiter re-initialised as in
 * iamf_decoder_internal_configure, IAMF_decoder.c:3809-3815).  Synchronous. */
int iamf_hip_batch_reset(iamf_hip_batch *b);

/* bytes one output sample occupies for a format (2, 3, 4) */
int iamf_hip_format_bytes(int out_format);
/* library / build identification string (static storage) */
const char *iamf_hip_version(void);

/* ------------------------------------------------------------------------------------------
 * Diagnostics (not part of the render path).  Launches a kernel with the render kernels' HBM traffic
 * SHAPE and no compute: one 256-thread workgroup per stream, per 1024-sample chunk every lane reads
 * `rows` x 16 B (a frame of rows x 4 KiB per chunk, prefetched one chunk ahead) and writes `pieces` x 16 B
 * (1 KiB contiguous per wave and store instruction).  rows = input channels, pieces = out_channels / 2 for
 * s16.  Time it on the caller's stream to know what the memory system delivers for that shape on THOSE
 * buffers (bench.py: roofline.same_traffic_no_compute).  The buffers' contents are read / overwritten.
 * ---------------------------------------------------------------------------------------- */
int iamf_hip_probe_traffic(int n_streams, int chunks, int rows, int pieces, const void *d_in,
                           int64_t in_stream_stride_bytes, void *d_out, int64_t out_stream_stride_bytes,
                           void *stream);
/* Where the stream buffers live matters (NOTEBOOK.md 3): a batch whose input and PCM output lie in
 * device-memory regions of the same kind runs ~14 % slower than one whose buffers lie in regions of different kinds,
 * and hipMalloc does not say which is which.  This times iamf_hip_probe_traffic on every (input candidate, output
 * candidate) pair — one untimed and three timed launches each, the median — and reports the fastest pair in
 * *best_in / *best_out (indices into the candidate arrays); ms, if not NULL, receives the n_in x n_out medians in
 * milliseconds, row-major.  Synchronous; the candidates' contents are read / overwritten.  At most 256 candidates
 * of each. */
int iamf_hip_pick_buffer_pair(int n_streams, int chunks, int rows, int pieces, const void *const *d_in_candidates,
                              int n_in, int64_t in_stream_stride_bytes, void *const *d_out_candidates, int n_out,
                              int64_t out_stream_stride_bytes, void *stream, int *best_in, int *best_out, float *ms);

/* Self-test (diagnostic): the demixer's quotients through a shared reciprocal (q = n r, e = fma(-d, q, n), q' = fma(e, r,
 * q); iac_amd/csrc/render_common.hpp w4_quot) against the IEEE division n / d for ALL 2^32 f32 numerators, on the
 * device.  counts[0] = numerators inside the range the kernels use the fast form for (2^-100 <= |n| < 2^126), counts[1] =
 * those of them whose quotient differs in any bit (must be 0), counts[2] = numerators outside the range that differ
 * (the kernels divide those the IEEE way).  Synchronous. */
int iamf_hip_selftest_shared_divisor(float divisor, uint64_t counts[3]);

/* ------------------------------------------------------------------------------------------
 * Decoder facade extension.  The reference chooses at BUILD time whether scene-based elements feed
 * the LFE of the output layout (-DDISABLE_LFE_HOA=0; default: compiled out, ae_rdr.h:63-65).  This
 * library carries both builds: the switch is per decoder handle (an IAMF_DecoderHandle of this
 * library's IAMF_decoder.h), to be set before IAMF_decoder_configure; its default is off, or on if
 * the environment has IAMF_HIP_LFE_HOA=1 when the handle is opened.
 * ---------------------------------------------------------------------------------------- */
int iamf_hip_decoder_set_hoa_lfe(void *decoder_handle, int enable);
/* Which build of the reference the handle behaves as: IAMF_HIP_VARIANT_SAMSUNG_TV = its tables
 * (m2m_rdr.c:36-830), 12-channel PCM stride (IAMF_decoder.c:3492-3495,3510-3512), always the top layer of a
 * scalable element (:1782-1822).  Before IAMF_decoder_configure; default VARIANT_DEFAULT, or SAMSUNG_TV if
 * the environment has IAMF_HIP_SAMSUNG_TV=1 when the handle is opened. */
int iamf_hip_decoder_set_variant(void *decoder_handle, int variant);

/* ------------------------------------------------------------------------------------------
 * One node, several GPUs, from C (BASELINE north_star: "independent IAMF streams shard embarrassingly
 * across the 8 GPUs of one node with RCCL over xGMI only for the final batched gather").  All state of the path is per
 * decoder handle in the reference (IAMF_decoder_private.h:342-345, audio_effect_peak_limiter.h:48-73), so the streams
 * of a job are split into contiguous blocks, one batch per device, no exchange while rendering.  A shard owns, per
 * device: the batch, a HIP stream for rendering, a second one for the gather, and a host thread that issues that
 * device's launches.  RCCL is loaded with dlopen at the first gather (the library does not link it; a host without it
 * gets IAMF_HIP_ERR_UNIMPLEMENTED from the gather and can still render; IAMF_HIP_RCCL_LIB in the environment names the
 * library to load instead of the system's librccl).
 * ---------------------------------------------------------------------------------------- */
typedef struct iamf_hip_shard iamf_hip_shard;
/* block of streams of device `index` of `n_devices`: sizes differ by at most one, the larger blocks first */
int iamf_hip_shard_split(int n_streams, int n_devices, int index, int *first, int *count);
/* cfg->n_streams = streams of the whole job; devices = n_devices distinct HIP device ordinals (NULL = 0 .. n_devices - 1) */
int iamf_hip_shard_create(const iamf_hip_batch_config *cfg, const int *devices, int n_devices, iamf_hip_shard **out);
void iamf_hip_shard_destroy(iamf_hip_shard *s);
int iamf_hip_shard_devices(const iamf_hip_shard *s);
int iamf_hip_shard_info(const iamf_hip_shard *s, int index, int *device, int *first, int *count);
/* the batch of device `index` (for the per-stream setters: make that device current first) and its render stream */
iamf_hip_batch *iamf_hip_shard_batch(iamf_hip_shard *s, int index);
void *iamf_hip_shard_render_stream(iamf_hip_shard *s, int index);
/* iamf_hip_batch_render / _flush on every device at once: d_in[i] / d_pcm[i] are device i's buffers (its block of streams,
 * strides as for the batch).  Returns sample-frames emitted per stream.  Asynchronous on the devices' render streams. */
int iamf_hip_shard_render(iamf_hip_shard *s, const float *const *d_in, int64_t in_stream_stride, int64_t in_frame_stride,
                          int32_t n_frames, void *const *d_pcm, int64_t pcm_stream_stride_bytes);
int iamf_hip_shard_flush(iamf_hip_shard *s, void *const *d_pcm, int64_t pcm_stream_stride_bytes);
/* The job's one exchange: every device's PCM regions (pcm_stream_stride_bytes per stream) -> d_dst on device
 * `root_index`, stream s of the job at d_dst + s * dst_stream_stride_bytes (the two strides must be equal).  ncclSend /
 * ncclRecv in one group on the gather streams, behind the last render: it overlaps the next iamf_hip_shard_render, which
 * in turn waits for it before overwriting the buffers it reads.  iamf_hip_shard_sync waits for everything. */
int iamf_hip_shard_gather(iamf_hip_shard *s, int root_index, void *d_dst, int64_t dst_stream_stride_bytes,
                          void *const *d_pcm, int64_t pcm_stream_stride_bytes);
/* The same exchange for the ROWS only: bytes_per_stream bytes of every stream's region (what the last render or flush
 * emitted: sample-frames x channels x bytes per sample) instead of the whole region with its padding — a flush moves 240
 * sample-frames per stream, not a call's region.  The strides may differ (each >= bytes_per_stream): rows that are not
 * back to back are packed / spread through staging buffers of the shard on the gather streams. */
int iamf_hip_shard_gather_rows(iamf_hip_shard *s, int root_index, void *d_dst, int64_t dst_stream_stride_bytes,
                               void *const *d_pcm, int64_t pcm_stream_stride_bytes, int64_t bytes_per_stream);
int iamf_hip_shard_sync(iamf_hip_shard *s);
/* What the gather cost device `index` (any pointer may be NULL): bytes it sent in the last gather / in all gathers, bytes it
 * received in the last one (the root: all devices' rows), milliseconds its gather stream spent on its share (pack, send /
 * receive, spread) in the last gather / in all.  Waits for that device's last gather.  SURVEY 8(e): the root's inbound
 * links bound the exchange — this is where a caller reads what each peer put on the wire. */
int iamf_hip_shard_times(iamf_hip_shard *s, int index, int64_t *last_sent_bytes, int64_t *total_sent_bytes,
                         int64_t *last_received_bytes, double *last_gather_ms, double *total_gather_ms);
/* "major.minor.patch" of the RCCL found on this host, "" if none (static storage) */
const char *iamf_hip_shard_rccl_version(void);

/* ------------------------------------------------------------------------------------------
 * LPCM sub-stream packets -> planar f32 element PCM, on the device.
 * Replaces, for a batch of streams, the reference's LPCM decoder (src/iamf_dec/pcm/IAMF_pcm_decoder.c:64-83, 133-149:
 * sample = integer / 2^(bits-1); 16 / 24 / 32 bit; little-endian, or big-endian with the reference's own byte order for
 * 24 bit, bitstream.c:204-208) and the re-ordering from audio-layer order to the renderer's channel order behind it
 * (IAMF_decoder.c:2230-2260).  The caller uploads each stream's packets as they are in the bitstream into a raw region
 * of `raw_stream_stride` bytes per stream, at offsets of its choosing, and describes where output channel c finds its
 * samples: sample i of channel c = the `sample_bytes` bytes at  src_offset[c] + (first + i) * src_step[c]  (src_step = bytes
 * from one sample of the channel to the next: sample_bytes for a mono sub-stream, twice that for a coupled one);
 * src_offset[c] < 0 = a channel no sub-stream carries: silence.  d_first_count: on the device, per stream two int32,
 * `first_count_stride` int32 from one stream's pair to the next (2 for a packed array; raw_stream_stride / 4 for pairs
 * kept at the head of each stream's raw region, so that one upload carries both; 0: one pair for all streams): the first sample to take (a trimmed
 * start, iamf_frame_trim IAMF_decoder.c:1361-1381) and the number of samples to write (0: the stream is left alone);
 * first + count <= frame_size is the caller's to guarantee.  Output:
 * d_out[stream * out_stream_stride + c * frame_size + i], i < count.  Asynchronous on `stream`.
 * Returns IAMF_HIP_OK, IAMF_HIP_ERR_BAD_ARG (a layout that could read outside a stream's raw region, frame_size not a
 * multiple of 4, ...) or IAMF_HIP_ERR_DEVICE.
 * ---------------------------------------------------------------------------------------- */
#define IAMF_HIP_LPCM_MAX_CHANNELS 32
typedef struct iamf_hip_lpcm_layout {
  int32_t sample_bytes;    /* 2, 3, 4 */
  int32_t little_endian;   /* 0: big-endian (codec config sample_format_flags, IAMF_pcm_decoder.c:52-60) */
  int32_t channels;        /* rows written per stream */
  int32_t frame_size;      /* floats per row */
  int32_t src_offset[IAMF_HIP_LPCM_MAX_CHANNELS];
  int32_t src_step[IAMF_HIP_LPCM_MAX_CHANNELS];
} iamf_hip_lpcm_layout;
int iamf_hip_lpcm_unpack(const iamf_hip_lpcm_layout *layout, const void *d_raw, int64_t raw_stream_stride,
                         const int32_t *d_first_count, int64_t first_count_stride, float *d_out, int64_t out_stream_stride,
                         int32_t n_streams, void *stream);

/* Render with element 0 handed over as LPCM packets instead of planar f32: iamf_hip_lpcm_unpack and iamf_hip_batch_render_ex
 * in one call, and for the calls of the headline kernel in one KERNEL — the reference decodes a packet into its f32 decoder
 * buffer (pcm/IAMF_pcm_decoder.c:64-83) and renders from there (IAMF_decoder.c:2550-2640); on the device that f32 copy is
 * 64 of the path's 68 bytes per sample-frame, so the render kernel reads the packets' 16-bit samples itself (32 + 4 bytes)
 * and converts them where it loads them.  d_raw: [n_streams][n_frames] packet rows, frame f of stream s at
 * d_raw + s * raw_stream_stride + f * raw_frame_stride (bytes), described by `layout` exactly as for iamf_hip_lpcm_unpack
 * (offsets relative to the frame's row; layout.channels = the element's channels, layout.frame_size = the batch's).
 * first_sample: with args->n_frames == 1 and args->n_samples > 0, the frame position of the first sample to render (a
 * trimmed start); else 0.  `args` as for iamf_hip_batch_render_ex with d_in == NULL (the in_* strides are not used).
 * Fused when: 16-bit little-endian, every channel a contiguous run (src_step == 2) at an offset that is a multiple of 8
 * bytes (with first_sample), strides multiples of 8, d_raw 16-byte aligned, and the batch is a plain matrix render of a
 * 1 / 4 / 9 / 16-channel element into one or two channels with the limiter on (no second element, ramps, demixer, down-mixer,
 * de-mapping, FIR or LFE generator).  Every other input — 24 / 32 bit, big-endian, coupled sub-streams, wide layouts —
 * takes iamf_hip_lpcm_unpack into a buffer of the batch and then the f32 kernels.  The PCM is the same bit for bit either
 * way (tests/test_gpu_lpcm.py); IAMF_HIP_LPCM_UNFUSED=1 in the environment forces the second way.
 * Returns what iamf_hip_batch_render_ex returns. */
typedef struct iamf_hip_lpcm_input {
  const void *d_raw;
  int64_t raw_stream_stride, raw_frame_stride;   /* bytes */
  int32_t first_sample;
  iamf_hip_lpcm_layout layout;
} iamf_hip_lpcm_input;
int iamf_hip_batch_render_lpcm(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *args);
/* ... for the streams [stream0, stream0 + n_streams) of the batch only, as iamf_hip_batch_render_range */
int iamf_hip_batch_render_lpcm_range(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *args,
                                     int32_t stream0, int32_t n_streams);

/* Host -> device by a kernel that reads pinned host memory (hipHostMalloc) over PCIe, 16 bytes per lane: the bytes of a
 * hipMemcpyAsync without leaving the compute queue, for callers that put a small upload between kernels (a pinned
 * hipMemcpyAsync costs ~9 us per call and the hand-over between copy engine and compute queue ~12 us each way on
 * MI355X; the kernel moves 8 MB as fast as the engine).  Both pointers 16-byte aligned, `bytes` a multiple of 16.
 * IAMF_HIP_OK, IAMF_HIP_ERR_BAD_ARG or IAMF_HIP_ERR_DEVICE. */
int iamf_hip_upload_by_kernel(const void *h_pinned, void *d_dst, size_t bytes, void *stream);
/* A one-lane kernel on `stream` that stores `seq` to a word of PINNED host memory behind a system-scope fence: a host
 * that has just queued a few microseconds of work and wants its result spins on the word instead of calling
 * hipStreamSynchronize (MI355X, tools/debug/sync_probe.hip: 14.8 instead of 18.1 us per launch-and-wait).  What was queued
 * before it on the stream has completed and is visible to the host when the word reads `seq`.  The caller bounds its
 * spin and falls back to hipStreamSynchronize (which also reports a device error).  IAMF_HIP_OK / _BAD_ARG / _DEVICE. */
int iamf_hip_stream_signal(void *stream, volatile uint32_t *h_pinned_flag, uint32_t seq);
/* f32 sample-frames [stream][sample][channels] — what a batch with out_format IAMF_HIP_FMT_F32 writes — to planar
 * [stream][channel][dst_channel_stride], the form a batch reads element PCM in: how one batch's rendered frame becomes
 * another batch's (second) element, e.g. a presentation both of whose elements need a per-stream stage (the reference
 * mixes rendered frames, iamf_mixer_mix, IAMF_decoder.c:2702-2733).  Strides in floats; channels <= 24.
 * IAMF_HIP_OK / _BAD_ARG / _DEVICE. */
int iamf_hip_deinterleave_f32(const float *d_src, int64_t src_stream_stride, int32_t channels, int32_t n_streams,
                              int32_t n_samples, float *d_dst, int64_t dst_stream_stride, int64_t dst_channel_stride, void *stream);

/* ------------------------------------------------------------------------------------------
 * A group of decoder handles: callers of the reference API get the batch renderer's throughput.
 * The reference renders one handle, one frame per call (IAMF_decoder_decode, include/IAMF_decoder.h:82-99, driver loop
 * src/iamf_dec/IAMF_decoder.c:3303-3525).  N configured handles (IAMF_DecoderHandle of this library's IAMF_decoder.h) of ONE
 * topology — same codec configuration, elements, output layout, bit depth and limiter settings; their mix gains,
 * loudness and parameter streams may differ — that have not decoded yet hand their rendering to one batch:
 *     iamf_hip_decoder_group_decode(g, data, sizes, rsizes, pcm, results)
 * is, for every i, exactly  results[i] = IAMF_decoder_decode(handles[i], data[i], sizes[i], &rsizes[i], pcm[i])
 * (data[i] == NULL flushes handle i; rsizes may be NULL): parsing and the parameter timelines run per handle on
 * `host_threads` threads (0 = up to 16), which also copy each handle's LPCM packets, as they are, into pinned staging; then
 * ONE upload, the device unpacks (iamf_hip_lpcm_unpack above), one render launch over the streams that completed a
 * temporal unit, one download.  The handles need not advance in step.  While grouped, a handle refuses
 * IAMF_decoder_decode / _configure / _close with IAMF_ERR_INVALID_STATE; destroying the group releases the handles
 * (which are then closed with IAMF_decoder_close as usual).  Returns IAMF_OK, or for create: IAMF_ERR_BAD_ARG (handles
 * of different topologies), IAMF_ERR_INVALID_STATE (not configured / already decoding / a resampler inherited from an earlier IA sequence).  The group's own return value reports device failures only
 * (IAMF_ERR_INTERNAL): such a round is half applied — the library waits for what it had in flight, and every later
 * _decode of the group returns IAMF_ERR_INVALID_STATE; the valid call after a failure is _destroy.
 * ---------------------------------------------------------------------------------------- */
typedef struct iamf_hip_decoder_group iamf_hip_decoder_group;
int iamf_hip_decoder_group_create(void *const *handles, int n, int host_threads, iamf_hip_decoder_group **out);
/* where the group's calls spent their time so far, seconds: [0] host parsing + packet staging (the thread pool), [1] enqueueing
 * the uploads, the unpack and render launches and the download, [2] waiting for the device, [3] handing every handle its PCM;
 * *rounds (may be NULL) = calls of iamf_hip_decoder_group_decode.  IAMF_OK or IAMF_ERR_BAD_ARG. */
int iamf_hip_decoder_group_times(const iamf_hip_decoder_group *g, double *seconds4, int64_t *rounds);
int iamf_hip_decoder_group_decode(iamf_hip_decoder_group *g, const uint8_t *const *data, const int32_t *sizes,
                                  uint32_t *rsizes, void *const *pcm, int32_t *results);
void iamf_hip_decoder_group_destroy(iamf_hip_decoder_group *g);

#ifdef __cplusplus
}
#endif
#endif /* IAMF_HIP_H */
