/*
 * oracle/iamf_oracle.h — CPU restatement of the IAMF post-decode rendering path.
 *
 * THIS IS TEST INFRASTRUCTURE (the parity checker and the timed CPU baseline).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (iac_amd/, include/) never links or calls anything in oracle/.
 *
 * Every function restates one piece of the reference (Samsung/iac, libiamf) in plain scalar
 * C with the same f32 operation order, and names the reference lines it follows.  Built with
 * -O3 -ffp-contract=off so that x86-64 evaluates exactly the IEEE-754 f32 sequence the
 * reference binary evaluates.  Parity status: PINNED — tests/test_oracle_golden.py checks it
 * bit-for-bit against fixtures in tests/golden/ that oracle/gen_golden.py produced by calling
 * the real reference (oracle/_ref/libiamf_ref.so, compiled from /root/reference).
 */
#ifndef IAMF_ORACLE_H
#define IAMF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CH 24      /* reference common/audio_defines.h:58 MAX_OUTPUT_CHANNELS */
#define ORC_MAX_DELAY 4096 /* reference common/audio_defines.h:59 MAX_DELAYSIZE       */

/* ---- static rendering matrices (data blob dumped from the reference, see dump_tables.c) ---- */
typedef struct {
  int kind;     /* 0 = HOA->layout (h2m), 1 = layout->layout (m2m) */
  int in_id;    /* h2m: ambisonics order 0..3 ; m2m: rendering id of the input layout  */
  int out_id;   /* rendering id of the output layout (0x020, 0x9A3, ...)               */
  int channels; /* h2m: the table's `channels` field ; m2m: n                          */
  int lfe1, lfe2;
  int m, n;     /* inputs, computed outputs */
  const float *mat;
} orc_matrix;

int orc_tables_load(const char *path);
int orc_tables_count(void);
int orc_tables_entry(int idx, orc_matrix *out);
/* reference src/iamf_dec/h2m_rdr.c:1070-1081 */
int orc_get_h2m(int order, int out_id, orc_matrix *out);
/* reference src/iamf_dec/m2m_rdr.c:1786-1804 */
int orc_get_m2m(int in_id, int out_id, orc_matrix *out);

/* ---- element renderers; planar contiguous buffers: in[m][ns], out[ch][ns] ---- */
/* reference src/iamf_dec/h2m_rdr.c:1088-1150 (DISABLE_LFE_HOA == 1 branch) */
void orc_render_h2m(const orc_matrix *mx, const float *in, float *out, int ns);
/* reference src/iamf_dec/m2m_rdr.c:1820-1840 */
void orc_render_m2m(const orc_matrix *mx, const float *in, float *out, int ns);

/* ---- HOA LFE generator (iamf_oracle_lfe.c) ---- reference h2m_rdr.c:1151-1239, compiled in only
 * with the reference's switch -DDISABLE_LFE_HOA=0; struct = lfe_filter_t, ae_rdr.h:92-96 */
typedef struct {
  int init;
  float c, a1, a2, a3, b1, b2;
  float ih[2], oh[2];
} orc_lfe;
void orc_lfe_init(orc_lfe *f, float cutoff_freq, float sample_rate);
float orc_lfe_update(orc_lfe *f, float input);
void orc_render_h2m_lfe(const orc_matrix *mx, const float *in, float *out, int ns, orc_lfe *lfe);
int orc_sizeof_lfe(void);

/* ---- gains, mixer, loudness ---- */
/* reference src/iamf_dec/IAMF_decoder.c:1392-1397 (constant) and :1401-1405 (per-sample) */
void orc_frame_gain_const(float *data, int channels, int ns, float gain);
void orc_frame_gain_ramp(float *data, int channels, int ns, const float *gains);
/* reference src/iamf_dec/IAMF_decoder.c:639-645 / :647-664 (host-side ramp builders) */
void orc_mix_gain_linear(float s, float e, int d, int o, int l, float *g);
void orc_mix_gain_quad(float s, float e, int d, float c, int ct, int o, int l, float *g);
/* reference src/iamf_dec/IAMF_decoder.c:2719-2730 */
void orc_mix(float *dst, const float *const *frames, int n_elements, int channels, int ns);
/* reference src/iamf_dec/IAMF_decoder.c:3206-3221 */
void orc_loudness(float *block, int ns, int channels, float gain);
/* reference src/common/fixedp11_5.c:72 */
float orc_db2lin(float db);

/* ---- peak limiter ---- reference src/iamf_dec/audio_effect_peak_limiter.c:73-271 */
typedef struct {
  int ch, delay, pad, started;
  float thr, atk, rel, inc;
  float g, gs, ge, tc;
  int head;   /* ring position of the oldest sample (reference: entryIndex) */
  int maxpos; /* ring position known to hold the window maximum, -1 = unknown */
  float pk[ORC_MAX_DELAY + 1];
  float dl[ORC_MAX_CH][ORC_MAX_DELAY + 1];
} orc_limiter;

void orc_limiter_init(orc_limiter *lim, float threshold_db, int rate, int channels,
                      float atk_sec, float rel_sec, int delay);
/* in/out planar [ch][ns]; returns the number of samples emitted (ns, or ns - pad once) */
int orc_limiter_process(orc_limiter *lim, const float *in, float *out, int ns);

/* ---- float -> interleaved PCM ---- reference src/iamf_dec/IAMF_decoder.c:100-167 */
void orc_pack(void *dst, const float *src, int ns, int channels, int bit_depth, int stride);

/* ---- parametric down-mixer ---- reference src/iamf_dec/downmix_renderer.c:53-242 */
typedef struct orc_downmixer orc_downmixer;
orc_downmixer *orc_dmx_open(int in_layout, int out_layout);
void orc_dmx_close(orc_downmixer *d);
int orc_dmx_set_mode_weight(orc_downmixer *d, int mode, int w_idx);
int orc_dmx_downmix(orc_downmixer *d, const float *in, float *out, int s, int duration, int size);
int orc_dmx_w_idx(const orc_downmixer *d);

/* demixer of scalable channel audio (iamf_oracle_demix.c; reference src/iamf_dec/demixer.c) */
typedef struct orc_demixer orc_demixer;
orc_demixer *orc_demixer_open(int frame_size);
void orc_demixer_close(orc_demixer *d);
int orc_demixer_set_frame_offset(orc_demixer *d, unsigned offset);
int orc_demixer_set_channel_layout(orc_demixer *d, int layout);
int orc_demixer_set_channels_order(orc_demixer *d, const int *chs, int count);
int orc_demixer_set_output_gain(orc_demixer *d, const int *chs, const float *gain, int count);
int orc_demixer_set_demixing_info(orc_demixer *d, int mode, int w_idx);
int orc_demixer_set_recon_gain(orc_demixer *d, int count, const int *chs, const float *gain, unsigned flags);
int orc_demixer_demix(orc_demixer *d, float *dst, float *src, int size);
void orc_demixer_state(const orc_demixer *d, int *mode, int *last_mode, int *w_idx, int *last_w_idx, int *skip);
const float *orc_demixer_window(const orc_demixer *d, int stop);
float orc_demixer_last_sfavg(const orc_demixer *d, int ch);

/* ---- speex-derived resampler (oracle/iamf_oracle_resample.c) ---- reference resample.c */
typedef struct orc_resampler orc_resampler;
orc_resampler *orc_resampler_open(int channels, int in_rate, int out_rate, int quality);
void orc_resampler_close(orc_resampler *r);
/* planar in[ch][ns] -> planar out[ch][ret]; mirrors iamf_resample, IAMF_decoder.c:3223-3248 */
int orc_resample(orc_resampler *r, const float *in, float *out, int ns);
int orc_resample_flush(orc_resampler *r, float *out);
int orc_resample_out_capacity(const orc_resampler *r, int ns);
int orc_resample_flush_capacity(const orc_resampler *r);

/* ---- one stream, render -> pack: the stage order of IAMF_decoder.c:3335-3500 ---- */
typedef struct {
  orc_matrix mx;       /* element renderer matrix (kind selects h2m / m2m) */
  float element_gain;  /* constant element mix gain (linear); applied iff != 1 and > 0 */
  float output_gain;   /* constant output mix gain (linear); same rule */
  float loudness_gain; /* db2lin(target - loudness) or 1.0f when normalisation is off */
  int loudness_on;     /* normalization_loudness != 0 */
  int limiter_on;
  int out_channels;    /* channels of the output layout */
  int bit_depth;       /* 16 / 24 / 32 */
  orc_limiter lim;
  float *buf_a, *buf_b; /* scratch [ORC_MAX_CH][max_ns] */
  int max_ns;
  int lfe_on;          /* HOA LFE generator (IAMF_decoder.c:2625-2633): set with orc_stream_enable_lfe */
  orc_lfe lfe;
} orc_stream;
/* after orc_stream_open: h2m element into a layout with an LFE slot, reference built -DDISABLE_LFE_HOA=0 */
void orc_stream_enable_lfe(orc_stream *s, int rate);

int orc_stream_open(orc_stream *s, const orc_matrix *mx, int out_channels, float element_gain,
                    float output_gain, int loudness_on, float loudness_gain, int limiter_on,
                    float threshold_db, int rate, int bit_depth, int max_ns);
void orc_stream_close(orc_stream *s);
/* in: planar f32 [m][ns]; pcm: interleaved out; returns sample-frames written */
int orc_stream_frame(orc_stream *s, const float *in, int ns, void *pcm);
/* end of stream: pushes `delay` zeros through the limiter (IAMF_decoder.c:3250-3301) */
int orc_stream_flush(orc_stream *s, void *pcm);

/* whole stream in one call (in: [n_frames][m][ns]); for the timed CPU baseline */
long orc_stream_run_frames(const orc_matrix *mx, int out_channels, int limiter_on, float threshold_db,
                           int rate, int bit_depth, const float *in, int n_frames, int ns, void *pcm);

int orc_sizeof_limiter(void);
int orc_sizeof_stream(void);

#ifdef __cplusplus
}
#endif
#endif
