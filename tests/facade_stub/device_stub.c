/*
 * TEST INFRASTRUCTURE — not a CPU path of the product.
 *
 * Stand-ins for the HIP runtime and for the batch ABI (include/iamf_hip.h), so that the plain-C host
 * side of the decoder facade (OBU parser, LPCM unpack, parameter timeline, call protocol) can be
 * compiled with gcc -fsanitize=address,undefined and fed malformed bitstreams on a machine without a
 * GPU (tests/test_facade_malformed.py).  Nothing here renders: "device" memory is malloc, copies
 * are memcpy (so ASan sees every host-side length), a render call reports the samples it was
 * asked for and writes zeros.  It is never linked into libiamf_hip.so.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "iamf_hip.h"

hipError_t hipMalloc(void **p, size_t n) { return (*p = calloc(1, n ? n : 1)) ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned f) { (void)f; return hipMalloc(p, n); }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st) {
  (void)k; (void)st; memcpy(d, s, n); return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { (void)k; memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { (void)st; memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }

struct iamf_hip_batch { iamf_hip_batch_config cfg; int *pad_left; /* per stream: limiter delay still to withhold */ int m2; };
struct iamf_hip_resampler { int ch, in, out; };

static const int k_ch[] = {2, 6, 8, 10, 11, 12, 14, 24, 8, 12};
int iamf_hip_layout_channels(int id) {
  switch (id) {
    case IAMF_HIP_SS_A: case IAMF_HIP_L_STEREO: case IAMF_HIP_L_BINAURAL: return 2;
    case IAMF_HIP_L_MONO: return 1;
    case IAMF_HIP_SS_B: case IAMF_HIP_L_51: case IAMF_HIP_L_312: return 6;
    case IAMF_HIP_SS_C: case IAMF_HIP_L_512: case IAMF_HIP_L_71: case IAMF_HIP_SS_I: return 8;
    case IAMF_HIP_SS_D: case IAMF_HIP_L_514: case IAMF_HIP_L_712: return 10;
    case IAMF_HIP_SS_E: return 11;
    case IAMF_HIP_SS_F: case IAMF_HIP_SS_J: case IAMF_HIP_L_714: return 12;
    case IAMF_HIP_SS_G: return 14;
    case IAMF_HIP_SS_H: return 24;
  }
  (void)k_ch;
  return 0;
}
static float g_zero[24 * 24];
static int fill_matrix(int kind, int m, int out_id, iamf_hip_matrix *mx) {
  memset(mx, 0, sizeof(*mx));
  mx->kind = kind; mx->out_id = out_id; mx->channels = iamf_hip_layout_channels(out_id);
  mx->lfe1 = mx->lfe2 = -1; mx->m = m; mx->n = mx->channels; mx->mat = g_zero;
  return mx->channels ? 0 : IAMF_HIP_ERR_BAD_ARG;
}
int iamf_hip_get_h2m_matrix(int order, int out_id, iamf_hip_matrix *mx) {
  return fill_matrix(IAMF_HIP_KIND_H2M, (order + 1) * (order + 1), out_id, mx);
}
int iamf_hip_get_m2m_matrix(int in_id, int out_id, iamf_hip_matrix *mx) {
  return fill_matrix(IAMF_HIP_KIND_M2M, iamf_hip_layout_channels(in_id), out_id, mx);
}
int iamf_hip_get_m2m_matrix_variant(int v, int in_id, int out_id, iamf_hip_matrix *mx) {
  (void)v;
  return iamf_hip_get_m2m_matrix(in_id, out_id, mx);
}
int iamf_hip_format_bytes(int f) { return f == 16 ? 2 : f == 24 ? 3 : (f == 32 || f == -32) ? 4 : 0; }

int iamf_hip_batch_create(const iamf_hip_batch_config *c, iamf_hip_batch **out) {
  if (c->frame_size <= 0 || c->out_channels <= 0 || c->out_channels > 24 || c->n_streams <= 0) return IAMF_HIP_ERR_BAD_ARG;
  *out = (iamf_hip_batch *)calloc(1, sizeof(**out));
  (*out)->cfg = *c;
  (*out)->pad_left = (int *)calloc((size_t)c->n_streams, sizeof(int));
  for (int s = 0; s < c->n_streams; ++s) (*out)->pad_left[s] = c->limiter_enable ? 240 : 0;
  return 0;
}
void iamf_hip_batch_destroy(iamf_hip_batch *b) {
  if (b) free(b->pad_left);
  free(b);
}
int iamf_hip_batch_set_gains(iamf_hip_batch *b, const float *a, const float *c, const float *d) {
  (void)b; (void)a; (void)c; (void)d; return 0;
}
int iamf_hip_batch_set_second_element(iamf_hip_batch *b, const iamf_hip_matrix *m, const float *g) {
  (void)g;
  if (!m || m->m <= 0 || m->m > 24) return IAMF_HIP_ERR_BAD_ARG;
  b->m2 = m->m;
  return 0;
}
int iamf_hip_batch_lfe_advance(iamf_hip_batch *b, const float *in, int64_t ss, int32_t n, void *st, int32_t s0, int32_t cnt) {
  (void)st;
  if (!b || !in || n <= 0 || n > b->cfg.frame_size || s0 < 0 || cnt <= 0 || s0 + cnt > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  float acc = 0;
  for (int s = s0; s < s0 + cnt; ++s)
    for (int i = 0; i < n; ++i) acc += ((const volatile float *)in)[(int64_t)s * ss + i];
  (void)acc;
  return IAMF_HIP_OK;
}
int iamf_hip_batch_share_lfe_state(iamf_hip_batch *b, iamf_hip_batch *o) {
  return b && o && b != o && b->cfg.lfe_hoa && o->cfg.lfe_hoa && b->cfg.n_streams == o->cfg.n_streams ? 0 : IAMF_HIP_ERR_BAD_ARG;
}
int iamf_hip_batch_set_projection(iamf_hip_batch *b, const float *p, int l) { (void)b; (void)p; (void)l; return 0; }
int iamf_hip_batch_set_demixer(iamf_hip_batch *b, const iamf_hip_demix_config *c) { (void)b; (void)c; return 0; }
static int emit(iamf_hip_batch *b, int s, void *pcm, int64_t cap, int n) {
  int skip = n < b->pad_left[s] ? n : b->pad_left[s];
  b->pad_left[s] -= skip;
  n -= skip;
  const int sc = b->cfg.pcm_stride_channels > 0 ? b->cfg.pcm_stride_channels : b->cfg.out_channels;
  int64_t need = ((int64_t)n * sc + (n > 0 && b->cfg.out_channels > sc ? b->cfg.out_channels - sc : 0)) *
                 iamf_hip_format_bytes(b->cfg.out_format);
  if (need > cap) return IAMF_HIP_ERR_BAD_ARG;
  memset((char *)pcm + (int64_t)s * cap, 0, (size_t)need); /* a real write: ASan checks the "device" buffer the facade sized */
  return n;
}
int iamf_hip_batch_render(iamf_hip_batch *b, const float *in, int64_t ss, int64_t fs, int32_t nf, void *pcm,
                          int64_t cap, void *st) {
  (void)in; (void)ss; (void)fs; (void)st;
  int r = 0;
  for (int s = 0; s < b->cfg.n_streams; ++s) r = emit(b, s, pcm, cap, nf * b->cfg.frame_size);
  return r;
}
int iamf_hip_batch_render_range(iamf_hip_batch *b, const iamf_hip_render_args *a, int32_t s0, int32_t cnt) {
  int n = a->n_samples ? a->n_samples : a->n_frames * b->cfg.frame_size, r = 0;
  if (s0 < 0 || cnt <= 0 || s0 + cnt > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  for (int s = s0; s < s0 + cnt; ++s) {
    /* touch what a kernel would read (an LFE element's rows start lfe_pre_samples in front of d_in) */
    const volatile float *in = (const volatile float *)a->d_in - a->lfe_pre_samples + (int64_t)s * a->in_stream_stride;
    float acc = 0;
    for (int64_t i = 0; i < (int64_t)b->cfg.matrix.m * b->cfg.frame_size && b->cfg.matrix.kind != IAMF_HIP_KIND_DMX; ++i) acc += in[i];
    (void)acc;
    if (a->d_element_ramp) acc += ((const volatile float *)a->d_element_ramp)[(int64_t)s * a->ramp_stream_stride + n - 1];
    if (a->d_in2) /* the second element: m2 planar rows of the frame */
      for (int c = 0; c < b->m2; ++c)
        for (int i = 0; i < n; ++i) acc += ((const volatile float *)a->d_in2)[(int64_t)s * a->in2_stream_stride + (int64_t)c * b->cfg.frame_size + i];
    r = emit(b, s, a->d_pcm, a->pcm_stream_stride_bytes, n);
    if (r < 0) return r;
  }
  return r;
}
int iamf_hip_batch_render_ex(iamf_hip_batch *b, const iamf_hip_render_args *a) {
  return iamf_hip_batch_render_range(b, a, 0, b->cfg.n_streams);
}
int iamf_hip_batch_flush_range(iamf_hip_batch *b, void *pcm, int64_t cap, void *st, int32_t s0, int32_t cnt) {
  (void)st;
  int r = 0;
  for (int s = s0; s < s0 + cnt; ++s) {
    int n = b->cfg.limiter_enable ? 240 - b->pad_left[s] : 0;
    b->pad_left[s] = 0;
    r = emit(b, s, pcm, cap, n);
  }
  return r;
}
int iamf_hip_batch_flush(iamf_hip_batch *b, void *pcm, int64_t cap, void *st) {
  return iamf_hip_batch_flush_range(b, pcm, cap, st, 0, b->cfg.n_streams);
}
int iamf_hip_deinterleave_f32(const float *src, int64_t sss, int32_t ch, int32_t ns, int32_t n, float *dst, int64_t dss, int64_t dcs,
                              void *st) {
  (void)st;
  if (!src || !dst || ch <= 0 || ch > 24 || ns <= 0 || n < 0 || dcs < n) return IAMF_HIP_ERR_BAD_ARG;
  for (int s = 0; s < ns; ++s)
    for (int c = 0; c < ch; ++c)
      for (int i = 0; i < n; ++i) dst[s * dss + c * dcs + i] = src[s * sss + (int64_t)i * ch + c];
  return IAMF_HIP_OK;
}
int iamf_hip_stream_signal(void *st, volatile uint32_t *flag, uint32_t seq) { (void)st; *flag = seq; return IAMF_HIP_OK; }
int iamf_hip_upload_by_kernel(const void *h, void *d, size_t n, void *st) {
  (void)st;
  if (!h || !d || !n || (n & 15)) return IAMF_HIP_ERR_BAD_ARG;
  memcpy(d, h, n);
  return IAMF_HIP_OK;
}
/* the device unpacker's reads and writes, byte for byte, so that ASan sees the group's raw rows and layout */
int iamf_hip_lpcm_unpack(const iamf_hip_lpcm_layout *lay, const void *d_raw, int64_t raw_stride, const int32_t *fc, int64_t fcs,
                         float *out, int64_t out_stride, int32_t n, void *st) {
  (void)st;
  if (!lay || !d_raw || !fc || !out || n <= 0 || lay->channels <= 0 || lay->channels > IAMF_HIP_LPCM_MAX_CHANNELS) return IAMF_HIP_ERR_BAD_ARG;
  for (int s = 0; s < n; ++s)
    for (int c = 0; c < lay->channels; ++c)
      for (int i = 0; i < fc[s * fcs + 1]; ++i) {
        float v = 0.f;
        if (lay->src_offset[c] >= 0) {
          const uint8_t *p = (const uint8_t *)d_raw + (int64_t)s * raw_stride + lay->src_offset[c] + (int64_t)(fc[s * fcs] + i) * lay->src_step[c];
          int acc = 0;
          for (int k = 0; k < lay->sample_bytes; ++k) acc += p[k];
          v = (float)acc;
        }
        out[(int64_t)s * out_stride + (int64_t)c * lay->frame_size + i] = v;
      }
  return IAMF_HIP_OK;
}
/* the fused LPCM entry: touches, byte for byte, the samples the render kernel would read from the packet rows */
int iamf_hip_batch_render_lpcm_range(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *a, int32_t s0,
                                     int32_t cnt) {
  int n = a->n_samples ? a->n_samples : a->n_frames * b->cfg.frame_size, r = 0;
  if (!in || !in->d_raw || a->d_in || s0 < 0 || cnt <= 0 || s0 + cnt > b->cfg.n_streams || in->first_sample < 0 ||
      (in->first_sample > 0 && a->n_frames != 1) || in->first_sample + (a->n_samples ? a->n_samples : b->cfg.frame_size) > b->cfg.frame_size)
    return IAMF_HIP_ERR_BAD_ARG;
  for (int s = s0; s < s0 + cnt; ++s) {
    int acc = 0;
    for (int f = 0; f < a->n_frames; ++f)
      for (int c = 0; c < in->layout.channels; ++c)
        for (int i = 0; i < (a->n_samples ? a->n_samples : b->cfg.frame_size) && in->layout.src_offset[c] >= 0; ++i) {
          const volatile uint8_t *p = (const volatile uint8_t *)in->d_raw + (int64_t)s * in->raw_stream_stride + (int64_t)f * in->raw_frame_stride +
                                      in->layout.src_offset[c] + (int64_t)(in->first_sample + i) * in->layout.src_step[c];
          for (int k = 0; k < in->layout.sample_bytes; ++k) acc += p[k];
        }
    (void)acc;
    r = emit(b, s, a->d_pcm, a->pcm_stream_stride_bytes, n);
    if (r < 0) return r;
  }
  return r;
}
int iamf_hip_batch_render_lpcm(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *a) {
  return iamf_hip_batch_render_lpcm_range(b, in, a, 0, b->cfg.n_streams);
}
int iamf_hip_resampler_create(int ns, int ch, int in, int out, iamf_hip_resampler **r) {
  (void)ns;
  *r = (iamf_hip_resampler *)calloc(1, sizeof(**r));
  (*r)->ch = ch; (*r)->in = in; (*r)->out = out;
  return 0;
}
void iamf_hip_resampler_destroy(iamf_hip_resampler *r) { free(r); }
int iamf_hip_resampler_out_capacity(const iamf_hip_resampler *r, int n) { return (int)((int64_t)n * r->out / r->in) + 2; }
int iamf_hip_resampler_flush_capacity(const iamf_hip_resampler *r) { (void)r; return 64; }
int iamf_hip_resampler_process(iamf_hip_resampler *r, const float *in, int64_t iss, int n, float *out, int64_t oss, void *st) {
  (void)in; (void)iss; (void)st;
  int m = (int)((int64_t)n * r->out / r->in);
  if ((int64_t)m * r->ch > oss) return IAMF_HIP_ERR_BAD_ARG;
  memset(out, 0, sizeof(float) * (size_t)m * r->ch);
  return m;
}
int iamf_hip_resampler_flush(iamf_hip_resampler *r, float *out, int64_t oss, void *st) {
  (void)r; (void)out; (void)oss; (void)st; return 0;
}
int iamf_hip_resampler_process_range(iamf_hip_resampler *r, const float *in, int64_t iss, int n, float *out, int64_t oss, void *st,
                                     int32_t s0, int32_t cnt) {
  (void)in; (void)iss; (void)st;
  int m = (int)((int64_t)n * r->out / r->in);
  if ((int64_t)m * r->ch > oss) return IAMF_HIP_ERR_BAD_ARG;
  for (int s = s0; s < s0 + cnt; ++s) memset(out + (int64_t)s * oss, 0, sizeof(float) * (size_t)m * r->ch); /* the rows a kernel would write */
  return m;
}
int iamf_hip_resampler_flush_range(iamf_hip_resampler *r, float *out, int64_t oss, void *st, int32_t s0, int32_t cnt) {
  (void)r; (void)out; (void)oss; (void)st; (void)s0; (void)cnt; return 0;
}
int iamf_hip_resampler_same_state(const iamf_hip_resampler *r, int32_t a, int32_t b) { (void)r; (void)a; (void)b; return 1; }
void iamf_hip_dmx_state_init(iamf_hip_dmx_state *s) { memset(s, 0, sizeof(*s)); }
int iamf_hip_dmx_set_mode_weight(iamf_hip_dmx_state *s, int mode, int w) { (void)w; s->mode = mode; return 0; }
void iamf_hip_dmx_coefficients(const iamf_hip_dmx_state *s, float *c) { (void)s; memset(c, 0, 5 * sizeof(float)); }
int iamf_hip_dmx_valid(int in, int out) { return in > out && in >= 0 && out >= 0 && in < 9 && out < 9; }
void iamf_hip_demix_state_init(iamf_hip_demix_state *s) { memset(s, 0, sizeof(*s)); }
int iamf_hip_demix_set_info(iamf_hip_demix_state *s, int mode, int w) { (void)w; s->mode = mode; return 0; }
void iamf_hip_demix_frame_fill(iamf_hip_demix_state *s, int n, const int32_t *ch, const float *g, iamf_hip_demix_frame *f) {
  (void)s; memset(f, 0, sizeof(*f));
  f->n_recon = n;
  for (int i = 0; i < n && i < 12; ++i) { f->recon_ch[i] = ch[i]; f->recon_cur[i] = g[i]; }
}
