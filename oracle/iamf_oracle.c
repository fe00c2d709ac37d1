/*
 * oracle/iamf_oracle.c — CPU restatement of the reference's post-decode rendering path.
 * TEST INFRASTRUCTURE ONLY (see iamf_oracle.h).  Plain scalar C, f32 arithmetic in the
 * reference's operation order; compile with -ffp-contract=off.
 */
#include "iamf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * rendering-matrix blob
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t kind, in_id, out_id;
  int32_t channels, lfe1, lfe2, m, n;
  uint32_t offset;
} blob_entry;

static struct {
  int n;
  blob_entry *ents;
  float *data;
} g_tab;

int orc_tables_load(const char *path) {
  FILE *f = fopen(path, "rb");
  char magic[8];
  uint32_t n;
  long pos, end;
  if (!f) return -1;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "IARDRTB1", 8) || fread(&n, 4, 1, f) != 1) {
    fclose(f);
    return -2;
  }
  free(g_tab.ents);
  free(g_tab.data);
  g_tab.ents = (blob_entry *)malloc(sizeof(blob_entry) * n);
  if (fread(g_tab.ents, sizeof(blob_entry), n, f) != n) {
    fclose(f);
    return -3;
  }
  pos = ftell(f);
  fseek(f, 0, SEEK_END);
  end = ftell(f);
  fseek(f, pos, SEEK_SET);
  g_tab.data = (float *)malloc((size_t)(end - pos));
  if (fread(g_tab.data, 1, (size_t)(end - pos), f) != (size_t)(end - pos)) {
    fclose(f);
    return -4;
  }
  fclose(f);
  g_tab.n = (int)n;
  return 0;
}

int orc_tables_count(void) { return g_tab.n; }

int orc_tables_entry(int idx, orc_matrix *out) {
  const blob_entry *e;
  if (idx < 0 || idx >= g_tab.n) return -1;
  e = &g_tab.ents[idx];
  out->kind = (int)e->kind;
  out->in_id = (int)e->in_id;
  out->out_id = (int)e->out_id;
  out->channels = e->channels;
  out->lfe1 = e->lfe1;
  out->lfe2 = e->lfe2;
  out->m = e->m;
  out->n = e->n;
  out->mat = g_tab.data + e->offset;
  return 0;
}

static int find_entry(int kind, int in_id, int out_id, orc_matrix *out) {
  /* first match in table order, like the reference's linear searches */
  for (int i = 0; i < g_tab.n; ++i) {
    const blob_entry *e = &g_tab.ents[i];
    if ((int)e->kind == kind && (int)e->in_id == in_id && (int)e->out_id == out_id)
      return orc_tables_entry(i, out);
  }
  return -1;
}

int orc_get_h2m(int order, int out_id, orc_matrix *out) { return find_entry(0, order, out_id, out); }
int orc_get_m2m(int in_id, int out_id, orc_matrix *out) { return find_entry(1, in_id, out_id, out); }

/* ------------------------------------------------------------------------------------------
 * element renderers
 * ---------------------------------------------------------------------------------------- */

/* h2m_rdr.c:1103-1112: per sample, per output n: acc = 0; acc += mat[n*m+k] * in[k] for k
 * ascending (f32 product, then f32 add).  h2m_rdr.c:1114-1150: the n computed feeds are then
 * spread over the output slots so that slot indices lfe1 / lfe2 stay free (the comparison in
 * the reference is against the SOURCE index i), and those slots are zeroed.  Slots at or
 * above the last destination are not touched. */
void orc_render_h2m(const orc_matrix *mx, const float *in, float *out, int ns) {
  const int m = mx->m, n = mx->n;
  int dest[ORC_MAX_CH];
  for (int i = 0; i < n; ++i) {
    int d = i;
    if (mx->lfe1 >= 0 && mx->lfe1 <= i) ++d;
    if (mx->lfe2 >= 0 && mx->lfe2 <= i) ++d;
    dest[i] = (mx->lfe1 >= 0 || mx->lfe2 >= 0) ? d : i;
  }
  for (int o = 0; o < n; ++o) {
    const float *row = mx->mat + o * m;
    float *dst = out + (size_t)dest[o] * ns;
    for (int i = 0; i < ns; ++i) {
      float acc = 0.f;
      for (int k = 0; k < m; ++k) {
        float p = row[k] * in[(size_t)k * ns + i];
        acc = acc + p;
      }
      dst[i] = acc;
    }
  }
  if (mx->lfe1 >= 0) memset(out + (size_t)mx->lfe1 * ns, 0, sizeof(float) * ns);
  if (mx->lfe2 >= 0) memset(out + (size_t)mx->lfe2 * ns, 0, sizeof(float) * ns);
}

/* m2m_rdr.c:1826-1837: out[n] = sum over inputs k ascending of mat[k*n_size+n] * in[k]. */
void orc_render_m2m(const orc_matrix *mx, const float *in, float *out, int ns) {
  const int m = mx->m, n = mx->n;
  for (int o = 0; o < n; ++o) {
    float *dst = out + (size_t)o * ns;
    for (int i = 0; i < ns; ++i) {
      float acc = 0.f;
      for (int k = 0; k < m; ++k) {
        float p = mx->mat[k * n + o] * in[(size_t)k * ns + i];
        acc = acc + p;
      }
      dst[i] = acc;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * gains, mixer, loudness
 * ---------------------------------------------------------------------------------------- */

/* IAMF_decoder.c:1392-1397: a constant gain is applied only when it is not 1 and positive. */
void orc_frame_gain_const(float *data, int channels, int ns, float gain) {
  if (gain != 1.f && gain > 0.f) {
    const int count = channels * ns;
    for (int i = 0; i < count; ++i) data[i] = data[i] * gain;
  }
}

/* IAMF_decoder.c:1401-1405 */
void orc_frame_gain_ramp(float *data, int channels, int ns, const float *gains) {
  for (int c = 0; c < channels; ++c)
    for (int i = 0; i < ns; ++i) data[(size_t)c * ns + i] = data[(size_t)c * ns + i] * gains[i];
}

/* IAMF_decoder.c:639-645: g = s + (e - s) * i / d, all f32, i and d converted from int */
void orc_mix_gain_linear(float s, float e, int d, int o, int l, float *g) {
  for (int i = o, k = 0; i < o + l; ++i, ++k) {
    float t = (e - s) * (float)i;
    t = t / (float)d;
    g[k] = s + t;
  }
}

/* IAMF_decoder.c:647-664: quadratic Bezier; `a` solved in double and narrowed to f32, the
 * polynomial mixes an f32 coefficient, a double square, and f32 products summed in double. */
void orc_mix_gain_quad(float s, float e, int d, float c, int ct, int o, int l, float *g) {
  int64_t alpha = (int64_t)d - 2 * (int64_t)ct;
  float a = 1.0f;
  for (int i = o, k = 0; i < o + l; ++i, ++k) {
    if (alpha) {
      double disc = pow((double)ct, 2.0) + (double)(alpha * (int64_t)i);
      a = (float)((sqrt(disc) - (double)ct) / (double)alpha);
    } else {
      a = (float)i;
      a = a / (float)(2 * ct);
    }
    {
      float k2 = s + e - 2 * c;
      float lin = 2 * a * (c - s);
      double v = (double)k2 * pow((double)a, 2.0) + (double)lin + (double)s;
      g[k] = (float)v;
    }
  }
}

/* IAMF_decoder.c:2719-2730: memset 0, then += each element frame in order */
void orc_mix(float *dst, const float *const *frames, int n_elements, int channels, int ns) {
  const int count = channels * ns;
  for (int i = 0; i < count; ++i) dst[i] = 0.f;
  for (int e = 0; e < n_elements; ++e)
    for (int i = 0; i < count; ++i) dst[i] = dst[i] + frames[e][i];
}

/* IAMF_decoder.c:3206-3221 */
void orc_loudness(float *block, int ns, int channels, float gain) {
  if (!ns || gain == 1.0f) return;
  for (int i = 0; i < channels * ns; ++i) block[i] = block[i] * gain;
}

/* fixedp11_5.c:72 */
float orc_db2lin(float db) { return powf(10.0f, 0.05f * db); }

/* ------------------------------------------------------------------------------------------
 * peak limiter
 * ---------------------------------------------------------------------------------------- */

/* audio_effect_peak_limiter.c:73-92 + :211-235 */
void orc_limiter_init(orc_limiter *lim, float threshold_db, int rate, int channels,
                      float atk_sec, float rel_sec, int delay) {
  memset(lim, 0, sizeof(*lim));
  lim->g = 1.0f;
  lim->gs = -1.0f;
  lim->ge = -1.0f;
  lim->tc = -1.0f;
  lim->maxpos = -1;
  lim->thr = (float)pow(10, threshold_db / 20); /* f32 divide, double pow, narrowed */
  lim->atk = atk_sec;
  lim->rel = rel_sec;
  lim->inc = (float)1 / (float)rate;
  lim->ch = channels;
  lim->delay = delay;
  lim->pad = delay;
}

/* audio_effect_peak_limiter.c:267-271 */
static float ease(float x) {
  if (1.0 < x) return 1.0f;
  if (x < 0) return 0.0f;
  return 1.0f - powf(x - 1, 2.0);
}

/* audio_effect_peak_limiter.c:237-265 */
static float limiter_gain_step(orc_limiter *l, float peak) {
  if (l->tc != -1 && l->tc < l->atk) {
    float r;
    l->tc += l->inc;
    r = ease(l->tc / l->atk);
    l->g = l->gs - r * (l->gs - l->ge);
  } else if (l->tc != -1 && l->tc < l->rel + l->atk) {
    float r;
    l->tc += l->inc;
    r = ease((l->tc - l->atk) / l->rel);
    l->g = l->ge + r * (1.0f - l->ge);
  } else {
    l->g = 1.0;
  }
  if (peak * l->g > l->thr) {
    l->gs = l->g;
    l->ge = l->thr / peak;
    l->tc = 0.0f;
  }
  return l->g;
}

/* audio_effect_peak_limiter.c:94-204.  Per sample: window maximum of the |x| ring BEFORE the
 * new sample enters, gain step, emit delayed sample * gain, store the new sample and its
 * cross-channel |x| maximum.  The reference's peak_pos cache (:117-133,171-176) always yields
 * the true ring maximum; it is kept here (same rescan order) so the CPU cost is comparable. */
int orc_limiter_process(orc_limiter *lim, const float *in, float *out, int ns) {
  const int D = lim->delay;
  if (!in) return 0;
  for (int k = 0; k < ns; ++k) {
    const int idx = (k + lim->head) % D;
    float peak = 0.f, gain, pm = 0.f;
    if (lim->maxpos < 0) {
      for (int i = 0; i < D; ++i) {
        const int p = (i + k + lim->head) % D;
        if (lim->pk[p] > peak) {
          peak = lim->pk[p];
          lim->maxpos = p;
        }
      }
    } else {
      peak = lim->pk[lim->maxpos];
    }
    gain = limiter_gain_step(lim, peak);
    for (int c = 0; c < lim->ch; ++c) {
      float a;
      out[(size_t)c * ns + k] = lim->dl[c][idx] * gain;
      lim->dl[c][idx] = in[(size_t)c * ns + k];
      a = (float)fabs(lim->dl[c][idx]);
      if (a > pm) pm = a;
    }
    if (lim->maxpos == idx)
      lim->maxpos = -1;
    else if (lim->maxpos < 0 || lim->pk[lim->maxpos] < pm)
      lim->maxpos = idx;
    lim->pk[idx] = pm;
  }
  lim->head = (lim->head + ns) % D;

  if (!lim->started) { /* :185-201 the first `delay` outputs are dropped once */
    if (lim->pad >= ns) {
      lim->pad -= ns;
      ns = 0;
    } else {
      int w = 0;
      for (int c = 0; c < lim->ch; ++c)
        for (int k = lim->pad; k < ns; ++k) out[w++] = out[(size_t)c * ns + k];
      ns -= lim->pad;
      lim->pad = 0;
      lim->started = 1;
    }
  }
  return ns;
}

/* ------------------------------------------------------------------------------------------
 * float -> PCM
 * ---------------------------------------------------------------------------------------- */

/* IAMF_decoder.c:100-119: scale, clamp (max with low bound then min with high bound), lrintf */
static int32_t to_pcm(float x, float scale, float lo, float hi) {
  x = x * scale;
  x = x > lo ? x : lo;
  x = x < hi ? x : hi;
  return (int32_t)lrintf(x);
}

/* IAMF_decoder.c:121-167 */
void orc_pack(void *dst, const float *src, int ns, int channels, int bit_depth, int stride) {
  if (bit_depth == 16) {
    int16_t *d = (int16_t *)dst;
    memset(d, 0, (size_t)2 * ns * stride);
    for (int c = 0; c < channels; ++c)
      for (int i = 0; i < ns; ++i)
        d[(size_t)i * stride + c] =
            src ? (int16_t)to_pcm(src[(size_t)ns * c + i], 32768.f, -32768.f, 32767.f) : 0;
  } else if (bit_depth == 24) {
    uint8_t *d = (uint8_t *)dst;
    memset(d, 0, (size_t)3 * ns * stride);
    for (int c = 0; c < channels; ++c)
      for (int i = 0; i < ns; ++i) {
        int32_t v = src ? to_pcm(src[(size_t)ns * c + i], 8388608.f, -8388608.f, 8388607.f) : 0;
        uint8_t *p = d + ((size_t)i * stride + c) * 3;
        p[0] = (uint8_t)(v & 0xff);
        p[1] = (uint8_t)((v >> 8) & 0xff);
        p[2] = (uint8_t)(((v >> 16) & 0x7f) | ((v >> 24) & 0x80));
      }
  } else if (bit_depth == 32) {
    int32_t *d = (int32_t *)dst;
    memset(d, 0, (size_t)4 * ns * stride);
    for (int c = 0; c < channels; ++c)
      for (int i = 0; i < ns; ++i)
        d[(size_t)i * stride + c] =
            src ? to_pcm(src[(size_t)ns * c + i], 2147483648.f, -2147483648.f, 2147483647.f) : 0;
  }
}

/* ------------------------------------------------------------------------------------------
 * parametric down-mixer
 * ---------------------------------------------------------------------------------------- */

/* channel ids follow reference IAMF_types.h:61-90 (L5/R5 alias L7/R7) */
enum {
  CH_NONE = 0, CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR,
  CH_HBL, CH_HBR, CH_MONO, CH_L2, CH_R2, CH_TL, CH_TR, CH_L3, CH_R3, CH_SL5, CH_SR5, CH_HL,
  CH_HR, CH_COUNT, CH_L5 = CH_L7, CH_R5 = CH_R7
};

/* playback channel order per layout: reference IAMF_utils.c:111-133 */
static const int k_layout_count[10] = {1, 2, 6, 8, 10, 8, 10, 12, 6, 2};
static const int k_layout_ch[10][12] = {
    {CH_MONO},
    {CH_L2, CH_R2},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5, CH_HL, CH_HR},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5, CH_HFL, CH_HFR, CH_HBL, CH_HBR},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HL, CH_HR},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR, CH_HBL, CH_HBR},
    {CH_L3, CH_R3, CH_C, CH_LFE, CH_TL, CH_TR},
    {CH_L2, CH_R2},
};
/* surround / top counts: reference IAMF_utils.c:157-161 */
static const int k_layout_surround[10] = {1, 2, 5, 5, 5, 7, 7, 7, 3, 2};
static const int k_layout_top[10] = {0, 0, 0, 2, 4, 0, 2, 4, 2, 0};

/* scale selectors for the two-term sums of downmix_renderer.c:65-75,164-171 */
enum { SC_CONST, SC_ALPHA, SC_BETA, SC_GAMMA, SC_DELTA, SC_GAMMA_W };
typedef struct {
  int dst, src0, sc0, src1, sc1;
  float k0, k1;
} dmx_rule;
static const dmx_rule k_rules[] = {
    {CH_MONO, CH_R2, SC_CONST, CH_L2, SC_CONST, 0.5f, 0.5f},
    {CH_L2, CH_L3, SC_CONST, CH_C, SC_CONST, 1.f, (float)0.707},
    {CH_R2, CH_R3, SC_CONST, CH_C, SC_CONST, 1.f, (float)0.707},
    {CH_TL, CH_HL, SC_CONST, CH_SL5, SC_GAMMA_W, 1.f, 0.f},
    {CH_TR, CH_HR, SC_CONST, CH_SR5, SC_GAMMA_W, 1.f, 0.f},
    {CH_L3, CH_L5, SC_CONST, CH_SL5, SC_DELTA, 1.f, 0.f},
    {CH_R3, CH_R5, SC_CONST, CH_SR5, SC_DELTA, 1.f, 0.f},
    {CH_SL5, CH_SL7, SC_ALPHA, CH_BL7, SC_BETA, 0.f, 0.f},
    {CH_SR5, CH_SR7, SC_ALPHA, CH_BR7, SC_BETA, 0.f, 0.f},
    {CH_HL, CH_HFL, SC_CONST, CH_HBL, SC_GAMMA, 1.f, 0.f},
    {CH_HR, CH_HFR, SC_CONST, CH_HBR, SC_GAMMA, 1.f, 0.f},
};

/* IAMF_utils.c:236-240 and fixedp11_5.c:81-82 */
static const struct { float a, b, g, d; int woff; } k_mix[7] = {
    {1.0, 1.0, (float)0.707, (float)0.707, -1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, -1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, -1}, {0, 0, 0, 0, 0},
    {1.0, 1.0, (float)0.707, (float)0.707, 1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, 1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, 1}};
static const float k_w[11] = {0.0, (float)0.0179, (float)0.0391, (float)0.0658, (float)0.1038, 0.25,
                              (float)0.3962, (float)0.4342, (float)0.4609, (float)0.4821, 0.5};

struct orc_downmixer {
  int mode, w_idx;
  int n_in, n_out;
  int ch_in[12], ch_out[12];
  int is_input[CH_COUNT];
  const float *data[CH_COUNT];
  float alpha, beta, gamma, delta, gamma_w;
  int woff;
};

/* downmix_renderer.c:131-178 */
orc_downmixer *orc_dmx_open(int in, int out) {
  orc_downmixer *d;
  if (in == out || in < 0 || in >= 9 || out < 0 || out >= 9) return 0; /* binaural (9) is not valid */
  if (k_layout_top[in] && !k_layout_top[out]) return 0;
  if (k_layout_surround[in] < k_layout_surround[out] || k_layout_top[in] < k_layout_top[out]) return 0;
  d = (orc_downmixer *)calloc(1, sizeof(*d));
  d->n_in = k_layout_count[in];
  d->n_out = k_layout_count[out];
  for (int i = 0; i < d->n_in; ++i) {
    d->ch_in[i] = k_layout_ch[in][i];
    d->is_input[d->ch_in[i]] = 1;
  }
  for (int i = 0; i < d->n_out; ++i) d->ch_out[i] = k_layout_ch[out][i];
  d->mode = -1;
  d->w_idx = -1;
  return d;
}

void orc_dmx_close(orc_downmixer *d) { free(d); }
int orc_dmx_w_idx(const orc_downmixer *d) { return d->w_idx; }

static float w_of(int idx) { return idx < 0 ? k_w[0] : idx > 10 ? k_w[10] : k_w[idx]; }

/* downmix_renderer.c:180-216 (+ fixedp11_5.c:83-99) */
int orc_dmx_set_mode_weight(orc_downmixer *d, int mode, int w_idx) {
  if (!d || mode < 0 || mode == 3 || mode >= 7) return -1;
  if (d->mode != mode) {
    d->mode = mode;
    d->alpha = k_mix[mode].a;
    d->beta = k_mix[mode].b;
    d->gamma = k_mix[mode].g;
    d->delta = k_mix[mode].d;
    d->woff = k_mix[mode].woff;
  }
  if (w_idx < 0 || w_idx > 10) {
    int nw = d->woff > 0 ? (d->w_idx + 1 < 10 ? d->w_idx + 1 : 10) : (d->w_idx - 1 > 0 ? d->w_idx - 1 : 0);
    d->w_idx = nw;
    d->gamma_w = d->gamma * w_of(nw);
  } else if (d->w_idx != w_idx) {
    d->w_idx = w_idx;
    d->gamma_w = d->gamma * w_of(w_idx);
  }
  return 0;
}

static float dmx_scale(const orc_downmixer *d, int sc, float k) {
  switch (sc) {
    case SC_ALPHA: return d->alpha;
    case SC_BETA: return d->beta;
    case SC_GAMMA: return d->gamma;
    case SC_DELTA: return d->delta;
    case SC_GAMMA_W: return d->gamma_w;
    default: return k;
  }
}

/* downmix_renderer.c:115-129: input channels are taken as they are; a derived channel is the
 * f32 sum (from 0) of its two scaled sources, each source evaluated recursively. */
static float dmx_eval(const orc_downmixer *d, int c, int i) {
  if (d->data[c]) return d->data[c][i];
  for (unsigned r = 0; r < sizeof(k_rules) / sizeof(k_rules[0]); ++r) {
    if (k_rules[r].dst == c) {
      float sum = 0.f;
      sum = sum + dmx_eval(d, k_rules[r].src0, i) * dmx_scale(d, k_rules[r].sc0, k_rules[r].k0);
      sum = sum + dmx_eval(d, k_rules[r].src1, i) * dmx_scale(d, k_rules[r].sc1, k_rules[r].k1);
      return sum;
    }
  }
  return 0.f;
}

/* downmix_renderer.c:218-242 */
int orc_dmx_downmix(orc_downmixer *d, const float *in, float *out, int s, int duration, int size) {
  int e;
  if (!d || !in || !out || !size || s >= size) return -1;
  memset(d->data, 0, sizeof(d->data));
  for (int i = 0; i < d->n_in; ++i) d->data[d->ch_in[i]] = in + (size_t)size * i;
  e = s + duration;
  if (e > size) e = size;
  for (int o = 0; o < d->n_out; ++o)
    for (int j = s; j < e; ++j) out[(size_t)size * o + j] = dmx_eval(d, d->ch_out[o], j);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * one stream: render -> gains -> mix -> loudness -> limiter -> pack
 * ---------------------------------------------------------------------------------------- */
int orc_stream_open(orc_stream *s, const orc_matrix *mx, int out_channels, float element_gain,
                    float output_gain, int loudness_on, float loudness_gain, int limiter_on,
                    float threshold_db, int rate, int bit_depth, int max_ns) {
  memset(s, 0, sizeof(*s));
  s->mx = *mx;
  s->out_channels = out_channels;
  s->element_gain = element_gain;
  s->output_gain = output_gain;
  s->loudness_on = loudness_on;
  s->loudness_gain = loudness_gain;
  s->limiter_on = limiter_on;
  s->bit_depth = bit_depth;
  s->max_ns = max_ns;
  s->buf_a = (float *)calloc((size_t)ORC_MAX_CH * max_ns, sizeof(float));
  s->buf_b = (float *)calloc((size_t)ORC_MAX_CH * max_ns, sizeof(float));
  if (!s->buf_a || !s->buf_b) return -1;
  /* IAMF_decoder.c:3809-3815: attack 1 ms, release 200 ms, look-ahead 240 */
  if (limiter_on) orc_limiter_init(&s->lim, threshold_db, rate, out_channels, 0.001f, 0.200f, 240);
  return 0;
}

/* IAMF_decoder.c:2630-2633: only if the output layout has an LFE; lfefilter_init(plfe, 120, rate) */
void orc_stream_enable_lfe(orc_stream *s, int rate) {
  if (s->mx.kind != 0 || (s->mx.lfe1 < 0 && s->mx.lfe2 < 0)) return;
  s->lfe_on = 1;
  orc_lfe_init(&s->lfe, 120, (float)rate);
}

void orc_stream_close(orc_stream *s) {
  free(s->buf_a);
  free(s->buf_b);
  s->buf_a = s->buf_b = 0;
}

/* stage order of iamf_decoder_internal_decode, IAMF_decoder.c:3374 (render), :3425-3433
 * (element gain), :3459 (mix), :3463-3469 (output gain), :3480-3484 (loudness), :3486-3490
 * (limiter), :3492-3500 (pack).  Single element; resampling is a separate oracle entry. */
int orc_stream_frame(orc_stream *s, const float *in, int ns, void *pcm) {
  const int ch = s->out_channels;
  const float *one[1];
  float *cur;
  int n = ns;
  memset(s->buf_a, 0, sizeof(float) * (size_t)ch * ns);
  if (s->mx.kind == 0)
    orc_render_h2m_lfe(&s->mx, in, s->buf_a, ns, s->lfe_on ? &s->lfe : 0);
  else
    orc_render_m2m(&s->mx, in, s->buf_a, ns);
  orc_frame_gain_const(s->buf_a, ch, ns, s->element_gain);
  one[0] = s->buf_a;
  orc_mix(s->buf_b, one, 1, ch, ns);
  orc_frame_gain_const(s->buf_b, ch, ns, s->output_gain);
  if (s->loudness_on) orc_loudness(s->buf_b, ns, ch, s->loudness_gain);
  cur = s->buf_b;
  if (s->limiter_on) {
    n = orc_limiter_process(&s->lim, s->buf_b, s->buf_a, ns);
    cur = s->buf_a;
  }
  orc_pack(pcm, cur, n, ch, s->bit_depth, ch);
  return n;
}

/* IAMF_decoder.c:3250-3301 without a resampler: `delay` zero samples go through the limiter */
int orc_stream_flush(orc_stream *s, void *pcm) {
  const int ch = s->out_channels;
  int n;
  if (!s->limiter_on) return 0;
  n = s->lim.delay;
  memset(s->buf_b, 0, sizeof(float) * (size_t)ch * n);
  n = orc_limiter_process(&s->lim, s->buf_b, s->buf_a, n);
  orc_pack(pcm, s->buf_a, n, ch, s->bit_depth, ch);
  return n;
}

/* sizes for callers that allocate the structs from another language (tests via ctypes) */
int orc_sizeof_limiter(void) { return (int)sizeof(orc_limiter); }
int orc_sizeof_stream(void) { return (int)sizeof(orc_stream); }

/* Timed-baseline helper: one whole stream (open, n_frames frames, flush, close) without leaving
 * C, so that callers can run one stream per thread.  Returns sample-frames emitted. */
long orc_stream_run_frames(const orc_matrix *mx, int out_channels, int limiter_on, float threshold_db,
                           int rate, int bit_depth, const float *in, int n_frames, int ns, void *pcm) {
  orc_stream s;
  long total = 0;
  if (orc_stream_open(&s, mx, out_channels, 1.0f, 1.0f, 0, 1.0f, limiter_on, threshold_db, rate,
                      bit_depth, ns) != 0)
    return -1;
  for (int f = 0; f < n_frames; ++f)
    total += orc_stream_frame(&s, in + (size_t)f * mx->m * ns, ns, pcm);
  total += orc_stream_flush(&s, pcm);
  orc_stream_close(&s);
  return total;
}
