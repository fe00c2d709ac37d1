/*
 * oracle/iamf_oracle_resample.c — CPU restatement of the reference's sample-rate converter
 * (src/iamf_dec/resample.c, a speexdsp derivative) at quality 4, float path.
 * TEST INFRASTRUCTURE ONLY (see iamf_oracle.h).
 *
 * The reference feeds the converter in 160-sample pieces through a per-channel memory
 * (resample.c:930-972, 786-811); because every piece is consumed completely whenever the output
 * buffer is not the limit, that is the same as running over one continuous stream, which is how
 * it is written here: output k reads N consecutive inputs starting at position pos, with
 * (pos, frac) advanced by (int_advance, frac_advance) per output (resample.c:296-303).
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../iac_amd/data/resample_window_q4.h"
#include "iamf_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct orc_resampler {
  int ch, in_rate, out_rate;
  unsigned num, den, filt_len, oversample, int_adv, frac_adv;
  int direct;
  float cutoff;
  float *table;
  float *mem;   /* [ch][filt_len - 1 + cap] : history then the samples of the current call */
  int cap;
  int last_sample; /* same for every channel (they are fed identically) */
  unsigned frac;
};

/* resample.c:194-214: cubic interpolation of the window table; float/double mix as written */
static double window_at(float x) {
  float y, frac;
  double interp[4];
  int ind;
  y = x * IAMF_RS_Q4_WINDOW_OVERSAMPLE;
  ind = (int)floor(y);
  frac = (y - ind);
  interp[3] = -0.1666666667 * frac + 0.1666666667 * (frac * frac * frac);
  interp[2] = frac + 0.5 * (frac * frac) - 0.5 * (frac * frac * frac);
  interp[0] = -0.3333333333 * frac + 0.5 * (frac * frac) - 0.1666666667 * (frac * frac * frac);
  interp[1] = 1.f - interp[3] - interp[2] - interp[0];
  return interp[0] * iamf_rs_q4_window[ind] + interp[1] * iamf_rs_q4_window[ind + 1] +
         interp[2] * iamf_rs_q4_window[ind + 2] + interp[3] * iamf_rs_q4_window[ind + 3];
}

/* resample.c:218-231 */
static float sinc_at(float cutoff, float x, int N) {
  float xx = x * cutoff;
  if (fabs(x) < 1e-6)
    return cutoff;
  else if (fabs(x) > .5 * N)
    return 0;
  return cutoff * sin(M_PI * xx) / (M_PI * xx) * window_at(fabs(2. * x / N));
}

static unsigned gcd_u(unsigned a, unsigned b) {
  while (b) {
    unsigned t = a;
    a = b;
    b = t % b;
  }
  return a;
}

/* resample.c:527-611 (filter design) + :1016-1046 (rate fraction) + :1098-1119 (skip_zeros) */
orc_resampler *orc_resampler_open(int channels, int in_rate, int out_rate, int quality) {
  orc_resampler *r;
  unsigned g;
  if (quality != 4 || channels <= 0 || in_rate <= 0 || out_rate <= 0) return 0;
  r = (orc_resampler *)calloc(1, sizeof(*r));
  r->ch = channels;
  r->in_rate = in_rate;
  r->out_rate = out_rate;
  g = gcd_u((unsigned)in_rate, (unsigned)out_rate);
  r->num = (unsigned)in_rate / g;
  r->den = (unsigned)out_rate / g;
  r->int_adv = r->num / r->den;
  r->frac_adv = r->num % r->den;
  r->oversample = IAMF_RS_Q4_OVERSAMPLE;
  r->filt_len = IAMF_RS_Q4_BASE_LENGTH;
  if (r->num > r->den) { /* down-sampling */
    r->cutoff = IAMF_RS_Q4_DOWN_BW * r->den / r->num;
    r->filt_len = (unsigned)((unsigned long long)r->filt_len * r->num / r->den);
    r->filt_len = ((r->filt_len - 1) & (~0x7U)) + 8;
    if (2 * r->den < r->num) r->oversample >>= 1;
    if (4 * r->den < r->num) r->oversample >>= 1;
    if (8 * r->den < r->num) r->oversample >>= 1;
    if (16 * r->den < r->num) r->oversample >>= 1;
    if (r->oversample < 1) r->oversample = 1;
  } else {
    r->cutoff = IAMF_RS_Q4_UP_BW;
  }
  r->direct = r->filt_len * r->den <= r->filt_len * r->oversample + 8;
  if (r->direct) {
    r->table = (float *)malloc(sizeof(float) * r->filt_len * r->den);
    for (unsigned i = 0; i < r->den; i++)
      for (int j = 0; j < (int)r->filt_len; j++)
        r->table[i * r->filt_len + j] =
            sinc_at(r->cutoff, ((j - (int)r->filt_len / 2 + 1) - ((float)i) / r->den), (int)r->filt_len);
  } else {
    r->table = (float *)malloc(sizeof(float) * (r->filt_len * r->oversample + 8));
    for (int i = -4; i < (int)(r->oversample * r->filt_len + 4); i++)
      r->table[i + 4] = sinc_at(r->cutoff, (i / (float)r->oversample - r->filt_len / 2), (int)r->filt_len);
  }
  r->cap = 0;
  r->mem = 0;
  r->last_sample = (int)(r->filt_len / 2); /* speex_resampler_skip_zeros, IAMF_decoder.c:1902 */
  r->frac = 0;
  return r;
}

void orc_resampler_close(orc_resampler *r) {
  if (!r) return;
  free(r->table);
  free(r->mem);
  free(r);
}

static void ensure_cap(orc_resampler *r, int ns) {
  if (ns > r->cap) {
    const int hist = (int)r->filt_len - 1;
    float *m = (float *)calloc((size_t)r->ch * (hist + ns), sizeof(float));
    if (r->mem)
      for (int c = 0; c < r->ch; ++c) memcpy(m + (size_t)c * (hist + ns), r->mem + (size_t)c * (hist + r->cap), sizeof(float) * hist);
    free(r->mem);
    r->mem = m;
    r->cap = ns;
  }
}

/* resample.c:246-256 */
static void cubic_coef(float frac, float interp[4]) {
  interp[0] = -0.16667f * frac + 0.16667f * frac * frac * frac;
  interp[1] = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
  interp[3] = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
  interp[2] = 1. - interp[0] - interp[1] - interp[3];
}

/* one channel over [history | ns new samples]; resample.c:258-308 (direct) / :356-412 (interpolated) */
static int run_channel(const orc_resampler *r, const float *mem, int in_len, float *out, int out_len,
                       int *last_sample_io, unsigned *frac_io) {
  const int N = (int)r->filt_len;
  int last_sample = *last_sample_io, n = 0;
  unsigned frac = *frac_io;
  while (!(last_sample >= in_len || n >= out_len)) {
    const float *iptr = &mem[last_sample];
    float sum;
    if (r->direct) {
      const float *sinct = &r->table[frac * N];
      sum = 0;
      for (int j = 0; j < N; j++) sum += sinct[j] * iptr[j];
    } else {
      const int offset = frac * r->oversample / r->den;
      const float fr = ((float)((frac * r->oversample) % r->den)) / r->den;
      float interp[4];
      float accum[4] = {0, 0, 0, 0};
      for (int j = 0; j < N; j++) {
        const float cur = iptr[j];
        accum[0] += cur * r->table[4 + (j + 1) * r->oversample - offset - 2];
        accum[1] += cur * r->table[4 + (j + 1) * r->oversample - offset - 1];
        accum[2] += cur * r->table[4 + (j + 1) * r->oversample - offset];
        accum[3] += cur * r->table[4 + (j + 1) * r->oversample - offset + 1];
      }
      cubic_coef(fr, interp);
      sum = interp[0] * accum[0] + interp[1] * accum[1] + interp[2] * accum[2] + interp[3] * accum[3];
    }
    /* resample.c:84,959: the float output is clamped to [-1, 1] */
    out[n++] = (float)(sum < -1.0 ? -1.0 : (sum > 1.0 ? 1.0 : sum));
    last_sample += (int)r->int_adv;
    frac += r->frac_adv;
    if (frac >= r->den) {
      frac -= r->den;
      last_sample++;
    }
  }
  *last_sample_io = last_sample;
  *frac_io = frac;
  return n;
}

static int process(orc_resampler *r, const float *in, int ns, float *out, int out_cap, int out_len) {
  const int hist = (int)r->filt_len - 1;
  int n = 0, ls = r->last_sample, consumed;
  unsigned fr = r->frac;
  ensure_cap(r, ns);
  for (int c = 0; c < r->ch; ++c) {
    float *m = r->mem + (size_t)c * (hist + r->cap);
    ls = r->last_sample;
    fr = r->frac;
    if (in)
      memcpy(m + hist, in + (size_t)c * ns, sizeof(float) * ns);
    else
      memset(m + hist, 0, sizeof(float) * ns);
    n = run_channel(r, m, ns, out + (size_t)c * out_cap, out_len, &ls, &fr);
    /* resample.c:801-809: drop what was consumed, keep N-1 samples of history */
    consumed = ls < ns ? ls : ns;
    memmove(m, m + consumed, sizeof(float) * hist);
  }
  consumed = ls < ns ? ls : ns;
  r->last_sample = ls - consumed;
  r->frac = fr;
  return n;
}

/* iamf_resample, IAMF_decoder.c:3223-3248: out capacity ns*(out/in + 1) with INTEGER division.
 * out is planar with row stride = that capacity; returns samples per channel. */
int orc_resample(orc_resampler *r, const float *in, float *out, int ns) {
  const int cap = ns * (r->out_rate / r->in_rate + 1);
  return process(r, in, ns, out, cap, cap);
}

int orc_resample_out_capacity(const orc_resampler *r, int ns) { return ns * (r->out_rate / r->in_rate + 1); }

/* end of stream (rest_flag == 2, IAMF_decoder.c:3227-3232): input latency zeros in, at most
 * output latency samples out (resample.c:1098-1105); out row stride = that output latency */
int orc_resample_flush(orc_resampler *r, float *out) {
  const int in_lat = (int)(r->filt_len / 2);
  const int out_lat = (int)(((r->filt_len / 2) * r->den + (r->num >> 1)) / r->num);
  return process(r, 0, in_lat, out, out_lat, out_lat);
}

int orc_resample_flush_capacity(const orc_resampler *r) {
  return (int)(((r->filt_len / 2) * r->den + (r->num >> 1)) / r->num);
}
