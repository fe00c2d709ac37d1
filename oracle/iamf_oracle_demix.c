/* iamf_oracle_demix.c — TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's demixer for
 * scalable channel audio (src/iamf_dec/demixer.c).  Scalar C, -ffp-contract=off.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may use it; the product never links it.
 *
 * PARITY: PINNED — tests/golden/demix.npz holds the outputs of the real demixer_* symbols of
 * oracle/_ref/libiamf_ref.so on tests/demix_cases.py's cases; tests/test_oracle_golden.py checks
 * this file against them bit for bit.
 *
 * Per frame (demixer.c:636-664): gain-up of the listed channels in place (:426-435); the missing
 * channels of the target layout by the de-mix chain (:123-424), each step with the reference's
 * mix of float and double arithmetic; recon-gain smoothing of the listed channels (:447-478);
 * copy-out in playback order.  The first `skip` samples of a frame use the previous mode/weight.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "iamf_oracle.h"

enum {
  CH_INVALID, CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR,
  CH_HBL, CH_HBR, CH_MONO, CH_L2, CH_R2, CH_TL, CH_TR, CH_L3, CH_R3, CH_SL5, CH_SR5, CH_HL,
  CH_HR, CH_COUNT, CH_L5 = CH_L7, CH_R5 = CH_R7
};
enum { MX_S_L, MX_S_R, MX_S5_L, MX_S5_R, MX_T_L, MX_T_R, MX_COUNT }; /* demixer.c:113-121 */

static const int k_count[9] = {1, 2, 6, 8, 10, 8, 10, 12, 6};
static const int k_playback[9][12] = { /* IAMF_utils.c:117-133 */
    {CH_MONO},
    {CH_L2, CH_R2},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5, CH_HL, CH_HR},
    {CH_L5, CH_R5, CH_C, CH_LFE, CH_SL5, CH_SR5, CH_HFL, CH_HFR, CH_HBL, CH_HBR},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HL, CH_HR},
    {CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR, CH_HBL, CH_HBR},
    {CH_L3, CH_R3, CH_C, CH_LFE, CH_TL, CH_TR},
};
/* demixer.c:66-78 (double literals narrowed to float) and fixedp11_5.c:81-82 */
static const struct { float alpha, beta, gamma, delta; int woff; } k_mat[8] = {
    {1.0, 1.0, (float)0.707, (float)0.707, -1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, -1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, -1}, {0, 0, 0, 0, 0},
    {1.0, 1.0, (float)0.707, (float)0.707, 1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, 1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, 1}, {0, 0, 0, 0, 0}};
static const float k_w[11] = {0.0, (float)0.0179, (float)0.0391, (float)0.0658, (float)0.1038, 0.25,
                              (float)0.3962, (float)0.4342, (float)0.4609, (float)0.4821, 0.5};
static float w_of(int idx) { return idx < 0 ? k_w[0] : idx > 10 ? k_w[10] : k_w[idx]; }

struct orc_demixer {
  float *ch[CH_COUNT];
  int w_idx, last_mode, last_w_idx, mode;
  int fs, skip;
  float *hann, *start, *stop, *large;
  float last_sf[CH_COUNT], last_sfavg[CH_COUNT];
  int layout, chs_in[12], chs_out[12], chs_count;
  int gain_ch[12], gain_count;
  float gain[12];
  int re_ch[12], re_count;
  float re_gain[12];
  unsigned re_flags;
};

/* demixer.c:476-529 */
orc_demixer *orc_demixer_open(int frame_size) {
  orc_demixer *d = (orc_demixer *)calloc(1, sizeof(*d));
  int wl = frame_size / 8;
  d->fs = frame_size;
  d->layout = -1;
  d->hann = (float *)malloc(sizeof(float) * (wl > 0 ? wl : 1));
  d->start = (float *)malloc(sizeof(float) * frame_size);
  d->stop = (float *)malloc(sizeof(float) * frame_size);
  d->large = (float *)malloc(sizeof(float) * MX_COUNT * frame_size);
  for (int i = 0; i < wl; ++i) d->hann[i] = (0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(wl - 1))));
  for (int i = 0; i < frame_size; ++i) {
    d->start[i] = 1;
    d->stop[i] = 0;
  }
  for (int i = 0; i < CH_COUNT; ++i) d->last_sf[i] = d->last_sfavg[i] = 1.0;
  return d;
}

void orc_demixer_close(orc_demixer *d) {
  if (!d) return;
  free(d->hann);
  free(d->start);
  free(d->stop);
  free(d->large);
  free(d);
}

/* demixer.c:541-567 */
int orc_demixer_set_frame_offset(orc_demixer *d, unsigned offset) {
  int wl = d->fs / 8, ov = wl / 2, pre = offset % d->fs;
  d->skip = pre;
  if (pre + ov > d->fs) return 0;
  for (int i = 0; i < pre; ++i) {
    d->start[i] = 0;
    d->stop[i] = 1;
  }
  for (int i = pre, j = 0; j < ov; ++i, ++j) {
    d->start[i] = d->hann[j];
    d->stop[i] = d->hann[j + ov];
  }
  for (int i = pre + ov; i < d->fs; ++i) {
    d->start[i] = 1;
    d->stop[i] = 0;
  }
  return 0;
}

int orc_demixer_set_channel_layout(orc_demixer *d, int layout) {
  if (layout < 0 || layout > 8) return -1;
  for (int c = 0; c < k_count[layout]; ++c) d->chs_out[c] = k_playback[layout][c];
  d->layout = layout;
  return 0;
}

int orc_demixer_set_channels_order(orc_demixer *d, const int *chs, int count) {
  memcpy(d->chs_in, chs, sizeof(int) * count);
  d->chs_count = count;
  return 0;
}

int orc_demixer_set_output_gain(orc_demixer *d, const int *chs, const float *gain, int count) {
  for (int i = 0; i < count; ++i) {
    d->gain_ch[i] = chs[i];
    d->gain[i] = gain[i];
  }
  d->gain_count = count;
  return 0;
}

/* demixer.c:592-618 */
int orc_demixer_set_demixing_info(orc_demixer *d, int mode, int w_idx) {
  if (mode < 0 || mode == 3 || mode > 6) return -1;
  if (w_idx < 0 || w_idx > 10) {
    d->last_mode = d->mode;
    d->mode = mode;
    d->last_w_idx = d->w_idx;
    if (k_mat[mode].woff > 0)
      d->w_idx = d->last_w_idx + 1 < 10 ? d->last_w_idx + 1 : 10;
    else
      d->w_idx = d->last_w_idx - 1 > 0 ? d->last_w_idx - 1 : 0;
  } else {
    if (mode != d->mode) d->last_mode = d->mode = mode;
    if (d->w_idx != w_idx) d->last_w_idx = d->w_idx = w_idx;
  }
  return 0;
}

/* demixer.c:620-634 */
int orc_demixer_set_recon_gain(orc_demixer *d, int count, const int *chs, const float *gain, unsigned flags) {
  if (flags && (flags ^ d->re_flags)) {
    for (int i = 0; i < count; ++i) d->re_ch[i] = chs[i];
    d->re_count = count;
    d->re_flags = flags;
  }
  for (int i = 0; i < count; ++i) d->re_gain[i] = gain[i];
  return 0;
}

static int dmx_s2(orc_demixer *d) { /* demixer.c:126-147 */
  float *r;
  if (!d->ch[CH_L2]) return -1;
  if (d->ch[CH_R2]) return 0;
  if (!d->ch[CH_MONO]) return -1;
  r = &d->large[d->fs * MX_S_R];
  for (int i = 0; i < d->fs; ++i) r[i] = 2 * d->ch[CH_MONO][i] - d->ch[CH_L2][i];
  d->ch[CH_R2] = r;
  return 0;
}

static int dmx_s3(orc_demixer *d) { /* demixer.c:152-181: the 0.707 literal makes this double arithmetic */
  float *l, *r;
  if (d->ch[CH_R3]) return 0;
  if (dmx_s2(d)) return -1;
  if (!d->ch[CH_C]) return -1;
  l = &d->large[MX_S_L * d->fs];
  r = &d->large[MX_S_R * d->fs];
  for (int i = 0; i < d->fs; i++) {
    l[i] = d->ch[CH_L2][i] - 0.707 * d->ch[CH_C][i];
    r[i] = d->ch[CH_R2][i] - 0.707 * d->ch[CH_C][i];
  }
  d->ch[CH_L3] = l;
  d->ch[CH_R3] = r;
  return 0;
}

static int dmx_s5(orc_demixer *d) { /* demixer.c:186-230 */
  float *l, *r;
  int i = 0;
  if (d->ch[CH_SR5]) return 0;
  if (dmx_s3(d)) return -1;
  if (!d->ch[CH_L5] || !d->ch[CH_R5]) return -1;
  l = &d->large[MX_S5_L * d->fs];
  r = &d->large[MX_S5_R * d->fs];
  for (; i < d->skip; i++) {
    l[i] = (d->ch[CH_L3][i] - d->ch[CH_L5][i]) / k_mat[d->last_mode].delta;
    r[i] = (d->ch[CH_R3][i] - d->ch[CH_R5][i]) / k_mat[d->last_mode].delta;
  }
  for (; i < d->fs; i++) {
    l[i] = (d->ch[CH_L3][i] - d->ch[CH_L5][i]) / k_mat[d->mode].delta;
    r[i] = (d->ch[CH_R3][i] - d->ch[CH_R5][i]) / k_mat[d->mode].delta;
  }
  d->ch[CH_SL5] = l;
  d->ch[CH_SR5] = r;
  return 0;
}

static int dmx_s7(orc_demixer *d) { /* demixer.c:236-284 */
  float *l, *r;
  int i = 0;
  if (d->ch[CH_BR7]) return 0;
  if (dmx_s5(d) < 0) return -1;
  if (!d->ch[CH_SL7] || !d->ch[CH_SR7]) return -1;
  l = &d->large[MX_S_L * d->fs];
  r = &d->large[MX_S_R * d->fs];
  for (; i < d->skip; i++) {
    l[i] = (d->ch[CH_SL5][i] - d->ch[CH_SL7][i] * k_mat[d->last_mode].alpha) / k_mat[d->last_mode].beta;
    r[i] = (d->ch[CH_SR5][i] - d->ch[CH_SR7][i] * k_mat[d->last_mode].alpha) / k_mat[d->last_mode].beta;
  }
  for (; i < d->fs; i++) {
    l[i] = (d->ch[CH_SL5][i] - d->ch[CH_SL7][i] * k_mat[d->mode].alpha) / k_mat[d->mode].beta;
    r[i] = (d->ch[CH_SR5][i] - d->ch[CH_SR7][i] * k_mat[d->mode].alpha) / k_mat[d->mode].beta;
  }
  d->ch[CH_BL7] = l;
  d->ch[CH_BR7] = r;
  return 0;
}

static int dmx_h2(orc_demixer *d) { /* demixer.c:290-335 */
  float *l, *r, w, lastW;
  int i = 0;
  if (d->ch[CH_HR]) return 0;
  if (!d->ch[CH_TL] || !d->ch[CH_TR]) return -1;
  if (dmx_s5(d)) return -1;
  w = w_of(d->w_idx);
  lastW = w_of(d->last_w_idx);
  l = &d->large[MX_T_L * d->fs];
  r = &d->large[MX_T_R * d->fs];
  for (; i < d->skip; i++) {
    l[i] = d->ch[CH_TL][i] - k_mat[d->last_mode].delta * lastW * d->ch[CH_SL5][i];
    r[i] = d->ch[CH_TR][i] - k_mat[d->last_mode].delta * lastW * d->ch[CH_SR5][i];
  }
  for (; i < d->fs; i++) {
    l[i] = d->ch[CH_TL][i] - k_mat[d->mode].delta * w * d->ch[CH_SL5][i];
    r[i] = d->ch[CH_TR][i] - k_mat[d->mode].delta * w * d->ch[CH_SR5][i];
  }
  d->ch[CH_HL] = l;
  d->ch[CH_HR] = r;
  return 0;
}

static int dmx_h4(orc_demixer *d) { /* demixer.c:340-377 */
  float *l, *r;
  int i = 0;
  if (d->ch[CH_HBR]) return 0;
  if (dmx_h2(d)) return -1;
  if (!d->ch[CH_HFR] || !d->ch[CH_HFL]) return -1;
  l = &d->large[MX_T_L * d->fs];
  r = &d->large[MX_T_R * d->fs];
  for (; i < d->skip; i++) {
    l[i] = (d->ch[CH_HL][i] - d->ch[CH_HFL][i]) / k_mat[d->last_mode].gamma;
    r[i] = (d->ch[CH_HR][i] - d->ch[CH_HFR][i]) / k_mat[d->last_mode].gamma;
  }
  for (; i < d->fs; i++) {
    l[i] = (d->ch[CH_HL][i] - d->ch[CH_HFL][i]) / k_mat[d->mode].gamma;
    r[i] = (d->ch[CH_HR][i] - d->ch[CH_HFR][i]) / k_mat[d->mode].gamma;
  }
  d->ch[CH_HBL] = l;
  d->ch[CH_HBR] = r;
  return 0;
}

static int dmx_channel(orc_demixer *d, int ch) { /* demixer.c:379-424 */
  if (d->ch[ch]) return 0;
  switch (ch) {
    case CH_R2: return dmx_s2(d);
    case CH_L3: case CH_R3: return dmx_s3(d);
    case CH_SL5: case CH_SR5: return dmx_s5(d);
    case CH_BL7: case CH_BR7: return dmx_s7(d);
    case CH_HL: case CH_HR: return dmx_h2(d);
    case CH_HBL: case CH_HBR: return dmx_h4(d);
    default: return -1;
  }
}

/* demixer.c:636-664; src is modified in place by the gain-up, like the reference */
int orc_demixer_demix(orc_demixer *d, float *dst, float *src, int size) {
  if (size != d->fs || d->layout < 0 || k_count[d->layout] != d->chs_count) return -1;
  memset(d->ch, 0, sizeof(d->ch));
  for (int c = 0; c < d->chs_count; ++c) d->ch[d->chs_in[c]] = src + (size_t)size * c;
  for (int c = 0; c < d->gain_count; ++c) /* :426-435 */
    for (int i = 0; i < d->fs; ++i)
      if (d->ch[d->gain_ch[c]]) d->ch[d->gain_ch[c]][i] *= d->gain[c];
  for (int c = 0; c < d->chs_count; ++c)
    if (dmx_channel(d, d->chs_out[c]) < 0) return -1;
  { /* :447-478 */
    float N = 7;
    for (int c = 0; c < d->re_count; c++) {
      int ch = d->re_ch[c];
      float sf = d->re_gain[c], sfavg, filt;
      float *out = d->ch[ch];
      if (N > 0)
        sfavg = (2 / (N + 1)) * sf + (1 - 2 / (N + 1)) * d->last_sfavg[ch];
      else
        sfavg = sf;
      for (int i = 0; i < d->fs; i++) {
        filt = d->last_sfavg[ch] * d->stop[i] + sfavg * d->start[i];
        out[i] *= filt;
      }
      d->last_sf[ch] = sf;
      d->last_sfavg[ch] = sfavg;
    }
  }
  for (int c = 0; c < d->chs_count; ++c) {
    int ch = d->chs_out[c];
    if (!d->ch[ch]) continue;
    memcpy(&dst[(size_t)c * size], d->ch[ch], sizeof(float) * size);
  }
  return 0;
}

/* accessors the GPU tests use to build the device-side per-frame records from this model's state */
void orc_demixer_state(const orc_demixer *d, int *mode, int *last_mode, int *w_idx, int *last_w_idx, int *skip) {
  *mode = d->mode;
  *last_mode = d->last_mode;
  *w_idx = d->w_idx;
  *last_w_idx = d->last_w_idx;
  *skip = d->skip;
}
const float *orc_demixer_window(const orc_demixer *d, int stop) { return stop ? d->stop : d->start; }
float orc_demixer_last_sfavg(const orc_demixer *d, int ch) { return d->last_sfavg[ch]; }
