/*
 * oracle/iamf_oracle_lfe.c — CPU restatement of the reference's HOA LFE generator.
 * TEST INFRASTRUCTURE (see iamf_oracle.h).  Parity: PINNED by tests/test_oracle_golden.py::test_lfe_*
 * against tests/golden/lfe.npz, which oracle/gen_golden_lfe.py produced from the reference built with
 * its own switch -DDISABLE_LFE_HOA=0 (oracle/_ref_lfe, `make -C oracle ref_lfe`).
 *
 * What the reference does (src/iamf_dec/h2m_rdr.c:1151-1239, call site IAMF_decoder.c:2625-2636):
 * a 2nd-order Butterworth low-pass at 120 Hz runs over ambisonics channel 0 (W) of the element; its
 * output, scaled, replaces the silence the default build writes into the layout's LFE slot(s).
 */
#include <math.h>
#include <string.h>

#include "iamf_oracle.h"

/* lfefilter_init, h2m_rdr.c:1192-1212.  `1 / sample_rate + 1.0e-10` is float / float + double, stored
 * to a float; `M_PI * cutoff * delta` is evaluated in double and narrowed for tanf. */
void orc_lfe_init(orc_lfe *f, float cutoff_freq, float sample_rate) {
  float delta_time = 1 / sample_rate + 1.0e-10;
  memset(f, 0, sizeof(*f));
  f->init = 1;
  if (cutoff_freq <= 0) return;
  f->c = 1.0f / (float)tanf(M_PI * cutoff_freq * delta_time);
  f->a1 = 1.0f / (1.0f + f->c + f->c * f->c);
  f->a2 = 2.0f * f->a1;
  f->a3 = f->a1;
  f->b1 = 2.0f * (1.0f - f->c * f->c) * f->a1;
  f->b2 = (1.0f - f->c + f->c * f->c) * f->a1;
}

/* lfefilter_update, h2m_rdr.c:1215-1237: one f32 expression, evaluated left to right */
float orc_lfe_update(orc_lfe *f, float input) {
  float output;
  if (f->init != 1) orc_lfe_init(f, 120, 48000.0f);
  output = f->a1 * input + f->a2 * f->ih[0] + f->a3 * f->ih[1] - f->b1 * f->oh[0] - f->b2 * f->oh[1];
  f->ih[1] = f->ih[0];
  f->ih[0] = input;
  f->oh[1] = f->oh[0];
  f->oh[0] = output;
  return output;
}

/* IAMF_element_renderer_render_H2M with DISABLE_LFE_HOA == 0 and a filter (h2m_rdr.c:1088-1187):
 * the matrix part and the slot spreading are orc_render_h2m's; then slot lfe1 takes the filtered W
 * times 0.5 (n <= 2) or divided by sqrt(n) — both double expressions narrowed on the store — and slot
 * lfe2 a copy of slot lfe1, or its own filter pass if there is no lfe1. */
void orc_render_h2m_lfe(const orc_matrix *mx, const float *in, float *out, int ns, orc_lfe *lfe) {
  orc_render_h2m(mx, in, out, ns);
  if (!lfe) return;
  if (mx->lfe1 >= 0) {
    float *dst = out + (size_t)mx->lfe1 * ns;
    for (int j = 0; j < ns; ++j) {
      float o = orc_lfe_update(lfe, in[j]);
      dst[j] = mx->n <= 2 ? o * 0.5 : o / sqrt(mx->n);
    }
  }
  if (mx->lfe2 >= 0) {
    float *dst = out + (size_t)mx->lfe2 * ns;
    for (int j = 0; j < ns; ++j) {
      if (mx->lfe1 >= 0) {
        dst[j] = out[(size_t)mx->lfe1 * ns + j];
      } else {
        float o = orc_lfe_update(lfe, in[j]);
        dst[j] = mx->n <= 2 ? o * 0.5 : o / sqrt(mx->n);
      }
    }
  }
}

int orc_sizeof_lfe(void) { return (int)sizeof(orc_lfe); }
