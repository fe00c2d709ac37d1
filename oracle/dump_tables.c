/*
 * oracle/dump_tables.c — TEST/BUILD INFRASTRUCTURE (runs only where /root/reference exists).
 *
 * Links the REAL reference (oracle/_ref/libiamf_ref.so) and dumps, through its own lookup
 * functions, every static rendering matrix it can return:
 *   IAMF_element_renderer_get_H2M_matrix   (reference src/iamf_dec/h2m_rdr.c:1070-1081)
 *   IAMF_element_renderer_get_M2M_matrix   (reference src/iamf_dec/m2m_rdr.c:1786-1804)
 * The result is a binary blob of coefficient DATA (no source text) that both the oracle and
 * the product load: iac_amd/data/rdr_tables.bin.  Exactly m*n floats are taken from each
 * `mat` pointer, i.e. exactly the floats render_H2M / render_M2M would read (this preserves
 * the mis-strided 2nd-order -> Sound System H table).
 *
 * Blob layout (little endian):
 *   char     magic[8] = "IARDRTB1"
 *   uint32   n_entries
 *   entry[n] { uint32 kind(0=H2M,1=M2M), in_id, out_id; int32 channels, lfe1, lfe2, m, n;
 *              uint32 offset (in floats, into data[]) }
 *   float    data[]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ae_rdr.h"

typedef struct {
  uint32_t kind, in_id, out_id;
  int32_t channels, lfe1, lfe2, m, n;
  uint32_t offset;
} entry_t;

static const int out_systems[] = {BS2051_A, BS2051_B, BS2051_C, BS2051_D, BS2051_E,
                                  BS2051_F, BS2051_G, BS2051_H, BS2051_I, BS2051_J,
                                  IAMF_312,  IAMF_712,  IAMF_BINAURAL, IAMF_MONO};
static const int in_systems[] = {IAMF_MONO, IAMF_STEREO, IAMF_312, IAMF_51,  IAMF_512,
                                 IAMF_514,  IAMF_71,     IAMF_712, IAMF_714, IAMF_BINAURAL};

int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "rdr_tables.bin";
  entry_t ents[512];
  int ne = 0;
  float *data = malloc(sizeof(float) * 1 << 20);
  uint32_t nd = 0;
  const int nout = sizeof(out_systems) / sizeof(out_systems[0]);
  const int nin = sizeof(in_systems) / sizeof(in_systems[0]);

  for (int order = 0; order <= 3; ++order) {
    for (int o = 0; o < nout; ++o) {
      IAMF_HOA_LAYOUT hin = {(HOA_ORDER)order, 0};
      IAMF_PREDEFINED_SP_LAYOUT pout = {(IAMF_SOUND_SYSTEM)out_systems[o], 0, 0};
      struct h2m_rdr_t h;
      if (IAMF_element_renderer_get_H2M_matrix(&hin, &pout, &h) != 0) continue;
      entry_t e = {0, (uint32_t)order, (uint32_t)out_systems[o], h.channels, h.lfe1, h.lfe2, h.m, h.n, nd};
      memcpy(data + nd, h.mat, sizeof(float) * h.m * h.n);
      nd += h.m * h.n;
      ents[ne++] = e;
    }
  }
  for (int i = 0; i < nin; ++i) {
    for (int o = 0; o < nout; ++o) {
      IAMF_PREDEFINED_SP_LAYOUT pin = {(IAMF_SOUND_SYSTEM)in_systems[i], 0, 0};
      IAMF_PREDEFINED_SP_LAYOUT pout = {(IAMF_SOUND_SYSTEM)out_systems[o], 0, 0};
      IAMF_SP_LAYOUT lin, lout;
      struct m2m_rdr_t mm;
      memset(&lin, 0, sizeof(lin));
      memset(&lout, 0, sizeof(lout));
      lin.sp_layout.predefined_sp = &pin;
      lout.sp_layout.predefined_sp = &pout;
      if (IAMF_element_renderer_get_M2M_matrix(&lin, &lout, &mm) != 0) continue;
      entry_t e = {1, (uint32_t)in_systems[i], (uint32_t)out_systems[o], mm.n, -1, -1, mm.m, mm.n, nd};
      memcpy(data + nd, mm.mat, sizeof(float) * mm.m * mm.n);
      nd += mm.m * mm.n;
      ents[ne++] = e;
    }
  }

  FILE *f = fopen(path, "wb");
  if (!f) return 1;
  uint32_t n32 = (uint32_t)ne;
  fwrite("IARDRTB1", 1, 8, f);
  fwrite(&n32, 4, 1, f);
  fwrite(ents, sizeof(entry_t), ne, f);
  fwrite(data, sizeof(float), nd, f);
  fclose(f);
  fprintf(stderr, "dump_tables: %d entries, %u floats -> %s\n", ne, nd, path);
  return 0;
}
