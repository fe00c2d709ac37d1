/* TEST INFRASTRUCTURE: drives iamf_hip_decoder_group (built with sanitizers against device_stub.c) over one .iamf
 * file with N handles that do NOT advance in step: in round r handle i is starved (one byte: no complete OBU) when
 * (r + i) % 3 == 0, and a handle that has eaten its whole stream flushes while the others go on.  Prints every
 * handle's total; tests/test_facade_malformed.py compares them with the single-handle driver's.
 * usage: group_driver file.iamf <sound system id | b> bits N threads [other.iamf]
 * (other.iamf: the odd handles are configured from that stream instead — what group_create says about the mix is the test) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "IAMF_decoder.h"
#include "iamf_hip.h"

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)malloc(size ? size : 1);
  if (fread(buf, 1, size, f) != (size_t)size) return 2;
  fclose(f);
  const int bits = atoi(argv[3]), N = atoi(argv[4]), threads = atoi(argv[5]);
  uint8_t *alt = 0;
  long alt_size = 0;
  if (argc > 6) {
    FILE *fa = fopen(argv[6], "rb");
    if (!fa) return 2;
    fseek(fa, 0, SEEK_END);
    alt_size = ftell(fa);
    fseek(fa, 0, SEEK_SET);
    alt = (uint8_t *)malloc(alt_size ? alt_size : 1);
    if (fread(alt, 1, alt_size, fa) != (size_t)alt_size) return 2;
    fclose(fa);
  }
  int ch = 2;
  void **h = (void **)calloc(N, sizeof(void *));
  uint32_t *used = (uint32_t *)calloc(N, sizeof(uint32_t)), *rs = (uint32_t *)calloc(N, sizeof(uint32_t));
  long *total = (long *)calloc(N, sizeof(long));
  int *done = (int *)calloc(N, sizeof(int));
  void **pcm = (void **)calloc(N, sizeof(void *));
  const uint8_t **data = (const uint8_t **)calloc(N, sizeof(uint8_t *));
  int32_t *sizes = (int32_t *)calloc(N, sizeof(int32_t)), *res = (int32_t *)calloc(N, sizeof(int32_t));
  for (int i = 0; i < N; ++i) {
    IAMF_DecoderHandle d = IAMF_decoder_open();
    h[i] = d;
    IAMF_decoder_set_bit_depth(d, bits);
    if (argv[2][0] == 'b') {
      IAMF_decoder_output_layout_set_binaural(d);
    } else {
      IAMF_decoder_output_layout_set_sound_system(d, (IAMF_SoundSystem)atoi(argv[2]));
      ch = IAMF_layout_sound_system_channels_count((IAMF_SoundSystem)atoi(argv[2]));
    }
    uint32_t r0 = 0;
    int r = (alt && (i & 1)) ? IAMF_decoder_configure(d, alt, (uint32_t)alt_size, &r0)
                             : IAMF_decoder_configure(d, buf, (uint32_t)size, &r0);
    if (i == 0) printf("configure %d rsize %u\n", r, r0);
    if (r != IAMF_OK) return 0;
    used[i] = r0;
    pcm[i] = malloc((size_t)(bits / 8) * IAMF_decoder_get_stream_info(d)->max_frame_size * ch);
  }
  iamf_hip_decoder_group *g = 0;
  int rc = iamf_hip_decoder_group_create(h, N, threads, &g);
  printf("group_create %d\n", rc);
  if (rc) {
    for (int i = 0; i < N; ++i) {
      IAMF_decoder_close((IAMF_DecoderHandle)h[i]);
      free(pcm[i]);
    }
    free(h); free(used); free(rs); free(total); free(done); free(pcm); free(data); free(sizes); free(res); free(buf); free(alt);
    return 0;
  }
  { /* a grouped handle refuses the single-handle entry points */
    uint32_t x = 0;
    printf("single_decode_while_grouped %d close %d\n", IAMF_decoder_decode((IAMF_DecoderHandle)h[0], buf, 4, &x, pcm[0]),
           IAMF_decoder_close((IAMF_DecoderHandle)h[0]));
  }
  int left = N;
  for (int round = 0; left > 0 && round < 100000; ++round) {
    for (int i = 0; i < N; ++i) {
      if (done[i]) { /* finished earlier: starve (a second flush returns 0 anyway) */
        data[i] = buf;
        sizes[i] = 1;
      } else if (used[i] >= (uint32_t)size) {
        data[i] = 0;
        sizes[i] = 0;
      } else if ((round + i) % 3 == 0) {
        data[i] = buf + used[i];
        sizes[i] = 1;
      } else {
        data[i] = buf + used[i];
        sizes[i] = (int32_t)(size - used[i]);
      }
    }
    rc = iamf_hip_decoder_group_decode(g, data, sizes, rs, pcm, res);
    if (rc) {
      printf("group_decode %d\n", rc);
      break;
    }
    for (int i = 0; i < N; ++i) {
      if (done[i]) continue;
      if (!data[i]) {
        printf("h%d flush %d\n", i, res[i]);
        if (res[i] > 0) total[i] += res[i];
        done[i] = 1;
        --left;
        continue;
      }
      if (sizes[i] == 1) continue;
      if (i == 0) printf("decode %d rsize %u\n", res[i], rs[i]);
      if (res[i] > 0) total[i] += res[i];
      used[i] += rs[i];
      if (!rs[i] || res[i] == IAMF_ERR_INVALID_STATE) used[i] = (uint32_t)size; /* as the single driver: stop feeding */
    }
  }
  for (int i = 0; i < N; ++i) printf("total h%d %ld\n", i, total[i]);
  iamf_hip_decoder_group_destroy(g);
  for (int i = 0; i < N; ++i) {
    IAMF_decoder_close((IAMF_DecoderHandle)h[i]);
    free(pcm[i]);
  }
  free(h); free(used); free(rs); free(total); free(done); free(pcm); free(data); free(sizes); free(res); free(buf); free(alt);
  return 0;
}
