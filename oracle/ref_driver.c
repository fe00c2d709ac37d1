/* ref_driver.c — TEST INFRASTRUCTURE: drives the REAL reference decoder (oracle/_ref/libiamf_ref.so,
 * built from /root/reference's own sources) over one in-memory .iamf stream inside a single C call,
 * the way test/tools/iamfplayer/player/iamfplayer.c:571-600 drives it, so that bench.py's
 * cpu_baseline can time it on many threads without Python between the frames.
 * Built by `make -C oracle tools` into oracle/_ref/libref_driver.so; includes only the reference's
 * public header. */
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#include "IAMF_decoder.h"

/* sound_system < 0 selects binaural.  Returns the sample-frames decoded (or a negative IAMF error);
 * *seconds = time spent in the IAMF_decoder_decode loop (configure excluded). */
long refdrv_decode(const uint8_t *stream, uint32_t size, int sound_system, int bit_depth, float limiter_db,
                   double *seconds) {
  IAMF_DecoderHandle d = IAMF_decoder_open();
  uint32_t used = 0, rs = 0;
  long total = 0;
  int n, ch;
  void *pcm;
  struct timespec t0, t1;
  if (!d) return -1;
  IAMF_decoder_peak_limiter_set_threshold(d, limiter_db);
  IAMF_decoder_set_bit_depth(d, (uint32_t)bit_depth);
  if (sound_system < 0) {
    IAMF_decoder_output_layout_set_binaural(d);
    ch = 2;
  } else {
    IAMF_decoder_output_layout_set_sound_system(d, (IAMF_SoundSystem)sound_system);
    ch = IAMF_layout_sound_system_channels_count((IAMF_SoundSystem)sound_system);
  }
  IAMF_decoder_set_pts(d, 0, 90000);
  n = IAMF_decoder_configure(d, stream, size, &rs);
  if (n != IAMF_OK) {
    IAMF_decoder_close(d);
    return n;
  }
  used = rs;
  pcm = malloc((size_t)(bit_depth / 8) * 6144 * 6 * (size_t)ch);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (used < size) {
    rs = 0;
    n = IAMF_decoder_decode(d, stream + used, (int32_t)(size - used), &rs, pcm);
    if (n < 0 || !rs) break;
    total += n;
    used += rs;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  free(pcm);
  IAMF_decoder_close(d);
  return total;
}
