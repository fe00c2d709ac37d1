/* TEST INFRASTRUCTURE: drives the facade (built with sanitizers against device_stub.c) over one .iamf
 * file the way iamfplayer does (test/tools/iamfplayer/player/iamfplayer.c:380-431,571-600) and prints
 * every return code.  usage: facade_driver file.iamf <sound system id | b> [bit depth] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "IAMF_decoder.h"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)malloc(size ? size : 1); /* exact size: ASan sees any read past the stream */
  if (fread(buf, 1, size, f) != (size_t)size) return 2;
  fclose(f);
  int bits = argc > 3 ? atoi(argv[3]) : 16, ch = 2;
  IAMF_DecoderHandle d = IAMF_decoder_open();
  IAMF_decoder_set_bit_depth(d, bits);
  if (argv[2][0] == 'b') {
    IAMF_decoder_output_layout_set_binaural(d);
  } else {
    IAMF_decoder_output_layout_set_sound_system(d, (IAMF_SoundSystem)atoi(argv[2]));
    ch = IAMF_layout_sound_system_channels_count((IAMF_SoundSystem)atoi(argv[2]));
  }
  uint32_t used = 0, rs = 0;
  long total = 0;
  int configs = 0;
  void *pcm = 0;
  for (;;) {
    rs = 0;
    int r = IAMF_decoder_configure(d, buf + used, (uint32_t)size - used, &rs);
    printf("configure %d rsize %u\n", r, rs);
    if (r != IAMF_OK) break;
    ++configs;
    used += rs;
    IAMF_StreamInfo *info = IAMF_decoder_get_stream_info(d);
    free(pcm);
    pcm = malloc((size_t)(bits / 8) * info->max_frame_size * ch); /* what the API tells a caller to allocate */
    int again = 0;
    while (used < (uint32_t)size) {
      rs = 0;
      int n = IAMF_decoder_decode(d, buf + used, (int32_t)(size - used), &rs, pcm);
      printf("decode %d rsize %u\n", n, rs);
      if (n == IAMF_ERR_INVALID_STATE) { used += rs; again = 1; break; }
      if (n > 0) total += n;
      used += rs;
      if (!rs) break;
    }
    if (!again) {
      int n = IAMF_decoder_decode(d, 0, 0, &rs, pcm);
      printf("flush %d\n", n);
      if (n > 0) total += n;
      break;
    }
  }
  printf("total %ld configs %d\n", total, configs);
  IAMF_decoder_close(d);
  free(pcm);
  free(buf);
  return 0;
}
