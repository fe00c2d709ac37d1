/*
 * iamfplayer_hip — command-line player on libiamf_hip.so, the counterpart of the reference's
 * test/tools/iamfplayer (player/iamfplayer.c:529-662 for raw .iamf input): it talks only to
 * include/IAMF_decoder.h, so the same source links against the reference's libiamf as well.
 *
 *   iamfplayer_hip [-o2] [-s<0..12|b>] [-d<16|24|32>] [-r<rate>] [-p<dB>] [-l<LKFS>] [-disable_limiter]
 *                  [-out <file.wav>] <input.iamf>
 * -o2 (raw IAMF bitstream input) is the only input mode; the WAV is written next to the cwd as
 * ss<N>_<stem>.wav / binaural_<stem>.wav like the reference does (iamfplayer.c:323-358).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "IAMF_decoder.h"

#define BLOCK_SIZE 184320 /* iamfplayer.c:372 */

static void wr32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void wr16(FILE *f, uint16_t v) { fwrite(&v, 2, 1, f); }

static void wav_header(FILE *f, int ch, int rate, int bits, uint32_t data_bytes) {
  fseek(f, 0, SEEK_SET);
  fwrite("RIFF", 1, 4, f);
  wr32(f, 36 + data_bytes);
  fwrite("WAVEfmt ", 1, 8, f);
  wr32(f, 16);
  wr16(f, 1);
  wr16(f, (uint16_t)ch);
  wr32(f, (uint32_t)rate);
  wr32(f, (uint32_t)(rate * ch * bits / 8));
  wr16(f, (uint16_t)(ch * bits / 8));
  wr16(f, (uint16_t)bits);
  fwrite("data", 1, 4, f);
  wr32(f, data_bytes);
}

int main(int argc, char **argv) {
  int ss = 0, binaural = 0, bits = 16, limiter = 1, rate = 0;
  float peak = -1.0f, loud = 0.0f;
  const char *in = 0, *out = 0;
  char name[512];
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-out") && i + 1 < argc) out = argv[++i];
    else if (!strncmp(argv[i], "-o", 2)) continue; /* -o2: raw IAMF bitstream, the only input mode */
    else if (!strcmp(argv[i], "-sb")) binaural = 1;
    else if (!strncmp(argv[i], "-s", 2)) ss = atoi(argv[i] + 2);
    else if (!strcmp(argv[i], "-disable_limiter")) limiter = 0;
    else if (!strncmp(argv[i], "-d", 2)) bits = atoi(argv[i] + 2);
    else if (!strncmp(argv[i], "-r", 2)) rate = atoi(argv[i] + 2);
    else if (!strncmp(argv[i], "-p", 2)) peak = (float)atof(argv[i] + 2);
    else if (!strncmp(argv[i], "-l", 2)) loud = (float)atof(argv[i] + 2);
    else in = argv[i];
  }
  if (!in) {
    fprintf(stderr, "usage: %s [-o2] [-s<n>|-sb] [-d<bits>] [-r<rate>] [-p<dB>] [-l<LKFS>] [-disable_limiter] [-out f.wav] in.iamf\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(in, "rb");
  if (!f) return perror(in), 1;
  IAMF_DecoderHandle dec = IAMF_decoder_open();
  if (!dec) return fprintf(stderr, "IAMF decoder can't created.\n"), 1;
  if (!limiter) IAMF_decoder_peak_limiter_enable(dec, 0);
  else IAMF_decoder_peak_limiter_set_threshold(dec, peak);
  IAMF_decoder_set_normalization_loudness(dec, loud);
  IAMF_decoder_set_bit_depth(dec, (uint32_t)bits);
  if (rate > 0 && IAMF_decoder_set_sampling_rate(dec, (uint32_t)rate) != IAMF_OK) return fprintf(stderr, "Invalid sampling rate %d\n", rate), 1;
  int channels;
  if (binaural) {
    IAMF_decoder_output_layout_set_binaural(dec);
    channels = IAMF_layout_binaural_channels_count();
  } else {
    IAMF_decoder_output_layout_set_sound_system(dec, (IAMF_SoundSystem)ss);
    channels = IAMF_layout_sound_system_channels_count((IAMF_SoundSystem)ss);
  }
  if (!out) {
    const char *b = strrchr(in, '/');
    b = b ? b + 1 : in;
    char stem[256];
    snprintf(stem, sizeof(stem), "%s", b);
    char *dot = strrchr(stem, '.');
    if (dot) *dot = 0;
    if (binaural) snprintf(name, sizeof(name), "binaural_%s.wav", stem);
    else snprintf(name, sizeof(name), "ss%d_%s.wav", ss, stem);
    out = name;
  }
  FILE *w = fopen(out, "wb");
  if (!w) return perror(out), 1;
  wav_header(w, channels, rate ? rate : 48000, bits, 0);

  uint8_t *block = (uint8_t *)malloc(BLOCK_SIZE);
  void *pcm = 0;
  uint32_t used = 0, size = 0, rsize = 0, data_bytes = 0, frames = 0;
  int state = 0, end = 0, ret = 0;
  uint64_t samples = 0;
  while (1) { /* iamfplayer.c:551-662 */
    size_t n = fread(block + used, 1, BLOCK_SIZE - used, f);
    if (n == 0) end = 1;
    size = used + (uint32_t)n;
    used = 0;
    if (state <= 0) {
      if (end) break;
      rsize = 0;
      IAMF_decoder_set_pts(dec, 0, 90000);
      ret = IAMF_decoder_configure(dec, block, size, &rsize);
      if (ret == IAMF_OK) {
        IAMF_StreamInfo *info = IAMF_decoder_get_stream_info(dec);
        state = 1;
        if (!pcm) pcm = malloc((size_t)bits / 8 * info->max_frame_size * channels);
      } else if (ret != IAMF_ERR_BUFFER_TOO_SMALL || !rsize) {
        fprintf(stderr, "errno: %d, fail to configure decoder.\n", ret);
        break;
      }
      used += rsize;
    }
    if (state > 0) {
      while (1) {
        rsize = 0;
        ret = end ? IAMF_decoder_decode(dec, 0, 0, &rsize, pcm) : IAMF_decoder_decode(dec, block + used, (int32_t)(size - used), &rsize, pcm);
        if (ret > 0) {
          ++frames;
          samples += (uint64_t)ret;
          fwrite(pcm, (size_t)bits / 8 * channels, (size_t)ret, w);
          data_bytes += (uint32_t)(bits / 8 * channels * ret);
        }
        if (end) break;
        used += rsize;
        if (ret == IAMF_ERR_INVALID_STATE) state = ret;
        if (ret < 0 || used >= size || !rsize) break;
      }
    }
    if (end) break;
    memmove(block, block + used, size - used);
    used = size - used;
  }
  wav_header(w, channels, rate ? rate : 48000, bits, data_bytes);
  fclose(w);
  fclose(f);
  fprintf(stdout, "===================== Get %u frames\n", frames);
  fprintf(stdout, "%s: %llu sample-frames, %d channels, %d bit -> %s\n", in, (unsigned long long)samples, channels, bits, out);
  free(block);
  free(pcm);
  IAMF_decoder_close(dec);
  return ret < 0 && ret != IAMF_ERR_INVALID_STATE ? 1 : 0;
}
