/*
 * IAMF_decoder.h — the IAMF decoder API exported by libiamf_hip.so.  Source-compatible with the
 * reference header (Samsung/iac include/IAMF_decoder.h:36-245): the same 19 functions, handle
 * type and PODs, so a caller of the reference (e.g. test/tools/iamfplayer) links unchanged.
 *
 * What runs where.  Bitstream handling (OBU parsing, LPCM unpacking, parameter timeline) stays
 * on the host; everything from the decoded planar f32 element PCM to the interleaved integer PCM
 * — the section iamf_stream_render .. iamf_decoder_plane2stride_out of the reference's
 * iamf_decoder_internal_decode (src/iamf_dec/IAMF_decoder.c:3374-3500) — runs on the GPU through
 * the batch ABI of iamf_hip.h with one stream per handle.  No GPU, no decoder: every call that
 * needs the device fails with IAMF_ERR_INTERNAL, there is no CPU renderer in this library.
 *
 * Scope of this implementation: `ipcm` (LPCM) substreams; single-layer channel-based elements and
 * ambisonics (mono mapping) elements; one sub-mix of one or two elements; mix-gain and demixing
 * parameter blocks.  Opus/AAC/FLAC, multi-layer scalable audio and projection-mode ambisonics
 * return IAMF_ERR_UNIMPLEMENTED from IAMF_decoder_configure.
 */
#ifndef IAMF_DECODER_H
#define IAMF_DECODER_H

#include <stdint.h>

#include "IAMF_defines.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct IAMF_StreamInfo {
  uint32_t max_frame_size;
} IAMF_StreamInfo;

typedef struct IAMF_Decoder *IAMF_DecoderHandle;

/* reference IAMF_decoder.h:60,66 */
IAMF_DecoderHandle IAMF_decoder_open(void);
int IAMF_decoder_close(IAMF_DecoderHandle handle);

/* Consumes descriptor OBUs (reference :82).  With rsize != NULL the data may be a prefix of the
 * stream: returns IAMF_ERR_BUFFER_TOO_SMALL until the first non-descriptor OBU has been seen and
 * reports the consumed bytes; rsize == NULL means `data` holds the complete descriptors.
 * (handle, NULL, 0, NULL) re-applies a changed output layout. */
int IAMF_decoder_configure(IAMF_DecoderHandle handle, const uint8_t *data, uint32_t size, uint32_t *rsize);

/* Decodes one temporal unit (reference :99).  Returns the number of sample-frames written to
 * `pcm` (interleaved, bit depth as set), 0 if more data is needed, or a negative IAMF_ERR_*.
 * data == NULL drains the limiter / resampler delay.  The caller owns `pcm`:
 * bit_depth/8 * max_frame_size * channels bytes. */
int IAMF_decoder_decode(IAMF_DecoderHandle handle, const uint8_t *data, int32_t size, uint32_t *rsize,
                        void *pcm);

int IAMF_decoder_set_mix_presentation_id(IAMF_DecoderHandle handle, uint64_t id);              /* :108 */
int IAMF_decoder_output_layout_set_sound_system(IAMF_DecoderHandle handle, IAMF_SoundSystem ss); /* :117 */
int IAMF_decoder_output_layout_set_binaural(IAMF_DecoderHandle handle);                        /* :125 */
int IAMF_layout_sound_system_channels_count(IAMF_SoundSystem ss);                              /* :132 */
int IAMF_layout_binaural_channels_count(void);                                                 /* :138 */
/* malloc'ed string the caller frees (reference :144) */
char *IAMF_decoder_get_codec_capability(void);
int IAMF_decoder_set_normalization_loudness(IAMF_DecoderHandle handle, float loudness);        /* :152 */
int IAMF_decoder_set_bit_depth(IAMF_DecoderHandle handle, uint32_t bit_depth);                 /* :161 */
int IAMF_decoder_peak_limiter_enable(IAMF_DecoderHandle handle, uint32_t enable);              /* :170 */
int IAMF_decoder_peak_limiter_set_threshold(IAMF_DecoderHandle handle, float db);              /* :179 */
float IAMF_decoder_peak_limiter_get_threshold(IAMF_DecoderHandle handle);                      /* :187 */
int IAMF_decoder_set_sampling_rate(IAMF_DecoderHandle handle, uint32_t rate);                  /* :195 */
IAMF_StreamInfo *IAMF_decoder_get_stream_info(IAMF_DecoderHandle handle);                      /* :202 */

typedef struct IAMF_Param {
  int parameter_length;
  uint32_t parameter_definition_type;
  union {
    uint32_t dmixp_mode;
  };
} IAMF_Param;

typedef enum IAMF_SoundMode {
  IAMF_SOUND_MODE_NONE = -2,
  IAMF_SOUND_MODE_NA = -1,
  IAMF_SOUND_MODE_STEREO,
  IAMF_SOUND_MODE_MULTICHANNEL,
  IAMF_SOUND_MODE_BINAURAL
} IAMF_SoundMode;

typedef struct IAMF_extradata {
  IAMF_SoundSystem output_sound_system;
  uint32_t number_of_samples;
  uint32_t bitdepth;
  uint32_t sampling_rate;
  IAMF_SoundMode output_sound_mode;

  int num_loudness_layouts;
  IAMF_Layout *loudness_layout;
  IAMF_LoudnessInfo *loudness;

  uint32_t num_parameters;
  IAMF_Param *param;
} IAMF_extradata;

int IAMF_decoder_set_pts(IAMF_DecoderHandle handle, int64_t pts, uint32_t time_base);          /* :237 */
/* deep copies: the caller frees loudness_layout, loudness and param (reference :239) */
int IAMF_decoder_get_last_metadata(IAMF_DecoderHandle handle, int64_t *pts, IAMF_extradata *metadata);

#ifdef __cplusplus
}
#endif
#endif /* IAMF_DECODER_H */
