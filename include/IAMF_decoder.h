/*
 * IAMF_decoder.h — the IAMF decoder API exported by libiamf_hip.so.  Source-compatible with the
 * reference header (Samsung/iac include/IAMF_decoder.h:36-245): the same 19 functions, handle
 * type and PODs, so a caller of the reference (e.g. test/tools/iamfplayer) links unchanged.
 *
 * What runs where.  Bitstream handling (OBU parsing, LPCM unpacking, parameter timeline, layer
 * selection) stays on the host; everything from the decoded planar f32 element PCM to the
 * interleaved integer PCM — demixer, iamf_stream_render .. iamf_decoder_plane2stride_out of the
 * reference's iamf_decoder_internal_decode (src/iamf_dec/IAMF_decoder.c:3347-3500) — runs on the
 * GPU through the batch ABI of iamf_hip.h with one stream per handle.  No GPU, no decoder: every
 * call that needs the device fails with IAMF_ERR_INTERNAL, there is no CPU renderer in this library.
 *
 * Scope of this implementation: `ipcm` (LPCM) substreams; channel-based elements with one or more
 * scalable layers (output gains, recon gains, demixing), ambisonics elements in mono and projection
 * mode; one sub-mix of one or two elements; mix-gain, demixing and recon-gain parameter blocks.
 * Opus / AAC / FLAC return IAMF_ERR_UNIMPLEMENTED from IAMF_decoder_configure.
 */
#ifndef IAMF_DECODER_H
#define IAMF_DECODER_H

#include <stdint.h>

#include "IAMF_defines.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct IAMF_Decoder *IAMF_DecoderHandle;
typedef struct IAMF_StreamInfo { uint32_t max_frame_size; } IAMF_StreamInfo;

/* ---- life cycle (reference IAMF_decoder.h:60,66) ---- */
IAMF_DecoderHandle IAMF_decoder_open(void);
int IAMF_decoder_close(IAMF_DecoderHandle h);

/* ---- settings, before IAMF_decoder_configure (reference :108-195) ---- */
int IAMF_decoder_set_mix_presentation_id(IAMF_DecoderHandle h, uint64_t mix_presentation_id);
int IAMF_decoder_output_layout_set_sound_system(IAMF_DecoderHandle h, IAMF_SoundSystem sound_system);
int IAMF_decoder_output_layout_set_binaural(IAMF_DecoderHandle h);
int IAMF_decoder_set_normalization_loudness(IAMF_DecoderHandle h, float lkfs);
int IAMF_decoder_set_bit_depth(IAMF_DecoderHandle h, uint32_t bits);          /* 16, 24 or 32 */
int IAMF_decoder_peak_limiter_enable(IAMF_DecoderHandle h, uint32_t on);
int IAMF_decoder_peak_limiter_set_threshold(IAMF_DecoderHandle h, float dbfs);
float IAMF_decoder_peak_limiter_get_threshold(IAMF_DecoderHandle h);
int IAMF_decoder_set_sampling_rate(IAMF_DecoderHandle h, uint32_t hz);        /* 8/12/16/24/32/44.1/48 k */
int IAMF_decoder_set_pts(IAMF_DecoderHandle h, int64_t pts, uint32_t time_base);  /* :237 */

/* Consumes descriptor OBUs (reference :82).  With consumed != NULL the data may be a prefix of the
 * stream: returns IAMF_ERR_BUFFER_TOO_SMALL until the first non-descriptor OBU has been seen and
 * reports the consumed bytes; consumed == NULL means `obus` holds the complete descriptors.
 * (h, NULL, 0, NULL) re-applies a changed output layout. */
int IAMF_decoder_configure(IAMF_DecoderHandle h, const uint8_t *obus, uint32_t n_bytes, uint32_t *consumed);

/* Decodes one temporal unit (reference :99).  Returns the number of sample-frames written to
 * `pcm_out` (interleaved, bit depth as set), 0 if more data is needed, or a negative IAMF_ERR_*.
 * obus == NULL drains the limiter / resampler delay.  The caller owns `pcm_out`:
 * bit_depth/8 * max_frame_size * channels bytes. */
int IAMF_decoder_decode(IAMF_DecoderHandle h, const uint8_t *obus, int32_t n_bytes, uint32_t *consumed, void *pcm_out);

/* ---- queries (reference :132-144,202) ---- */
int IAMF_layout_sound_system_channels_count(IAMF_SoundSystem sound_system);
int IAMF_layout_binaural_channels_count(void);
char *IAMF_decoder_get_codec_capability(void); /* malloc'ed string, the caller frees it */
IAMF_StreamInfo *IAMF_decoder_get_stream_info(IAMF_DecoderHandle h);

/* ---- metadata of the last decoded temporal unit (reference :206-239) ---- */
typedef enum IAMF_SoundMode {
  IAMF_SOUND_MODE_NONE = -2, IAMF_SOUND_MODE_NA = -1,
  IAMF_SOUND_MODE_STEREO = 0, IAMF_SOUND_MODE_MULTICHANNEL = 1, IAMF_SOUND_MODE_BINAURAL = 2
} IAMF_SoundMode;

typedef struct IAMF_Param {
  int parameter_length;
  uint32_t parameter_definition_type;
  union { uint32_t dmixp_mode; };
} IAMF_Param;

typedef struct IAMF_extradata {
  IAMF_SoundSystem output_sound_system;
  uint32_t number_of_samples, bitdepth, sampling_rate;
  IAMF_SoundMode output_sound_mode;
  int num_loudness_layouts;
  IAMF_Layout *loudness_layout;
  IAMF_LoudnessInfo *loudness;
  uint32_t num_parameters;
  IAMF_Param *param;
} IAMF_extradata;

/* deep copies: the caller frees loudness_layout, loudness and param */
int IAMF_decoder_get_last_metadata(IAMF_DecoderHandle h, int64_t *pts, IAMF_extradata *out);

#ifdef __cplusplus
}
#endif
#endif /* IAMF_DECODER_H */
