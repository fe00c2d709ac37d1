/*
 * IAMF_defines.h — public constants and PODs of the IAMF decoder API as implemented by
 * libiamf_hip.so.  Source-compatible with the reference header of the same name
 * (Samsung/iac include/IAMF_defines.h:36-215): identical identifiers, values and layouts, so code
 * written against the reference compiles and links unchanged.  Re-declared here, not copied.
 */
#ifndef IAMF_DEFINES_H
#define IAMF_DEFINES_H

#include <stdint.h>

/* audio element kinds (reference :44-49) */
typedef enum {
  AUDIO_ELEMENT_INVALID = -1,
  AUDIO_ELEMENT_CHANNEL_BASED,
  AUDIO_ELEMENT_SCENE_BASED,
  AUDIO_ELEMENT_COUNT
} AudioElementType;

typedef enum AmbisonicsMode { AMBISONICS_MONO, AMBISONICS_PROJECTION } AmbisonicsMode;

/* layout(): 2-bit type, then 4-bit sound system for loudspeaker layouts (reference :56-60,86-119) */
typedef enum IAMF_LayoutType {
  IAMF_LAYOUT_TYPE_NOT_DEFINED = 0,
  IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION = 2,
  IAMF_LAYOUT_TYPE_BINAURAL,
} IAMF_LayoutType;

/* ITU-R BS.2051 sound systems A..J plus the IAMF extensions (reference :62-78) */
typedef enum IAMF_SoundSystem {
  SOUND_SYSTEM_INVALID = -1,
  SOUND_SYSTEM_A,       /* 0+2+0 */
  SOUND_SYSTEM_B,       /* 0+5+0 */
  SOUND_SYSTEM_C,       /* 2+5+0 */
  SOUND_SYSTEM_D,       /* 4+5+0 */
  SOUND_SYSTEM_E,       /* 4+5+1 */
  SOUND_SYSTEM_F,       /* 3+7+0 */
  SOUND_SYSTEM_G,       /* 4+9+0 */
  SOUND_SYSTEM_H,       /* 9+10+3 */
  SOUND_SYSTEM_I,       /* 0+7+0 */
  SOUND_SYSTEM_J,       /* 4+7+0 */
  SOUND_SYSTEM_EXT_712, /* 2+7+0 */
  SOUND_SYSTEM_EXT_312, /* 2+3+0 */
  SOUND_SYSTEM_MONO,    /* 0+1+0 */
  SOUND_SYSTEM_END
} IAMF_SoundSystem;

typedef enum IAMF_ParameterType {
  IAMF_PARAMETER_TYPE_MIX_GAIN = 0,
  IAMF_PARAMETER_TYPE_DEMIXING,
  IAMF_PARAMETER_TYPE_RECON_GAIN,
} IAMF_ParameterType;

typedef enum IAMF_AnimationType {
  ANIMATION_TYPE_INVALID = -1,
  ANIMATION_TYPE_STEP,
  ANIMATION_TYPE_LINEAR,
  ANIMATION_TYPE_BEZIER
} IAMF_AnimationType;

typedef struct IAMF_Layout {
  union {
    struct {
      uint8_t reserved : 2;
      uint8_t sound_system : 4;
      uint8_t type : 2;
    } sound_system;
    struct {
      uint8_t reserved : 6;
      uint8_t type : 2;
    } binaural;
    struct {
      uint8_t reserved : 6;
      uint8_t type : 2;
    };
  };
} IAMF_Layout;

typedef struct _anchor_loudness_t {
  uint8_t anchor_element;
  int16_t anchored_loudness;
} anchor_loudness_t;

/* loudness_info(): Q7.8 dB values (reference :121-154) */
typedef struct IAMF_LoudnessInfo {
  uint8_t info_type;
  int16_t integrated_loudness;
  int16_t digital_peak;
  int16_t true_peak;
  uint8_t num_anchor_loudness;
  anchor_loudness_t *anchor_loudness;
} IAMF_LoudnessInfo;

typedef enum {
  IAMF_CODEC_UNKNOWN = 0,
  IAMF_CODEC_OPUS,
  IAMF_CODEC_AAC,
  IAMF_CODEC_FLAC,
  IAMF_CODEC_PCM,
  IAMF_CODEC_COUNT
} IAMF_CodecID;

/* error codes (reference :181-190) */
enum {
  IAMF_OK = 0,
  IAMF_ERR_BAD_ARG = -1,
  IAMF_ERR_BUFFER_TOO_SMALL = -2,
  IAMF_ERR_INTERNAL = -3,
  IAMF_ERR_INVALID_PACKET = -4,
  IAMF_ERR_INVALID_STATE = -5,
  IAMF_ERR_UNIMPLEMENTED = -6,
  IAMF_ERR_ALLOC_FAIL = -7,
};

/* loudspeaker_layout of a channel-based layer (reference :196-209) */
typedef enum {
  IA_CHANNEL_LAYOUT_INVALID = -1,
  IA_CHANNEL_LAYOUT_MONO = 0, /* 1.0.0 */
  IA_CHANNEL_LAYOUT_STEREO,   /* 2.0.0 */
  IA_CHANNEL_LAYOUT_510,      /* 5.1.0 */
  IA_CHANNEL_LAYOUT_512,      /* 5.1.2 */
  IA_CHANNEL_LAYOUT_514,      /* 5.1.4 */
  IA_CHANNEL_LAYOUT_710,      /* 7.1.0 */
  IA_CHANNEL_LAYOUT_712,      /* 7.1.2 */
  IA_CHANNEL_LAYOUT_714,      /* 7.1.4 */
  IA_CHANNEL_LAYOUT_312,      /* 3.1.2 */
  IA_CHANNEL_LAYOUT_BINAURAL, /* binaural */
  IA_CHANNEL_LAYOUT_COUNT
} IAChannelLayoutType;

#endif /* IAMF_DEFINES_H */
