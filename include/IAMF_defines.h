/*
 * IAMF_defines.h — public constants and PODs of the IAMF decoder API as implemented by
 * libiamf_hip.so.  Source-compatible with the reference header of the same name
 * (Samsung/iac include/IAMF_defines.h:36-215): the same identifiers with the same values and the
 * same struct layouts, so code written against the reference compiles and links unchanged.
 * Written for this library (values listed explicitly), not taken over from the reference file.
 */
#ifndef IAMF_DEFINES_H
#define IAMF_DEFINES_H

#include <stdint.h>

/* ---- result codes of every API function (reference :181-190) ---- */
enum {
  IAMF_OK = 0,
  IAMF_ERR_BAD_ARG = -1, IAMF_ERR_BUFFER_TOO_SMALL = -2, IAMF_ERR_INTERNAL = -3, IAMF_ERR_INVALID_PACKET = -4,
  IAMF_ERR_INVALID_STATE = -5, IAMF_ERR_UNIMPLEMENTED = -6, IAMF_ERR_ALLOC_FAIL = -7,
};

/* ---- output layouts: ITU-R BS.2051 sound systems A..J (loudspeakers as top+middle+bottom) plus the
 *      IAMF extensions (reference :62-78); layout() = 2-bit type, then the 4-bit sound system for
 *      loudspeaker layouts (:56-60,86-119) ---- */
typedef enum IAMF_SoundSystem {
  SOUND_SYSTEM_INVALID = -1,
  SOUND_SYSTEM_A = 0 /* 0+2+0 */, SOUND_SYSTEM_B = 1 /* 0+5+0 */, SOUND_SYSTEM_C = 2 /* 2+5+0 */,
  SOUND_SYSTEM_D = 3 /* 4+5+0 */, SOUND_SYSTEM_E = 4 /* 4+5+1 */, SOUND_SYSTEM_F = 5 /* 3+7+0 */,
  SOUND_SYSTEM_G = 6 /* 4+9+0 */, SOUND_SYSTEM_H = 7 /* 9+10+3 */, SOUND_SYSTEM_I = 8 /* 0+7+0 */,
  SOUND_SYSTEM_J = 9 /* 4+7+0 */, SOUND_SYSTEM_EXT_712 = 10 /* 2+7+0 */, SOUND_SYSTEM_EXT_312 = 11 /* 2+3+0 */,
  SOUND_SYSTEM_MONO = 12 /* 0+1+0 */, SOUND_SYSTEM_END = 13
} IAMF_SoundSystem;

typedef enum IAMF_LayoutType {
  IAMF_LAYOUT_TYPE_NOT_DEFINED = 0, IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION = 2, IAMF_LAYOUT_TYPE_BINAURAL = 3,
} IAMF_LayoutType;

typedef struct IAMF_Layout { /* one byte, the bit-fields of layout() */
  union {
    struct { uint8_t reserved : 2; uint8_t sound_system : 4; uint8_t type : 2; } sound_system;
    struct { uint8_t reserved : 6; uint8_t type : 2; } binaural;
    struct { uint8_t reserved : 6; uint8_t type : 2; };
  };
} IAMF_Layout;

/* ---- loudness_info(): Q7.8 dB values (reference :121-154) ---- */
typedef struct _anchor_loudness_t { uint8_t anchor_element; int16_t anchored_loudness; } anchor_loudness_t;
typedef struct IAMF_LoudnessInfo {
  uint8_t info_type;
  int16_t integrated_loudness, digital_peak, true_peak;
  uint8_t num_anchor_loudness;
  anchor_loudness_t *anchor_loudness;
} IAMF_LoudnessInfo;

/* ---- descriptors: element kinds (reference :44-49), ambisonics modes, loudspeaker_layout of a
 *      channel-based layer (:196-209), parameter and animation types, codecs ---- */
typedef enum {
  AUDIO_ELEMENT_INVALID = -1, AUDIO_ELEMENT_CHANNEL_BASED = 0, AUDIO_ELEMENT_SCENE_BASED = 1, AUDIO_ELEMENT_COUNT = 2
} AudioElementType;

typedef enum AmbisonicsMode { AMBISONICS_MONO = 0, AMBISONICS_PROJECTION = 1 } AmbisonicsMode;

typedef enum {
  IA_CHANNEL_LAYOUT_INVALID = -1,
  IA_CHANNEL_LAYOUT_MONO = 0 /* 1.0.0 */, IA_CHANNEL_LAYOUT_STEREO = 1 /* 2.0.0 */, IA_CHANNEL_LAYOUT_510 = 2,
  IA_CHANNEL_LAYOUT_512 = 3, IA_CHANNEL_LAYOUT_514 = 4, IA_CHANNEL_LAYOUT_710 = 5, IA_CHANNEL_LAYOUT_712 = 6,
  IA_CHANNEL_LAYOUT_714 = 7, IA_CHANNEL_LAYOUT_312 = 8, IA_CHANNEL_LAYOUT_BINAURAL = 9, IA_CHANNEL_LAYOUT_COUNT = 10
} IAChannelLayoutType;

typedef enum IAMF_ParameterType {
  IAMF_PARAMETER_TYPE_MIX_GAIN = 0, IAMF_PARAMETER_TYPE_DEMIXING = 1, IAMF_PARAMETER_TYPE_RECON_GAIN = 2,
} IAMF_ParameterType;

typedef enum IAMF_AnimationType {
  ANIMATION_TYPE_INVALID = -1, ANIMATION_TYPE_STEP = 0, ANIMATION_TYPE_LINEAR = 1, ANIMATION_TYPE_BEZIER = 2
} IAMF_AnimationType;

typedef enum {
  IAMF_CODEC_UNKNOWN = 0, IAMF_CODEC_OPUS = 1, IAMF_CODEC_AAC = 2, IAMF_CODEC_FLAC = 3, IAMF_CODEC_PCM = 4,
  IAMF_CODEC_COUNT = 5
} IAMF_CodecID;

#endif /* IAMF_DEFINES_H */
