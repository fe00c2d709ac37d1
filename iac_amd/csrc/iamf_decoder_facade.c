/*
 * iamf_decoder_facade.c — the reference's public decoder API (include/IAMF_decoder.h) on top of
 * the GPU batch renderer (include/iamf_hip.h), one stream per handle.  Plain C host code.
 *
 * Host side (this file): OBU parsing (wire format as the reference parses it in
 * src/iamf_dec/IAMF_OBU.c:79-138,256-607,641-932,990-1248), LPCM unpacking
 * (pcm/IAMF_pcm_decoder.c:60-151), the audio-layer -> playback channel order of single-layer
 * channel-based elements (IAMF_utils.c:117-133,181-196), the mix-gain / demixing parameter
 * timeline (IAMF_decoder.c:857-982, downmix_renderer.c:180-216) and the call protocol of
 * IAMF_decoder.c:3726-4168.  Device side: everything from iamf_stream_render to
 * iamf_decoder_plane2stride_out (IAMF_decoder.c:3374-3500).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "IAMF_decoder.h"
#include "iamf_hip.h"

#define MAX_ELEMENTS 8
#define MAX_PRESENTATIONS 8
#define MAX_SUBSTREAMS 24
#define MAX_SEGMENTS 16
#define MAX_PARAMS 16
#define MAX_FRAME_SIZE (1u << 20) /* samples; LPCM frames of real streams are <= 8192 */

/* ---- byte reader: every field the path needs is byte aligned or the leading bits of a byte ---- */
typedef struct {
  const uint8_t *p;
  uint32_t size, pos;
  int err;
} Rd;

static uint8_t rd_u8(Rd *r) {
  if (r->pos >= r->size) {
    r->err = 1;
    return 0;
  }
  return r->p[r->pos++];
}
static uint16_t rd_u16(Rd *r) {
  uint16_t a = rd_u8(r);
  return (uint16_t)((a << 8) | rd_u8(r));
}
static uint32_t rd_u32(Rd *r) {
  uint32_t a = rd_u16(r);
  return (a << 16) | rd_u16(r);
}
static uint64_t rd_leb128(Rd *r) { /* bitstream.c:136-157 */
  uint64_t v = 0;
  for (int i = 0; i < 8; ++i) {
    uint8_t b = rd_u8(r);
    v |= ((uint64_t)(b & 0x7f)) << (7 * i);
    if (!(b & 0x80)) return v;
  }
  r->err = 1;
  return v;
}
static void rd_skip(Rd *r, uint64_t n) {
  if (r->pos + n > r->size)
    r->err = 1;
  else
    r->pos += (uint32_t)n;
}
static void rd_string(Rd *r) {
  while (r->pos < r->size && r->p[r->pos]) r->pos++;
  rd_skip(r, 1);
}

typedef struct {
  int type, redundant, trimming, extension;
  uint64_t trim_end, trim_start;
  const uint8_t *payload;
  uint32_t payload_size, size;
} Obu;

/* IAMF_OBU_split, IAMF_OBU.c:79-138 */
static uint32_t obu_split(const uint8_t *data, uint32_t size, Obu *o) {
  Rd r = {data, size, 0, 0};
  uint8_t h;
  uint64_t len;
  if (size < 2) return 0;
  h = rd_u8(&r);
  o->type = h >> 3;
  o->redundant = (h >> 2) & 1;
  o->trimming = (h >> 1) & 1;
  o->extension = h & 1;
  len = rd_leb128(&r);
  if (r.err || len + r.pos > size) return 0;
  o->size = r.pos + (uint32_t)len;
  r.size = o->size;
  o->trim_end = o->trim_start = 0;
  if (o->trimming) {
    o->trim_end = rd_leb128(&r);
    o->trim_start = rd_leb128(&r);
  }
  if (o->extension) rd_skip(&r, rd_leb128(&r));
  if (r.err) return 0;
  o->payload = data + r.pos;
  o->payload_size = o->size - r.pos;
  return o->size;
}

/* ---- parsed descriptors ---- */
typedef struct {
  uint64_t id, rate;
  int mode; /* 1: the parameter blocks carry duration / intervals themselves */
  uint64_t duration, constant_interval, nb_segments, intervals[MAX_SEGMENTS];
} ParamDef;

typedef struct {
  int anim;
  float start, end, control, control_rel; /* linear gains, relative time in [0,1] */
  uint64_t interval;
} GainSeg;

#define MAX_QUEUE 64
#define MAX_LAYERS 6
/* recon-gain segment (IAMF_OBU.c:1142-1195): per layer the flags and the gains of the flagged channels */
typedef struct {
  uint32_t flags[MAX_LAYERS];
  float gain[MAX_LAYERS][12];
} ReconSeg;
/* One parameter's timeline, kept the way the reference's database keeps it (IAMF_decoder.c:984-1125):
 * a FIFO of segments; `timestamp` is the start time of the oldest queued segment and only moves
 * when iamf_database_parameters_time_elapse pops fully elapsed segments after a frame. */
typedef struct {
  uint64_t id;
  int type; /* IAMF_PARAMETER_TYPE_* */
  ParamDef def;
  int use_default; /* mix gain: no block received yet */
  uint64_t timestamp, duration, elapse;
  int qn;
  GainSeg q[MAX_QUEUE]; /* demixing segments use .anim as the mode */
  ReconSeg *rq;         /* recon-gain parameters: [MAX_QUEUE], parallel to q */
} Param;

/* one layer of a scalable channel audio element (IAMF_OBU.c:491-530) */
typedef struct {
  int layout, out_gain_flag, recon_flag, nsub, ncoupled;
  int gain_flags; /* 6 bits over IAOutputGainChannel */
  int16_t gain_q;
} Layer;

typedef struct {
  uint64_t id;
  int type; /* AudioElementType */
  int nsub;
  uint64_t sub_ids[MAX_SUBSTREAMS];
  int layout, nb_coupled;              /* channel-based */
  int amb_channels;
  uint8_t amb_map[MAX_SUBSTREAMS];     /* scene-based, mono mapping */
  int amb_projection, amb_coupled;     /* scene-based, projection mode */
  float proj[MAX_SUBSTREAMS * 2 * 16]; /* [decoded channel][ambisonics channel] */
  int has_demix;
  uint64_t demix_pid;
  int demix_default_mode, demix_default_w;
  int channels;
  int nlayers;
  Layer layer[MAX_LAYERS];
  int has_recon;
  uint64_t recon_pid;
} Element;

enum { MAX_ANCHORS = 16 };
typedef struct {
  uint64_t id;
  int nel;
  uint64_t el_id[2];
  ParamDef el_gain_def[2];
  int16_t el_gain_q[2];
  ParamDef out_gain_def;
  int16_t out_gain_q;
  int nlayouts;
  int layout_type[8], layout_ss[8];
  IAMF_LoudnessInfo loud[8];              /* anchor_loudness stays NULL here: the values live in anchors[] */
  anchor_loudness_t anchors[8][MAX_ANCHORS];
  int swapped; /* the two element entries above were exchanged for the batch (setup_pipeline): batch order != stream order */
} Presentation;

/* The per-stream stage one element can need in front of its renderer: the demixer of a scalable / output-gained channel
 * element (IAMF_decoder.c:2324-2386) or the parametric down-mixer (:2448-2478), with what the parameter blocks left for it */
typedef struct {
  int use_dmx, dmx_mode;
  iamf_hip_dmx_state dmx;
  int use_demix, demix_layer, demix_nsub, demix_nch;
  iamf_hip_demix_state dmst;
  uint32_t rec_flags;
  int rec_n;
  int32_t rec_ch[12];
  float rec_gain[12];
  uint32_t layer_rec_flags[MAX_LAYERS]; /* latest recon-gain block, per layer (ctx->conf_s[i].recon_gain) */
  float layer_rec_gain[MAX_LAYERS][12];
  iamf_hip_demix_frame *h_demix; /* pinned */
  iamf_hip_dmx_frame *h_dmx;     /* pinned */
  iamf_hip_demix_config dc_sig;  /* the demixer configuration, if use_demix */
} Pre;

struct IAMF_Decoder {
  /* user settings (IAMF_decoder.c:3726-3744, 3960-4130) */
  int out_type; /* IAMF_LayoutType */
  IAMF_SoundSystem out_ss;
  uint32_t bit_depth, out_rate;
  int limiter_on;
  float limiter_db, norm_loudness;
  int64_t mix_id;
  int64_t pts;
  uint32_t pts_base;
  /* descriptors */
  int have_header, have_codec;
  uint32_t frame_size, sample_size, rate;
  int little_endian;
  int nel, npr, nparam;
  Element el[MAX_ELEMENTS];
  Presentation pr[MAX_PRESENTATIONS];
  Param param[MAX_PARAMS];
  /* runtime */
  int configured;
  Presentation *sel;
  Element *sel_el[2];
  Param *el_gain_p[2], *out_gain_p;
  IAMF_StreamInfo info;
  int out_channels;
  float mix_loudness;
  iamf_hip_batch *batch, *batch3; /* batch3: limiter stage behind the resampler */
  iamf_hip_resampler *rs;
  int need_cfg;     /* ctx->need_configure: bit 0 the output layout, bit 1 the mix presentation changed since the last configure */
  int frame_channels0; /* channels of the layout the presentation was enabled with: the mixed frame's (setup_pipeline) */
  int rs_channels, reopen_rs, rs_inherited; /* the channels `rs` was opened for; the next setup_pipeline re-opens it (TV layout switch) */
  Pre pre[2]; /* [0]: the stage in front of element 0 of `batch`; [1]: of element 1, in front of `aux` */
  /* Both elements need a stage: element 1 is rendered (stage, matrix, its mix gain) by a batch of its own into f32 and
   * handed to `batch` as a second element with the identity matrix and gain 1 (setup_pipeline) */
  iamf_hip_batch *aux;
  int dmx_last;      /* the element (batch order) whose down-mixer was opened last: dmx_shared_coefficients */
  float dmx_static_s; /* the TL / TR factor the reference's shared table holds */
  float *d_aux_il, *d_aux_pl; /* device: aux's frame as it writes it [sample][channel], and planar for `batch` */
  float aux_gain_set;
  iamf_hip_batch_config aux_sig; /* what `aux` was created from (mat pointer zeroed), and its matrix: as cfg_sig / cfg_mat */
  const float *aux_mat;
  /* packets of the temporal unit being assembled */
  uint8_t *pkt[2][MAX_SUBSTREAMS];
  uint32_t pkt_len[2][MAX_SUBSTREAMS];
  uint32_t pkt_cap[2][MAX_SUBSTREAMS]; /* bytes allocated: a packet buffer only ever grows (no allocator call per frame) */
  /* while the handle is in a group: where sub-stream s's packet goes instead (its slot in the group's pinned upload row),
   * the slot's size, and whether the current packet is there (iamf_decoder_group.inc) */
  uint8_t *pkt_ext[2][MAX_SUBSTREAMS];
  uint32_t pkt_ext_cap[2][MAX_SUBSTREAMS];
  uint8_t pkt_in_ext[2][MAX_SUBSTREAMS];
  int pkt_have[2][MAX_SUBSTREAMS];
  uint64_t tu_trim_start, tu_trim_end;
  uint64_t timestamp; /* stream time of the next frame, samples */
  /* buffers */
  /* pinned host memory that the kernels read and write directly (a frame is 64 KB in, 4 KB out: over PCIe inside the
   * one render launch, instead of three copies around it — each ~9 us of call and ~12 us of engine hand-over) */
  float *h_in[2], *h_ramp[4], *d_mid, *d_res; /* ramps: element 0, element 1 for `batch`, output, element 1 for `aux` */
  /* element 0's packets as they are, pinned, for the fused LPCM form of the render kernel (iamf_hip_batch_render_lpcm):
   * one mono-coded ambisonics element in 16-bit little-endian LPCM into one or two channels with the limiter on */
  uint8_t *h_raw;
  int lp_ok;
  iamf_hip_lpcm_layout lp_layout;
  float gain_set[2]; /* element / output constant gains the batch holds (iamf_hip_batch_set_gains synchronises: only on change) */
  float *tmp; /* [MAX_SUBSTREAMS * 2][frame_size] unpack scratch */
  void *h_pcm; /* pinned: the render kernels write it, decode copies the caller's share out */
  volatile uint32_t *h_done; /* pinned word a one-lane kernel writes behind the frame's launches (facade_wait) */
  uint32_t done_seq;
  size_t pcm_cap;
  hipStream_t stream;
  int flushed;
  uint32_t last_frame;
  int pcm_stride, pcm_extra; /* channels per PCM sample-frame; surplus elements past the last frame (TV, > 12 ch) */
  int tv;           /* behaves as the reference built -DSAMSUNG_TV (iamf_hip_decoder_set_variant) */
  int lfe_hoa;      /* HOA LFE generator on: the reference built -DDISABLE_LFE_HOA=0 (ae_rdr.h:63-65) */
  struct iamf_hip_decoder_group *group; /* the handle renders through a group's batch (iamf_decoder_group.inc) */
  iamf_hip_batch_config cfg_sig;        /* what setup_pipeline created the batch from (mat pointer zeroed) ... */
  const float *cfg_mat;                 /* ... and its matrix */
  /* IAMF_decoder_get_last_metadata (IAMF_decoder.c:3619-3706,4150-4168) */
  uint64_t meta_duration;   /* ctx->duration: samples returned since IAMF_decoder_set_pts */
  int el_dmx_mode[2];       /* cctx->dmx_mode of the presentation's elements (batch order), -1 = none yet */
  uint32_t meta_dmixp;      /* ctx->metadata.param->dmixp_mode */
  int started;      /* a configure call with data has been made: status left INIT */
  int need_reconf;  /* a new IA sequence header was met while decoding: status RECONFIGURE */
};

/* ---- small tables ---- */
static const int k_ss_channels[] = {2, 6, 8, 10, 11, 12, 14, 24, 8, 12, 10, 6, 1}; /* IAMF_decoder.c:208-219 */
static const int k_ss_rid[] = {IAMF_HIP_SS_A, IAMF_HIP_SS_B, IAMF_HIP_SS_C, IAMF_HIP_SS_D, IAMF_HIP_SS_E,
                               IAMF_HIP_SS_F, IAMF_HIP_SS_G, IAMF_HIP_SS_H, IAMF_HIP_SS_I, IAMF_HIP_SS_J,
                               IAMF_HIP_L_712, IAMF_HIP_L_312, IAMF_HIP_L_MONO}; /* :221-226 */
static const int k_layer_rid[] = {IAMF_HIP_L_MONO, IAMF_HIP_L_STEREO, IAMF_HIP_L_51, IAMF_HIP_L_512,
                                  IAMF_HIP_L_514, IAMF_HIP_L_71, IAMF_HIP_L_712, IAMF_HIP_L_714,
                                  IAMF_HIP_L_312, IAMF_HIP_L_BINAURAL}; /* :256-261 */
static const int k_ss_layout[] = {IA_CHANNEL_LAYOUT_STEREO, IA_CHANNEL_LAYOUT_510, IA_CHANNEL_LAYOUT_512,
                                  IA_CHANNEL_LAYOUT_514, -1, -1, -1, -1, IA_CHANNEL_LAYOUT_710,
                                  IA_CHANNEL_LAYOUT_714, IA_CHANNEL_LAYOUT_712, IA_CHANNEL_LAYOUT_312,
                                  IA_CHANNEL_LAYOUT_MONO}; /* :228-238 */
static const int k_layout_channels[] = {1, 2, 6, 8, 10, 8, 10, 12, 6, 2};
/* playback channel p of a layer layout is audio-layer channel k_al_of_pl[layout][p]
 * (IAMF_utils.c:117-133 vs :181-196) */
static const int k_al_of_pl[9][12] = {
    {0}, {0, 1}, {0, 1, 4, 5, 2, 3}, {0, 1, 6, 7, 2, 3, 4, 5}, {0, 1, 8, 9, 2, 3, 4, 5, 6, 7},
    {0, 1, 6, 7, 2, 3, 4, 5}, {0, 1, 8, 9, 2, 3, 4, 5, 6, 7}, {0, 1, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9},
    {0, 1, 4, 5, 2, 3}};

/* IAChannel ids (IAMF_types.h:61-90) */
enum {
  CH_INVALID, CH_L7, CH_R7, CH_C, CH_LFE, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR, CH_HBL,
  CH_HBR, CH_MONO, CH_L2, CH_R2, CH_TL, CH_TR, CH_L3, CH_R3, CH_SL5, CH_SR5, CH_HL, CH_HR,
  CH_L5 = CH_L7, CH_R5 = CH_R7
};
static const int k_layout_surround[9] = {1, 2, 5, 5, 5, 7, 7, 7, 3}; /* IAMF_utils.c:157-161 */
static const int k_layout_top[9] = {0, 0, 0, 2, 4, 0, 2, 4, 2};
static const int k_al_channels[9][12] = { /* audio-layer order, IAMF_utils.c:181-196 */
    {CH_MONO},
    {CH_L2, CH_R2},
    {CH_L5, CH_R5, CH_SL5, CH_SR5, CH_C, CH_LFE},
    {CH_L5, CH_R5, CH_SL5, CH_SR5, CH_HL, CH_HR, CH_C, CH_LFE},
    {CH_L5, CH_R5, CH_SL5, CH_SR5, CH_HFL, CH_HFR, CH_HBL, CH_HBR, CH_C, CH_LFE},
    {CH_L7, CH_R7, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_C, CH_LFE},
    {CH_L7, CH_R7, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HL, CH_HR, CH_C, CH_LFE},
    {CH_L7, CH_R7, CH_SL7, CH_SR7, CH_BL7, CH_BR7, CH_HFL, CH_HFR, CH_HBL, CH_HBR, CH_C, CH_LFE},
    {CH_L3, CH_R3, CH_TL, CH_TR, CH_C, CH_LFE}};
/* layer layout -> IAMF_SoundSystem (iamf_layer_layout_convert_sound_system, IAMF_decoder.c:268-275) */
static const int k_layout_ss[9] = {SOUND_SYSTEM_MONO, SOUND_SYSTEM_A, SOUND_SYSTEM_B, SOUND_SYSTEM_C, SOUND_SYSTEM_D,
                                   SOUND_SYSTEM_I, SOUND_SYSTEM_EXT_712, SOUND_SYSTEM_J, SOUND_SYSTEM_EXT_312};

/* iamf_channel_layout_get_new_channels, IAMF_decoder.c:450-531: the channels a layer adds, in
 * the order its sub-streams decode to */
static int layer_new_channels(int last, int cur, int32_t *out) {
  int n = 0;
  if (last < 0) {
    for (int i = 0; i < k_layout_channels[cur]; ++i) out[n++] = k_al_channels[cur][i];
    return n;
  }
  {
    const int s1 = k_layout_surround[last], s2 = k_layout_surround[cur];
    const int t1 = k_layout_top[last], t2 = k_layout_top[cur];
    if (s1 < 5 && 5 <= s2) { out[n++] = CH_L5; out[n++] = CH_R5; }
    if (s1 < 7 && 7 <= s2) { out[n++] = CH_SL7; out[n++] = CH_SR7; }
    if (t2 != t1 && t2 == 4) { out[n++] = CH_HFL; out[n++] = CH_HFR; }
    if (t2 - t1 == 4) {
      out[n++] = CH_HBL; out[n++] = CH_HBR;
    } else if (!t1 && t2 - t1 == 2) {
      if (s2 < 5) { out[n++] = CH_TL; out[n++] = CH_TR; } else { out[n++] = CH_HL; out[n++] = CH_HR; }
    }
    if (s1 < 3 && 3 <= s2) { out[n++] = CH_C; out[n++] = CH_LFE; }
    if (s1 < 2 && 2 <= s2) out[n++] = CH_L2;
  }
  return n;
}

/* iamf_output_gain_channel_map, IAMF_decoder.c:533-600; g over IAOutputGainChannel
 * (RTF, LTF, RS, LS, R, L; IAMF_decoder_private.h:62-70) */
static int output_gain_channel(int layout, int g) {
  const int s = k_layout_surround[layout];
  switch (g) {
    case 5: return layout == 0 ? CH_MONO : layout == 1 ? CH_L2 : layout == 8 ? CH_L3 : CH_INVALID;
    case 4: return layout == 1 ? CH_R2 : layout == 8 ? CH_R3 : CH_INVALID;
    case 3: return s == 5 ? CH_SL5 : CH_INVALID;
    case 2: return s == 5 ? CH_SR5 : CH_INVALID;
    case 1: return s < 5 ? CH_TL : CH_HL;
    case 0: return s < 5 ? CH_TR : CH_HR;
    default: return CH_INVALID;
  }
}

/* iamf_recon_channels_get_flags, IAMF_decoder.c:371-407 (bits over IAReconChannel, IAMF_types.h:37-60) */
static uint32_t recon_default_flags(int l1, int l2) {
  uint32_t f = 0;
  int s1, s2, t1, t2;
  if (l1 == l2) return 0;
  s1 = k_layout_surround[l1]; s2 = k_layout_surround[l2];
  t1 = k_layout_top[l1]; t2 = k_layout_top[l2];
  if (s1 != s2) {
    if (s2 <= 3) f |= (1u << 0) | (1u << 2);
    else if (s2 == 5) f |= (1u << 3) | (1u << 4);
    else if (s2 == 7) f |= (1u << 7) | (1u << 8);
  }
  if (t2 != t1 && t2 == 4) f |= (1u << 9) | (1u << 10);
  if (s2 == 5 && t1 && t2 == t1) f |= (1u << 5) | (1u << 6);
  return f;
}

/* iamf_recon_channels_order_update, IAMF_decoder.c:409-448 */
static int recon_channel_order(int layout, uint32_t flags, int32_t *out) {
  static const int map[9][12] = {
      {CH_MONO, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
      {CH_L2, 0, CH_R2, 0, 0, 0, 0, 0, 0, 0, 0, 0},
      {CH_L5, CH_C, CH_R5, CH_SL5, CH_SR5, 0, 0, 0, 0, 0, 0, CH_LFE},
      {CH_L5, CH_C, CH_R5, CH_SL5, CH_SR5, CH_HL, CH_HR, 0, 0, 0, 0, CH_LFE},
      {CH_L5, CH_C, CH_R5, CH_SL5, CH_SR5, CH_HFL, CH_HFR, 0, 0, CH_HBL, CH_HBR, CH_LFE},
      {CH_L7, CH_C, CH_R7, CH_SL7, CH_SR7, 0, 0, CH_BL7, CH_BR7, 0, 0, CH_LFE},
      {CH_L7, CH_C, CH_R7, CH_SL7, CH_SR7, CH_HL, CH_HR, CH_BL7, CH_BR7, 0, 0, CH_LFE},
      {CH_L7, CH_C, CH_R7, CH_SL7, CH_SR7, CH_HFL, CH_HFR, CH_BL7, CH_BR7, CH_HBL, CH_HBR, CH_LFE},
      {CH_L3, CH_C, CH_R3, 0, 0, CH_TL, CH_TR, 0, 0, 0, 0, CH_LFE}};
  int n = 0;
  for (int c = 0; c < 12; ++c) /* recon_channel_order[] is the enum order */
    if (flags & (1u << c)) out[n++] = map[layout][c];
  return n;
}

static int popcount32(uint32_t v) {
  int n = 0;
  for (; v; ++n) v &= v - 1;
  return n;
}

static float q_to_float(int16_t q, int frac) { return ((float)q) * powf(2.0f, (float)-frac); } /* fixedp11_5.c:45 */
static float qf_to_float(uint8_t q, int frac) { return ((float)q / (pow(2.0f, (float)frac) - 1.0)); } /* :53 */
static float db2lin(float db) { return powf(10.0f, 0.05f * db); }                               /* :72 */

/* ---- descriptor parsing ---- */
static int parse_param_def(Rd *r, ParamDef *d) { /* IAMF_OBU.c:358-389 */
  memset(d, 0, sizeof(*d));
  d->id = rd_leb128(r);
  d->rate = rd_leb128(r);
  d->mode = rd_u8(r) >> 7;
  if (!d->mode) {
    d->duration = rd_leb128(r);
    d->constant_interval = rd_leb128(r);
    if (!d->constant_interval) {
      d->nb_segments = rd_leb128(r);
      if (d->nb_segments > MAX_SEGMENTS) return IAMF_ERR_UNIMPLEMENTED;
      for (uint64_t i = 0; i < d->nb_segments; ++i) d->intervals[i] = rd_leb128(r);
    } else {
      d->nb_segments = (d->duration + d->constant_interval - 1) / d->constant_interval;
      if (d->nb_segments > MAX_SEGMENTS) return IAMF_ERR_UNIMPLEMENTED;
    }
  }
  return r->err ? IAMF_ERR_INVALID_PACKET : IAMF_OK;
}

static Param *param_get(struct IAMF_Decoder *d, const ParamDef *def, int type) {
  for (int i = 0; i < d->nparam; ++i)
    if (d->param[i].id == def->id) return &d->param[i];
  if (d->nparam >= MAX_PARAMS) return 0;
  Param *p = &d->param[d->nparam++];
  memset(p, 0, sizeof(*p));
  p->id = def->id;
  p->type = type;
  p->def = *def;
  p->use_default = type == IAMF_PARAMETER_TYPE_MIX_GAIN;
  return p;
}

static int parse_codec_config(struct IAMF_Decoder *d, const Obu *o) { /* IAMF_OBU.c:303-343 */
  Rd r = {o->payload, o->payload_size, 0, 0};
  char cc[4];
  uint64_t fs64;
  rd_leb128(&r);
  for (int i = 0; i < 4; ++i) cc[i] = (char)rd_u8(&r);
  fs64 = rd_leb128(&r);
  d->frame_size = fs64 > MAX_FRAME_SIZE ? 0 : (uint32_t)fs64;
  rd_u16(&r); /* roll distance */
  if (memcmp(cc, "ipcm", 4)) return IAMF_ERR_UNIMPLEMENTED; /* Opus / AAC / FLAC: upstream of this path */
  d->little_endian = rd_u8(&r) & 1; /* pcm/IAMF_pcm_decoder.c:60-62 */
  d->sample_size = rd_u8(&r);
  d->rate = rd_u32(&r);
  if (r.err || !fs64 || (d->sample_size != 16 && d->sample_size != 24 && d->sample_size != 32))
    return IAMF_ERR_INVALID_PACKET;
  if (fs64 > MAX_FRAME_SIZE) return IAMF_ERR_UNIMPLEMENTED; /* every per-frame buffer is sized from it */
  d->have_codec = 1;
  return IAMF_OK;
}

static int parse_element(struct IAMF_Decoder *d, const Obu *o) { /* IAMF_OBU.c:391-607 */
  Rd r = {o->payload, o->payload_size, 0, 0};
  Element e;
  uint64_t np;
  memset(&e, 0, sizeof(e));
  e.id = rd_leb128(&r);
  e.type = rd_u8(&r) >> 5;
  rd_leb128(&r); /* codec config id */
  {
    /* untrusted: compare as uint64 before narrowing; an element without sub-streams would make every
     * temporal unit "complete" and leave the unpacker without a sample count */
    const uint64_t nsub = rd_leb128(&r);
    if (r.err || nsub < 1) return IAMF_ERR_INVALID_PACKET;
    if (nsub > MAX_SUBSTREAMS) return IAMF_ERR_UNIMPLEMENTED;
    e.nsub = (int)nsub;
  }
  for (int i = 0; i < e.nsub; ++i) e.sub_ids[i] = rd_leb128(&r);
  np = rd_leb128(&r);
  for (uint64_t i = 0; i < np; ++i) {
    uint64_t type = rd_leb128(&r);
    if (type == IAMF_PARAMETER_TYPE_DEMIXING || type == IAMF_PARAMETER_TYPE_RECON_GAIN) {
      ParamDef def;
      int rc = parse_param_def(&r, &def);
      if (rc) return rc;
      if (type == IAMF_PARAMETER_TYPE_DEMIXING) {
        e.has_demix = 1;
        e.demix_pid = def.id;
        e.demix_default_mode = rd_u8(&r) >> 5;
        e.demix_default_w = rd_u8(&r) >> 4;
        if (!param_get(d, &def, IAMF_PARAMETER_TYPE_DEMIXING)) return IAMF_ERR_ALLOC_FAIL;
      } else {
        Param *rp = param_get(d, &def, IAMF_PARAMETER_TYPE_RECON_GAIN);
        if (!rp) return IAMF_ERR_ALLOC_FAIL;
        if (!rp->rq) rp->rq = (ReconSeg *)calloc(MAX_QUEUE, sizeof(ReconSeg));
        if (!rp->rq) return IAMF_ERR_ALLOC_FAIL;
        e.has_recon = 1;
        e.recon_pid = def.id;
      }
    } else {
      rd_skip(&r, rd_leb128(&r));
    }
  }
  if (e.type == AUDIO_ELEMENT_CHANNEL_BASED) {
    int layers = rd_u8(&r) >> 5, subs = 0;
    if (layers < 1 || layers > MAX_LAYERS) return IAMF_ERR_UNIMPLEMENTED;
    e.nlayers = layers;
    for (int i = 0; i < layers; ++i) { /* IAMF_OBU.c:505-530 */
      Layer *l = &e.layer[i];
      uint8_t b = rd_u8(&r);
      l->layout = b >> 4;
      l->out_gain_flag = (b >> 3) & 1;
      l->recon_flag = (b >> 2) & 1;
      l->nsub = rd_u8(&r);
      l->ncoupled = rd_u8(&r);
      if (l->out_gain_flag) {
        l->gain_flags = rd_u8(&r) >> 2;
        l->gain_q = (int16_t)rd_u16(&r);
      }
      if (l->layout > IA_CHANNEL_LAYOUT_312) return IAMF_ERR_UNIMPLEMENTED;
      if (l->nsub < 1 || l->ncoupled > l->nsub) return IAMF_ERR_INVALID_PACKET;
      subs += l->nsub;
    }
    if (subs != e.nsub) return IAMF_ERR_INVALID_PACKET;
    /* the highest layer until an output layout selects another (iamf_stream_new, IAMF_decoder.c:1721-1724) */
    e.layout = e.layer[layers - 1].layout;
    e.nb_coupled = e.layer[0].ncoupled;
    e.channels = k_layout_channels[e.layout];
    if (layers == 1 && e.nsub + e.nb_coupled != e.channels) return IAMF_ERR_INVALID_PACKET;
  } else if (e.type == AUDIO_ELEMENT_SCENE_BASED) {
    uint64_t mode = rd_leb128(&r);
    if (mode == AMBISONICS_MONO) {
      e.amb_channels = rd_u8(&r);
      if (rd_u8(&r) != e.nsub) return IAMF_ERR_INVALID_PACKET; /* substream_count of the mono mapping */
      if (e.amb_channels > MAX_SUBSTREAMS) return IAMF_ERR_INVALID_PACKET;
      for (int i = 0; i < e.amb_channels; ++i) e.amb_map[i] = rd_u8(&r);
    } else if (mode == AMBISONICS_PROJECTION) { /* IAMF_OBU.c:562-583, IAMF_core_decoder.c:228-252 */
      int subs, l_in;
      e.amb_projection = 1;
      e.amb_channels = rd_u8(&r);
      subs = rd_u8(&r);
      e.amb_coupled = rd_u8(&r);
      l_in = subs + e.amb_coupled;
      if (e.amb_channels > 16 || l_in > MAX_SUBSTREAMS * 2 || subs != e.nsub || e.amb_coupled > subs)
        return IAMF_ERR_INVALID_PACKET;
      for (int i = 0; i < l_in * e.amb_channels; ++i) e.proj[i] = q_to_float((int16_t)rd_u16(&r), 15);
    } else {
      return IAMF_ERR_UNIMPLEMENTED;
    }
    e.channels = e.amb_channels;
    if (e.channels != 1 && e.channels != 4 && e.channels != 9 && e.channels != 16) return IAMF_ERR_UNIMPLEMENTED;
  } else {
    return IAMF_ERR_UNIMPLEMENTED;
  }
  if (r.err) return IAMF_ERR_INVALID_PACKET;
  for (int i = 0; i < d->nel; ++i)
    if (d->el[i].id == e.id) return IAMF_OK; /* "already in database": the first one stays (iamf_database_element_add, :1136-1140) */
  if (d->nel >= MAX_ELEMENTS) return IAMF_ERR_UNIMPLEMENTED;
  d->el[d->nel++] = e;
  return IAMF_OK;
}

static int parse_presentation(struct IAMF_Decoder *d, const Obu *o) { /* IAMF_OBU.c:641-932 */
  Rd r = {o->payload, o->payload_size, 0, 0};
  Presentation p;
  uint64_t labels;
  memset(&p, 0, sizeof(p));
  p.id = rd_leb128(&r);
  labels = rd_leb128(&r);
  for (uint64_t i = 0; i < 2 * labels; ++i) rd_string(&r);
  if (rd_leb128(&r) != 1) return IAMF_ERR_UNIMPLEMENTED; /* one sub-mix, as the reference */
  p.nel = (int)rd_leb128(&r);
  if (p.nel < 1 || p.nel > 2) return IAMF_ERR_UNIMPLEMENTED;
  for (int i = 0; i < p.nel; ++i) {
    int rc;
    p.el_id[i] = rd_leb128(&r);
    for (uint64_t k = 0; k < labels; ++k) rd_string(&r);
    rd_u8(&r); /* headphones_rendering_mode: only matters with an external binauraliser */
    rd_skip(&r, rd_leb128(&r));
    if ((rc = parse_param_def(&r, &p.el_gain_def[i]))) return rc;
    p.el_gain_q[i] = (int16_t)rd_u16(&r);
  }
  {
    int rc = parse_param_def(&r, &p.out_gain_def);
    if (rc) return rc;
    p.out_gain_q = (int16_t)rd_u16(&r);
  }
  p.nlayouts = (int)rd_leb128(&r);
  if (p.nlayouts > 8) return IAMF_ERR_UNIMPLEMENTED;
  for (int i = 0; i < p.nlayouts; ++i) {
    uint8_t b = rd_u8(&r);
    p.layout_type[i] = b >> 6;
    p.layout_ss[i] = (b >> 2) & 15;
    p.loud[i].info_type = rd_u8(&r);
    p.loud[i].integrated_loudness = (int16_t)rd_u16(&r);
    p.loud[i].digital_peak = (int16_t)rd_u16(&r);
    if (p.loud[i].info_type & 1) p.loud[i].true_peak = (int16_t)rd_u16(&r);
    if (p.loud[i].info_type & 2) { /* IAMF_OBU.c:890-912 */
      int n = rd_u8(&r);
      if (n > MAX_ANCHORS) return IAMF_ERR_UNIMPLEMENTED; /* the specification defines three anchor elements */
      p.loud[i].num_anchor_loudness = (uint8_t)n;
      for (int k = 0; k < n; ++k) {
        p.anchors[i][k].anchor_element = rd_u8(&r);
        p.anchors[i][k].anchored_loudness = (int16_t)rd_u16(&r);
      }
    }
    if (p.loud[i].info_type & ~3) rd_skip(&r, rd_leb128(&r));
  }
  if (r.err) return IAMF_ERR_INVALID_PACKET;
  /* every mix presentation OBU is kept, whatever its id (iamf_object_set_add, :1215): a lookup by id finds the first, the
   * matching score is taken over all of them (setup_pipeline) */
  if (d->npr >= MAX_PRESENTATIONS) return IAMF_ERR_UNIMPLEMENTED;
  d->pr[d->npr++] = p;
  return IAMF_OK;
}

/* IAMF_OBU.c:990-1215 (mix gain and demixing segments) */
static int parse_parameter_block(struct IAMF_Decoder *d, const Obu *o) {
  Rd r = {o->payload, o->payload_size, 0, 0};
  uint64_t id = rd_leb128(&r), duration, cinterval, nseg, left;
  Param *p = 0;
  for (int i = 0; i < d->nparam; ++i)
    if (d->param[i].id == id) p = &d->param[i];
  if (!p) return IAMF_OK; /* not a parameter of the selected presentation */
  if (!p->def.mode) {
    duration = p->def.duration;
    cinterval = p->def.constant_interval;
    nseg = p->def.nb_segments;
  } else {
    duration = rd_leb128(&r);
    cinterval = rd_leb128(&r);
    nseg = cinterval ? (duration + cinterval - 1) / cinterval : rd_leb128(&r);
  }
  if (nseg > MAX_SEGMENTS || p->qn + (int)nseg > MAX_QUEUE) return IAMF_ERR_UNIMPLEMENTED;
  left = duration;
  for (uint64_t i = 0; i < nseg; ++i) {
    GainSeg g;
    uint64_t iv = 0;
    memset(&g, 0, sizeof(g));
    if (!cinterval) iv = p->def.mode ? rd_leb128(&r) : p->def.intervals[i];
    if (!iv) iv = cinterval < left ? cinterval : left; /* iamf_parameter_get_segment_interval */
    left -= iv;
    g.interval = iv;
    if (p->type == IAMF_PARAMETER_TYPE_MIX_GAIN) {
      g.anim = (int)rd_leb128(&r);
      g.start = db2lin(q_to_float((int16_t)rd_u16(&r), 8));
      if (g.anim != ANIMATION_TYPE_STEP) {
        g.end = db2lin(q_to_float((int16_t)rd_u16(&r), 8));
        if (g.anim == ANIMATION_TYPE_BEZIER) {
          g.control = db2lin(q_to_float((int16_t)rd_u16(&r), 8));
          g.control_rel = qf_to_float(rd_u8(&r), 8);
        }
      }
    } else if (p->type == IAMF_PARAMETER_TYPE_DEMIXING) {
      g.anim = rd_u8(&r) >> 5;
    } else if (p->type == IAMF_PARAMETER_TYPE_RECON_GAIN) { /* IAMF_OBU.c:1142-1195 */
      const Element *re = 0;
      ReconSeg *rs = &p->rq[p->qn];
      for (int k = 0; k < d->nel; ++k)
        if (d->el[k].has_recon && d->el[k].recon_pid == p->id) re = &d->el[k];
      memset(rs, 0, sizeof(*rs));
      for (int k = 0; re && k < re->nlayers; ++k) {
        int n;
        if (!re->layer[k].recon_flag) continue;
        rs->flags[k] = (uint32_t)rd_leb128(&r);
        n = popcount32(rs->flags[k]);
        if (n > 12) return IAMF_ERR_INVALID_PACKET;
        for (int t = 0; t < n; ++t) rs->gain[k][t] = qf_to_float(rd_u8(&r), 8);
      }
    }
    if (r.err) return IAMF_ERR_INVALID_PACKET;
    /* iamf_database_parameter_add, IAMF_decoder.c:1042-1071 */
    p->q[p->qn++] = g;
    p->duration += iv;
  }
  p->use_default = 0;
  /* iamf_stream_decoder_update_parameter, IAMF_decoder.c:2141-2148: a recon-gain block updates the
   * per-layer gains from the segment covering the middle of the current frame */
  for (int e = 0; e < 2 && p->type == IAMF_PARAMETER_TYPE_RECON_GAIN; ++e) {
    const Element *se = d->sel && e < d->sel->nel ? d->sel_el[e] : 0;
    const uint64_t pts = d->timestamp + d->frame_size / 2;
    if (!se || !se->has_recon || se->recon_pid != p->id) continue;
    if (pts > p->timestamp && pts <= p->timestamp + p->duration) {
      uint64_t start = pts - p->timestamp;
      for (int i = 0; i < p->qn; ++i) {
        if (start < p->q[i].interval) { /* iamf_stream_scale_decoder_update_recon_gain, :2238-2274 */
          for (int k = 0; k < se->nlayers; ++k) {
            if (!se->layer[k].recon_flag) continue;
            d->pre[e].layer_rec_flags[k] = p->rq[i].flags[k];
            memcpy(d->pre[e].layer_rec_gain[k], p->rq[i].gain[k], sizeof(d->pre[e].layer_rec_gain[k]));
          }
          break;
        }
        start -= p->q[i].interval;
      }
    }
  }
  /* iamf_stream_decoder_update_parameter, IAMF_decoder.c:2130-2151: a demixing block sets the mode of every selected
   * channel-based element that names the parameter, from the segment covering the middle of the current frame */
  for (int e = 0; e < 2 && p->type == IAMF_PARAMETER_TYPE_DEMIXING; ++e) {
    if (!d->sel || e >= d->sel->nel || !d->sel_el[e] || !d->sel_el[e]->has_demix || d->sel_el[e]->demix_pid != p->id) continue;
    const uint64_t pts = d->timestamp + d->frame_size / 2;
    int mode = IAMF_ERR_INTERNAL;
    if (pts > p->timestamp && pts <= p->timestamp + p->duration) { /* :799-841 */
      uint64_t start = pts - p->timestamp;
      for (int i = 0; i < p->qn; ++i) {
        if (start < p->q[i].interval) {
          mode = p->q[i].anim;
          break;
        }
        start -= p->q[i].interval;
      }
    }
    d->el_dmx_mode[e] = mode;
    d->pre[e].dmx_mode = mode; /* what the element's demixer / down-mixer takes at its next frame */
  }
  return IAMF_OK;
}

/* iamf_database_parameters_time_elapse, IAMF_decoder.c:1089-1125 */
static void params_time_elapse(struct IAMF_Decoder *d, uint64_t duration) {
  for (int i = 0; i < d->nparam; ++i) {
    Param *p = &d->param[i];
    uint64_t e = duration;
    if (d->rate != p->def.rate) { /* time_transform, :90-94 */
      double r = (double)duration * p->def.rate;
      e = (uint64_t)(r / d->rate + 0.5f);
    }
    p->elapse += e;
    while (p->qn && p->q[0].interval <= p->elapse) {
      p->timestamp += p->q[0].interval;
      p->duration -= p->q[0].interval;
      p->elapse -= p->q[0].interval;
      memmove(&p->q[0], &p->q[1], sizeof(GainSeg) * (size_t)(p->qn - 1));
      if (p->rq) memmove(&p->rq[0], &p->rq[1], sizeof(ReconSeg) * (size_t)(p->qn - 1));
      p->qn--;
    }
  }
}

/* ---- per-frame mix gains: iamf_database_parameter_get_mix_gain_unit, IAMF_decoder.c:857-982,
 *      with the ramp builders of :639-664 ---- */
static void gain_linear(float s, float e, int d, int o, int l, float *g) {
  for (int i = o, k = 0; i < o + l; ++i, ++k) g[k] = s + (e - s) * i / d;
}
static void gain_quad(float s, float e, int d, float c, int ct, int o, int l, float *g) {
  int64_t alpha = d - 2 * ct;
  float a = 1.0f;
  for (int i = o, k = 0; i < o + l; ++i, ++k) {
    if (alpha) {
      a = (sqrt(pow(ct, 2) + alpha * i) - ct) / alpha;
    } else {
      a = i;
      a /= (2 * ct);
    }
    g[k] = (s + e - 2 * c) * pow(a, 2) + 2 * a * (c - s) + s;
  }
}

/* returns 0: constant gain in *constant; 1: per-sample gains in g[0..duration); -1: the unit
 * does not cover the frame, so iamf_frame_gain refuses it and NO gain is applied
 * (IAMF_decoder.c:1386-1390) */
static int mix_gain_unit(const Param *p, float default_gain, uint64_t pt, int duration, int rate, float *constant,
                         float *g) {
  uint64_t start = 0;
  float ratio = 1.f;
  int count = 0, have_array = 0, use_default = 0;
  int64_t sgd = 0;
  int left = duration;
  *constant = default_gain;
  if (!p) return 0;
  if (pt < p->timestamp)
    use_default = 1;
  else
    start = pt - p->timestamp;
  if (p->use_default || use_default) return 0;
  if ((uint64_t)rate != p->def.rate) ratio = (rate + 0.1f) / p->def.rate;
  for (int i = 0; i < p->qn; ++i) {
    const GainSeg *seg = &p->q[i];
    int64_t minterval = seg->interval * ratio;
    sgd += minterval;
    if ((int64_t)start < sgd) {
      if (seg->anim == ANIMATION_TYPE_STEP) {
        if (!count && (int64_t)(start + duration) <= sgd) {
          *constant = seg->start;
          return 0;
        } else if (!count) {
          have_array = 1;
          count = (int)(sgd - start);
          for (int k = 0; k < count; ++k) g[k] = seg->start;
          start = sgd;
        } else {
          int e = count + (int)minterval;
          if (e >= duration)
            e = duration;
          else
            start = sgd;
          for (int k = count; k < e; ++k) g[k] = seg->start;
          count = e;
        }
      } else {
        int ss = (int)(sgd - minterval), d2, off = (int)start - ss;
        have_array = 1;
        if ((int64_t)(start + left) <= sgd) {
          d2 = left;
        } else {
          d2 = (int)(sgd - start);
          start = sgd;
          left -= d2;
        }
        /* The reference counts what is `left` of the frame only in this branch: behind a STEP sub-block that filled part of
         * the frame it still takes the whole frame for what is left and writes past its gains[duration] (IAMF_decoder.c:
         * 921-960; found by tests/e2e_fuzz.py "params": it dies of heap corruption on such streams, or goes on with a
         * damaged heap).  Here the ramp ends where the frame ends. */
        if (d2 > duration - count) d2 = duration - count;
        if (d2 <= 0) break;
        if (seg->anim == ANIMATION_TYPE_LINEAR)
          gain_linear(seg->start, seg->end, (int)minterval, off, d2, g + count);
        else
          gain_quad(seg->start, seg->end, (int)minterval, seg->control,
                    seg->control_rel * (minterval + .1f), off, d2, g + count);
        count += d2;
      }
    }
    if (count == duration) break;
  }
  (void)have_array;
  if (count < duration) return -1;
  return 1;
}

/* ---- lifecycle ---- */
IAMF_DecoderHandle IAMF_decoder_open(void) { /* IAMF_decoder.c:3726-3744 */
  struct IAMF_Decoder *d = (struct IAMF_Decoder *)calloc(1, sizeof(*d));
  if (!d) return 0;
  d->limiter_on = 1;
  d->limiter_db = -1.0f; /* LIMITER_MaximumTruePeak */
  d->mix_id = -1;
  d->out_type = IAMF_LAYOUT_TYPE_NOT_DEFINED;
  d->pts_base = 90000;
  {
    /* the reference decides this at build time (DISABLE_LFE_HOA, default 1 = generator compiled out);
     * here it is a run-time switch with the same default: the environment, or iamf_hip_decoder_set_hoa_lfe */
    const char *e = getenv("IAMF_HIP_LFE_HOA");
    d->lfe_hoa = e && e[0] == '1';
    e = getenv("IAMF_HIP_SAMSUNG_TV");
    d->tv = e && e[0] == '1';
  }
  d->out_rate = 48000; /* OUTPUT_SAMPLERATE, IAMF_decoder.c:56,3734: other stream rates are resampled to it */
  return d;
}

static void free_runtime(struct IAMF_Decoder *d) {
  if (d->aux) iamf_hip_batch_destroy(d->aux); /* (before `batch`, whose LFE filter state it may share) */
  d->aux = 0;
  if (d->batch) iamf_hip_batch_destroy(d->batch);
  if (d->batch3) iamf_hip_batch_destroy(d->batch3);
  d->batch = d->batch3 = 0; /* (the resampler stays: setup_pipeline decides what becomes of it) */
  if (d->h_raw) (void)hipHostFree(d->h_raw);
  d->h_raw = 0;
  d->lp_ok = 0;
  for (int e = 0; e < 2; ++e) {
    if (d->h_in[e]) (void)hipHostFree(d->h_in[e]);
    d->h_in[e] = 0;
    for (int s = 0; s < MAX_SUBSTREAMS; ++s) {
      free(d->pkt[e][s]);
      d->pkt[e][s] = 0;
      d->pkt_cap[e][s] = 0;
      d->pkt_have[e][s] = 0;
    }
  }
  for (int i = 0; i < 4; ++i) {
    if (d->h_ramp[i]) (void)hipHostFree(d->h_ramp[i]);
    d->h_ramp[i] = 0;
  }
  free(d->tmp);
  d->tmp = 0;
  if (d->d_mid) (void)hipFree(d->d_mid);
  if (d->d_res) (void)hipFree(d->d_res);
  for (int e = 0; e < 2; ++e) {
    if (d->pre[e].h_dmx) (void)hipHostFree(d->pre[e].h_dmx);
    if (d->pre[e].h_demix) (void)hipHostFree(d->pre[e].h_demix);
    d->pre[e].h_dmx = 0;
    d->pre[e].h_demix = 0;
  }
  if (d->d_aux_il) (void)hipFree(d->d_aux_il);
  if (d->d_aux_pl) (void)hipFree(d->d_aux_pl);
  d->d_aux_il = d->d_aux_pl = 0;
  if (d->h_pcm) (void)hipHostFree(d->h_pcm);
  if (d->h_done) (void)hipHostFree((void *)d->h_done);
  d->h_done = 0;
  d->d_mid = d->d_res = 0;
  d->h_pcm = 0;
  if (d->stream) (void)hipStreamDestroy(d->stream);
  d->stream = 0;
  d->configured = 0;
}

/* iamf_database_reset + iamf_database_init (IAMF_decoder.c:1182-1196, 3801-3806): everything parsed
 * from the previous IA sequence goes — descriptors, the parameter timelines with their queues and
 * timestamps, the per-element runtime derived from them; the user's settings stay */
static void reset_descriptors(struct IAMF_Decoder *d) {
  free_runtime(d);
  for (int i = 0; i < d->nparam; ++i) free(d->param[i].rq);
  memset(d->param, 0, sizeof(d->param));
  memset(d->el, 0, sizeof(d->el));
  memset(d->pr, 0, sizeof(d->pr));
  d->nel = d->npr = d->nparam = 0;
  d->have_header = d->have_codec = 0;
  d->frame_size = d->sample_size = d->rate = 0;
  d->sel = 0;
  d->sel_el[0] = d->sel_el[1] = 0;
  d->el_gain_p[0] = d->el_gain_p[1] = d->out_gain_p = 0;
  memset(d->pre, 0, sizeof(d->pre)); /* (free_runtime has released what it pointed to) */
  d->pre[0].dmx_mode = d->pre[1].dmx_mode = -1;
  d->el_dmx_mode[0] = d->el_dmx_mode[1] = -1;
  d->meta_dmixp = 0;
  d->tu_trim_start = d->tu_trim_end = 0;
  d->timestamp = 0;
  d->last_frame = 0;
  d->need_reconf = 0;
}

int IAMF_decoder_close(IAMF_DecoderHandle d) {
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->group) return IAMF_ERR_INVALID_STATE; /* destroy the group first: it renders for this handle */
  free_runtime(d);
  if (d->rs) iamf_hip_resampler_destroy(d->rs);
  for (int i = 0; i < d->nparam; ++i) free(d->param[i].rq);
  free(d);
  return IAMF_OK;
}

/* ---- configuration ---- */
static int out_rendering_id(const struct IAMF_Decoder *d) {
  return d->out_type == IAMF_LAYOUT_TYPE_BINAURAL ? IAMF_HIP_L_BINAURAL : k_ss_rid[d->out_ss];
}

static int element_matrix(const struct IAMF_Decoder *d, const Element *e, iamf_hip_matrix *mx) {
  const int out_id = out_rendering_id(d);
  if (e->type == AUDIO_ELEMENT_SCENE_BASED) {
    int order = e->channels == 1 ? 0 : e->channels == 4 ? 1 : e->channels == 9 ? 2 : 3; /* IAMF_decoder.c:2403-2413 */
    return iamf_hip_get_h2m_matrix(order, out_id, mx);
  }
  return iamf_hip_get_m2m_matrix_variant(d->tv ? IAMF_HIP_VARIANT_SAMSUNG_TV : IAMF_HIP_VARIANT_DEFAULT,
                                         e->channels == 1 ? IAMF_HIP_L_MONO : k_layer_rid[e->layout], out_id, mx);
}

/* channels per frame handed to the device: decoded channels for projection-mode ambisonics */
static int element_in_channels(const Element *e) {
  return e->amb_projection ? e->nsub + e->amb_coupled : e->channels;
}

/* iamf_stream_set_output_layout, IAMF_decoder.c:1776-1822: which layer of a scalable element is decoded */
static int select_layer(const struct IAMF_Decoder *d, const Element *e) {
  if (e->nlayers == 1) return 0;
  if (d->tv) return e->nlayers - 1; /* -DSAMSUNG_TV: "default to use the highest layout" (:1782,1818-1820) */
  if (d->out_type == IAMF_LAYOUT_TYPE_BINAURAL) return e->nlayers - 1;
  for (int i = 0; i < e->nlayers; ++i)
    if (k_layout_ss[e->layer[i].layout] == (int)d->out_ss) return i;
  for (int i = 0; i < e->nlayers; ++i)
    if (k_layout_channels[e->layer[i].layout] > k_ss_channels[d->out_ss]) return i;
  return e->nlayers - 1;
}

static Element *find_element(struct IAMF_Decoder *d, uint64_t id) {
  for (int i = 0; i < d->nel; ++i)
    if (d->el[i].id == id) return &d->el[i];
  return 0;
}

/* What element ei needs in front of its renderer, and the matrix its batch is created with: the parametric down-mixer when
 * the element carries demixing info and the target is a smaller IAMF layout (iamf_stream_renderer_enable_downmix,
 * IAMF_decoder.c:2448-2478), the demixer when it is scalable or carries output gains (:2351-2386) */
static int pre_decide(struct IAMF_Decoder *d, int ei, iamf_hip_matrix *mx) {
  Element *e = d->sel_el[ei];
  Pre *q = &d->pre[ei];
  const int out_layout = d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION ? k_ss_layout[d->out_ss] : -1;
  q->use_dmx = q->use_demix = 0;
  if (e->type == AUDIO_ELEMENT_CHANNEL_BASED) {
    int lay = select_layer(d, e), gains = 0;
    for (int k = 0; k <= lay; ++k) gains |= e->layer[k].out_gain_flag;
    if (e->nlayers > 1 || gains) {
      q->use_demix = 1;
      q->demix_layer = lay;
    }
  }
  if (e->type == AUDIO_ELEMENT_CHANNEL_BASED && e->has_demix && out_layout >= 0 && iamf_hip_dmx_valid(e->layout, out_layout)) {
    q->use_dmx = 1;
    mx->kind = IAMF_HIP_KIND_DMX;
    mx->in_id = e->layout;
    mx->out_id = out_layout;
    iamf_hip_dmx_state_init(&q->dmx);
    q->dmx_mode = -1;
    iamf_hip_dmx_set_mode_weight(&q->dmx, e->demix_default_mode, e->demix_default_w);
    return 0;
  }
  return element_matrix(d, e, mx);
}

/* ... and its configuration on the batch that renders the element (projection de-mapping, IAMF_core_decoder.c:116-130;
 * iamf_stream_scale_demixer_configure, IAMF_decoder.c:2351-2386) plus the pinned per-frame records */
static int pre_attach(struct IAMF_Decoder *d, int ei, iamf_hip_batch *batch) {
  const Element *e = d->sel_el[ei];
  Pre *q = &d->pre[ei];
  memset(&q->dc_sig, 0, sizeof(q->dc_sig));
  if (e->amb_projection && iamf_hip_batch_set_projection(batch, e->proj, element_in_channels(e))) return IAMF_ERR_INTERNAL;
  if (q->use_demix) {
    iamf_hip_demix_config dc;
    int last = -1;
    memset(&dc, 0, sizeof(dc));
    dc.layout = e->layout;
    q->demix_nsub = 0;
    for (int l = 0; l <= q->demix_layer; ++l) {
      const Layer *L = &e->layer[l];
      int n = layer_new_channels(last, L->layout, dc.chs_in + dc.n_in);
      if (n != L->nsub + L->ncoupled || dc.n_in + n > 12) return IAMF_ERR_INVALID_PACKET;
      dc.n_in += n;
      q->demix_nsub += L->nsub;
      last = L->layout;
    }
    if (dc.n_in != k_layout_channels[e->layout]) return IAMF_ERR_INVALID_PACKET;
    /* The layers' output gains, in layer order (iamf_stream_scale_demixer_configure, IAMF_decoder.c:2365-2380).  dmx_gainup
     * (demixer.c:421-430) multiplies a channel only if it was DECODED, so entries naming a channel that is derived later
     * are dropped here (the batch dropped them anyway): that bounds the list by what the layers carry, and more than 12
     * entries that do name decoded channels are refused — until the second half of round 4 this loop wrote dc.gain_ch[] /
     * dc.gain[] without a bound (six layers can name 24 channels).
     * The reference collects all mapped entries in chs[12] / gains[12] on its stack, also without a bound, and stores the
     * mapped channel BEFORE testing it: with exactly 12 entries collected and one more flag bit set, chs[12] =
     * IA_CH_INVALID lands on its neighbour — gains[0] in the gcc build here: the first entry's gain becomes 0.  Found by
     * tests/e2e_fuzz.py 'wide' seed 7214 (flags 43 / 63 / 27 / 27 over stereo, 3.1.2, 5.1.2, 7.1.4: one channel of the
     * reference's PCM rendered from a zeroed Rtf); not reproduced: e2e_fuzz.reference_gain_list_overflows() names such
     * streams, tools/debug/fuzz_bisect.py took this one apart. */
    for (int l = 0; l <= q->demix_layer; ++l) {
      const Layer *L = &e->layer[l];
      if (!L->out_gain_flag) continue;
      for (int c = 0; c < 6; ++c)
        if (L->gain_flags & (1 << c)) {
          int ch = output_gain_channel(L->layout, c), decoded = 0;
          if (ch == CH_INVALID) continue;
          for (int k = 0; k < dc.n_in; ++k) decoded |= dc.chs_in[k] == ch;
          if (!decoded) continue;
          if (dc.n_gain >= 12) return IAMF_ERR_UNIMPLEMENTED;
          dc.gain_ch[dc.n_gain] = ch;
          dc.gain[dc.n_gain++] = db2lin(q_to_float(L->gain_q, 8));
        }
    }
    q->demix_nch = dc.n_in;
    /* iamf_stream_scale_decoder_set_default_recon_gain, :2202-2236 */
    q->rec_flags = q->demix_layer > 0 ? recon_default_flags(e->layer[0].layout, e->layout) : 0;
    q->rec_n = recon_channel_order(e->layout, q->rec_flags, q->rec_ch);
    for (int i = 0; i < 12; ++i) q->rec_gain[i] = 1.f;
    memset(q->layer_rec_flags, 0, sizeof(q->layer_rec_flags));
    dc.frame_offset = 0; /* LPCM has no decoder delay: demixer_set_frame_offset(0), IAMF_decoder.c:2175-2183 */
    if (iamf_hip_batch_set_demixer(batch, &dc)) return IAMF_ERR_INTERNAL;
    q->dc_sig = dc;
    iamf_hip_demix_state_init(&q->dmst);
    if (e->has_demix) iamf_hip_demix_set_info(&q->dmst, e->demix_default_mode, e->demix_default_w);
    q->dmx_mode = -1;
    if (hipHostMalloc((void **)&q->h_demix, sizeof(iamf_hip_demix_frame), 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  }
  if (hipHostMalloc((void **)&q->h_dmx, sizeof(iamf_hip_dmx_frame), 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  return IAMF_OK;
}

/* What a down-mixer of this handle multiplies with.  The reference keeps the dependency tables of its down-mixer
 * (chsl5, chl3, chhl, chtl ...: downmix_renderer.c:65-75) in STATIC arrays that every DMRenderer shares: DMRenderer_open
 * points their `sp` fields at the opened instance's own mix_factors (:165-172) and DMRenderer_set_mode_weight stores
 * gamma * w in the TL / TR entries' `s` (:199-211).  With one down-mixer that is its own state.  With two in one
 * presentation, BOTH read alpha / beta / gamma / delta of the one opened LAST — the later element in presentation order —
 * as they are at that moment (the earlier element renders before the later one's update of the frame: it sees the mode of
 * the frame before), and gamma * w as whoever stored it last (each element's own update precedes its own render).
 * Reproduced per handle, and pinned by `l714dmx_plus_l714dmx_C` / `scalable_plus_l714dmx_312`; the reference shares the
 * tables across HANDLES of a process as well — not reproduced: that depends on how a caller interleaves its decoders. */
static void dmx_shared_coefficients(const struct IAMF_Decoder *d, float out[5]) {
  const iamf_hip_dmx_state *last = &d->pre[d->dmx_last].dmx;
  out[0] = last->alpha;
  out[1] = last->beta;
  out[2] = last->gamma;
  out[3] = last->delta;
  out[4] = d->dmx_static_s;
}

/* The stage's records of the frame at hand: iamf_stream_render's down-mixer update (IAMF_decoder.c:2574-2583) into *fr,
 * iamf_stream_scale_decoder_demix's (:2324-2349; demixer_set_recon_gain, demixer.c:620-634) into *dm */
static void pre_frame_parts(struct IAMF_Decoder *d, int ei, iamf_hip_dmx_frame *fr, iamf_hip_demix_frame *dm, int renderer,
                            int decoder) {
  const Element *e = d->sel_el[ei];
  Pre *q = &d->pre[ei];
  if (q->use_dmx && renderer) {
    fr->offset = 0;
    dmx_shared_coefficients(d, fr->prev);
    if (q->dmx_mode > -1) {
      iamf_hip_dmx_set_mode_weight(&q->dmx, q->dmx_mode, -1);
      if (e->layout != IA_CHANNEL_LAYOUT_312) d->dmx_static_s = q->dmx.gamma_w;
    }
    dmx_shared_coefficients(d, fr->cur);
  }
  if (q->use_demix && decoder) {
    if (e->layer[q->demix_layer].recon_flag) {
      const uint32_t lf = q->layer_rec_flags[q->demix_layer];
      const int cnt = popcount32(lf);
      if (lf && (lf ^ q->rec_flags)) {
        q->rec_n = recon_channel_order(e->layout, lf, q->rec_ch);
        q->rec_flags = lf;
      }
      for (int i = 0; i < cnt && i < 12; ++i) q->rec_gain[i] = q->layer_rec_gain[q->demix_layer][i];
    }
    if (q->dmx_mode > -1) iamf_hip_demix_set_info(&q->dmst, q->dmx_mode, -1);
    iamf_hip_demix_frame_fill(&q->dmst, q->rec_n, q->rec_ch, q->rec_gain, dm);
  }
}

static void pre_frame(struct IAMF_Decoder *d, int ei, iamf_hip_dmx_frame *fr, iamf_hip_demix_frame *dm) {
  pre_frame_parts(d, ei, fr, dm, 1, 1);
}

/* A temporal unit that is trimmed away completely returns 0 samples, but not before the reference has decoded it — the
 * demixer of a scalable element has run, its recon-gain smoothing and weight index have moved on (iamf_stream_decoder_decode,
 * IAMF_decoder.c:3347) — and, unless the start or the end trim alone covers the whole frame (:3354-3357), rendered it: the
 * parametric down-mixer has stepped its weight (:2574-2583) and the HOA LFE generator's filter has run over the frame
 * (:2625-2636).  Returns 1 if the unit was rendered in the reference (the caller then advances the LFE generator). */
static int64_t time_transform(int64_t t1, int s1, int s2);
static int dropped_unit_advance(struct IAMF_Decoder *d, int ns, iamf_hip_dmx_frame *fr[2], iamf_hip_demix_frame *dm[2]) {
  /* the caller's clock moves on by the samples trimmed from the start, kept frame or not (IAMF_decoder.c:3410-3415 stands
   * in front of the "nothing left" exit :3417-3422) */
  if (d->tu_trim_start > 0) d->pts += time_transform((int64_t)d->tu_trim_start, (int)d->rate, (int)d->pts_base);
  const int rendered = !(d->tu_trim_start == (uint64_t)d->frame_size || d->tu_trim_end == (uint64_t)d->frame_size) && ns > 0;
  for (int k = 0; k < d->sel->nel; ++k) {
    const int e = d->sel->swapped ? d->sel->nel - 1 - k : k;
    if (e == 0 || d->aux) pre_frame_parts(d, e, fr[e], dm[e], rendered, 1);
  }
  return rendered;
}

/* iamf_target_layout_matching_calculation, IAMF_decoder.c:2997-3028: 100 for the output layout itself, else 50 plus / minus
 * the difference in channel count (a target the reference knows no channel count for counts as 0 channels) */
static int layout_match_score(const struct IAMF_Decoder *d, int type, int ss) {
  const int out_ch = d->out_type == IAMF_LAYOUT_TYPE_BINAURAL ? 2 : k_ss_channels[d->out_ss];
  int chs = 0;
  if (type == d->out_type && (type == IAMF_LAYOUT_TYPE_BINAURAL ||
                              (type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION && ss == (int)d->out_ss)))
    return 100;
  if (type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION)
    chs = ss >= 0 && ss < (int)(sizeof(k_ss_channels) / sizeof(k_ss_channels[0])) ? k_ss_channels[ss] : 0;
  else if (type == IAMF_LAYOUT_TYPE_BINAURAL)
    chs = 2;
  return out_ch < chs ? 50 + (chs - out_ch) : 50 - (out_ch - chs);
}

static int setup_pipeline(struct IAMF_Decoder *d) {
  iamf_hip_batch_config cfg;
  Presentation *p = 0;
  int resample, aux = 0; /* aux: element 1 needs a stage of its own */
  free_runtime(d);
  d->pre[1].use_dmx = d->pre[1].use_demix = 0;
  if (!d->have_header || !d->have_codec || !d->nel || !d->npr) return IAMF_ERR_BUFFER_TOO_SMALL;
  if (d->out_type == IAMF_LAYOUT_TYPE_NOT_DEFINED) return IAMF_ERR_BAD_ARG;
  /* iamf_decoder_get_best_mix_presentation, IAMF_decoder.c:3083-3111: the only one; else the one the caller named; else the
   * FIRST one with the highest matching score over its layouts (iamf_mix_presentation_matching_calculation, :3059-3081) */
  if (d->npr == 1) {
    p = &d->pr[0];
  } else {
    int best = 0;
    for (int i = 0; i < d->npr && d->mix_id >= 0 && !p; ++i)   /* iamf_database_get_mix_presentation: the first with that id */
      if (d->pr[i].id == (uint64_t)d->mix_id) p = &d->pr[i];
    const int named = p != 0;
    for (int i = 0; i < d->npr && !named; ++i) {
      int score = 0;
      for (int l = 0; l < d->pr[i].nlayouts; ++l) {
        const int s = layout_match_score(d, d->pr[i].layout_type[l], d->pr[i].layout_ss[l]);
        if (s > score) score = s;
      }
      if (score > best) {
        best = score;
        p = &d->pr[i];
      }
    }
    if (!p) return IAMF_ERR_INTERNAL; /* (the reference: no presentation, IAMF_ERR_INTERNAL from configure) */
  }
  d->sel = p;
  /* iamf_mix_presentation_get_best_loudness, :3030-3057: the loudness of the FIRST layout with the highest score, 0 without one */
  d->mix_loudness = 0.f;
  {
    int best = 0;
    for (int l = 0; l < p->nlayouts; ++l) {
      const int s = layout_match_score(d, p->layout_type[l], p->layout_ss[l]);
      if (s > best) {
        best = s;
        d->mix_loudness = q_to_float(p->loud[l].integrated_loudness, 8);
      }
    }
  }
  for (int i = 0; i < p->nel; ++i) {
    d->sel_el[i] = find_element(d, p->el_id[i]);
    if (!d->sel_el[i]) return IAMF_ERR_INTERNAL;
    d->el_gain_p[i] = param_get(d, &p->el_gain_def[i], IAMF_PARAMETER_TYPE_MIX_GAIN);
  }
  d->out_gain_p = param_get(d, &p->out_gain_def, IAMF_PARAMETER_TYPE_MIX_GAIN);
  d->out_channels = d->out_type == IAMF_LAYOUT_TYPE_BINAURAL ? 2 : k_ss_channels[d->out_ss];
  /* The mixed frame keeps the channel count of the layout the presentation was ENABLED with (pst->frame.channels,
   * IAMF_decoder.c:3171); the -DSAMSUNG_TV layout switch does not touch it, and iamf_frame_gain applies the output mix
   * gain to that many channels (:1383-1408,3462-3469): after a switch to a wider layout the channels beyond it go without. */
  if (!d->frame_channels0) d->frame_channels0 = d->out_channels;
  /* IAMF_decoder.c:3189-3199: a new presentation TAKES the resampler of the one before it (iamf_presentation_take_resampler)
   * — its rates, its history, its phase — and opens one only if there was none and the rates differ.  A second IA sequence
   * on a handle that resampled the first one therefore goes on through the SAME resampler: no latency is skipped again,
   * and the first sequence's ratio stays even if the new stream's rate is another (a reference quirk; its buffer sized
   * for the first sequence's frames is why it dies on some such streams).  Kept as long as the channel count is the same;
   * the -DSAMSUNG_TV layout switch closes and re-opens it (:3872-3876, reopen_rs). */
  if (d->rs && (d->reopen_rs || d->rs_channels != d->out_channels)) {
    iamf_hip_resampler_destroy(d->rs);
    d->rs = 0;
  }
  d->reopen_rs = 0;
  d->rs_inherited = d->rs != 0; /* (a group builds its own resampler: such a handle stays single, iamf_hip_decoder_group_create) */
  resample = d->rs != 0 || d->out_rate != d->rate;
  d->info.max_frame_size = d->frame_size <= 1024 ? 6144 : 6 * d->frame_size; /* :1628-1630 */

  memset(&cfg, 0, sizeof(cfg));
  cfg.n_streams = 1;
  cfg.frame_size = (int32_t)d->frame_size;
  cfg.sample_rate = (int32_t)d->rate;
  cfg.out_channels = d->out_channels;
  cfg.out_gain_channels = d->frame_channels0 < d->out_channels ? d->frame_channels0 : 0;
  /* -DSAMSUNG_TV: every PCM frame is written with a 12-channel stride (IAMF_decoder.c:3492-3495) */
  d->pcm_stride = d->tv ? 12 : d->out_channels;
  d->pcm_extra = d->out_channels > d->pcm_stride ? d->out_channels - d->pcm_stride : 0;
  cfg.projection = IAMF_HIP_PROJ_EXACT; /* a single decoder handle is not throughput bound */
  /* Which layer of every channel-based element is decoded (iamf_stream_set_output_layout, IAMF_decoder.c:1776-1822) */
  for (int i = 0; i < p->nel; ++i) {
    Element *e = d->sel_el[i];
    if (e->type != AUDIO_ELEMENT_CHANNEL_BASED) continue;
    e->layout = e->layer[select_layer(d, e)].layout;
    e->channels = k_layout_channels[e->layout];
  }
  /* The batch has ONE pre-stage in front of ONE element: the demixer of a scalable / output-gained channel element
   * (IAMF_decoder.c:2351-2386), the parametric down-mixer (:2448-2478), the projection de-mapping (IAMF_core_decoder.c:
   * 116-130) or the LFE generator's source (:2625-2636); its second element is a plain matrix.  The reference runs all
   * of these per STREAM and then mixes  0 + frame_0 + frame_1  (iamf_mixer_mix, :2702-2733) — an f32 sum that does not
   * depend on the order of the two frames (0 + a is a; a + b is b + a in IEEE arithmetic; two -0 terms give +0 either
   * way, which no PCM word can tell).  So a presentation whose SECOND element needs the pre-stage is rendered with its
   * two elements exchanged: every per-element item (mix-gain definition and default, parameter streams, packets) is
   * indexed through the presentation entry that is exchanged here.  Both elements needing one: element 1 gets a batch
   * of its own (`aux`, below).  The LFE generator exists in `batch` only, so a scene-based element that feeds it is
   * always element 0. */
  {
    const int out_layout = d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION ? k_ss_layout[d->out_ss] : -1;
    int needs[2] = {0, 0};
    for (int i = 0; i < p->nel; ++i) {
      const Element *e = d->sel_el[i];
      if (e->type == AUDIO_ELEMENT_CHANNEL_BASED) {
        int lay = select_layer(d, e), gains = 0;
        for (int k = 0; k <= lay; ++k) gains |= e->layer[k].out_gain_flag;
        needs[i] = e->nlayers > 1 || gains || (e->has_demix && out_layout >= 0 && iamf_hip_dmx_valid(e->layout, out_layout));
      } else {
        needs[i] = e->amb_projection || (d->lfe_hoa && d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION);
      }
    }
    const int lfe_gen = d->lfe_hoa && d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION;
    const int scene0 = d->sel_el[0]->type == AUDIO_ELEMENT_SCENE_BASED;
    const int scene1 = p->nel == 2 && d->sel_el[1]->type == AUDIO_ELEMENT_SCENE_BASED;
    /* Two scene-based elements and the LFE generator: the reference has ONE filter per output layout (IAMF_decoder.c:
     * 2629-2632), both W channels run through it in turn, in presentation order.  `aux` renders before `batch` and the two
     * share the filter state (iamf_hip_batch_share_lfe_state): the EARLIER element of the presentation must be element 1. */
    if (p->nel == 2 && needs[1] &&
        (!needs[0] || (lfe_gen && scene1 && !scene0) || (lfe_gen && scene1 && scene0 && !p->swapped))) {
      const uint64_t id = p->el_id[0];
      const ParamDef pd = p->el_gain_def[0];
      const int16_t q = p->el_gain_q[0];
      Element *e = d->sel_el[0];
      Param *gp = d->el_gain_p[0];
      p->el_id[0] = p->el_id[1];
      p->el_gain_def[0] = p->el_gain_def[1];
      p->el_gain_q[0] = p->el_gain_q[1];
      d->sel_el[0] = d->sel_el[1];
      d->el_gain_p[0] = d->el_gain_p[1];
      p->el_id[1] = id;
      p->el_gain_def[1] = pd;
      p->el_gain_q[1] = q;
      d->sel_el[1] = e;
      d->el_gain_p[1] = gp;
      p->swapped ^= 1; /* (a second configuration of the same descriptors finds them exchanged already) */
      aux = needs[0];
    } else if (p->nel == 2 && needs[1]) {
      aux = 1;
    }
  }
  /* IAMF_decoder.c:2625-2633: scene-based element, LFE generator compiled in, layout with an LFE */
  if (d->lfe_hoa && d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION)
    cfg.lfe_hoa = d->sel_el[0]->type == AUDIO_ELEMENT_SCENE_BASED;
  d->el_dmx_mode[0] = d->el_dmx_mode[1] = -1; /* cctx->dmx_mode = INVALID_VALUE at stream creation, :1728 */
  if (pre_decide(d, 0, &cfg.matrix)) return IAMF_ERR_INTERNAL;
  if (resample) {
    cfg.out_format = IAMF_HIP_FMT_F32;
    cfg.limiter_enable = 0;
  } else {
    cfg.out_format = (int32_t)d->bit_depth;
    cfg.pcm_stride_channels = d->pcm_stride;
    cfg.limiter_enable = d->limiter_on;
    cfg.limiter_threshold_db = d->limiter_db;
    cfg.loudness_enable = d->norm_loudness != 0.f;
  }
  if (!iamf_hip_format_bytes(cfg.out_format)) return IAMF_ERR_BAD_ARG; /* bit depth never set: IAMF_decoder.c:3726 */
  if (iamf_hip_batch_create(&cfg, &d->batch)) return IAMF_ERR_INTERNAL;
  d->cfg_sig = cfg; /* a group of handles is formed from handles whose batches were created alike */
  d->cfg_mat = cfg.matrix.mat;
  d->cfg_sig.matrix.mat = 0;
  {
    int rc = pre_attach(d, 0, d->batch);
    if (rc) return rc;
  }
  if (p->nel == 2 && !aux) {
    iamf_hip_matrix m2;
    float one = 1.f;
    if (element_matrix(d, d->sel_el[1], &m2) || iamf_hip_batch_set_second_element(d->batch, &m2, &one))
      return IAMF_ERR_INTERNAL;
  }
  if (aux) {
    /* Element 1 needs a stage of its own: a second batch renders it — stage, matrix, its mix gain; no limiter, no
     * loudness — into f32 sample-frames, and `batch` takes those as its second element through the identity matrix with
     * gain 1:  0 + 1 * y[c] + 0 * y[..]  is y[c] for every finite y, so what reaches the mixer is what the reference's
     * renderer and iamf_frame_gain left for this element (IAMF_decoder.c:2536-2651, 639-664), bit for bit. */
    iamf_hip_batch_config ca;
    iamf_hip_matrix m2;
    float eye[24 * 24], one = 1.f;
    int rc;
    memset(&ca, 0, sizeof(ca));
    ca.n_streams = 1;
    ca.frame_size = (int32_t)d->frame_size;
    ca.sample_rate = (int32_t)d->rate;
    ca.out_channels = d->out_channels;
    ca.out_format = IAMF_HIP_FMT_F32;
    ca.projection = IAMF_HIP_PROJ_EXACT;
    ca.lfe_hoa = cfg.lfe_hoa && d->sel_el[1]->type == AUDIO_ELEMENT_SCENE_BASED; /* (then element 0 is scene-based too) */
    if (pre_decide(d, 1, &ca.matrix) || iamf_hip_batch_create(&ca, &d->aux)) return IAMF_ERR_INTERNAL;
    /* both batches have a generator only if both matrices have an LFE slot — the same output layout: both or neither */
    if (ca.lfe_hoa && iamf_hip_batch_share_lfe_state(d->aux, d->batch) == IAMF_HIP_ERR_INVALID_STATE) return IAMF_ERR_INTERNAL;
    d->aux_sig = ca;
    d->aux_mat = ca.matrix.mat;
    d->aux_sig.matrix.mat = 0;
    rc = pre_attach(d, 1, d->aux);
    if (rc) return rc;
    d->aux_gain_set = 1.f;
    for (int i = 0; i < d->out_channels; ++i)
      for (int j = 0; j < d->out_channels; ++j) eye[i * d->out_channels + j] = i == j ? 1.f : 0.f;
    memset(&m2, 0, sizeof(m2));
    m2.kind = IAMF_HIP_KIND_M2M;
    m2.m = m2.n = m2.channels = d->out_channels;
    m2.lfe1 = m2.lfe2 = -1;
    m2.mat = eye;
    if (iamf_hip_batch_set_second_element(d->batch, &m2, &one)) return IAMF_ERR_INTERNAL;
    if (hipMalloc((void **)&d->d_aux_il, sizeof(float) * d->frame_size * d->out_channels) != hipSuccess ||
        hipMalloc((void **)&d->d_aux_pl, sizeof(float) * d->frame_size * d->out_channels) != hipSuccess)
      return IAMF_ERR_ALLOC_FAIL;
  }
  d->dmx_last = 0;
  d->dmx_static_s = 0.f;
  for (int k = 0; k < p->nel; ++k) { /* DMRenderer_open + DMRenderer_set_mode_weight(default), in presentation order */
    const int e = p->swapped ? p->nel - 1 - k : k;
    if (!d->pre[e].use_dmx || (e == 1 && !aux)) continue;
    d->dmx_last = e;
    if (d->sel_el[e]->layout != IA_CHANNEL_LAYOUT_312) d->dmx_static_s = d->pre[e].dmx.gamma_w;
  }
  if (resample) {
    float eye[24 * 24];
    iamf_hip_batch_config c3;
    for (int i = 0; i < d->out_channels; ++i)
      for (int j = 0; j < d->out_channels; ++j) eye[i * d->out_channels + j] = i == j ? 1.f : 0.f;
    if (!d->rs) {
      if (iamf_hip_resampler_create(1, d->out_channels, (int)d->rate, (int)d->out_rate, &d->rs)) return IAMF_ERR_INTERNAL;
      d->rs_channels = d->out_channels;
    }
    memset(&c3, 0, sizeof(c3));
    c3.n_streams = 1;
    c3.frame_size = 1; /* interleaved f32 in */
    c3.sample_rate = (int32_t)d->out_rate;
    c3.out_channels = d->out_channels;
    c3.out_format = (int32_t)d->bit_depth;
    c3.pcm_stride_channels = d->pcm_stride;
    c3.matrix.kind = IAMF_HIP_KIND_M2M;
    c3.matrix.m = c3.matrix.n = c3.matrix.channels = d->out_channels;
    c3.matrix.lfe1 = c3.matrix.lfe2 = -1;
    c3.matrix.mat = eye;
    c3.limiter_enable = d->limiter_on;
    c3.limiter_threshold_db = d->limiter_db;
    c3.loudness_enable = d->norm_loudness != 0.f;
    c3.projection = IAMF_HIP_PROJ_EXACT;
    if (!iamf_hip_format_bytes(c3.out_format)) return IAMF_ERR_BAD_ARG;
    if (iamf_hip_batch_create(&c3, &d->batch3)) return IAMF_ERR_INTERNAL;
  }
  {
    iamf_hip_batch *lb = d->batch3 ? d->batch3 : d->batch;
    float one = 1.f, lg = db2lin(d->norm_loudness - d->mix_loudness);
    if (iamf_hip_batch_set_gains(lb, d->batch3 ? &one : 0, d->batch3 ? &one : 0, &lg)) return IAMF_ERR_INTERNAL;
  }
  if (hipStreamCreate(&d->stream) != hipSuccess) return IAMF_ERR_INTERNAL;
  d->tmp = (float *)malloc(sizeof(float) * MAX_SUBSTREAMS * 2 * d->frame_size);
  if (!d->tmp) return IAMF_ERR_ALLOC_FAIL;
  for (int e = 0; e < p->nel; ++e) {
    size_t bytes = sizeof(float) * (size_t)element_in_channels(d->sel_el[e]) * d->frame_size;
    if (hipHostMalloc((void **)&d->h_in[e], bytes, 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
    memset(d->h_in[e], 0, bytes);
  }
  for (int i = 0; i < 4; ++i)
    if (hipHostMalloc((void **)&d->h_ramp[i], sizeof(float) * d->frame_size, 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  {
    /* the headline's kind of stream: its packets go to the render kernel as they are (the reference's LPCM decode,
     * pcm/IAMF_pcm_decoder.c:64-83, happens where the kernel loads them).  The library decides per call whether the
     * fused kernel takes it and unpacks on the device otherwise; here: whether asking is worth it */
    const Element *e0 = d->sel_el[0];
    iamf_hip_lpcm_layout *L = &d->lp_layout;
    d->lp_ok = p->nel == 1 && e0->type == AUDIO_ELEMENT_SCENE_BASED && !e0->amb_projection && !d->pre[0].use_dmx && !d->pre[0].use_demix &&
               !resample && d->sample_size == 16 && d->little_endian && d->limiter_on && d->out_channels <= 2 &&
               d->pcm_stride == d->out_channels && (d->frame_size & 63) == 0 && e0->nsub > 0 && e0->nsub <= MAX_SUBSTREAMS &&
               (e0->channels == 1 || e0->channels == 4 || e0->channels == 9 || e0->channels == 16) && !getenv("IAMF_HIP_FACADE_UNPACK");
    memset(L, 0, sizeof(*L));
    L->sample_bytes = 2;
    L->little_endian = 1;
    L->channels = e0->channels;
    L->frame_size = (int32_t)d->frame_size;
    for (int c = 0; c < e0->channels && d->lp_ok; ++c) {
      if (e0->amb_map[c] >= e0->nsub) d->lp_ok = 0; /* a channel no sub-stream carries: the f32 form zeroes it */
      L->src_offset[c] = (int32_t)(e0->amb_map[c] * d->frame_size * 2);
      L->src_step[c] = 2;
    }
    if (d->lp_ok && hipHostMalloc((void **)&d->h_raw, (size_t)e0->nsub * d->frame_size * 2 + 256, 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  }
  d->pcm_cap = (size_t)4 * ((size_t)d->info.max_frame_size * d->pcm_stride + d->pcm_extra);
  if (hipHostMalloc(&d->h_pcm, d->pcm_cap, 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  if (hipHostMalloc((void **)&d->h_done, 64, 0) != hipSuccess) return IAMF_ERR_ALLOC_FAIL;
  *d->h_done = 0;
  d->done_seq = 0;
  d->gain_set[0] = d->gain_set[1] = 1.f; /* what iamf_hip_batch_create starts with */
  if (resample) {
    int cap = iamf_hip_resampler_out_capacity(d->rs, (int)d->frame_size) + 512;
    if (hipMalloc((void **)&d->d_mid, sizeof(float) * d->frame_size * d->out_channels) != hipSuccess ||
        hipMalloc((void **)&d->d_res, sizeof(float) * cap * d->out_channels) != hipSuccess)
      return IAMF_ERR_ALLOC_FAIL;
  }
  d->timestamp = 0;
  d->flushed = 0;
  d->configured = 1;
  return IAMF_OK;
}

int IAMF_decoder_configure(IAMF_DecoderHandle d, const uint8_t *data, uint32_t size, uint32_t *rsize) {
  uint32_t pos = 0;
  int saw_data = 0, rc;
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->group) return IAMF_ERR_INVALID_STATE;
  if (rsize) *rsize = 0;
  if (data && size > 0) {
    /* status RECEIVE (a configuration completed) or RECONFIGURE (decode met a new sequence header) ->
     * the database is reset before the new descriptors are read (IAMF_decoder.c:3796-3806) */
    if (d->configured || d->need_reconf) reset_descriptors(d);
    d->started = 1;
  }
  while (data && pos < size) { /* iamf_decoder_internal_read_descriptors_OBUs, IAMF_decoder.c:2784-2831 */
    Obu o;
    uint32_t n = obu_split(data + pos, size - pos, &o);
    if (!n) break;
    rc = IAMF_OK;
    /* a redundant copy of a descriptor is skipped once all four kinds have been seen (:2800-2803) */
    if (o.redundant && d->have_header && d->have_codec && d->nel && d->npr) {
      pos += n;
      continue;
    }
    switch (o.type) {
      case 31:
        if (o.payload_size < 6 || memcmp(o.payload, "iamf", 4) || o.payload[4] > 1) return IAMF_ERR_INVALID_PACKET;
        d->have_header = 1;
        break;
      case 0: rc = parse_codec_config(d, &o); break;
      case 1: rc = parse_element(d, &o); break;
      case 2: rc = parse_presentation(d, &o); break;
      default: saw_data = 1; break;
    }
    if (rc) return rc;
    if (saw_data) break;
    pos += n;
  }
  if (rsize) *rsize = pos;
  if (data && rsize && !saw_data) return IAMF_ERR_BUFFER_TOO_SMALL; /* IAMF_decoder.c:2778,3923-3930 */
  if (!(data && size > 0) && d->configured && !d->need_reconf && !d->need_cfg)
    return IAMF_ERR_BAD_ARG; /* "Decoder need configure with descriptor obus": nothing was changed (IAMF_decoder.c:3883-3886) */
  if (!(data && size > 0) && d->configured && d->tv && !d->need_reconf && d->need_cfg == 1) {
    d->need_cfg = 0;
    /* The -DSAMSUNG_TV build's run-time output-layout switch (IAMF_decoder.c:3837-3881): configure without descriptors
     * after IAMF_decoder_output_layout_set_*.  The reference re-opens the RENDERERS (and the resampler) for the new
     * layout and re-initialises the limiter — its 240 delayed samples are gone, the next frame withholds 240 again —
     * while the decoders, the parameter database and the stream time go on: everything that belongs to them is kept
     * across the rebuild of the pipeline. */
    const uint64_t ts = d->timestamp;
    const int el_mode[2] = {d->el_dmx_mode[0], d->el_dmx_mode[1]};
    const Element *was_el[2] = {d->sel_el[0], d->sel ? (d->sel->nel > 1 ? d->sel_el[1] : 0) : 0};
    const int had_limiter = d->limiter_on && d->configured;
    Pre was[2]; /* (values only: the pinned records they point to go with the old pipeline) */
    memcpy(was, d->pre, sizeof(was));
    d->reopen_rs = 1;
    rc = setup_pipeline(d);
    if (rc == IAMF_OK) {
      d->timestamp = ts;
      if (had_limiter) { /* IAMF_decoder.c:3837-3842: the limiter's delayed samples are counted as delivered */
        d->meta_duration += 240;
        d->last_frame += 240;
      }
      /* the stream contexts (demixing modes, recon gains, the demixers' histories) live on: every element of the new
       * pipeline takes up what the SAME element left, whichever position it had (a layout may exchange the two the other
       * way round, another mix presentation may bring other elements) */
      for (int e = 0; e < d->sel->nel; ++e) {
        int o = -1;
        Pre *q = &d->pre[e];
        for (int k = 0; k < 2; ++k)
          if (was_el[k] && was_el[k] == d->sel_el[e]) o = k;
        if (o < 0) continue;
        d->el_dmx_mode[e] = el_mode[o];
        q->dmx_mode = was[o].dmx_mode;
        memcpy(q->layer_rec_flags, was[o].layer_rec_flags, sizeof(q->layer_rec_flags));
        memcpy(q->layer_rec_gain, was[o].layer_rec_gain, sizeof(q->layer_rec_gain));
        if (was[o].use_demix && q->use_demix) { /* the demixer is part of the stream decoder, which is not re-opened */
          q->dmst = was[o].dmst;
          q->rec_flags = was[o].rec_flags;
          q->rec_n = was[o].rec_n;
          memcpy(q->rec_ch, was[o].rec_ch, sizeof(q->rec_ch));
          memcpy(q->rec_gain, was[o].rec_gain, sizeof(q->rec_gain));
        }
      }
    }
    return rc;
  }
  d->frame_channels0 = 0; /* the presentation is enabled anew: its mixed frame takes the layout's channel count */
  rc = setup_pipeline(d);
  if (rc == IAMF_OK) d->need_cfg = 0;
  return rc;
}

/* ---- one temporal unit ---- */
static float lpcm_sample(const struct IAMF_Decoder *d, const uint8_t *p) { /* pcm/IAMF_pcm_decoder.c:64-83,133-149 */
  if (d->sample_size == 16) {
    int16_t v = d->little_endian ? (int16_t)(p[0] | (p[1] << 8)) : (int16_t)(p[1] | (p[0] << 8));
    return v / (float)(1 << 15);
  } else if (d->sample_size == 24) {
    /* big-endian 24-bit: the reference's reads24be (bitstream.c:204-208) assembles the first two bytes
     * little-endian, i.e. byte 1 is the most significant one; a drop-in has to read it the same way */
    int32_t v = d->little_endian ? (p[0] | (p[1] << 8) | (p[2] << 16)) : (p[2] | (p[0] << 8) | (p[1] << 16));
    if (v & 0x800000) v |= ~0xffffff;
    return v / (float)(1 << 23);
  } else {
    uint32_t u = d->little_endian ? ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24))
                                  : ((uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24));
    return (int32_t)u / (float)(1U << 31);
  }
}

/* n samples of one channel, `step` bytes apart, -> f32: lpcm_sample with the format decided once per row (the per-sample
 * form cost 1.5-3 ns a sample, most of a single-handle call once the copies around the launch were gone) */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__) && !defined(__SANITIZE_ADDRESS__)
__attribute__((target_clones("avx2", "default"), optimize("O3")))
#endif
static void lpcm_row(const struct IAMF_Decoder *d, const uint8_t *p, int step, int n, float *out) {
  if (d->sample_size == 16 && d->little_endian) {
    if (step == 2) {
      for (int i = 0; i < n; ++i) {
        int16_t v;
        memcpy(&v, p + 2 * (size_t)i, 2);
        out[i] = v / (float)(1 << 15);
      }
    } else {
      for (int i = 0; i < n; ++i) {
        int16_t v;
        memcpy(&v, p + (size_t)i * step, 2);
        out[i] = v / (float)(1 << 15);
      }
    }
  } else if (d->sample_size == 16) {
    for (int i = 0; i < n; ++i) {
      const uint8_t *q = p + (size_t)i * step;
      out[i] = (int16_t)(q[1] | (q[0] << 8)) / (float)(1 << 15);
    }
  } else {
    for (int i = 0; i < n; ++i) out[i] = lpcm_sample(d, p + (size_t)i * step);
  }
}

/* substreams -> planar f32 in the order the renderer expects; returns samples per channel */
static int unpack_element(struct IAMF_Decoder *d, int ei) {
  const Element *e = d->sel_el[ei];
  const int bps = (int)d->sample_size / 8, fs = (int)d->frame_size;
  float *dst = d->h_in[ei];
  int ns = -1;
  float *tmp = d->tmp;
  /* audio-layer order: coupled pairs first, then singles */
  int c = 0;
  const int demix = d->pre[ei].use_demix;
  const int nsub = demix ? d->pre[ei].demix_nsub : e->nsub;
  int lay = 0, lay_s0 = 0; /* layer of sub-stream s and its first sub-stream */
  for (int s = 0; s < nsub; ++s) {
    int w;
    if (demix) { /* every layer: coupled sub-streams first (iamf_stream_scale_decoder_decode, :2276-2322) */
      while (s >= lay_s0 + e->layer[lay].nsub) lay_s0 += e->layer[lay++].nsub;
      w = s - lay_s0 < e->layer[lay].ncoupled ? 2 : 1;
    } else {
      w = ((e->type == AUDIO_ELEMENT_CHANNEL_BASED && s < e->nb_coupled) ||
           (e->amb_projection && s < e->amb_coupled)) ? 2 : 1;
    }
    const int n = (int)(d->pkt_len[ei][s] / (uint32_t)(w * bps));
    if (ns < 0) ns = n;
    if (n != ns || n > fs) return IAMF_ERR_INVALID_PACKET;
    for (int k = 0; k < w; ++k) lpcm_row(d, d->pkt[ei][s] + (size_t)k * bps, w * bps, n, tmp + (size_t)(c + k) * fs);
    c += w;
  }
  if (ns < 0) return IAMF_ERR_INVALID_PACKET; /* no sub-stream at all: nothing fixed the sample count */
  if (e->amb_projection || demix) { /* decoded channel order goes to the device: de-mapping / demixer run there */
    for (int l = 0; l < c; ++l) memcpy(dst + (size_t)l * fs, tmp + (size_t)l * fs, sizeof(float) * ns);
    return ns;
  }
  for (int p = 0; p < e->channels; ++p) {
    int src = e->type == AUDIO_ELEMENT_CHANNEL_BASED ? k_al_of_pl[e->layout][p] : e->amb_map[p];
    if (src < c)
      memcpy(dst + (size_t)p * fs, tmp + (size_t)src * fs, sizeof(float) * ns);
    else
      memset(dst + (size_t)p * fs, 0, sizeof(float) * ns);
  }
  return ns;
}

/* element 0's packets as they are -> the pinned raw row (sub-stream s at s * frame_size * 2); unpack_element's checks;
 * returns samples per channel */
static int stage_lpcm_row(struct IAMF_Decoder *d) {
  const Element *e = d->sel_el[0];
  const int fs = (int)d->frame_size;
  int ns = -1;
  for (int s = 0; s < e->nsub; ++s) {
    const int n = (int)(d->pkt_len[0][s] / 2u);
    if (ns < 0) ns = n;
    if (n != ns || n > fs) return IAMF_ERR_INVALID_PACKET;
    memcpy(d->h_raw + (size_t)s * fs * 2, d->pkt[0][s], (size_t)n * 2);
  }
  return ns < 0 ? IAMF_ERR_INVALID_PACKET : ns;
}

static int tu_complete(const struct IAMF_Decoder *d) { /* IAMF_decoder.c:2854-2869 */
  for (int e = 0; e < d->sel->nel; ++e)
    for (int s = 0; s < d->sel_el[e]->nsub; ++s)
      if (!d->pkt_have[e][s]) return 0;
  return 1;
}

/* Wait for what this call queued on the handle's stream.  A frame is ~10 us of device work: a one-lane kernel behind it
 * writes a pinned word and the host spins on it (14.8 instead of 18.1 us per launch-and-wait on MI355X,
 * tools/debug/sync_probe.hip); a spin that gets long — or IAMF_HIP_FACADE_SYNC=1 — ends in hipStreamSynchronize, which is
 * also what reports a device error. */
static int facade_wait(struct IAMF_Decoder *d) {
  static int plain = -1;
  if (plain < 0) plain = getenv("IAMF_HIP_FACADE_SYNC") ? 1 : 0;
  if (!plain && d->h_done && iamf_hip_stream_signal(d->stream, d->h_done, ++d->done_seq) == IAMF_HIP_OK) {
    for (int spin = 0; spin < 400000; ++spin)
      if (*d->h_done == d->done_seq) return IAMF_OK;
  }
  return hipStreamSynchronize(d->stream) == hipSuccess ? IAMF_OK : IAMF_ERR_INTERNAL;
}

/* time_transform, IAMF_decoder.c:91-95 */
static int64_t time_transform(int64_t t1, int s1, int s2) {
  double r;
  if (s1 == s2) return t1;
  r = (double)(t1 * s2);
  return (int64_t)(r / s1 + 0.5f);
}

/* What a decoded temporal unit leaves for IAMF_decoder_get_last_metadata, as the reference's frame loop does
 * (IAMF_decoder.c:3410-3415: the caller's pts moves on by the samples trimmed from the START of a frame;
 * :3436-3442: the demixing mode of the last channel-based element, in presentation order, that has one) */
static void meta_note_frame(struct IAMF_Decoder *d, int s0, uint64_t trim_end) {
  const int fs = (int)d->frame_size;
  if (s0 > 0 && s0 != fs && trim_end != (uint64_t)fs) d->pts += time_transform(s0, (int)d->rate, (int)d->pts_base);
  for (int k = 0; k < d->sel->nel; ++k) {
    const int e = d->sel->swapped ? d->sel->nel - 1 - k : k;
    if (d->sel_el[e]->type == AUDIO_ELEMENT_CHANNEL_BASED && d->el_dmx_mode[e] >= 0) d->meta_dmixp = (uint32_t)d->el_dmx_mode[e];
  }
}
/* ctx->duration / ctx->last_frame_size, IAMF_decoder.c:3521-3522 */
static void meta_note_output(struct IAMF_Decoder *d, int n) {
  d->meta_duration += (uint64_t)(n > 0 ? n : 0);
  d->last_frame = (uint32_t)(n > 0 ? n : 0);
}

static int render_tu(struct IAMF_Decoder *d, void *pcm) {
  iamf_hip_render_args a;
  const int fs = (int)d->frame_size, bytes = (int)d->bit_depth / 8;
  int ns = 0, n, keep, s0;
  float cgain[3];
  int ramp[3] = {0, 0, 0};
  int lp = d->lp_ok && !d->group; /* the packets go to the render kernel as they are (decided per frame below) */
  memset(&a, 0, sizeof(a));
  if (lp) {
    int r = stage_lpcm_row(d);
    if (r < 0) return r;
    ns = r;
  } else {
    for (int e = 0; e < d->sel->nel; ++e) {
      int r = unpack_element(d, e);
      if (r < 0) return r;
      ns = r;
    }
  }
  /* iamf_frame_trim (IAMF_decoder.c:1361-1381): rendering is memoryless, so trimming the
   * element PCM before the renderer equals trimming the rendered frame */
  for (int e = 0; e < d->sel->nel; ++e)
    for (int s = 0; s < d->sel_el[e]->nsub; ++s) d->pkt_have[e][s] = 0;
  /* the trims are LEB128 fields of the OBU header: compared as uint64 BEFORE narrowing.  The reference
   * refuses start < 0 || end < 0 || samples - start - end < 0 with IAMF_ERR_BAD_ARG and moves the
   * stream time on (iamf_frame_trim :1364-1370, :3424-3428) */
  if (d->tu_trim_start > (uint64_t)ns || d->tu_trim_end > (uint64_t)ns ||
      d->tu_trim_start + d->tu_trim_end > (uint64_t)ns) {
    d->timestamp += fs;
    return IAMF_ERR_BAD_ARG;
  }
  s0 = (int)d->tu_trim_start;
  keep = ns - s0 - (int)d->tu_trim_end;
  if (keep <= 0) {
    iamf_hip_dmx_frame *fr[2] = {d->pre[0].h_dmx, d->pre[1].h_dmx};
    iamf_hip_demix_frame *dm[2] = {d->pre[0].h_demix, d->pre[1].h_demix};
    if (dropped_unit_advance(d, ns, fr, dm)) {
      /* the LFE generator's filter over the frame the reference rendered before it cut it (element 1's batch first: it is
       * the earlier element of the presentation when both feed the generator) */
      if (d->aux && d->aux_sig.lfe_hoa &&
          iamf_hip_batch_lfe_advance(d->aux, d->h_in[1], (int64_t)element_in_channels(d->sel_el[1]) * fs, ns, d->stream, 0, 1))
        return IAMF_ERR_INTERNAL;
      if (d->cfg_sig.lfe_hoa &&
          iamf_hip_batch_lfe_advance(d->batch, d->h_in[0], (int64_t)element_in_channels(d->sel_el[0]) * fs, ns, d->stream, 0, 1))
        return IAMF_ERR_INTERNAL;
      if ((d->cfg_sig.lfe_hoa || (d->aux && d->aux_sig.lfe_hoa)) && facade_wait(d)) return IAMF_ERR_INTERNAL; /* h_in is reused */
    }
    d->timestamp += fs;
    return 0;
  }
  meta_note_frame(d, s0, d->tu_trim_end);
  /* mix gains of this frame */
  {
    const uint64_t pt = d->timestamp + (uint64_t)s0;
    float el_def[2] = {db2lin(q_to_float(d->sel->el_gain_q[0], 8)), db2lin(q_to_float(d->sel->el_gain_q[1], 8))};
    float og_def = db2lin(q_to_float(d->sel->out_gain_q, 8));
    ramp[0] = mix_gain_unit(d->el_gain_p[0], el_def[0], pt, keep, (int)d->rate, &cgain[0], d->h_ramp[0]);
    ramp[1] = d->sel->nel > 1 ? mix_gain_unit(d->el_gain_p[1], el_def[1], pt, keep, (int)d->rate, &cgain[1], d->h_ramp[1]) : 0;
    ramp[2] = mix_gain_unit(d->out_gain_p, og_def, pt, keep, (int)d->rate, &cgain[2], d->h_ramp[2]);
    if (d->sel->nel < 2) cgain[1] = 1.f;
    for (int i = 0; i < 3; ++i) {
      if (ramp[i] < 0) { /* unit refused: no gain at all */
        ramp[i] = 0;
        cgain[i] = 1.f;
      }
      if (!ramp[i] && i == 1) /* element 1 takes its constant through the ramp: x * positive constant is the same product */
        for (int k = 0; k < keep; ++k) d->h_ramp[i][k] = cgain[i];
    }
  }
  if (lp && ((s0 & 3) || (keep & 63) || ramp[0] || ramp[2])) { /* not a call of the fused kernel: the f32 form after all */
    int r = unpack_element(d, 0);
    if (r < 0) return r;
    lp = 0;
  }
  /* An element that feeds the HOA LFE generator keeps its whole frame: the reference renders first and trims the result
   * (IAMF_decoder.c:3424-3430), so the generator's filter runs over the trimmed samples too — the batch gets a pointer to
   * the first kept sample and how many lie in front of and behind the call (iamf_hip_render_args::lfe_pre_samples). */
  const int lfe0 = d->cfg_sig.lfe_hoa, lfe1 = d->aux && d->aux_sig.lfe_hoa;
  if (!lp)
    for (int e = 0; e < d->sel->nel; ++e) {
      const int ch = element_in_channels(d->sel_el[e]);
      if (s0 && !(e == 0 ? lfe0 : lfe1))
        for (int c = 0; c < ch; ++c) memmove(d->h_in[e] + (size_t)c * fs, d->h_in[e] + (size_t)c * fs + s0, sizeof(float) * keep);
    }
  a.d_in = lp ? 0 : d->h_in[0] + (lfe0 ? s0 : 0);
  if (lfe0) {
    a.lfe_pre_samples = s0;
    a.lfe_post_samples = (int)d->tu_trim_end;
  }
  a.in_stream_stride = a.in_frame_stride = (int64_t)element_in_channels(d->sel_el[0]) * fs;
  if (d->sel->nel > 1 && !d->aux) {
    a.d_in2 = d->h_in[1];
    a.in2_stream_stride = a.in2_frame_stride = (int64_t)d->sel_el[1]->channels * fs;
  }
  /* constant gains go through iamf_frame_gain's rule (only if != 1 and > 0); ramps unconditionally */
  {
    float eg = ramp[0] ? 1.f : cgain[0], og = ramp[2] ? 1.f : cgain[2];
    if (eg != d->gain_set[0] || og != d->gain_set[1]) {
      if (iamf_hip_batch_set_gains(d->batch, &eg, &og, 0)) return IAMF_ERR_INTERNAL;
      d->gain_set[0] = eg;
      d->gain_set[1] = og;
    }
    if (ramp[0]) a.d_element_ramp = d->h_ramp[0];
    if (ramp[2]) a.d_output_ramp = d->h_ramp[2];
    if (d->sel->nel > 1) a.d_element2_ramp = d->h_ramp[1]; /* element 1: constant or ramp, both exact */
  }
  a.ramp_stream_stride = fs;
  for (int k = 0; k < d->sel->nel; ++k) { /* in the order the reference renders its streams (dmx_shared_coefficients) */
    const int e = d->sel->swapped ? d->sel->nel - 1 - k : k;
    if (e == 0 || d->aux) pre_frame(d, e, d->pre[e].h_dmx, d->pre[e].h_demix);
  }
  if (d->pre[0].use_dmx) a.d_dmx_frames = d->pre[0].h_dmx;
  if (d->pre[0].use_demix) {
    a.d_demix_frames = d->pre[0].h_demix;
    a.demix_sample0 = s0;
  }
  if (d->aux) { /* element 1 through its own batch -> f32 sample-frames -> planar, where `batch` reads its second element */
    iamf_hip_render_args b;
    const float eg = ramp[1] ? 1.f : cgain[1], one = 1.f;
    int n1;
    memset(&b, 0, sizeof(b));
    b.d_in = d->h_in[1] + (lfe1 ? s0 : 0);
    b.in_stream_stride = b.in_frame_stride = (int64_t)element_in_channels(d->sel_el[1]) * fs;
    if (lfe1) {
      b.lfe_pre_samples = s0;
      b.lfe_post_samples = (int)d->tu_trim_end;
    }
    if (eg != d->aux_gain_set) {
      if (iamf_hip_batch_set_gains(d->aux, &eg, &one, 0)) return IAMF_ERR_INTERNAL;
      d->aux_gain_set = eg;
    }
    if (ramp[1]) {
      memcpy(d->h_ramp[3], d->h_ramp[1], sizeof(float) * keep);
      b.d_element_ramp = d->h_ramp[3];
    }
    b.ramp_stream_stride = fs;
    for (int k = 0; k < keep; ++k) d->h_ramp[1][k] = 1.f; /* `batch` takes the frame as it is */
    if (d->pre[1].use_dmx) b.d_dmx_frames = d->pre[1].h_dmx;
    if (d->pre[1].use_demix) {
      b.d_demix_frames = d->pre[1].h_demix;
      b.demix_sample0 = s0;
    }
    b.n_frames = 1;
    b.n_samples = keep < fs ? keep : 0;
    b.stream = d->stream;
    b.d_pcm = d->d_aux_il;
    b.pcm_stream_stride_bytes = (int64_t)sizeof(float) * fs * d->out_channels;
    n1 = iamf_hip_batch_render_ex(d->aux, &b);
    if (n1 != keep) return IAMF_ERR_INTERNAL;
    if (iamf_hip_deinterleave_f32(d->d_aux_il, 0, d->out_channels, 1, keep, d->d_aux_pl, 0, fs, d->stream)) return IAMF_ERR_INTERNAL;
    a.d_in2 = d->d_aux_pl;
    a.in2_stream_stride = a.in2_frame_stride = (int64_t)d->out_channels * fs;
  }
  a.n_frames = 1;
  a.n_samples = keep < fs ? keep : 0;
  a.stream = d->stream;
  if (lp) {
    iamf_hip_lpcm_input in;
    memset(&in, 0, sizeof(in));
    in.d_raw = d->h_raw;
    in.raw_stream_stride = in.raw_frame_stride = (int64_t)d->sel_el[0]->nsub * fs * 2;
    in.first_sample = s0;
    in.layout = d->lp_layout;
    a.d_pcm = d->h_pcm;
    a.pcm_stream_stride_bytes = (int64_t)d->pcm_cap;
    n = iamf_hip_batch_render_lpcm(d->batch, &in, &a);
  } else if (!d->rs) {
    a.d_pcm = d->h_pcm;
    a.pcm_stream_stride_bytes = (int64_t)d->pcm_cap;
    n = iamf_hip_batch_render_ex(d->batch, &a);
  } else { /* render (f32) -> resample -> loudness / limiter / pack */
    int n2;
    a.d_pcm = d->d_mid;
    a.pcm_stream_stride_bytes = (int64_t)sizeof(float) * fs * d->out_channels;
    n = iamf_hip_batch_render_ex(d->batch, &a);
    if (n < 0) return IAMF_ERR_INTERNAL;
    n2 = iamf_hip_resampler_process(d->rs, d->d_mid, (int64_t)fs * d->out_channels, n, d->d_res,
                                    (int64_t)(iamf_hip_resampler_out_capacity(d->rs, fs) + 512) * d->out_channels, d->stream);
    if (n2 < 0) return IAMF_ERR_INTERNAL;
    n = n2 ? iamf_hip_batch_render(d->batch3, d->d_res, 0, d->out_channels, n2, d->h_pcm, (int64_t)d->pcm_cap, d->stream) : 0;
  }
  if (n < 0) return IAMF_ERR_INTERNAL;
  if (facade_wait(d)) return IAMF_ERR_INTERNAL;
  if (n > 0) memcpy(pcm, d->h_pcm, ((size_t)n * d->pcm_stride + d->pcm_extra) * bytes);
  params_time_elapse(d, (uint64_t)keep); /* IAMF_decoder.c:3471: the mixed frame's length */
  d->timestamp += fs;
  meta_note_output(d, n);
  return n;
}

static int flush_tail(struct IAMF_Decoder *d, void *pcm) { /* iamf_delay_buffer_handle, IAMF_decoder.c:3250-3301 */
  const int bytes = (int)d->bit_depth / 8;
  int n = 0;
  if (d->flushed && !d->rs) {
    /* Every further flush call of the reference pushes another 240 zeros through the limiter and hands out what they
     * displace (iamf_delay_buffer_handle, :3284-3299): the zeros of the flush before, times a positive gain — 240
     * sample-frames of zero PCM.  Behind a resampler every call also drains another output-latency's worth of its filter
     * tail (:3271-3282: rest_flag = 2, NULL input): the path below, again.  Either way the call ends at :3521-3522. */
    n = d->limiter_on ? 240 : 0;
    if (n) memset(pcm, 0, ((size_t)n * d->pcm_stride + d->pcm_extra) * bytes);
    meta_note_output(d, n);
    return n;
  }
  d->flushed = 1;
  if (!d->rs) {
    if (!d->limiter_on) {
      meta_note_output(d, 0);
      return 0;
    }
    n = iamf_hip_batch_flush(d->batch, d->h_pcm, (int64_t)d->pcm_cap, d->stream);
  } else {
    const int cap = iamf_hip_resampler_flush_capacity(d->rs);
    const int extra = d->limiter_on ? 240 : 0;
    int n2;
    if (hipMemsetAsync(d->d_res, 0, sizeof(float) * (cap + extra) * d->out_channels, d->stream) != hipSuccess) return IAMF_ERR_INTERNAL;
    n2 = iamf_hip_resampler_flush(d->rs, d->d_res, (int64_t)(cap + extra) * d->out_channels, d->stream);
    if (n2 < 0) return IAMF_ERR_INTERNAL;
    /* the reference drains the resampler straight into the limiter (:3271-3287): iamf_loudness_process belongs to the
     * frame loop (:3480-3484), the tail is NOT normalised.  (The limiter's delayed samples were scaled when they entered.) */
    if (d->norm_loudness != 0.f) {
      const float one = 1.f;
      if (iamf_hip_batch_set_gains(d->batch3, 0, 0, &one)) return IAMF_ERR_INTERNAL;
    }
    n = (n2 + extra) ? iamf_hip_batch_render(d->batch3, d->d_res, 0, d->out_channels, n2 + extra, d->h_pcm, (int64_t)d->pcm_cap, d->stream) : 0;
  }
  if (n < 0) return IAMF_ERR_INTERNAL;
  if (facade_wait(d)) return IAMF_ERR_INTERNAL;
  if (n > 0) memcpy(pcm, d->h_pcm, ((size_t)n * d->pcm_stride + d->pcm_extra) * bytes);
  meta_note_output(d, n);
  return n;
}

/* iamf_decoder_internal_parse_OBUs (IAMF_decoder.c:2871-2995): consumes OBUs until a temporal unit is complete
 * (*ready = 1, *rsize = bytes consumed) or the data ends (returns 0) or something is wrong (returns the error) */
static int decode_parse(struct IAMF_Decoder *d, const uint8_t *data, int32_t size, uint32_t *rsize, int *ready) {
  uint32_t pos = 0;
  *ready = 0;
  while (pos < (uint32_t)size) {
    Obu o;
    uint32_t n = obu_split(data + pos, (uint32_t)size - pos, &o);
    if (!n) break;
    pos += n;
    if (o.type == 31 && !o.redundant) { /* a new IA sequence: the caller must reconfigure (:2918-2921) */
      if (rsize) *rsize = pos - n;
      d->need_reconf = 1;
      return IAMF_ERR_INVALID_STATE;
    }
    if (o.type == 3) {
      int rc = parse_parameter_block(d, &o);
      if (rc) return rc;
    } else if (o.type >= 5 && o.type <= 23) {
      Rd r = {o.payload, o.payload_size, 0, 0};
      uint64_t sid = o.type == 5 ? rd_leb128(&r) : (uint64_t)(o.type - 6);
      for (int e = 0; e < d->sel->nel; ++e)
        for (int s = 0; s < d->sel_el[e]->nsub; ++s)
          if (d->sel_el[e]->sub_ids[s] == sid) {
            uint32_t len = o.payload_size - r.pos;
            uint8_t *b = d->pkt[e][s];
            if (d->pkt_ext[e][s] && len <= d->pkt_ext_cap[e][s]) { /* grouped: straight into the upload row */
              b = d->pkt_ext[e][s];
              d->pkt_in_ext[e][s] = 1;
            } else {
              d->pkt_in_ext[e][s] = 0;
              if (!b || len > d->pkt_cap[e][s]) {
                b = (uint8_t *)realloc(b, len ? len : 1);
                if (!b) return IAMF_ERR_ALLOC_FAIL;
                d->pkt[e][s] = b;
                d->pkt_cap[e][s] = len ? len : 1;
              }
            }
            memcpy(b, o.payload + r.pos, len);
            d->pkt_len[e][s] = len;
            d->pkt_have[e][s] = 1;
            if (e == 0 && s == 0) {
              d->tu_trim_start = o.trim_start;
              d->tu_trim_end = o.trim_end;
            }
          }
      if (tu_complete(d)) {
        if (rsize) *rsize = pos;
        *ready = 1;
        return 0;
      }
    }
  }
  if (rsize) *rsize = pos;
  return 0;
}

int IAMF_decoder_decode(IAMF_DecoderHandle d, const uint8_t *data, int32_t size, uint32_t *rsize, void *pcm) {
  int ready = 0, rc;
  if (!d || !pcm) return IAMF_ERR_BAD_ARG;
  if (rsize) *rsize = 0;
  if (d->group) return IAMF_ERR_INVALID_STATE; /* a grouped handle decodes through iamf_hip_decoder_group_decode */
  if (!d->configured || d->need_reconf) return IAMF_ERR_INVALID_STATE; /* status != RECEIVE, :3941-3942 */
  if (!data || size <= 0) return flush_tail(d, pcm); /* IAMF_decoder.c:3508-3519 */
  rc = decode_parse(d, data, size, rsize, &ready);
  if (rc || !ready) return rc;
  return render_tu(d, pcm);
}

/* extension (include/iamf_hip.h): the run-time form of the reference's DISABLE_LFE_HOA build switch */
int iamf_hip_decoder_set_hoa_lfe(void *handle, int enable) {
  struct IAMF_Decoder *d = (struct IAMF_Decoder *)handle;
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->configured) return IAMF_ERR_INVALID_STATE; /* like the sampling rate: before configuration */
  d->lfe_hoa = enable ? 1 : 0;
  return IAMF_OK;
}

int iamf_hip_decoder_set_variant(void *handle, int variant) {
  struct IAMF_Decoder *d = (struct IAMF_Decoder *)handle;
  if (!d || (variant != IAMF_HIP_VARIANT_DEFAULT && variant != IAMF_HIP_VARIANT_SAMSUNG_TV)) return IAMF_ERR_BAD_ARG;
  if (d->configured) return IAMF_ERR_INVALID_STATE;
  d->tv = variant == IAMF_HIP_VARIANT_SAMSUNG_TV;
  return IAMF_OK;
}

/* ---- setters / getters (IAMF_decoder.c:3948-4168) ---- */
/* the three setters note a CHANGE (ctx->need_configure, IAMF_decoder.c:3943-3996): configure without data has something to
 * do only then */
int IAMF_decoder_set_mix_presentation_id(IAMF_DecoderHandle d, uint64_t id) {
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->mix_id == (int64_t)id) return IAMF_OK;
  d->mix_id = (int64_t)id;
  d->need_cfg |= 2;
  return IAMF_OK;
}
int IAMF_decoder_output_layout_set_sound_system(IAMF_DecoderHandle d, IAMF_SoundSystem ss) {
  if (!d || ss <= SOUND_SYSTEM_INVALID || ss >= SOUND_SYSTEM_END) return IAMF_ERR_BAD_ARG;
  if (d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION && d->out_ss == ss) return IAMF_OK;
  d->out_type = IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION;
  d->out_ss = ss;
  d->need_cfg |= 1;
  return IAMF_OK;
}
int IAMF_decoder_output_layout_set_binaural(IAMF_DecoderHandle d) {
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->out_type == IAMF_LAYOUT_TYPE_BINAURAL) return IAMF_OK;
  d->out_type = IAMF_LAYOUT_TYPE_BINAURAL;
  d->need_cfg |= 1;
  return IAMF_OK;
}
int IAMF_layout_sound_system_channels_count(IAMF_SoundSystem ss) {
  if (ss <= SOUND_SYSTEM_INVALID || ss >= SOUND_SYSTEM_END) return IAMF_ERR_BAD_ARG;
  return k_ss_channels[ss];
}
int IAMF_layout_binaural_channels_count(void) { return 2; }
char *IAMF_decoder_get_codec_capability(void) {
  const char *s = "iamf.001.001.ipcm"; /* the one codec of this build */
  char *r = (char *)malloc(strlen(s) + 1);
  if (r) strcpy(r, s);
  return r;
}
int IAMF_decoder_set_normalization_loudness(IAMF_DecoderHandle d, float loudness) {
  if (!d) return IAMF_ERR_BAD_ARG;
  d->norm_loudness = loudness;
  return IAMF_OK;
}
int IAMF_decoder_set_bit_depth(IAMF_DecoderHandle d, uint32_t bit_depth) {
  if (!d || (bit_depth != 16 && bit_depth != 24 && bit_depth != 32)) return IAMF_ERR_BAD_ARG;
  d->bit_depth = bit_depth;
  return IAMF_OK;
}
int IAMF_decoder_peak_limiter_enable(IAMF_DecoderHandle d, uint32_t enable) {
  if (!d) return IAMF_ERR_BAD_ARG;
  d->limiter_on = enable ? 1 : 0;
  return IAMF_OK;
}
int IAMF_decoder_peak_limiter_set_threshold(IAMF_DecoderHandle d, float db) {
  if (!d) return IAMF_ERR_BAD_ARG;
  d->limiter_db = db;
  return IAMF_OK;
}
float IAMF_decoder_peak_limiter_get_threshold(IAMF_DecoderHandle d) { return d ? d->limiter_db : 0.f; }
int IAMF_decoder_set_sampling_rate(IAMF_DecoderHandle d, uint32_t rate) { /* IAMF_decoder.c:4112-4130 */
  static const uint32_t ok[] = {8000, 12000, 16000, 24000, 32000, 44100, 48000};
  if (!d) return IAMF_ERR_BAD_ARG;
  if (d->started) return IAMF_ERR_INVALID_STATE; /* only before the first configure call (status INIT, :4117-4120) */
  for (unsigned i = 0; i < sizeof(ok) / sizeof(ok[0]); ++i)
    if (ok[i] == rate) {
      d->out_rate = rate;
      return IAMF_OK;
    }
  return IAMF_ERR_BAD_ARG;
}
IAMF_StreamInfo *IAMF_decoder_get_stream_info(IAMF_DecoderHandle d) { return d ? &d->info : 0; }
int IAMF_decoder_set_pts(IAMF_DecoderHandle d, int64_t pts, uint32_t time_base) { /* IAMF_decoder.c:4136-4148 */
  if (!d) return IAMF_ERR_BAD_ARG;
  d->pts = pts;
  d->pts_base = time_base;
  d->meta_duration = 0;
  return IAMF_OK;
}

/* iamf_presentation_get_output_sound_mode + iamf_sound_mode_combine, IAMF_decoder.c:241-254,1555-1578 */
static IAMF_SoundMode sound_mode_of(const struct IAMF_Decoder *d) {
  IAMF_SoundMode mode = IAMF_SOUND_MODE_NONE;
  for (int i = 0; i < d->sel->nel; ++i) {
    const Element *e = d->sel_el[i];
    IAMF_SoundMode sm;
    if (e->type == AUDIO_ELEMENT_SCENE_BASED) /* iamf_layout_get_sound_mode of the output layout, :358-369 */
      sm = d->out_type == IAMF_LAYOUT_TYPE_BINAURAL ? IAMF_SOUND_MODE_BINAURAL
           : d->out_ss == SOUND_SYSTEM_A            ? IAMF_SOUND_MODE_STEREO
                                                    : IAMF_SOUND_MODE_MULTICHANNEL;
    else /* the layer that is decoded (cctx->layout, :1726) */
      sm = (e->layout == IA_CHANNEL_LAYOUT_MONO || e->layout == IA_CHANNEL_LAYOUT_STEREO) ? IAMF_SOUND_MODE_STEREO
           : e->layout == IA_CHANNEL_LAYOUT_BINAURAL                                      ? IAMF_SOUND_MODE_BINAURAL
                                                                                          : IAMF_SOUND_MODE_MULTICHANNEL;
    if (mode == IAMF_SOUND_MODE_NONE) mode = sm;
    else if (sm == IAMF_SOUND_MODE_NONE || mode == sm) ;
    else if (mode == IAMF_SOUND_MODE_BINAURAL || sm == IAMF_SOUND_MODE_BINAURAL) mode = IAMF_SOUND_MODE_NA;
    else mode = IAMF_SOUND_MODE_MULTICHANNEL;
  }
  return mode;
}

/* IAMF_decoder.c:3619-3706 (what the metadata holds), :4150-4168 (the call).  Everything handed out is the caller's to
 * free: loudness_layout, loudness, param — and every loudness[i].anchor_loudness.  (The reference hands out ITS OWN
 * anchor arrays through two shallow copies, :3644-3645 and :3688, and its player frees loudness[0]'s,
 * iamfplayer.c:309-321 — a double free there; a copy per call is the form of that contract that is safe to follow.) */
int IAMF_decoder_get_last_metadata(IAMF_DecoderHandle d, int64_t *pts, IAMF_extradata *m) {
  if (!d || !pts || !m) return IAMF_ERR_BAD_ARG;
  memset(m, 0, sizeof(*m));
  *pts = d->pts + time_transform((int64_t)(d->meta_duration - d->last_frame), (int)d->out_rate, (int)d->pts_base);
  m->number_of_samples = d->last_frame;
  if (!d->configured || !d->sel) return IAMF_OK; /* an open handle: the zeroed context (observed on the reference) */
  m->output_sound_system = d->out_type == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION ? d->out_ss : SOUND_SYSTEM_INVALID;
  m->bitdepth = d->bit_depth;
  m->sampling_rate = 48000; /* OUTPUT_SAMPLERATE, whatever IAMF_decoder_set_sampling_rate chose (:3628) */
  m->output_sound_mode = sound_mode_of(d);
  m->num_loudness_layouts = d->sel->nlayouts;
  if (d->sel->nlayouts) {
    m->loudness_layout = (IAMF_Layout *)calloc((size_t)d->sel->nlayouts, sizeof(IAMF_Layout));
    m->loudness = (IAMF_LoudnessInfo *)calloc((size_t)d->sel->nlayouts, sizeof(IAMF_LoudnessInfo));
    if (!m->loudness_layout || !m->loudness) return IAMF_ERR_ALLOC_FAIL;
    for (int i = 0; i < d->sel->nlayouts; ++i) {
      m->loudness_layout[i].type = (uint8_t)d->sel->layout_type[i]; /* iamf_layout_copy2, :324-331 */
      if (d->sel->layout_type[i] == IAMF_LAYOUT_TYPE_LOUDSPEAKERS_SS_CONVENTION)
        m->loudness_layout[i].sound_system.sound_system = (IAMF_SoundSystem)d->sel->layout_ss[i];
      m->loudness[i] = d->sel->loud[i];
      m->loudness[i].anchor_loudness = 0;
      if (d->sel->loud[i].num_anchor_loudness) {
        const size_t n = d->sel->loud[i].num_anchor_loudness;
        m->loudness[i].anchor_loudness = (anchor_loudness_t *)calloc(n, sizeof(anchor_loudness_t));
        if (!m->loudness[i].anchor_loudness) return IAMF_ERR_ALLOC_FAIL;
        memcpy(m->loudness[i].anchor_loudness, d->sel->anchors[i], n * sizeof(anchor_loudness_t));
      }
    }
  }
  for (int i = 0; i < d->sel->nel; ++i)
    if (d->sel_el[i]->has_demix) { /* an element with demixing info: one DEMIXING record (:3647-3662) */
      m->num_parameters = 1;
      m->param = (IAMF_Param *)calloc(1, sizeof(IAMF_Param));
      if (!m->param) return IAMF_ERR_ALLOC_FAIL;
      m->param->parameter_length = 8;
      m->param->parameter_definition_type = IAMF_PARAMETER_TYPE_DEMIXING;
      m->param->dmixp_mode = d->meta_dmixp;
      break;
    }
  return IAMF_OK;
}

#ifndef IAMF_FACADE_NO_GROUP
#include "iamf_decoder_group.inc"
#endif
