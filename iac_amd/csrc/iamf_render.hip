// iamf_render.hip — MI355X (gfx950) kernels + the C ABI of include/iamf_hip.h.
//
// One workgroup (4 waves) owns one IAMF stream for the whole call and walks its samples in
// chunks of 256, lane = sample.  Per chunk, fused in one pass over HBM:
//   element renderer (gain matrix, reference h2m_rdr.c:1103-1150 / m2m_rdr.c:1826-1837)
//   -> element gain -> mix -> output gain -> loudness (IAMF_decoder.c:1392-1397, 2719-2730,
//   3206-3221) -> look-ahead peak limiter (audio_effect_peak_limiter.c:94-271)
//   -> float->PCM interleave (IAMF_decoder.c:100-167).
// Rendered samples live only in an LDS ring (the limiter's 240-sample delay line); HBM sees
// the planar f32 input once and the packed PCM once.
//
// Arithmetic is IEEE f32 in the reference's operation order (this TU is compiled with
// -ffp-contract=off, correctly rounded division), so the VALU path is bit-exact against the
// CPU reference, not merely within +-1 LSB.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/iamf_hip.h"

namespace {

constexpr int kChunk = 256;      // samples per workgroup step = threads per workgroup
constexpr int kDelay = 240;      // limiter look-ahead (reference common/audio_defines.h:41)
constexpr int kRing = 512;       // LDS ring length (power of two >= kChunk + kDelay + 15)
constexpr int kSave = 256;       // samples of ring persisted per stream between calls
constexpr int kHead = 256;       // coefficient-table head kept in LDS
constexpr int kMaxOut = 24;      // reference MAX_OUTPUT_CHANNELS
constexpr int kMaxIn = 24;

struct LimState {  // per stream, persisted in HBM between calls
  float g;   // currentGain
  float gs;  // targetStartGain
  float ge;  // targetEndGain
  int n;     // increments of currentTC since the last trigger; >= n_end means idle
};

struct RenderParams {
  const float *in;          // planar f32 element PCM (device) or nullptr = zeros (flush)
  int64_t in_stream_stride; // floats
  int64_t in_frame_stride;  // floats
  uint8_t *pcm;             // packed output (device)
  int64_t pcm_stream_stride;  // bytes
  const float *matrix;      // device, feed-major [n_feeds][M]
  const float *gains;       // device [3][n_streams]: element, output, loudness
  const float *ctab;        // device limiter coefficient table [n_end + 1]
  LimState *lim;            // device [n_streams]
  float *ring_y;            // device [n_streams][out_ch][kSave]
  float *ring_pm;           // device [n_streams][kSave]
  int64_t pos0;             // samples of each stream consumed before this call
  int32_t total;            // samples to process in this call
  int32_t frame_size;
  int32_t n_streams;
  int32_t n_feeds;
  int32_t out_ch;
  int32_t out_format;
  int32_t limiter_on;
  int32_t loudness_on;
  int32_t n_atk, n_end;     // limiter table split points
  float thr;
  const int32_t *src_feed;  // device [out_ch]: output slot -> feed index, or -1 = silent slot
};

// One gain step evaluated for a hypothetical pre-state n_pre (no trigger since the state was
// set): audio_effect_peak_limiter.c:241-255 with currentTC = T[n_pre].
__device__ __forceinline__ float gain_at(int n_pre, float gs, float ge, float c, int n_atk, int n_end) {
  float g = 1.0f;
  if (n_pre < n_atk) {
    g = gs - c * (gs - ge);
  } else if (n_pre < n_end) {
    g = ge + c * (1.0f - ge);
  }
  return g;
}

__device__ __forceinline__ float to_scaled(float x, float scale, float lo, float hi) {
  x = x * scale;
  x = x > lo ? x : lo;
  x = x < hi ? x : hi;
  return rintf(x);  // v_rndne_f32: ties to even, like lrintf in the default rounding mode
}

template <int M>
__global__ __launch_bounds__(kChunk) void render_kernel(const RenderParams p) {
  extern __shared__ float lds[];
  const int out_ch = p.out_ch;
  float *ring_y = lds;                        // [out_ch][kRing]
  float *ring_pm = ring_y + out_ch * kRing;   // [kRing]  per-sample max |y| over channels
  float *ring_b16 = ring_pm + kRing;          // [kRing]  max of pm over the trailing 16 samples
  float *arr_p = ring_b16 + kRing;            // [kChunk] window maxima (serial fallback)
  float *arr_e = arr_p + kChunk;              // [kChunk] thr / p
  float *arr_g = arr_e + kChunk;              // [kChunk] gains (serial fallback)
  float *head = arr_g + kChunk;               // [kHead]  ctab[0..kHead)
  float *st = head + kHead;                   // [4]      limiter state exchange

  const int s = blockIdx.x;
  const int t = threadIdx.x;
  const int fs = p.frame_size;
  const float thr = p.thr;
  const int n_atk = p.n_atk, n_end = p.n_end;

  // ---- stream state -> LDS ----
  {
    const float *sy = p.ring_y + (int64_t)s * out_ch * kSave;
    const float *spm = p.ring_pm + (int64_t)s * kSave;
    // saved entry i (0..kSave) is global sample pos0 - kSave + i
    const int rp = (int)((p.pos0 - kSave + t) & (kRing - 1));
    const int rq = (int)((p.pos0 + t) & (kRing - 1));
    for (int c = 0; c < out_ch; ++c) {
      ring_y[c * kRing + rp] = sy[c * kSave + t];
      ring_y[c * kRing + rq] = 0.f;
    }
    ring_pm[rp] = spm[t];
    ring_pm[rq] = 0.f;
    ring_b16[rq] = 0.f;
    head[t] = (t <= n_end) ? p.ctab[t] : 1.0f;
  }
  __syncthreads();
  {
    // trailing-16 maxima of the restored part; entries older than the saved window count as 0
    const int64_t gk = p.pos0 - kSave + t;
    float b = 0.f;
    for (int j = 0; j < 16; ++j) {
      if (t - j >= 0) b = fmaxf(b, ring_pm[(int)((gk - j) & (kRing - 1))]);
    }
    ring_b16[(int)(gk & (kRing - 1))] = b;
  }
  LimState ls = p.lim[s];
  float g_cur = ls.g, gs = ls.gs, ge = ls.ge;
  int n_st = ls.n;
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);
  __syncthreads();

  const int64_t out_base = p.limiter_on ? (p.pos0 > kDelay ? p.pos0 - kDelay : 0) : p.pos0;
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;

  for (int c0 = 0; c0 < p.total; c0 += kChunk) {
    const int k = c0 + t;
    const bool valid = k < p.total;
    const int64_t gk = p.pos0 + k;
    const int rp = (int)(gk & (kRing - 1));

    // ---- load one sample of every input channel (coalesced: lane = sample) ----
    float x[M];
    if (valid && p.in) {
      const int f = k / fs;
      const int i = k - f * fs;
      const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = src[(int64_t)m * fs];
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = 0.f;
    }

    // ---- element renderer + gains; the rendered sample goes to the LDS delay ring ----
    float pm = 0.f;
    for (int c = 0; c < out_ch; ++c) {
      const int f = p.src_feed[c];
      float y = 0.f;
      if (f >= 0) {
        const float *row = p.matrix + f * M;  // wave-uniform -> scalar loads
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) acc = acc + row[m] * x[m];
        y = acc;
      }
      if (eg_on) y = y * eg;
      y = 0.f + y;  // iamf_mixer_mix: memset 0 then += (IAMF_decoder.c:2719-2730)
      if (og_on) y = y * og;
      if (lg_on) y = y * lg;
      if (valid) ring_y[c * kRing + rp] = y;
      pm = fmaxf(pm, fabsf(y));
    }

    float g = 1.0f;
    if (p.limiter_on) {
      if (valid) ring_pm[rp] = pm;
      __syncthreads();
      // trailing-16 maximum, then the 240-sample window [gk-240, gk-1] as 15 such blocks
      float b = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) b = fmaxf(b, ring_pm[(int)((gk - j) & (kRing - 1))]);
      if (valid) ring_b16[rp] = b;
      __syncthreads();
      float pk = 0.f;
#pragma unroll
      for (int j = 0; j < 15; ++j) pk = fmaxf(pk, ring_b16[(int)((gk - 1 - 16 * j) & (kRing - 1))]);
      const float e = thr / pk;

      // hypothesis: no trigger inside this chunk -> every gain follows from (n_st, gs, ge)
      int n_pre = n_st + t;
      if (n_pre > n_end) n_pre = n_end;
      const int ci = n_pre + 1 <= n_end ? n_pre + 1 : n_end;
      const float cf = ci < kHead ? head[ci] : p.ctab[ci];
      const float gh = gain_at(n_pre, gs, ge, cf, n_atk, n_end);
      const int trig = valid && (pk * gh > thr);
      const int cnt = p.total - c0 < kChunk ? p.total - c0 : kChunk;
      if (!__syncthreads_or(trig)) {
        g = gh;
        // state after the chunk = state after its last valid sample
        const int n_last = n_st + cnt - 1 < n_end ? n_st + cnt - 1 : n_end;  // pre-state of last
        if (n_last < n_end) {
          const int cl = n_last + 1;
          const float cfl = cl < kHead ? head[cl] : p.ctab[cl];
          g_cur = gain_at(n_last, gs, ge, cfl, n_atk, n_end);
          n_st = n_last + 1;
        } else {
          g_cur = 1.0f;
          n_st = n_end;
        }
      } else {
        // at least one trigger: run the recurrence serially over the chunk
        arr_p[t] = pk;
        arr_e[t] = e;
        __syncthreads();
        if (t == 0) {
          float lgc = g_cur, lgs = gs, lge = ge;
          int ln = n_st;
          for (int i = 0; i < cnt; ++i) {
            if (ln < n_end) {
              const int cl = ln + 1;
              const float c = cl < kHead ? head[cl] : p.ctab[cl];
              lgc = gain_at(ln, lgs, lge, c, n_atk, n_end);
              ln = ln + 1;
            } else {
              lgc = 1.0f;
            }
            const float pp = arr_p[i];
            if (pp * lgc > thr) {
              lgs = lgc;
              lge = arr_e[i];
              ln = 0;
            }
            arr_g[i] = lgc;
          }
          st[0] = lgc;
          st[1] = lgs;
          st[2] = lge;
          st[3] = __int_as_float(ln);
        }
        __syncthreads();
        g = arr_g[t];
        g_cur = st[0];
        gs = st[1];
        ge = st[2];
        n_st = __float_as_int(st[3]);
      }
    } else {
      __syncthreads();
    }

    // ---- emit: delayed sample * gain -> PCM ----
    const int64_t j = p.limiter_on ? gk - kDelay : gk;  // global index of the emitted sample
    if (valid && j >= 0) {
      const int rd = (int)(j & (kRing - 1));
      uint8_t *dst = pcm + (j - out_base) * (int64_t)out_ch * bytes;
      if (p.out_format == IAMF_HIP_FMT_S16) {
        if (out_ch == 2) {
          const float a = to_scaled(ring_y[rd] * g, 32768.f, -32768.f, 32767.f);
          const float bq = to_scaled(ring_y[kRing + rd] * g, 32768.f, -32768.f, 32767.f);
          const uint32_t w = (uint32_t)(uint16_t)(int16_t)(int)a | ((uint32_t)(uint16_t)(int16_t)(int)bq << 16);
          *reinterpret_cast<uint32_t *>(dst) = w;
        } else {
          int16_t *d16 = reinterpret_cast<int16_t *>(dst);
          for (int c = 0; c < out_ch; ++c)
            d16[c] = (int16_t)(int)to_scaled(ring_y[c * kRing + rd] * g, 32768.f, -32768.f, 32767.f);
        }
      } else if (p.out_format == IAMF_HIP_FMT_S24) {
        for (int c = 0; c < out_ch; ++c) {
          const int v = (int)to_scaled(ring_y[c * kRing + rd] * g, 8388608.f, -8388608.f, 8388607.f);
          dst[c * 3 + 0] = (uint8_t)(v & 0xff);
          dst[c * 3 + 1] = (uint8_t)((v >> 8) & 0xff);
          dst[c * 3 + 2] = (uint8_t)(((v >> 16) & 0x7f) | ((v >> 24) & 0x80));
        }
      } else if (p.out_format == IAMF_HIP_FMT_S32) {
        int32_t *d32 = reinterpret_cast<int32_t *>(dst);
        for (int c = 0; c < out_ch; ++c) {
          // the reference clamps against 2147483647.f (== 2^31 in f32) and narrows a long:
          // +full scale wraps to INT32_MIN (IAMF_decoder.c:114-119)
          const float r = to_scaled(ring_y[c * kRing + rd] * g, 2147483648.f, -2147483648.f, 2147483647.f);
          d32[c] = (int32_t)(long long)r;
        }
      } else {
        float *df = reinterpret_cast<float *>(dst);
        for (int c = 0; c < out_ch; ++c) df[c] = ring_y[c * kRing + rd] * g;
      }
    }
    __syncthreads();  // ring slots read above are overwritten by the next chunk
  }

  // ---- persist stream state ----
  {
    float *sy = p.ring_y + (int64_t)s * out_ch * kSave;
    float *spm = p.ring_pm + (int64_t)s * kSave;
    const int64_t end = p.pos0 + p.total;
    const int rp = (int)((end - kSave + t) & (kRing - 1));
    for (int c = 0; c < out_ch; ++c) sy[c * kSave + t] = ring_y[c * kRing + rp];
    spm[t] = ring_pm[rp];
    if (t == 0) {
      LimState o;
      o.g = g_cur;
      o.gs = gs;
      o.ge = ge;
      o.n = n_st;
      p.lim[s] = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      fprintf(stderr, "iamf_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
              __FILE__, __LINE__);                                                     \
      return IAMF_HIP_ERR_DEVICE;                                                      \
    }                                                                                  \
  } while (0)

struct TableEntry {
  uint32_t kind, in_id, out_id;
  int32_t channels, lfe1, lfe2, m, n;
  uint32_t offset;
};

extern "C" const uint8_t _binary_rdr_tables_bin_start[];
extern "C" const uint8_t _binary_rdr_tables_bin_end[];

struct Tables {
  int n = 0;
  const TableEntry *ents = nullptr;
  const float *data = nullptr;
  bool ok = false;
  Tables() {
    const uint8_t *b = _binary_rdr_tables_bin_start;
    const size_t sz = (size_t)(_binary_rdr_tables_bin_end - _binary_rdr_tables_bin_start);
    if (sz < 12 || memcmp(b, "IARDRTB1", 8) != 0) return;
    uint32_t cnt;
    memcpy(&cnt, b + 8, 4);
    n = (int)cnt;
    ents = reinterpret_cast<const TableEntry *>(b + 12);
    data = reinterpret_cast<const float *>(b + 12 + sizeof(TableEntry) * cnt);
    ok = true;
  }
};

const Tables &tables() {
  static Tables t;
  return t;
}

int find_matrix(int kind, int in_id, int out_id, iamf_hip_matrix *out) {
  const Tables &t = tables();
  if (!t.ok || !out) return -1;
  for (int i = 0; i < t.n; ++i) {  // first match in table order, as the reference searches
    const TableEntry &e = t.ents[i];
    if ((int)e.kind == kind && (int)e.in_id == in_id && (int)e.out_id == out_id) {
      out->kind = kind;
      out->in_id = in_id;
      out->out_id = out_id;
      out->channels = e.channels;
      out->lfe1 = e.lfe1;
      out->lfe2 = e.lfe2;
      out->m = e.m;
      out->n = e.n;
      out->mat = t.data + e.offset;
      return 0;
    }
  }
  return -1;
}

// audio_effect_peak_limiter.c:267-271
float ease(float x) {
  if (1.0 < x) return 1.0f;
  if (x < 0) return 0.0f;
  return 1.0f - powf(x - 1, 2.0);
}

}  // namespace

struct iamf_hip_batch {
  iamf_hip_batch_config cfg;
  int m = 0, n_feeds = 0;
  int32_t src_feed[kMaxOut];
  int32_t *d_src_feed = nullptr;
  float thr = 0.f;
  int n_atk = 0, n_end = 0;
  int64_t pos = 0;      // samples consumed per stream
  bool flushed = false;
  float *d_matrix = nullptr, *d_gains = nullptr, *d_ctab = nullptr, *d_ring_y = nullptr,
        *d_ring_pm = nullptr;
  LimState *d_lim = nullptr;
  std::vector<float> h_gains;
};

namespace {

int reset_state(iamf_hip_batch *b) {
  const int ns = b->cfg.n_streams;
  std::vector<LimState> init((size_t)ns);
  for (auto &l : init) {
    // audio_effect_peak_limiter.c:211-235: gain 1, targets -1, currentTC -1 (idle)
    l.g = 1.0f;
    l.gs = -1.0f;
    l.ge = -1.0f;
    l.n = b->n_end;
  }
  HIPCHK(hipMemcpy(b->d_lim, init.data(), sizeof(LimState) * ns, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(b->d_ring_y, 0, sizeof(float) * (size_t)ns * b->cfg.out_channels * kSave));
  HIPCHK(hipMemset(b->d_ring_pm, 0, sizeof(float) * (size_t)ns * kSave));
  b->pos = 0;
  b->flushed = false;
  return IAMF_HIP_OK;
}

typedef void (*kernel_fn)(const RenderParams);

template <int M>
void launch_m(const RenderParams &p, dim3 grid, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(render_kernel<M>, grid, dim3(kChunk), lds_bytes, st, p);
}

int launch(const RenderParams &p, int m, size_t lds_bytes, hipStream_t st) {
  dim3 grid((unsigned)p.n_streams);
  switch (m) {
#define CASE_M(v) case v: launch_m<v>(p, grid, lds_bytes, st); break;
    CASE_M(1) CASE_M(2) CASE_M(4) CASE_M(6) CASE_M(8) CASE_M(9) CASE_M(10) CASE_M(12) CASE_M(14) CASE_M(16) CASE_M(24)
#undef CASE_M
    default: return IAMF_HIP_ERR_UNIMPLEMENTED;
  }
  HIPCHK(hipGetLastError());
  return IAMF_HIP_OK;
}

int render_call(iamf_hip_batch *b, const float *d_in, int64_t ss, int64_t fsr, int total, void *d_pcm,
                int64_t pcm_stride, void *stream) {
  RenderParams p;
  memset(&p, 0, sizeof(p));
  p.in = d_in;
  p.in_stream_stride = ss;
  p.in_frame_stride = fsr;
  p.pcm = static_cast<uint8_t *>(d_pcm);
  p.pcm_stream_stride = pcm_stride;
  p.matrix = b->d_matrix;
  p.gains = b->d_gains;
  p.ctab = b->d_ctab;
  p.lim = b->d_lim;
  p.ring_y = b->d_ring_y;
  p.ring_pm = b->d_ring_pm;
  p.pos0 = b->pos;
  p.total = total;
  p.frame_size = b->cfg.frame_size;
  p.n_streams = b->cfg.n_streams;
  p.n_feeds = b->n_feeds;
  p.out_ch = b->cfg.out_channels;
  p.out_format = b->cfg.out_format;
  p.limiter_on = b->cfg.limiter_enable ? 1 : 0;
  p.loudness_on = b->cfg.loudness_enable ? 1 : 0;
  p.n_atk = b->n_atk;
  p.n_end = b->n_end;
  p.thr = b->thr;
  p.src_feed = b->d_src_feed;
  const size_t lds = sizeof(float) * ((size_t)(p.out_ch + 2) * kRing + 3 * kChunk + kHead + 4);
  const int r = launch(p, b->m, lds, static_cast<hipStream_t>(stream));
  if (r != IAMF_HIP_OK) return r;
  const int64_t before = p.limiter_on ? (b->pos > kDelay ? b->pos - kDelay : 0) : b->pos;
  b->pos += total;
  const int64_t after = p.limiter_on ? (b->pos > kDelay ? b->pos - kDelay : 0) : b->pos;
  return (int)(after - before);
}

}  // namespace

extern "C" {

int iamf_hip_get_h2m_matrix(int order, int out_id, iamf_hip_matrix *out) {
  return find_matrix(IAMF_HIP_KIND_H2M, order, out_id, out);
}

int iamf_hip_get_m2m_matrix(int in_id, int out_id, iamf_hip_matrix *out) {
  return find_matrix(IAMF_HIP_KIND_M2M, in_id, out_id, out);
}

int iamf_hip_layout_channels(int out_id) {
  switch (out_id) {
    case IAMF_HIP_SS_A: return 2;
    case IAMF_HIP_SS_B: return 6;
    case IAMF_HIP_SS_C: return 8;
    case IAMF_HIP_SS_D: return 10;
    case IAMF_HIP_SS_E: return 11;
    case IAMF_HIP_SS_F: return 12;
    case IAMF_HIP_SS_G: return 14;
    case IAMF_HIP_SS_H: return 24;
    case IAMF_HIP_SS_I: return 8;
    case IAMF_HIP_SS_J: return 12;
    case IAMF_HIP_L_712: return 10;
    case IAMF_HIP_L_312: return 6;
    case IAMF_HIP_L_MONO: return 1;
    case IAMF_HIP_L_BINAURAL: return 2;
    default: return 0;
  }
}

int iamf_hip_format_bytes(int f) {
  switch (f) {
    case IAMF_HIP_FMT_S16: return 2;
    case IAMF_HIP_FMT_S24: return 3;
    case IAMF_HIP_FMT_S32: return 4;
    case IAMF_HIP_FMT_F32: return 4;
    default: return 0;
  }
}

const char *iamf_hip_version(void) { return "iamf_hip 0.1 (gfx950, HIP)"; }

int iamf_hip_batch_create(const iamf_hip_batch_config *cfg, iamf_hip_batch **out) {
  if (!cfg || !out) return IAMF_HIP_ERR_BAD_ARG;
  *out = nullptr;
  const iamf_hip_matrix &mx = cfg->matrix;
  if (cfg->n_streams <= 0 || cfg->frame_size <= 0 || cfg->sample_rate <= 0 || !mx.mat ||
      mx.m <= 0 || mx.m > kMaxIn || mx.n <= 0 || mx.n > kMaxOut || cfg->out_channels <= 0 ||
      cfg->out_channels > kMaxOut || !iamf_hip_format_bytes(cfg->out_format))
    return IAMF_HIP_ERR_BAD_ARG;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return IAMF_HIP_ERR_DEVICE;

  iamf_hip_batch *b = new (std::nothrow) iamf_hip_batch();
  if (!b) return IAMF_HIP_ERR_ALLOC_FAIL;
  b->cfg = *cfg;
  b->m = mx.m;
  b->n_feeds = mx.n;

  // output slot <- feed: h2m_rdr.c:1114-1150 keeps slots lfe1/lfe2 free (the comparison is
  // against the SOURCE index) and zeroes them; every other slot that no feed reaches stays
  // silent.  m2m feeds map 1:1.
  for (int i = 0; i < kMaxOut; ++i) b->src_feed[i] = -1;
  for (int i = 0; i < mx.n; ++i) {
    int d = i;
    if (mx.kind == IAMF_HIP_KIND_H2M && (mx.lfe1 >= 0 || mx.lfe2 >= 0)) {
      if (mx.lfe1 >= 0 && mx.lfe1 <= i) ++d;
      if (mx.lfe2 >= 0 && mx.lfe2 <= i) ++d;
    }
    if (d < cfg->out_channels) b->src_feed[d] = i;
  }
  if (mx.kind == IAMF_HIP_KIND_H2M) {
    if (mx.lfe1 >= 0 && mx.lfe1 < cfg->out_channels) b->src_feed[mx.lfe1] = -1;
    if (mx.lfe2 >= 0 && mx.lfe2 < cfg->out_channels) b->src_feed[mx.lfe2] = -1;
  }

  // feed-major copy of the matrix: row f = coefficients of feed f over inputs 0..m-1
  std::vector<float> fm((size_t)mx.m * mx.n);
  for (int f = 0; f < mx.n; ++f)
    for (int k = 0; k < mx.m; ++k)
      fm[(size_t)f * mx.m + k] = mx.kind == IAMF_HIP_KIND_H2M ? mx.mat[f * mx.m + k] : mx.mat[k * mx.n + f];

  // limiter constants and the coefficient table.  currentTC only ever takes the values
  // T[n] = n-fold f32 accumulation of incTC from 0 (audio_effect_peak_limiter.c:82,242,248,263),
  // so the attack / release curve values are a function of n alone.
  const float atk = 0.001f, rel = 0.200f;  // common/audio_defines.h:39-40
  b->thr = (float)pow(10, cfg->limiter_threshold_db / 20);
  const float inc = (float)1 / (float)cfg->sample_rate;
  std::vector<float> T;
  T.push_back(0.0f);
  while (T.back() < rel + atk && T.size() < (1u << 22)) T.push_back(T.back() + inc);
  b->n_end = (int)T.size() - 1;
  b->n_atk = 0;
  while (b->n_atk < b->n_end && T[b->n_atk] < atk) ++b->n_atk;
  std::vector<float> ctab((size_t)b->n_end + 1, 1.0f);
  for (int k = 1; k <= b->n_end; ++k)
    ctab[k] = (k - 1) < b->n_atk ? ease(T[k] / atk) : ease((T[k] - atk) / rel);

  const int ns = cfg->n_streams;
  b->h_gains.assign((size_t)3 * ns, 1.0f);
#define CREATE_CHK(expr)                       \
  do {                                         \
    if ((expr) != hipSuccess) {                \
      iamf_hip_batch_destroy(b);               \
      return IAMF_HIP_ERR_DEVICE;              \
    }                                          \
  } while (0)
  CREATE_CHK(hipMalloc(&b->d_matrix, sizeof(float) * fm.size()));
  CREATE_CHK(hipMalloc(&b->d_gains, sizeof(float) * 3 * ns));
  CREATE_CHK(hipMalloc(&b->d_ctab, sizeof(float) * ctab.size()));
  CREATE_CHK(hipMalloc(&b->d_lim, sizeof(LimState) * ns));
  CREATE_CHK(hipMalloc(&b->d_ring_y, sizeof(float) * (size_t)ns * cfg->out_channels * kSave));
  CREATE_CHK(hipMalloc(&b->d_ring_pm, sizeof(float) * (size_t)ns * kSave));
  CREATE_CHK(hipMalloc(&b->d_src_feed, sizeof(int32_t) * kMaxOut));
  CREATE_CHK(hipMemcpy(b->d_src_feed, b->src_feed, sizeof(int32_t) * kMaxOut, hipMemcpyHostToDevice));
  CREATE_CHK(hipMemcpy(b->d_matrix, fm.data(), sizeof(float) * fm.size(), hipMemcpyHostToDevice));
  CREATE_CHK(hipMemcpy(b->d_gains, b->h_gains.data(), sizeof(float) * 3 * ns, hipMemcpyHostToDevice));
  CREATE_CHK(hipMemcpy(b->d_ctab, ctab.data(), sizeof(float) * ctab.size(), hipMemcpyHostToDevice));
#undef CREATE_CHK
  const int r = reset_state(b);
  if (r != IAMF_HIP_OK) {
    iamf_hip_batch_destroy(b);
    return r;
  }
  *out = b;
  return IAMF_HIP_OK;
}

void iamf_hip_batch_destroy(iamf_hip_batch *b) {
  if (!b) return;
  (void)hipFree(b->d_matrix);
  (void)hipFree(b->d_gains);
  (void)hipFree(b->d_ctab);
  (void)hipFree(b->d_lim);
  (void)hipFree(b->d_ring_y);
  (void)hipFree(b->d_ring_pm);
  (void)hipFree(b->d_src_feed);
  delete b;
}

int iamf_hip_batch_set_gains(iamf_hip_batch *b, const float *eg, const float *og, const float *lg) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  const int ns = b->cfg.n_streams;
  if (eg) memcpy(&b->h_gains[0], eg, sizeof(float) * ns);
  if (og) memcpy(&b->h_gains[(size_t)ns], og, sizeof(float) * ns);
  if (lg) memcpy(&b->h_gains[(size_t)2 * ns], lg, sizeof(float) * ns);
  HIPCHK(hipMemcpy(b->d_gains, b->h_gains.data(), sizeof(float) * 3 * ns, hipMemcpyHostToDevice));
  return IAMF_HIP_OK;
}

int iamf_hip_batch_render(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride,
                          int64_t in_frame_stride, int32_t n_frames, void *d_pcm,
                          int64_t pcm_stream_stride_bytes, void *stream) {
  if (!b || !d_in || !d_pcm || n_frames < 0) return IAMF_HIP_ERR_BAD_ARG;
  if (b->flushed) return IAMF_HIP_ERR_INVALID_STATE;
  if (n_frames == 0) return 0;
  const int64_t total = (int64_t)n_frames * b->cfg.frame_size;
  if (total > INT32_MAX) return IAMF_HIP_ERR_BAD_ARG;
  const int64_t need = total * b->cfg.out_channels * iamf_hip_format_bytes(b->cfg.out_format);
  if (b->cfg.n_streams > 1 && pcm_stream_stride_bytes < need) return IAMF_HIP_ERR_BUFFER_TOO_SMALL;
  return render_call(b, d_in, in_stream_stride, in_frame_stride, (int)total, d_pcm,
                     pcm_stream_stride_bytes, stream);
}

int iamf_hip_batch_flush(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stream_stride_bytes, void *stream) {
  if (!b || !d_pcm) return IAMF_HIP_ERR_BAD_ARG;
  if (b->flushed) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b->cfg.limiter_enable) return 0;
  const int r = render_call(b, nullptr, 0, 0, kDelay, d_pcm, pcm_stream_stride_bytes, stream);
  if (r >= 0) b->flushed = true;
  return r;
}

int iamf_hip_batch_reset(iamf_hip_batch *b) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  HIPCHK(hipDeviceSynchronize());
  return reset_state(b);
}

}  // extern "C"
