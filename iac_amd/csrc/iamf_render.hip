// iamf_render.hip — MI355X (gfx950) kernels + the C ABI of include/iamf_hip.h.
//
// One workgroup owns one IAMF stream for the whole call and walks its samples in chunks.  Per
// chunk, fused in one pass over HBM:
//   [demixer of scalable channel audio, demixer.c] [projection de-mapping, IAMF_core_decoder.c:116-130]
//   -> element renderer (gain matrix h2m_rdr.c:1103-1150 / m2m_rdr.c:1826-1837, parametric
//      down-mixer downmix_renderer.c, or the HRTF FIR of render_fir.hpp)
//   -> element gain -> mix -> output gain -> loudness (IAMF_decoder.c:1392-1397, 2719-2730,
//   3206-3221) -> look-ahead peak limiter (audio_effect_peak_limiter.c:94-271)
//   -> float->PCM interleave (IAMF_decoder.c:100-167).
// Rendered samples never touch HBM (LDS ring or registers hold the limiter's 240-sample delay
// line); HBM sees the planar f32 input once and the packed PCM once.
//
// Kernel families (launch() picks one per call; all share the persisted per-stream state, so
// consecutive calls of one stream may take different ones):
//   render_fast.hpp     1- and 2-channel layouts, 1024-sample chunks, 4 samples per lane (headline);
//                       variants: FIR (HRTF stage on the f32 MFMA in front, render_fir.hpp), DOWN
//                       (parametric down-mixer, render_downmix.hpp), IN2 (second element of <= 4
//                       channels and / or per-sample gain ramps)
//   render_wide4.hpp    even 6..24-channel layouts, s16, rendered samples kept in registers; VALU or
//                       MFMA projection; variants: DMX (demixer of scalable channel audio in front),
//                       DOWN, MIX (as IN2) (own translation units: iamf_render_wide4.hip, _mix.hip)
//   render_wide.hpp     any other multi-channel aligned call, 256-sample chunks
//   render_generic.hpp  everything else: ragged calls, flush, limiter off, wide second elements,
//                       exact two-stage projection, demixer + down-mixer together, other PCM
//                       formats of the stages above
//
// Arithmetic is IEEE f32 in the reference's operation order (compiled with -ffp-contract=off,
// correctly rounded division), so the VALU paths are bit-exact against the CPU reference, not
// merely within +-1 LSB; the MFMA projection variants are the documented exception.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/iamf_hip.h"

#define IAMF_FFT_HOST_TABLES   // render_fir_fft.hpp: the host-side table builder is compiled in this unit

extern "C" int iamf_hip_fir_m2b_has(int m);                                           // iamf_render_fir_m2b.hip
extern "C" int iamf_hip_fir_m2b_launch(const void *params, int m, hipStream_t st);    // iamf_render_fir_m2b.hip
extern "C" int iamf_hip_fir_m2b_launch_fft(const void *params, int m, hipStream_t st);  // iamf_render_fir_m2b.hip
extern "C" int iamf_hip_lpcm_unpack_frames(const iamf_hip_lpcm_layout *lay, const void *d_raw, int64_t raw_stream_stride,   // iamf_unpack.hip
                                           int64_t raw_frame_stride, int32_t n_frames, const int32_t *d_first_count,
                                           int64_t first_count_stride, float *d_out, int64_t out_stream_stride,
                                           int64_t out_frame_stride, int32_t n_streams, void *stream, int32_t uniform_first,
                                           int32_t uniform_count);
extern "C" int iamf_hip_fast_lpcm_has(int m, int oc);                                   // iamf_render_lpcm.hip
extern "C" int iamf_hip_fast_lpcm_launch(const void *params, int m, hipStream_t st);   // iamf_render_lpcm.hip
extern "C" int iamf_hip_wide4_has_mix(int m, int c);                                  // iamf_render_wide4_mix.hip
extern "C" int iamf_hip_wide4_mix_launch(const void *params, int m, hipStream_t st);  // iamf_render_wide4_mix.hip
extern "C" int iamf_hip_wide4_has_lfe(int m, int c);                                  // iamf_render_wide4_lfe.hip
extern "C" int iamf_hip_wide4_lfe_launch(const void *params, int m, hipStream_t st);  // iamf_render_wide4_lfe.hip
extern "C" int iamf_hip_wide4_has_downmixer(int m, int c);                            // iamf_render_wide4.hip
extern "C" int iamf_hip_wide4_has_demixer(int m, int c);                              // iamf_render_wide4.hip
extern "C" int iamf_hip_wide4_has(int m, int c);                                      // iamf_render_wide4.hip
extern "C" int iamf_hip_wide4_launch(const void *params, int m, hipStream_t st);      // iamf_render_wide4.hip

namespace {

#include "render_common.hpp"
#include "render_downmix.hpp"
#include "render_fir.hpp"
#include "render_fir16.hpp"
#include "render_fir_fft.hpp"
#include "render_fast.hpp"
#include "render_generic.hpp"
#include "render_nolim.hpp"
#include "render_wide.hpp"
#include "render_lfe.hpp"

// PCM with a fixed channel stride (the -DSAMSUNG_TV build: iamf_decoder_plane2stride_out with
// stride = SAMSUNG_SPECIFIC_CHANNELS = 12, IAMF_decoder.c:121-167,3492-3495).  The reference zeroes
// n * stride elements and then writes, channel by channel, element i * stride + c for every sample i:
// with more channels than the stride those land in the NEXT sample-frames' slots and the last writer
// wins (the highest channel), the last sample's surplus channels past the zeroed area.  One thread per
// output element reproduces exactly that.  src: natural interleaving [n][ch].
__global__ __launch_bounds__(256) void restride_kernel(const uint8_t *src, int64_t src_stream_stride, uint8_t *dst,
                                                       int64_t dst_stream_stride, int n, int ch, int sc, int bps, int stream0) {
  const int s = blockIdx.y + stream0;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_out = (int64_t)n * sc + (ch > sc ? ch - sc : 0);
  if (p >= n_out) return;
  const int c0 = (int)(p % sc);
  const int64_t i = p / sc;
  const uint8_t *from = nullptr;
  for (int k = (ch - 1 - c0) / sc; k >= 0 && ch > c0; --k) {   // the highest channel that maps here
    const int64_t ii = i - k;
    if (ii >= 0 && ii < n) {
      from = src + (int64_t)s * src_stream_stride + (ii * ch + c0 + (int64_t)sc * k) * bps;
      break;
    }
  }
  if (!from && p >= (int64_t)n * sc) return;   // past the zeroed area and nobody writes it
  uint8_t *to = dst + (int64_t)s * dst_stream_stride + p * bps;
  for (int b = 0; b < bps; ++b) to[b] = from ? from[b] : 0;
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
extern "C" const uint8_t _binary_rdr_tables_tv_bin_start[];   // the -DSAMSUNG_TV build's tables (m2m_rdr.c:36-830)
extern "C" const uint8_t _binary_rdr_tables_tv_bin_end[];

struct Tables {
  int n = 0;
  const TableEntry *ents = nullptr;
  const float *data = nullptr;
  bool ok = false;
  explicit Tables(bool tv) {
    const uint8_t *b = tv ? _binary_rdr_tables_tv_bin_start : _binary_rdr_tables_bin_start;
    const size_t sz = tv ? (size_t)(_binary_rdr_tables_tv_bin_end - _binary_rdr_tables_tv_bin_start)
                         : (size_t)(_binary_rdr_tables_bin_end - _binary_rdr_tables_bin_start);
    if (sz < 12 || memcmp(b, "IARDRTB1", 8) != 0) return;
    uint32_t cnt;
    memcpy(&cnt, b + 8, 4);
    n = (int)cnt;
    ents = reinterpret_cast<const TableEntry *>(b + 12);
    data = reinterpret_cast<const float *>(b + 12 + sizeof(TableEntry) * cnt);
    ok = true;
  }
};

const Tables &tables(bool tv) {
  static Tables t(false), t_tv(true);
  return tv ? t_tv : t;
}

int find_matrix(int kind, int in_id, int out_id, iamf_hip_matrix *out, bool tv = false) {
  const Tables &t = tables(tv);
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
  int device = 0;                    // the HIP device the batch was created on; every later call must run on it
  hipEvent_t done = nullptr;         // recorded behind every render / flush on the caller's stream: the synchronous
                                     // setters and destroy wait for IT, so the caller's stream may be gone by then
  bool rendered = false;
  int m = 0, n_feeds = 0;
  int32_t src_feed[kMaxOut];
  uint32_t nz_mask[6] = {0, 0, 0, 0, 0, 0};
  int sparse = 0;
  int32_t *d_src_feed = nullptr;
  float thr = 0.f;
  int n_atk = 0, n_end = 0;
  std::vector<int64_t> spos;        // samples consumed, per stream (streams advance together unless the range calls are used)
  std::vector<uint8_t> sflushed;    // per stream: the limiter's tail has been emitted
  bool any_rendered = false;        // the setters that must precede the first render check this
  bool lp_scale_ok = false;         // every non-zero weight w has w * 2^-15 exact and w * 2^-15 * sample normal (fused LPCM form)
  float *d_matrix = nullptr, *d_gains = nullptr, *d_ctab = nullptr, *d_ring_y = nullptr,
        *d_ring_pm = nullptr;
  LimState *d_lim = nullptr;
  std::vector<float> h_gains;
  // second element / down-mixer
  bool has2 = false;
  int m2 = 0;
  float *d_matrix2 = nullptr, *d_gains2 = nullptr;
  int32_t *d_src_feed2 = nullptr, *d_dmx_tab = nullptr;
  bool dmx = false;
  int dmx_n_in = 0, dmx_n_out = 0;
  float *d_pre = nullptr;
  int pre_l = 0;
  std::vector<float> h_fm;          // host copy of d_matrix ([n_feeds][m])
  float *d_matrix_pre = nullptr;    // [n_feeds][pre_l]: renderer matrix x de-mapping matrix (tolerance mode)
  uint32_t nz_mask_pre[6] = {0, 0, 0, 0, 0, 0};
  int sparse_pre = 0;
  bool demix = false;
  int demix_steps = 0, demix_skip = 0, demix_layout = 0, demix_gmask = 0, demix_w4 = 0;
  int32_t *d_demix_tab = nullptr;
  float *d_demix_ftab = nullptr;
  bool fir = false;
  int fir_taps = 0;
  float *d_fir_hist[2] = {nullptr, nullptr};
  void *d_fir_h16 = nullptr;    // split-f16 filter tables (render_fir16.hpp)
  float *d_fir_pq = nullptr, *d_fir_tw = nullptr;   // spectra and twiddles of the FFT stage (render_fir_fft.hpp)
  float *d_fir_zero = nullptr;                      // m * frame size zero floats for that stage
  float *d_fir_pre[2] = {nullptr, nullptr};         // the history at the input's channel stride (RenderParams::fir_pre), or null
  size_t fir_pre_bytes = 0;
  float *d_fir_y = nullptr;                         // [n_streams][2][total] f32: its output when it runs as its own kernel
  size_t fir_y_floats = 0;
  // iamf_hip_batch_render_lpcm, calls the fused kernel does not take: the unpacked element and the unpacker's {first, count}
  float *d_lp_in = nullptr;
  size_t lp_in_floats = 0;
  float *d_fir_id = nullptr;                        // 2 x 2 identity + slot map for the limiter / pack kernel behind it
  int32_t *d_fir_id_feed = nullptr;
  float fir_inv_scale = 1.f;
  int fir_cur = 0;
  // HOA LFE generator (render_lfe.hpp)
  bool lfe = false;
  float lfe_a1 = 0.f, lfe_a2 = 0.f, lfe_a3 = 0.f, lfe_b1 = 0.f, lfe_b2 = 0.f;
  double lfe_div = 0.0;
  float *d_lfe_state = nullptr, *d_lfe_next = nullptr, *d_lfe_u = nullptr;
  bool lfe_shared = false;   // d_lfe_state / d_lfe_next belong to another batch (iamf_hip_batch_share_lfe_state)
  size_t lfe_u_floats = 0;
  // fixed PCM channel stride (cfg.pcm_stride_channels): the kernels pack into d_nat, restride_kernel re-lays
  uint8_t *d_nat = nullptr;
  size_t nat_bytes = 0;
  uint8_t *d_dump = nullptr;   // RenderParams::dump
};

namespace {

// every entry point that touches device state: the caller's current device must be the batch's
bool on_batch_device(const iamf_hip_batch *b) {
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess && dev == b->device;
}

// the setters copy into buffers a queued render may still be reading (renders are asynchronous on the
// caller's stream, which may be non-blocking with respect to the null stream): wait for the batch's own
// event behind the last render (not for the stream handle, which the caller may have destroyed since)
int quiesce(iamf_hip_batch *b) {
  if (b->rendered) HIPCHK(hipEventSynchronize(b->done));
  return IAMF_HIP_OK;
}

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
  if (b->fir) {
    const size_t hb = sizeof(float) * (size_t)ns * b->m * 256;
    HIPCHK(hipMemset(b->d_fir_hist[0], 0, hb));
    HIPCHK(hipMemset(b->d_fir_hist[1], 0, hb));
    for (float *q : b->d_fir_pre)
      if (q) HIPCHK(hipMemset(q, 0, b->fir_pre_bytes));
  }
  if (b->lfe && !b->lfe_shared) {  // lfefilter_init zeroes both histories (h2m_rdr.c:1210-1211)
    HIPCHK(hipMemset(b->d_lfe_state, 0, sizeof(float) * 4 * (size_t)ns));
    HIPCHK(hipMemset(b->d_lfe_next, 0, sizeof(float) * 2 * (size_t)ns));
  }
  b->spos.assign((size_t)ns, 0);
  b->sflushed.assign((size_t)ns, 0);
  b->any_rendered = false;
  return IAMF_HIP_OK;
}

template <int M>
void launch_m(const RenderParams &p, dim3 grid, size_t lds_bytes, hipStream_t st) {
  hipLaunchKernelGGL(render_kernel<M>, grid, dim3(kChunk), lds_bytes, st, p);
}

// limiter off, one matrix-rendered element, constant gains (render_nolim.hpp): (streams, chunks of 1024 sample-frames)
template <int M>
void launch_nolim_m(const RenderParams &p, hipStream_t st) {
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  dim3 grid((unsigned)p.n_launch, (unsigned)((p.total + kNlChunk - 1) / kNlChunk));
  hipLaunchKernelGGL(render_nolim_kernel<M>, grid, dim3(256), (size_t)kNlChunk * p.out_ch * bytes, st, p);
}

template <int M>
void launch_fir_m(const RenderParams &p, dim3 grid, hipStream_t st) {
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 1>), 120 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 2>), 120 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 3>), 120 * 1024);
    opted.end();
  }
  // Three stages with one specification (render_fir.hpp): overlap-save FFT on the VALU (default, render_fir_fft.hpp),
  // split-f16 MFMA (IAMF_HIP_FIR_F16=1, render_fir16.hpp), f32 MFMA (IAMF_HIP_FIR_F32=1); they differ in the last bits
  const int stage = fir_stage_choice(p);
  if (stage == 3) {
    static_assert(fast_lds_floats(2, M, 3) * 4 <= 80 * 1024, "two workgroups per CU");
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 3>), grid, dim3(256), sizeof(float) * (size_t)fast_lds_floats(2, M, 3), st, p);
  } else if (stage == 2) {
    static_assert(fast_lds_floats(2, M, 2) * 4 <= 80 * 1024, "two workgroups per CU");
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 2>), grid, dim3(256), sizeof(float) * (size_t)fast_lds_floats(2, M, 2), st, p);
  } else {
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 1>), grid, dim3(512), sizeof(float) * (size_t)fast_lds_floats(2, M, 1), st, p);
  }
}

template <int M>
void launch_fft_m(const RenderParams &p, hipStream_t st) {   // render_fir_fft.hpp: fir_fft_kernel
  const dim3 g((unsigned)((p.total + kFftSpan - 1) / kFftSpan), (unsigned)p.n_launch);
  // (whole frames only: past a call that ends inside a frame the two-base fetch would read what the caller left in the rest
  //  of the frame — harmless to the samples that are kept unless it is a NaN, which a transform spreads over its block)
  if ((M & 1) == 0 && p.fir_pre && p.fir_pre_next && p.total % p.frame_size == 0 && !getenv("IAMF_HIP_FIR_GENERAL_FETCH"))
    hipLaunchKernelGGL((fir_fft_kernel<M, (M & 1) == 0>), g, dim3(256), sizeof(float) * (size_t)kFftLdsFloats, st, p, p.fir_y, 2 * (int64_t)p.total);
  else
    hipLaunchKernelGGL((fir_fft_kernel<M, false>), g, dim3(256), sizeof(float) * (size_t)kFftLdsFloats, st, p, p.fir_y, 2 * (int64_t)p.total);
}

template <int M>
void launch_fast_m(const RenderParams &p, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)fast_lds_floats(p.out_ch, M);
  // more than 64 KiB of dynamic LDS has to be opted into per kernel (gfx950 has 160 KiB per CU)
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 1>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2>), 80 * 1024);
    opted.end();
  }
  if (p.in2 || p.elem_ramp || p.elem2_ramp || p.out_ramp) {  // the mixing variant
    static OptIn opted2;
    if (opted2.begin()) {
      opted2.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 1, 0, false, true>), 80 * 1024);
      opted2.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 0, false, true>), 80 * 1024);
      opted2.end();
    }
    if (p.out_ch == 1)
      hipLaunchKernelGGL((render_fast_kernel<M, 1, 0, false, true>), grid, dim3(256), lds, st, p);
    else
      hipLaunchKernelGGL((render_fast_kernel<M, 2, 0, false, true>), grid, dim3(256), lds, st, p);
    return;
  }
  if (p.out_ch == 1)
    hipLaunchKernelGGL((render_fast_kernel<M, 1>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((render_fast_kernel<M, 2>), grid, dim3(256), lds, st, p);
}

// parametric down-mixer to mono / stereo: 7.1 -> {2, 1}, 5.1 -> {2, 1}, stereo -> mono
template <int M, int OC>
void launch_fast_down_mc(const RenderParams &p, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)fast_lds_floats(OC, M);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, OC, 0, true>), 80 * 1024);
    opted.end();
  }
  hipLaunchKernelGGL((render_fast_kernel<M, OC, 0, true>), grid, dim3(256), lds, st, p);
}
bool launch_fast_down(const RenderParams &p, int m, dim3 grid, hipStream_t st) {
  if (m == 8 && p.out_ch == 2) launch_fast_down_mc<8, 2>(p, grid, st);
  else if (m == 8 && p.out_ch == 1) launch_fast_down_mc<8, 1>(p, grid, st);
  else if (m == 6 && p.out_ch == 2) launch_fast_down_mc<6, 2>(p, grid, st);
  else if (m == 6 && p.out_ch == 1) launch_fast_down_mc<6, 1>(p, grid, st);
  else if (m == 2 && p.out_ch == 1) launch_fast_down_mc<2, 1>(p, grid, st);
  else return false;
  return true;
}

template <int M>
void launch_wide_m(const RenderParams &p, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)wide_lds_floats(p.out_ch, M);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide_kernel<M, false>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_wide_kernel<M, true>), 80 * 1024);
    opted.end();
  }
  if (p.use_mfma)
    hipLaunchKernelGGL((render_wide_kernel<M, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((render_wide_kernel<M, false>), grid, dim3(256), lds, st, p);
}

// The fast kernel takes aligned, limiter-on calls into 1- or 2-channel layouts; everything else
// (odd sizes, flush, limiter off, wide layouts) goes to the generic kernel.  Both are exact.
bool fast_path_ok(const RenderParams &p, bool down_mixer = false) {
  if (getenv("IAMF_HIP_FORCE_GENERIC") || p.og_ch < p.out_ch) return false;
  if (!p.limiter_on || !p.in || p.out_ch > 2 || p.n_end < kFWin) return false;
  if (p.pre_matrix || p.demix_on) return false;
  if (p.dmx_on && !(down_mixer && p.dmx_frames)) return false;
  if (p.elem_ramp || p.elem2_ramp || p.out_ramp) {  // per-sample gains: the mixing variant reads them 4 at a time
    if (p.dmx_on || p.fir_taps > 0 || (p.ramp_stream_stride & 3) || (p.elem2_ramp && !p.in2)) return false;
    if ((reinterpret_cast<uintptr_t>(p.elem_ramp) | reinterpret_cast<uintptr_t>(p.elem2_ramp) |
         reinterpret_cast<uintptr_t>(p.out_ramp)) & 15)
      return false;
  }
  if (p.in2 && (p.dmx_on || p.fir_taps > 0 || p.m2 > kFIn2 || (reinterpret_cast<uintptr_t>(p.in2) & 15) ||
                (p.in2_stream_stride & 3) || (p.in2_frame_stride & 3)))
    return false;  // a second element of up to 4 channels rides along (render_fast_kernel<.., IN2>)
  // (a position that is not a multiple of 16 — a trimmed first frame — from 240 samples on: render_fast.hpp `base`)
  if (((p.pos0 & 15) && p.pos0 < kDelay) || (p.total & 63) || (p.frame_size & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (p.in_stream_stride & 3) || (p.in_frame_stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(p.pcm) & 15) || (p.pcm_stream_stride & 15)) return false;
  {  // the kernel addresses a stream's input of one call with 32-bit byte offsets (buffer loads, render_fast.hpp)
    const int64_t frames = (int64_t)p.total / p.frame_size + 2;
    if (frames * p.in_frame_stride * 4 + 100 * (int64_t)p.frame_size >= (int64_t)1 << 31) return false;
    if (p.lpcm && frames * p.lpcm_frame_stride + ((int64_t)1 << 24) >= (int64_t)1 << 31) return false;
  }
  return true;
}

// The wide kernel: 3..24 output channels, limiter on, aligned calls.
// any_pos: the caller will launch render_wide4_kernel, which (like render_fast_kernel) places its ring per call; the
// 256-sample kernel of render_wide.hpp keeps absolute ring positions and needs the stream at a multiple of 16
bool wide_path_ok(const RenderParams &p, int m, bool with_stage = false, bool any_pos = false) {
  if (getenv("IAMF_HIP_FORCE_GENERIC") || p.og_ch < p.out_ch) return false;
  if (!p.limiter_on || !p.in || p.out_ch <= 2 || p.out_ch > kMaxOut || p.n_end < kWWin) return false;
  if (p.pre_matrix) return false;
  const bool mixing = p.in2 || p.elem_ramp || p.elem2_ramp || p.out_ramp;
  if (mixing && !with_stage) return false;
  if ((p.demix_on || p.dmx_on) && (!with_stage || mixing)) return false;  // demixer / down-mixer / mixer: wide4 variants only
  if (p.dmx_on && (!p.dmx_frames || p.demix_on)) return false;  // demixer AND down-mixer: generic kernel
  if (((p.pos0 & 15) && !(any_pos && p.pos0 >= kDelay)) || (p.total & 63)) return false;
  if ((reinterpret_cast<uintptr_t>(p.pcm) & 15) || (p.pcm_stream_stride & 15)) return false;
  return sizeof(float) * (size_t)wide_lds_floats(p.out_ch, m) <= 80 * 1024;
}

// The 4-samples-per-lane wide kernel (render_wide4.hpp): 16-bit PCM, whole 1024-sample chunks,
// 16-byte aligned planar input, an instantiated (inputs, outputs) pair.
bool wide4_path_ok(const RenderParams &p, int m) {
  if (getenv("IAMF_HIP_NO_WIDE4")) return false;
  // whole 1024-sample chunks; the last one may be short if it still holds the 256 samples of stream state
  if (p.out_format != IAMF_HIP_FMT_S16 || ((p.total & 1023) && (p.total & 1023) < 256) ||
      (p.frame_size & 3) || p.n_end < 1088)
    return false;
  if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (p.in_stream_stride & 3) || (p.in_frame_stride & 3)) return false;
  if (p.in2 || p.elem_ramp || p.elem2_ramp || p.out_ramp) {  // the mixing variant (render_wide4.hpp, MIX)
    if (p.in2 && (p.m2 > kFIn2 || (reinterpret_cast<uintptr_t>(p.in2) & 15) || (p.in2_stream_stride & 3) ||
                  (p.in2_frame_stride & 3)))
      return false;
    if ((p.ramp_stream_stride & 3) || (p.elem2_ramp && !p.in2) ||
        ((reinterpret_cast<uintptr_t>(p.elem_ramp) | reinterpret_cast<uintptr_t>(p.elem2_ramp) |
          reinterpret_cast<uintptr_t>(p.out_ramp)) & 15))
      return false;
    return iamf_hip_wide4_has_mix(m, p.out_ch) != 0;
  }
  if (p.dmx_on) return iamf_hip_wide4_has_downmixer(m, p.out_ch) != 0;
  if (p.demix_on)  // scalable channel audio: the variant with the demixer in front of the projection
    return p.demix_w4 && !p.use_mfma && (p.demix_i0 & 3) == 0 && iamf_hip_wide4_has_demixer(m, p.out_ch) != 0;
  return iamf_hip_wide4_has(m, p.out_ch) != 0;
}

int launch(const RenderParams &p, int m, size_t lds_bytes, hipStream_t st) {
  dim3 grid((unsigned)p.n_launch);
  if (p.lpcm) {  // element 0 as LPCM packets: render_call has checked that this is a call of the fast kernel
    if (!iamf_hip_fast_lpcm_launch(&p, m, st)) return IAMF_HIP_ERR_INVALID_STATE;
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  if (p.fir_taps > 0 && p.in) {  // HRTF renderer: aligned calls only (the flush goes to the generic kernel)
    if (!fast_path_ok(p)) return IAMF_HIP_ERR_UNIMPLEMENTED;
    // the FIR stage keeps input offsets of one stream as 32-bit integers
    if (((int64_t)(p.total / p.frame_size) + 1) * p.in_frame_stride >= (int64_t)1 << 31) return IAMF_HIP_ERR_BAD_ARG;
    if (fir_stage_choice(p) == 4) {
      // (1) the FFT stage for every hop of every stream -> y in HBM (overlap-save blocks are independent: one grid); (2)
      // gains, limiter, pack = the two-channel matrix kernel with the identity over y (one "frame" of `total` samples per
      // stream).  1024 streams x 64 frames: 1.5 ms (vector ALU) + 0.6 ms (the limiter's chain, four workgroups a CU).
      // Tried and dropped, both bit-exact: slices of STREAMS with the limiter kernel of one slice beside the stage of the
      // next on a second stream (23.8 against 33.2 Gsamples/s: the limiter kernel takes 0.6 ms for 256 streams as for
      // 1024), and slices of TIME the same way (31.1: four resident limiter workgroups and two stage workgroups each want
      // all 512 VGPRs of a SIMD lane, so the two kernels take turns instead of overlapping).
      switch (m) {
        case 1: launch_fft_m<1>(p, st); break;
        case 4: launch_fft_m<4>(p, st); break;
        case 9: launch_fft_m<9>(p, st); break;
        case 16: launch_fft_m<16>(p, st); break;
        default:
          if (!iamf_hip_fir_m2b_launch_fft(&p, m, st)) return IAMF_HIP_ERR_UNIMPLEMENTED;
      }
      HIPCHK(hipGetLastError());
      RenderParams q = p;
      q.in = p.fir_y;                               // planar [2][total]: channel stride = "frame size" = total
      q.in_stream_stride = 2 * (int64_t)p.total;
      q.in_frame_stride = 2 * (int64_t)p.total;
      q.frame_size = p.total;
      q.matrix = p.fir_id_matrix;
      q.src_feed = p.fir_id_feed;
      q.n_feeds = 2;
      q.fir_taps = 0;
      q.fir_hist = q.fir_hist_next = nullptr;
      launch_fast_m<2>(q, grid, st);
      return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
    }
    switch (m) {
      case 1: launch_fir_m<1>(p, grid, st); break;
      case 4: launch_fir_m<4>(p, grid, st); break;
      case 9: launch_fir_m<9>(p, grid, st); break;
      case 16: launch_fir_m<16>(p, grid, st); break;
      default:  // channel-based elements (M2B): the loudspeaker layouts' channel counts
        if (!iamf_hip_fir_m2b_launch(&p, m, st)) return IAMF_HIP_ERR_UNIMPLEMENTED;
    }
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  if (p.dmx_on && fast_path_ok(p, true) && launch_fast_down(p, m, grid, st)) {
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  // an LFE slot is filled by render_wide4_kernel<.., LFE> where that exists, else by the generic kernel
  if (p.lfe && !p.lfe_k0 && !p.demix_on && !p.dmx_on && !p.in2 && !p.elem_ramp && !p.elem2_ramp && !p.out_ramp && !p.pre_matrix &&
      wide_path_ok(p, m) && wide4_path_ok(p, m) && iamf_hip_wide4_has_lfe(m, p.out_ch) &&
      iamf_hip_wide4_lfe_launch(&p, m, st)) {
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  const bool fast = !p.lfe && fast_path_ok(p);
  const bool wide = !p.lfe && !fast && wide_path_ok(p, m);
  const bool mixing = p.in2 || p.elem_ramp || p.elem2_ramp || p.out_ramp;
  const bool wide_any = !p.lfe && !fast && wide_path_ok(p, m, false, true);
  if (!p.lfe && (wide_any || ((p.demix_on || p.dmx_on || mixing) && wide_path_ok(p, m, true, true))) && wide4_path_ok(p, m) &&
      (mixing ? iamf_hip_wide4_mix_launch(&p, m, st) : iamf_hip_wide4_launch(&p, m, st))) {
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  if (!p.limiter_on && p.in && !p.lfe && !p.pre_matrix && !p.demix_on && !p.dmx_on && !mixing && p.fir_taps == 0 &&
      p.og_ch >= p.out_ch && !getenv("IAMF_HIP_FORCE_GENERIC") && nolim_shape_ok(p)) {
    switch (m) {
#define CASE_N(v) case v: launch_nolim_m<v>(p, st); break;
      CASE_N(1) CASE_N(2) CASE_N(4) CASE_N(6) CASE_N(8) CASE_N(9) CASE_N(10) CASE_N(11) CASE_N(12) CASE_N(14) CASE_N(16) CASE_N(24)
#undef CASE_N
      default: return IAMF_HIP_ERR_UNIMPLEMENTED;
    }
    HIPCHK(hipGetLastError());
    return IAMF_HIP_OK;
  }
  switch (m) {
#define CASE_M(v)                              \
  case v:                                      \
    if (fast)                                  \
      launch_fast_m<v>(p, grid, st);           \
    else if (wide)                             \
      launch_wide_m<v>(p, grid, st);           \
    else                                       \
      launch_m<v>(p, grid, lds_bytes, st);     \
    break;
    CASE_M(1) CASE_M(2) CASE_M(4) CASE_M(6) CASE_M(8) CASE_M(9) CASE_M(10) CASE_M(12) CASE_M(14) CASE_M(16) CASE_M(24)
#undef CASE_M
    // 11 inputs: no element of the reference has them, but the stage behind the resampler takes the OUTPUT layout's
    // channels through the identity, and Sound System E has 11 (found by tests/test_gpu_fuzz_facade.py: resampling into
    // layout E was refused).  The general kernel only.
    case 11: launch_m<11>(p, grid, lds_bytes, st); break;
    default: return IAMF_HIP_ERR_UNIMPLEMENTED;
  }
  HIPCHK(hipGetLastError());
  return IAMF_HIP_OK;
}

// layout tables of the down-mixer: playback channel order per IAChannelLayoutType (reference
// IAMF_utils.c:117-133) and surround / top counts (:157-161)
const int kLayoutCount[9] = {1, 2, 6, 8, 10, 8, 10, 12, 6};
const int kLayoutCh[9][12] = {
    {kChMono},
    {kChL2, kChR2},
    {kChL7, kChR7, kChC, kChLFE, kChSL5, kChSR5},
    {kChL7, kChR7, kChC, kChLFE, kChSL5, kChSR5, kChHL, kChHR},
    {kChL7, kChR7, kChC, kChLFE, kChSL5, kChSR5, kChHFL, kChHFR, kChHBL, kChHBR},
    {kChL7, kChR7, kChC, kChLFE, kChSL7, kChSR7, kChBL7, kChBR7},
    {kChL7, kChR7, kChC, kChLFE, kChSL7, kChSR7, kChBL7, kChBR7, kChHL, kChHR},
    {kChL7, kChR7, kChC, kChLFE, kChSL7, kChSR7, kChBL7, kChBR7, kChHFL, kChHFR, kChHBL, kChHBR},
    {kChL3, kChR3, kChC, kChLFE, kChTL, kChTR},
};
const int kLayoutSurround[9] = {1, 2, 5, 5, 5, 7, 7, 7, 3};
const int kLayoutTop[9] = {0, 0, 0, 2, 4, 0, 2, 4, 2};

// streams [s0, s0 + cnt) of the batch; they must stand at the same position (samples consumed so far)
// Element 0 as LPCM packets for the fused kernel (render_fast_kernel<.., LP>, iamf_render_lpcm.hip).  render_call
// returns kNotFused — before it has changed or launched anything — when the call is not one that kernel takes; the caller
// (iamf_hip_batch_render_lpcm) then unpacks to f32 and renders as usual.
struct LpcmIn {
  const uint8_t *raw;
  int64_t stream_stride, frame_stride;
  int32_t off[16];
};
constexpr int kNotFused = INT32_MIN;

// HOA LFE generator over `ltotal` samples of the element rows at d_in: feed-forward part in parallel, the recurrence one lane
// per stream, both on the caller's stream (render_lfe.hpp); the output lies in b->d_lfe_u (t4 quads per stream), the filter
// state of streams [s0, s0 + cnt) has moved on.
int lfe_prepass(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride, int64_t in_frame_stride, int ltotal,
                hipStream_t st, int s0, int cnt, int *t4_out) {
  const int ns = b->cfg.n_streams, nb = (ns + 63) / 64, t4 = (ltotal + 3) / 4;
  const size_t need_u = (size_t)nb * t4 * 64 * 4;
  if (need_u > b->lfe_u_floats) {  // grows with the largest call seen
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(b->d_lfe_u);
    b->d_lfe_u = nullptr;
    b->lfe_u_floats = 0;
    HIPCHK(hipMalloc(&b->d_lfe_u, sizeof(float) * need_u));
    b->lfe_u_floats = need_u;
  }
  LfeParams lp;
  memset(&lp, 0, sizeof(lp));
  lp.in = d_in;
  lp.in_stream_stride = in_stream_stride;
  lp.in_frame_stride = in_frame_stride;
  lp.pre_matrix = b->d_pre;   // projection mode: W is channel 0 AFTER the de-mapping, in every mode
  lp.pre_l = b->pre_l;
  lp.pre_m = b->m;
  lp.frame_size = b->cfg.frame_size;
  lp.n_streams = ns;
  lp.total = ltotal;
  lp.t4 = t4;
  lp.a1 = b->lfe_a1; lp.a2 = b->lfe_a2; lp.a3 = b->lfe_a3; lp.b1 = b->lfe_b1; lp.b2 = b->lfe_b2;
  lp.state = b->d_lfe_state;
  lp.state_next = b->d_lfe_next;
  lp.u_t = reinterpret_cast<float4 *>(b->d_lfe_u);
  lp.s_first = s0;
  lp.s_count = cnt;
  hipLaunchKernelGGL(lfe_ff_kernel, dim3((unsigned)((t4 + kLfeTileQ - 1) / kLfeTileQ), (unsigned)nb), dim3(256), 0, st, lp);
  hipLaunchKernelGGL(lfe_chain_kernel, dim3((unsigned)nb), dim3(64), 0, st, lp);
  HIPCHK(hipGetLastError());
  *t4_out = t4;
  return IAMF_HIP_OK;
}

int render_call(iamf_hip_batch *b, const iamf_hip_render_args &a, int total, int s0, int cnt, const LpcmIn *lp = nullptr) {
  if (!on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (lp && (b->fir || b->lfe || b->d_pre || b->demix || b->dmx || b->has2 || a.d_element_ramp || a.d_element2_ramp ||
             a.d_output_ramp || !b->lp_scale_ok || getenv("IAMF_HIP_LPCM_UNFUSED")))
    return kNotFused;
  if (s0 < 0 || cnt <= 0 || s0 + cnt > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  if (a.lfe_pre_samples < 0 || a.lfe_post_samples < 0 ||
      ((a.lfe_pre_samples || a.lfe_post_samples) &&
       (a.n_frames != 1 || (int64_t)a.lfe_pre_samples + total + a.lfe_post_samples > b->cfg.frame_size)))
    return IAMF_HIP_ERR_BAD_ARG;
  const int64_t pos = b->spos[(size_t)s0];
  for (int i = s0; i < s0 + cnt; ++i)
    if (b->spos[(size_t)i] != pos || b->sflushed[(size_t)i]) return IAMF_HIP_ERR_INVALID_STATE;
  const bool whole = s0 == 0 && cnt == b->cfg.n_streams;
  // per-batch (not per-stream) state: the FIR history ping-pong.  (The LFE generator's pre-pass takes the range: streams
  // outside it keep their filter state, round 4.)
  if (!whole && b->fir) return IAMF_HIP_ERR_UNIMPLEMENTED;
  RenderParams p;
  memset(&p, 0, sizeof(p));
  p.stream0 = s0;
  p.n_launch = cnt;
  p.in = a.d_in;
  p.in_stream_stride = a.in_stream_stride;
  p.in_frame_stride = a.in_frame_stride;
  p.pcm = static_cast<uint8_t *>(a.d_pcm);
  p.pcm_stream_stride = a.pcm_stream_stride_bytes;
  p.matrix = b->d_matrix;
  p.gains = b->d_gains;
  p.ctab = b->d_ctab;
  p.lim = b->d_lim;
  p.ring_y = b->d_ring_y;
  p.ring_pm = b->d_ring_pm;
  p.pos0 = pos;
  p.total = total;
  p.frame_size = b->cfg.frame_size;
  p.n_streams = b->cfg.n_streams;
  p.n_feeds = b->n_feeds;
  p.out_ch = b->cfg.out_channels;
  p.og_ch = (b->cfg.out_gain_channels > 0 && b->cfg.out_gain_channels < b->cfg.out_channels) ? b->cfg.out_gain_channels : b->cfg.out_channels;
  p.out_format = b->cfg.out_format;
  p.limiter_on = b->cfg.limiter_enable ? 1 : 0;
  p.loudness_on = b->cfg.loudness_enable ? 1 : 0;
  bool tolerance = false;
  {
    const int proj = b->cfg.projection;
    const char *env = getenv("IAMF_HIP_PROJECTION");  // "exact" / "mfma" override for experiments
    bool mf = proj == IAMF_HIP_PROJ_MFMA || (proj == IAMF_HIP_PROJ_AUTO && b->cfg.matrix.kind == IAMF_HIP_KIND_H2M);
    if (env && !strcmp(env, "exact")) mf = false;
    if (env && !strcmp(env, "mfma")) mf = true;
    p.use_mfma = (mf && b->n_feeds <= 32) ? 1 : 0;
    tolerance = mf;
  }
  p.n_atk = b->n_atk;
  p.n_end = b->n_end;
  p.thr = b->thr;
  p.src_feed = b->d_src_feed;
  p.dump = b->d_dump;
  for (int g = 0; g < 6; ++g) p.nz_mask[g] = b->nz_mask[g];
  p.sparse = getenv("IAMF_HIP_DENSE") ? 0 : b->sparse;
  if (b->has2 && a.d_in2) {
    p.in2 = a.d_in2;
    p.in2_stream_stride = a.in2_stream_stride;
    p.in2_frame_stride = a.in2_frame_stride;
    p.matrix2 = b->d_matrix2;
    p.src_feed2 = b->d_src_feed2;
    p.gains2 = b->d_gains2;
    p.m2 = b->m2;
  }
  p.elem_ramp = a.d_element_ramp;
  p.elem2_ramp = a.d_element2_ramp;
  p.out_ramp = a.d_output_ramp;
  p.ramp_stream_stride = a.ramp_stream_stride;
  if (b->dmx) {
    p.dmx_on = 1;
    p.dmx_frames = a.d_in ? a.d_dmx_frames : nullptr;
    p.dmx_n_in = b->dmx_n_in;
    p.dmx_n_out = b->dmx_n_out;
    p.dmx_tab = b->d_dmx_tab;
    p.dmx_in_layout = b->cfg.matrix.in_id;
    p.dmx_out_layout = b->cfg.matrix.out_id;
  }
  int m_eff = b->m;
  if (b->d_pre && a.d_in) {
    const int l = b->pre_l;
    const bool inst = l == 1 || l == 2 || l == 4 || l == 6 || l == 8 || l == 9 || l == 10 || l == 12 || l == 14 ||
                      l == 16 || l == 24;
    if (tolerance && b->d_matrix_pre && inst && !p.in2) {  // one composed matrix, see iamf_hip_batch_set_projection
      p.matrix = b->d_matrix_pre;
      for (int g = 0; g < 6; ++g) p.nz_mask[g] = b->nz_mask_pre[g];
      p.sparse = getenv("IAMF_HIP_DENSE") ? 0 : b->sparse_pre;
      m_eff = l;
    } else {
      p.pre_matrix = b->d_pre;
      p.pre_l = b->pre_l;
    }
  }
  if (b->demix && a.d_in) {
    p.demix_on = 1;
    p.demix_steps = b->demix_steps;
    p.demix_skip = b->demix_skip;
    p.demix_tab = b->d_demix_tab;
    p.demix_ftab = b->d_demix_ftab;
    p.demix_frames = a.d_demix_frames;
    p.demix_i0 = a.n_frames == 1 ? a.demix_sample0 : 0;
    p.demix_layout = b->demix_layout;
    p.demix_gmask = b->demix_gmask;
    p.demix_w4 = b->demix_w4;
  }
  if (lp) {
    p.lpcm = lp->raw;
    p.lpcm_stream_stride = lp->stream_stride;
    p.lpcm_frame_stride = lp->frame_stride;
    for (int m = 0; m < 16; ++m) p.lpcm_off[m] = lp->off[m];
    if (!fast_path_ok(p) || !iamf_hip_fast_lpcm_has(m_eff, p.out_ch)) return kNotFused;
  }
  if (b->fir) {
    p.fir_taps = b->fir_taps;
    p.fir_hist = b->d_fir_hist[b->fir_cur];
    p.fir_hist_next = b->d_fir_hist[b->fir_cur ^ 1];
    p.fir_h16 = b->d_fir_h16;
    p.fir_inv_scale = b->fir_inv_scale;
    p.fir_pq = b->d_fir_pq;
    p.fir_tw = b->d_fir_tw;
    p.fir_zero = b->d_fir_zero;
    p.fir_pre = b->d_fir_pre[b->fir_cur];
    p.fir_pre_next = b->d_fir_pre[b->fir_cur ^ 1];
    if (a.d_in && b->d_fir_pq && (b->cfg.frame_size & 63) == 0 && !getenv("IAMF_HIP_FIR_FUSED") && !getenv("IAMF_HIP_FIR_F16") &&
        !getenv("IAMF_HIP_FIR_F32")) {
      // the FFT stage's output of this call, [n_streams][2][total] (grows with the largest call seen)
      const size_t need = (size_t)b->cfg.n_streams * 2 * (size_t)total;
      if (need > b->fir_y_floats) {
        HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(a.stream)));
        (void)hipFree(b->d_fir_y);
        b->d_fir_y = nullptr;
        b->fir_y_floats = 0;
        HIPCHK(hipMalloc(&b->d_fir_y, sizeof(float) * need));
        b->fir_y_floats = need;
      }
      p.fir_y = b->d_fir_y;
      p.fir_id_matrix = b->d_fir_id;
      p.fir_id_feed = b->d_fir_id_feed;
    }
  }
  if (b->lfe && a.d_in && total > 0) {
    // a trimmed frame: the filter also runs over what is cut off (iamf_hip_render_args::lfe_pre_samples; checked above)
    const int lpre = a.lfe_pre_samples, lpost = a.lfe_post_samples;
    int t4 = 0;
    const int rc = lfe_prepass(b, a.d_in - lpre, a.in_stream_stride, a.in_frame_stride, lpre + total + lpost,
                               static_cast<hipStream_t>(a.stream), s0, cnt, &t4);
    if (rc) return rc;
    p.lfe = b->d_lfe_u;
    p.lfe_t4 = t4;
    p.lfe_k0 = lpre;
    p.lfe_div = b->lfe_div;
    for (int c = 0; c < b->cfg.out_channels && c < 32; ++c)
      if (b->src_feed[c] == -2) p.lfe_mask |= 1 << c;
  }
  const size_t lds = sizeof(float) * ((size_t)(p.out_ch + 2) * kRing + 3 * kChunk + kHead + 4 +
                                      ((b->dmx || b->demix) ? (size_t)kChCount * kChunk : 0));
  const int sc = b->cfg.pcm_stride_channels;
  const bool restride = sc > 0 && sc != p.out_ch;
  const int bps = iamf_hip_format_bytes(p.out_format);
  if (restride) {  // pack naturally into scratch, then re-lay with the fixed channel stride
    const size_t per_stream = (size_t)(total > kDelay ? total : kDelay) * p.out_ch * bps;
    const size_t need = per_stream * (size_t)p.n_streams;
    if (need > b->nat_bytes) {
      HIPCHK(hipStreamSynchronize(static_cast<hipStream_t>(a.stream)));
      (void)hipFree(b->d_nat);
      b->d_nat = nullptr;
      b->nat_bytes = 0;
      HIPCHK(hipMalloc(&b->d_nat, need));
      b->nat_bytes = need;
    }
    p.pcm = b->d_nat;
    p.pcm_stream_stride = (int64_t)per_stream;
  }
  const int r = launch(p, m_eff, lds, static_cast<hipStream_t>(a.stream));
  if (r != IAMF_HIP_OK) return r;
  if (b->fir && p.in) b->fir_cur ^= 1;
  const int64_t before = p.limiter_on ? (pos > kDelay ? pos - kDelay : 0) : pos;
  for (int i = s0; i < s0 + cnt; ++i) b->spos[(size_t)i] = pos + total;
  b->any_rendered = true;
  const int64_t after = p.limiter_on ? (pos + total > kDelay ? pos + total - kDelay : 0) : pos + total;
  const int n_emit = (int)(after - before);
  if (restride && n_emit > 0) {
    const int64_t n_out = (int64_t)n_emit * sc + (p.out_ch > sc ? p.out_ch - sc : 0);
    hipLaunchKernelGGL(restride_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)cnt), dim3(256), 0,
                       static_cast<hipStream_t>(a.stream), b->d_nat, p.pcm_stream_stride,
                       static_cast<uint8_t *>(a.d_pcm), a.pcm_stream_stride_bytes, n_emit, p.out_ch, sc, bps, s0);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(b->done, static_cast<hipStream_t>(a.stream)));
  b->rendered = true;
  return n_emit;
}

// feed-major copy of a renderer matrix and its output-slot map (h2m_rdr.c:1114-1150 keeps the
// slots lfe1/lfe2 free - the comparison is against the SOURCE index - and zeroes them; every
// other slot no feed reaches stays silent; m2m feeds map 1:1)
void build_feed_map(const iamf_hip_matrix &mx, int out_channels, std::vector<float> &fm, int32_t *src_feed) {
  for (int i = 0; i < kMaxOut; ++i) src_feed[i] = -1;
  for (int i = 0; i < mx.n; ++i) {
    int d = i;
    if (mx.kind == IAMF_HIP_KIND_H2M && (mx.lfe1 >= 0 || mx.lfe2 >= 0)) {
      if (mx.lfe1 >= 0 && mx.lfe1 <= i) ++d;
      if (mx.lfe2 >= 0 && mx.lfe2 <= i) ++d;
    }
    if (d < out_channels) src_feed[d] = i;
  }
  if (mx.kind == IAMF_HIP_KIND_H2M) {
    if (mx.lfe1 >= 0 && mx.lfe1 < out_channels) src_feed[mx.lfe1] = -1;
    if (mx.lfe2 >= 0 && mx.lfe2 < out_channels) src_feed[mx.lfe2] = -1;
  }
  fm.assign((size_t)mx.m * mx.n, 0.f);
  for (int f = 0; f < mx.n; ++f)
    for (int k = 0; k < mx.m; ++k)
      fm[(size_t)f * mx.m + k] = mx.kind == IAMF_HIP_KIND_H2M ? mx.mat[f * mx.m + k] : mx.mat[k * mx.n + f];
}

}  // namespace

extern "C" {

int iamf_hip_get_h2m_matrix(int order, int out_id, iamf_hip_matrix *out) {
  return find_matrix(IAMF_HIP_KIND_H2M, order, out_id, out);
}

int iamf_hip_get_m2m_matrix(int in_id, int out_id, iamf_hip_matrix *out) {
  return find_matrix(IAMF_HIP_KIND_M2M, in_id, out_id, out);
}

int iamf_hip_get_m2m_matrix_variant(int variant, int in_id, int out_id, iamf_hip_matrix *out) {
  if (variant != IAMF_HIP_VARIANT_DEFAULT && variant != IAMF_HIP_VARIANT_SAMSUNG_TV) return -1;
  return find_matrix(IAMF_HIP_KIND_M2M, in_id, out_id, out, variant == IAMF_HIP_VARIANT_SAMSUNG_TV);
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
  iamf_hip_matrix mx = cfg->matrix;
  const bool dmx = mx.kind == IAMF_HIP_KIND_DMX;
  const bool fir = mx.kind == IAMF_HIP_KIND_FIR;
  if (fir && (cfg->fir_taps < 1 || cfg->fir_taps > 256 || mx.n != 2 || cfg->out_channels != 2 ||
              !cfg->limiter_enable ||
              (mx.m != 1 && mx.m != 4 && mx.m != 9 && mx.m != 16 && !iamf_hip_fir_m2b_has(mx.m))))
    return IAMF_HIP_ERR_BAD_ARG;
  if (dmx) {
    if (!iamf_hip_dmx_valid(mx.in_id, mx.out_id)) return IAMF_HIP_ERR_BAD_ARG;
    mx.m = kLayoutCount[mx.in_id];
    mx.n = kLayoutCount[mx.out_id];
  }
  if (cfg->n_streams <= 0 || cfg->frame_size <= 0 || cfg->sample_rate <= 0 || (!dmx && !mx.mat) ||
      mx.m <= 0 || mx.m > kMaxIn || mx.n <= 0 || mx.n > kMaxOut || cfg->out_channels <= 0 ||
      cfg->out_channels > kMaxOut || !iamf_hip_format_bytes(cfg->out_format) || cfg->projection < 0 ||
      cfg->projection > IAMF_HIP_PROJ_MFMA || (dmx && cfg->out_channels != mx.n) || cfg->pcm_stride_channels < 0 || cfg->out_gain_channels < 0 ||
      cfg->pcm_stride_channels > 64)
    return IAMF_HIP_ERR_BAD_ARG;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return IAMF_HIP_ERR_DEVICE;

  iamf_hip_batch *b = new (std::nothrow) iamf_hip_batch();
  if (!b) return IAMF_HIP_ERR_ALLOC_FAIL;
  b->cfg = *cfg;
  if (hipGetDevice(&b->device) != hipSuccess) {
    delete b;
    return IAMF_HIP_ERR_DEVICE;
  }
  b->m = mx.m;
  b->n_feeds = mx.n;
  b->dmx = dmx;

  b->fir = fir;
  b->fir_taps = fir ? cfg->fir_taps : 0;
  std::vector<float> fm(1, 0.f);
  int32_t dmx_tab[24];
  memset(dmx_tab, 0, sizeof(dmx_tab));
  if (dmx) {
    for (int i = 0; i < kMaxOut; ++i) b->src_feed[i] = -1;
    b->dmx_n_in = mx.m;
    b->dmx_n_out = mx.n;
    for (int i = 0; i < mx.m; ++i) dmx_tab[i] = kLayoutCh[mx.in_id][i];
    for (int i = 0; i < mx.n; ++i) dmx_tab[12 + i] = kLayoutCh[mx.out_id][i];
  } else if (fir) {
    for (int i = 0; i < kMaxOut; ++i) b->src_feed[i] = -1;  // the flush renders silence
    fm.assign(mx.mat, mx.mat + (size_t)2 * mx.m * cfg->fir_taps);
  } else {
    build_feed_map(mx, cfg->out_channels, fm, b->src_feed);
    // The fused LPCM kernel folds the LPCM decoder's "/ 32768" into the weights: (w * 2^-15) * (float)s has the bits of
    // w * (s * 2^-15) as long as neither the scaled weight nor a product leaves the normal range — scaling by a power of two
    // commutes with rounding there.  True of every table of the reference (|w| in 1e-4 .. 2); a caller's own matrix with
    // weights below 2^-100 takes the unfused path.
    b->lp_scale_ok = true;
    for (float w : fm)
      if (w != 0.f && !(fabsf(w) >= 0x1p-100f && fabsf(w) <= 0x1p100f)) b->lp_scale_ok = false;
    if (cfg->lfe_hoa && mx.kind == IAMF_HIP_KIND_H2M && (mx.lfe1 >= 0 || mx.lfe2 >= 0)) {
      // HOA LFE generator on (the reference built -DDISABLE_LFE_HOA=0): lfefilter_init(plfe, 120, rate),
      // h2m_rdr.c:1192-1212, in the reference's own expression types
      const float sample_rate = (float)cfg->sample_rate, cutoff_freq = 120;
      const float delta_time = 1 / sample_rate + 1.0e-10;
      const float c = 1.0f / (float)tanf(M_PI * cutoff_freq * delta_time);
      b->lfe = true;
      b->lfe_a1 = 1.0f / (1.0f + c + c * c);
      b->lfe_a2 = 2.0f * b->lfe_a1;
      b->lfe_a3 = b->lfe_a1;
      b->lfe_b1 = 2.0f * (1.0f - c * c) * b->lfe_a1;
      b->lfe_b2 = (1.0f - c + c * c) * b->lfe_a1;
      b->lfe_div = mx.n <= 2 ? 0.0 : sqrt((double)mx.n);
      if (mx.lfe1 >= 0 && mx.lfe1 < cfg->out_channels) b->src_feed[mx.lfe1] = -2;
      if (mx.lfe2 >= 0 && mx.lfe2 < cfg->out_channels) b->src_feed[mx.lfe2] = -2;
    }
    // sparsity of the matrix as the output slots see it (render_wide4.hpp skips all-zero weight batches)
    int set = 0, all = 0;
    for (int g = 0; g < 6 && 4 * g < cfg->out_channels; ++g)
      for (int k = 0; k < mx.m && k < 32; ++k) {
        bool nz = false;
        for (int c = 4 * g; c < 4 * g + 4 && c < cfg->out_channels; ++c)
          if (b->src_feed[c] >= 0 && fm[(size_t)b->src_feed[c] * mx.m + k] != 0.f) nz = true;
        ++all;
        if (nz) {
          ++set;
          b->nz_mask[g] |= 1u << k;
        }
      }
    b->sparse = 2 * set < all ? 1 : 0;
    b->h_fm = fm;
  }

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
  CREATE_CHK(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
  CREATE_CHK(hipMalloc(&b->d_matrix, sizeof(float) * fm.size()));
  CREATE_CHK(hipMalloc(&b->d_gains, sizeof(float) * 3 * ns));
  CREATE_CHK(hipMalloc(&b->d_ctab, sizeof(float) * ctab.size()));
  CREATE_CHK(hipMalloc(&b->d_lim, sizeof(LimState) * ns));
  CREATE_CHK(hipMalloc(&b->d_ring_y, sizeof(float) * (size_t)ns * cfg->out_channels * kSave));
  CREATE_CHK(hipMalloc(&b->d_ring_pm, sizeof(float) * (size_t)ns * kSave));
  if (b->lfe) {
    CREATE_CHK(hipMalloc(&b->d_lfe_state, sizeof(float) * 4 * (size_t)ns));
    CREATE_CHK(hipMalloc(&b->d_lfe_next, sizeof(float) * 2 * (size_t)ns));
  }
  CREATE_CHK(hipMalloc(&b->d_dump, (size_t)ns * 256 * 16));
  CREATE_CHK(hipMalloc(&b->d_src_feed, sizeof(int32_t) * kMaxOut));
  CREATE_CHK(hipMemcpy(b->d_src_feed, b->src_feed, sizeof(int32_t) * kMaxOut, hipMemcpyHostToDevice));
  if (fir) {
    const size_t hb = sizeof(float) * (size_t)ns * mx.m * 256;
    CREATE_CHK(hipMalloc(&b->d_fir_hist[0], hb));
    CREATE_CHK(hipMalloc(&b->d_fir_hist[1], hb));
    CREATE_CHK(hipMemset(b->d_fir_hist[0], 0, hb));
    CREATE_CHK(hipMemset(b->d_fir_hist[1], 0, hb));
    // Tables of render_fir16.hpp: per channel [ear][hi/lo][shift r][304] halves with
    // table_r[n] = hp[n + r], hp[j] = h[j - 15] * scale split as hi + lo * 2^-11.  The scale is the
    // power of two that puts the largest tap in [2^13, 2^14).
    const int taps = cfg->fir_taps;
    float hmax = 0.f;
    for (size_t i = 0; i < (size_t)2 * mx.m * taps; ++i) hmax = fmaxf(hmax, fabsf(mx.mat[i]));
    if (hmax > 0.f && hmax < 1e30f) {
      int e = 0;
      (void)frexpf(hmax, &e);  // hmax = f * 2^e, f in [0.5, 1)
      const float scale = ldexpf(1.f, 14 - e);
      std::vector<_Float16> tab((size_t)mx.m * 2 * 2 * 8 * kF16Taps, (_Float16)0.f);
      for (int ch = 0; ch < mx.m; ++ch)
        for (int ear = 0; ear < 2; ++ear)
          for (int r = 0; r < 8; ++r)
            for (int n = 0; n < kF16Taps; ++n) {
              const int k = n + r - 15;  // tap index of hp[n + r]
              if (k < 0 || k >= taps) continue;
              const float v = mx.mat[((size_t)ear * mx.m + ch) * taps + k] * scale;
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
              const size_t at = ((((size_t)ch * 2 + ear) * 2 + 0) * 8 + r) * kF16Taps + n;
              tab[at] = hi;
              tab[at + (size_t)8 * kF16Taps] = lo;
            }
      CREATE_CHK(hipMalloc(&b->d_fir_h16, tab.size() * sizeof(_Float16)));
      CREATE_CHK(hipMemcpy(b->d_fir_h16, tab.data(), tab.size() * sizeof(_Float16), hipMemcpyHostToDevice));
      b->fir_inv_scale = 1.f / (scale * kF16InScale);
    }
    {  // the FFT stage's tables (render_fir_fft.hpp): pair spectra in the transform's own bin order, twiddles
      std::vector<float> pq, tw;
      fft_build_tables(mx.mat, mx.m, taps, pq, tw);
      CREATE_CHK(hipMalloc(&b->d_fir_pq, pq.size() * sizeof(float)));
      CREATE_CHK(hipMemcpy(b->d_fir_pq, pq.data(), pq.size() * sizeof(float), hipMemcpyHostToDevice));
      CREATE_CHK(hipMalloc(&b->d_fir_tw, tw.size() * sizeof(float)));
      CREATE_CHK(hipMemcpy(b->d_fir_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
      const size_t zf = (size_t)(mx.m > 0 ? mx.m : 1) * (size_t)(cfg->frame_size > 64 ? cfg->frame_size : 64);
      CREATE_CHK(hipMalloc(&b->d_fir_zero, zf * sizeof(float)));
      CREATE_CHK(hipMemset(b->d_fir_zero, 0, zf * sizeof(float)));
      if (cfg->frame_size >= 1024 && cfg->frame_size % 256 == 0 && mx.m % 2 == 0) {   // fir_fft_kernel<M, true>
        const int g = cfg->frame_size / 256;
        b->fir_pre_bytes = sizeof(float) * (size_t)((ns + g - 1) / g) * mx.m * cfg->frame_size;
        for (float *&q : b->d_fir_pre) {
          CREATE_CHK(hipMalloc(&q, b->fir_pre_bytes));
          CREATE_CHK(hipMemset(q, 0, b->fir_pre_bytes));
        }
      }
      const float id[4] = {1.f, 0.f, 0.f, 1.f};
      int32_t idf[kMaxOut];
      for (int i = 0; i < kMaxOut; ++i) idf[i] = i < 2 ? i : -1;
      CREATE_CHK(hipMalloc(&b->d_fir_id, sizeof(id)));
      CREATE_CHK(hipMemcpy(b->d_fir_id, id, sizeof(id), hipMemcpyHostToDevice));
      CREATE_CHK(hipMalloc(&b->d_fir_id_feed, sizeof(idf)));
      CREATE_CHK(hipMemcpy(b->d_fir_id_feed, idf, sizeof(idf), hipMemcpyHostToDevice));
    }
  }
  CREATE_CHK(hipMalloc(&b->d_dmx_tab, sizeof(dmx_tab)));
  CREATE_CHK(hipMemcpy(b->d_dmx_tab, dmx_tab, sizeof(dmx_tab), hipMemcpyHostToDevice));
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
  int cur = -1;   // free on the batch's own device whatever is current, and restore
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != b->device && hipSetDevice(b->device) == hipSuccess;
  struct Restore {
    bool on;
    int dev;
    ~Restore() {
      if (on) (void)hipSetDevice(dev);
    }
  } restore{sw, cur};
  if (b->rendered) (void)hipEventSynchronize(b->done);
  if (b->done) (void)hipEventDestroy(b->done);
  (void)hipFree(b->d_matrix_pre);
  (void)hipFree(b->d_matrix);
  (void)hipFree(b->d_gains);
  (void)hipFree(b->d_ctab);
  (void)hipFree(b->d_lim);
  (void)hipFree(b->d_ring_y);
  (void)hipFree(b->d_ring_pm);
  (void)hipFree(b->d_src_feed);
  (void)hipFree(b->d_dmx_tab);
  (void)hipFree(b->d_pre);
  (void)hipFree(b->d_demix_tab);
  (void)hipFree(b->d_demix_ftab);
  (void)hipFree(b->d_fir_hist[0]);
  (void)hipFree(b->d_fir_hist[1]);
  (void)hipFree(b->d_fir_h16);
  (void)hipFree(b->d_fir_pq);
  (void)hipFree(b->d_fir_tw);
  (void)hipFree(b->d_fir_zero);
  (void)hipFree(b->d_fir_pre[0]);
  (void)hipFree(b->d_fir_pre[1]);
  (void)hipFree(b->d_fir_y);
  (void)hipFree(b->d_lp_in);
  (void)hipFree(b->d_fir_id);
  (void)hipFree(b->d_fir_id_feed);
  if (!b->lfe_shared) {
    (void)hipFree(b->d_lfe_state);
    (void)hipFree(b->d_lfe_next);
  }
  (void)hipFree(b->d_lfe_u);
  (void)hipFree(b->d_nat);
  (void)hipFree(b->d_dump);
  (void)hipFree(b->d_matrix2);
  (void)hipFree(b->d_gains2);
  (void)hipFree(b->d_src_feed2);
  delete b;
}

int iamf_hip_batch_set_gains(iamf_hip_batch *b, const float *eg, const float *og, const float *lg) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  if (!on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  {  // a render queued earlier may still be reading d_gains
    const int q = quiesce(b);
    if (q != IAMF_HIP_OK) return q;
  }
  const int ns = b->cfg.n_streams;
  if (eg) memcpy(&b->h_gains[0], eg, sizeof(float) * ns);
  if (og) memcpy(&b->h_gains[(size_t)ns], og, sizeof(float) * ns);
  if (lg) memcpy(&b->h_gains[(size_t)2 * ns], lg, sizeof(float) * ns);
  HIPCHK(hipMemcpy(b->d_gains, b->h_gains.data(), sizeof(float) * 3 * ns, hipMemcpyHostToDevice));
  return IAMF_HIP_OK;
}

// the argument checks of a range call: IAMF_HIP_OK and the call's samples per stream in *total_out (0: nothing to do), or the error
static int range_args_check(const iamf_hip_batch *b, const iamf_hip_render_args *a, int32_t stream0, int32_t n_streams, int64_t *total_out) {
  *total_out = 0;
  if (!b || !a || !a->d_in || !a->d_pcm || a->n_frames < 0) return IAMF_HIP_ERR_BAD_ARG;
  if (stream0 < 0 || n_streams <= 0 || stream0 + n_streams > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  if (a->n_frames == 0) return IAMF_HIP_OK;
  if (b->has2 && !a->d_in2) return IAMF_HIP_ERR_BAD_ARG;
  if (b->dmx && !a->d_dmx_frames) return IAMF_HIP_ERR_BAD_ARG;
  if (b->demix && a->d_in && (!a->d_demix_frames || a->demix_sample0 < 0 ||
                              a->demix_sample0 + (a->n_samples ? a->n_samples : 1) > b->cfg.frame_size))
    return IAMF_HIP_ERR_BAD_ARG;
  if ((a->d_element_ramp || a->d_element2_ramp || a->d_output_ramp) &&
      a->ramp_stream_stride < (int64_t)a->n_frames * b->cfg.frame_size && b->cfg.n_streams > 1)
    return IAMF_HIP_ERR_BAD_ARG;
  int64_t total = (int64_t)a->n_frames * b->cfg.frame_size;
  if (a->n_samples > 0) {
    if (a->n_frames != 1 || a->n_samples > b->cfg.frame_size) return IAMF_HIP_ERR_BAD_ARG;
    total = a->n_samples;
  }
  if (total > INT32_MAX) return IAMF_HIP_ERR_BAD_ARG;
  const int sc = b->cfg.pcm_stride_channels > 0 ? b->cfg.pcm_stride_channels : b->cfg.out_channels;
  const int64_t need = (total * sc + (b->cfg.out_channels > sc ? b->cfg.out_channels - sc : 0)) *
                       iamf_hip_format_bytes(b->cfg.out_format);
  if (b->cfg.n_streams > 1 && a->pcm_stream_stride_bytes < need) return IAMF_HIP_ERR_BUFFER_TOO_SMALL;
  *total_out = total;
  return IAMF_HIP_OK;
}

static int render_range_impl(iamf_hip_batch *b, const iamf_hip_render_args *a, int32_t stream0, int32_t n_streams, const LpcmIn *lp) {
  int64_t total = 0;
  const int rc = range_args_check(b, a, stream0, n_streams, &total);
  if (rc != IAMF_HIP_OK || total == 0) return rc;
  return render_call(b, *a, (int)total, stream0, n_streams, lp);
}

int iamf_hip_batch_render_range(iamf_hip_batch *b, const iamf_hip_render_args *a, int32_t stream0, int32_t n_streams) {
  return render_range_impl(b, a, stream0, n_streams, nullptr);
}

// Element 0 as LPCM packets.  Fused (render_fast_kernel<.., LP>: the packets' 16-bit samples are converted where the
// render kernel loads them) when the call is one of the headline kernel's and every channel is a contiguous run of
// little-endian 16-bit samples; in every other case exactly what the caller would do itself: iamf_hip_lpcm_unpack into a
// buffer of the batch, then the f32 path.  Both give the same PCM bit for bit.
int iamf_hip_batch_render_lpcm(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *args) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  return iamf_hip_batch_render_lpcm_range(b, in, args, 0, b->cfg.n_streams);
}

// the same for the streams [stream0, stream0 + n_streams) only (as iamf_hip_batch_render_range: buffers and strides are
// still indexed by the stream's number in the batch)
int iamf_hip_batch_render_lpcm_range(iamf_hip_batch *b, const iamf_hip_lpcm_input *in, const iamf_hip_render_args *args,
                                     int32_t stream0, int32_t n_streams) {
  if (!b || !in || !args || !in->d_raw || args->d_in || !args->d_pcm || args->n_frames < 0) return IAMF_HIP_ERR_BAD_ARG;
  if (stream0 < 0 || n_streams <= 0 || stream0 + n_streams > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  if (!on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (args->n_frames == 0) return 0;
  const iamf_hip_lpcm_layout &L = in->layout;
  const int fs = b->cfg.frame_size, ch = b->d_pre ? b->pre_l : b->m, ns = b->cfg.n_streams, nf = args->n_frames;
  if (L.sample_bytes < 2 || L.sample_bytes > 4 || L.channels != ch || ch > IAMF_HIP_LPCM_MAX_CHANNELS || L.frame_size != fs ||
      (fs & 3) || in->raw_frame_stride <= 0 || in->raw_stream_stride < (int64_t)nf * in->raw_frame_stride)
    return IAMF_HIP_ERR_BAD_ARG;
  const int first = in->first_sample, count = args->n_samples > 0 ? args->n_samples : fs;
  if (first < 0 || (first > 0 && (nf != 1 || args->n_samples <= 0)) || first + count > fs) return IAMF_HIP_ERR_BAD_ARG;
  bool fusable = L.sample_bytes == 2 && L.little_endian && ch <= 16 && (in->raw_frame_stride & 7) == 0 &&
                 (in->raw_stream_stride & 7) == 0 && (reinterpret_cast<uintptr_t>(in->d_raw) & 15) == 0;
  for (int c = 0; c < ch; ++c) {   // every byte a kernel may read lies inside the frame's packet row (as iamf_hip_lpcm_unpack)
    if (L.src_offset[c] < 0) {
      fusable = false;
      continue;
    }
    if (L.src_step[c] < L.sample_bytes ||
        (int64_t)L.src_offset[c] + (int64_t)(fs - 1) * L.src_step[c] + L.sample_bytes > in->raw_frame_stride)
      return IAMF_HIP_ERR_BAD_ARG;
    if (L.src_step[c] != 2 || ((L.src_offset[c] + 2 * first) & 7)) fusable = false;
  }
  iamf_hip_render_args a = *args;
  if (fusable) {
    LpcmIn lp;
    memset(&lp, 0, sizeof(lp));
    lp.raw = static_cast<const uint8_t *>(in->d_raw);
    lp.stream_stride = in->raw_stream_stride;
    lp.frame_stride = in->raw_frame_stride;
    for (int c = 0; c < ch; ++c) lp.off[c] = L.src_offset[c] + 2 * first;
    a.d_in = reinterpret_cast<const float *>(in->d_raw);   // not read by the fused kernel; non-null = "not a flush"
    a.in_stream_stride = a.in_frame_stride = 0;
    const int r = render_range_impl(b, &a, stream0, n_streams, &lp);
    if (r != kNotFused) return r;
  }
  // Unfused: the RANGE's packets -> planar f32 in a buffer of the batch (rows indexed by the stream's number in the batch, as
  // every buffer of a range call), then the f32 path.  Everything the render call would refuse is refused here first, so
  // that no unpack launch precedes an error; {first, count} travel in the unpacker's kernel arguments (a device
  // word updated by a blocking copy cost K host synchronisations per round of a group with K divergent runs).
  {
    a.d_in = reinterpret_cast<const float *>(in->d_raw);   // placeholder for the checks: set below
    int64_t total = 0;
    const int rc = range_args_check(b, &a, stream0, n_streams, &total);
    if (rc != IAMF_HIP_OK || total == 0) return rc;
  }
  hipStream_t st = static_cast<hipStream_t>(args->stream);
  const size_t need = (size_t)ns * nf * ch * fs;
  if (need > b->lp_in_floats) {   // grows with the largest call seen
    const int q = quiesce(b);     // a render queued earlier may still be reading the old buffer
    if (q != IAMF_HIP_OK) return q;
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(b->d_lp_in);
    b->d_lp_in = nullptr;
    b->lp_in_floats = 0;
    HIPCHK(hipMalloc(&b->d_lp_in, sizeof(float) * need));
    // rows of channels no sub-stream carries (src_offset < 0) are never written by the unpacker: keep them defined
    HIPCHK(hipMemset(b->d_lp_in, 0, sizeof(float) * need));
    b->lp_in_floats = need;
  }
  // as many frames per launch as the grid's third dimension takes
  const int fmax = 65535 / n_streams > 0 ? 65535 / n_streams : 0;
  if (fmax == 0) return IAMF_HIP_ERR_UNIMPLEMENTED;
  const int64_t lp_stream_stride = (int64_t)nf * ch * fs;
  for (int f = 0; f < nf; f += fmax) {
    const int r = iamf_hip_lpcm_unpack_frames(
        &L, static_cast<const uint8_t *>(in->d_raw) + (int64_t)stream0 * in->raw_stream_stride + (int64_t)f * in->raw_frame_stride,
        in->raw_stream_stride, in->raw_frame_stride, nf - f < fmax ? nf - f : fmax, nullptr, 0,
        b->d_lp_in + (int64_t)stream0 * lp_stream_stride + (size_t)f * ch * fs, lp_stream_stride, (int64_t)ch * fs, n_streams, st, first,
        count);
    if (r != IAMF_HIP_OK) return r;
  }
  a.d_in = b->d_lp_in;
  a.in_stream_stride = lp_stream_stride;
  a.in_frame_stride = (int64_t)ch * fs;
  return render_range_impl(b, &a, stream0, n_streams, nullptr);
}

int iamf_hip_batch_render_ex(iamf_hip_batch *b, const iamf_hip_render_args *a) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  return iamf_hip_batch_render_range(b, a, 0, b->cfg.n_streams);
}

int iamf_hip_batch_render(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride,
                          int64_t in_frame_stride, int32_t n_frames, void *d_pcm,
                          int64_t pcm_stream_stride_bytes, void *stream) {
  iamf_hip_render_args a;
  memset(&a, 0, sizeof(a));
  a.d_in = d_in;
  a.in_stream_stride = in_stream_stride;
  a.in_frame_stride = in_frame_stride;
  a.n_frames = n_frames;
  a.d_pcm = d_pcm;
  a.pcm_stream_stride_bytes = pcm_stream_stride_bytes;
  a.stream = stream;
  if (b && (b->has2 || b->dmx || b->demix)) return IAMF_HIP_ERR_BAD_ARG;  // those need iamf_hip_batch_render_ex
  return iamf_hip_batch_render_ex(b, &a);
}

int iamf_hip_batch_flush_range(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stream_stride_bytes, void *stream,
                               int32_t stream0, int32_t n_streams) {
  if (!b || !d_pcm) return IAMF_HIP_ERR_BAD_ARG;
  if (stream0 < 0 || n_streams <= 0 || stream0 + n_streams > b->cfg.n_streams) return IAMF_HIP_ERR_BAD_ARG;
  for (int i = stream0; i < stream0 + n_streams; ++i)
    if (b->sflushed[(size_t)i]) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b->cfg.limiter_enable) return 0;
  iamf_hip_render_args a;
  memset(&a, 0, sizeof(a));  // no inputs: 240 zero samples go through the limiter
  a.d_pcm = d_pcm;
  a.pcm_stream_stride_bytes = pcm_stream_stride_bytes;
  a.stream = stream;
  const int r = render_call(b, a, kDelay, stream0, n_streams);
  if (r >= 0)
    for (int i = stream0; i < stream0 + n_streams; ++i) b->sflushed[(size_t)i] = 1;
  return r;
}

int iamf_hip_batch_flush(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stream_stride_bytes, void *stream) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  return iamf_hip_batch_flush_range(b, d_pcm, pcm_stream_stride_bytes, stream, 0, b->cfg.n_streams);
}

int iamf_hip_batch_set_second_element(iamf_hip_batch *b, const iamf_hip_matrix *mx, const float *g2) {
  if (b && !on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b || !mx || !mx->mat || mx->m <= 0 || mx->m > kMaxIn || mx->n <= 0 || mx->n > kMaxOut ||
      mx->kind == IAMF_HIP_KIND_DMX || b->any_rendered)
    return IAMF_HIP_ERR_BAD_ARG;
  std::vector<float> fm;
  int32_t feed[kMaxOut];
  build_feed_map(*mx, b->cfg.out_channels, fm, feed);
  const int ns = b->cfg.n_streams;
  std::vector<float> gains((size_t)ns, 1.0f);
  if (g2) memcpy(gains.data(), g2, sizeof(float) * ns);
  (void)hipFree(b->d_matrix2);
  (void)hipFree(b->d_gains2);
  (void)hipFree(b->d_src_feed2);
  b->d_matrix2 = nullptr;
  b->d_gains2 = nullptr;
  b->d_src_feed2 = nullptr;
  HIPCHK(hipMalloc(&b->d_matrix2, sizeof(float) * fm.size()));
  HIPCHK(hipMalloc(&b->d_gains2, sizeof(float) * ns));
  HIPCHK(hipMalloc(&b->d_src_feed2, sizeof(feed)));
  HIPCHK(hipMemcpy(b->d_matrix2, fm.data(), sizeof(float) * fm.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_gains2, gains.data(), sizeof(float) * ns, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_src_feed2, feed, sizeof(feed), hipMemcpyHostToDevice));
  b->m2 = mx->m;
  b->has2 = true;
  return IAMF_HIP_OK;
}

// A frame of an element that feeds the HOA LFE generator is rendered and then trimmed away COMPLETELY (start + end trim =
// its length, neither of them the whole frame): the reference's filter has run over it (iamf_stream_render precedes
// iamf_frame_trim, IAMF_decoder.c:3372-3406), nothing is emitted.  Streams [first, first + count) only.
int iamf_hip_batch_lfe_advance(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride, int32_t n_samples, void *stream,
                               int32_t first, int32_t count) {
  if (!b || !d_in || n_samples <= 0 || n_samples > b->cfg.frame_size || first < 0 || count <= 0 || first + count > b->cfg.n_streams)
    return IAMF_HIP_ERR_BAD_ARG;
  if (!on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b->lfe) return IAMF_HIP_OK;   // a layout without an LFE slot: no generator, nothing to advance
  int t4 = 0;
  return lfe_prepass(b, d_in, in_stream_stride, in_stream_stride, n_samples, static_cast<hipStream_t>(stream), first, count, &t4);
}

// The reference keeps the LFE generator's filter in the OUTPUT LAYOUT (IAMF_decoder.c:2629-2632: plfe =
// &stream->final_layout->sp.lfe_f), not in the stream: two scene-based elements of one presentation push their W channels
// through the same two histories in turn.  Two batches that render such a pair share the state: `b` adopts `owner`'s; the
// caller issues the calls of a frame in presentation order on one HIP stream.
int iamf_hip_batch_share_lfe_state(iamf_hip_batch *b, iamf_hip_batch *owner) {
  if (!b || !owner || b == owner) return IAMF_HIP_ERR_BAD_ARG;
  if (!on_batch_device(b) || !on_batch_device(owner)) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b->lfe || !owner->lfe || owner->lfe_shared || b->lfe_shared || b->any_rendered || owner->any_rendered ||
      b->cfg.n_streams != owner->cfg.n_streams || b->cfg.sample_rate != owner->cfg.sample_rate)
    return IAMF_HIP_ERR_BAD_ARG;
  (void)hipFree(b->d_lfe_state);
  (void)hipFree(b->d_lfe_next);
  b->d_lfe_state = owner->d_lfe_state;
  b->d_lfe_next = owner->d_lfe_next;
  b->lfe_shared = true;
  return IAMF_HIP_OK;
}

int iamf_hip_batch_set_projection(iamf_hip_batch *b, const float *matrix, int l_in) {
  if (b && !on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b || !matrix || l_in <= 0 || l_in > kMaxIn || b->any_rendered || b->dmx || b->fir) return IAMF_HIP_ERR_BAD_ARG;
  (void)hipFree(b->d_pre);
  b->d_pre = nullptr;
  HIPCHK(hipMalloc(&b->d_pre, sizeof(float) * (size_t)l_in * b->m));
  HIPCHK(hipMemcpy(b->d_pre, matrix, sizeof(float) * (size_t)l_in * b->m, hipMemcpyHostToDevice));
  b->pre_l = l_in;
  // Tolerance mode (IAMF_HIP_PROJ_MFMA, or AUTO with an H2M matrix: results within +-1 LSB of the
  // reference): de-mapping and rendering are both linear, so one matrix W' = W * P^T (products and
  // sums in double, rounded once) takes the decoded channels straight to the output slots and the
  // call runs on the same kernels as a mono-mode element with l_in channels.  IAMF_HIP_PROJ_EXACT
  // keeps the reference's two stages and their f32 roundings (generic kernel).
  (void)hipFree(b->d_matrix_pre);
  b->d_matrix_pre = nullptr;
  if (!b->h_fm.empty()) {
    std::vector<float> w((size_t)b->n_feeds * l_in);
    for (int f = 0; f < b->n_feeds; ++f)
      for (int l = 0; l < l_in; ++l) {
        double acc = 0;
        for (int r = 0; r < b->m; ++r) acc += (double)b->h_fm[(size_t)f * b->m + r] * (double)matrix[(size_t)l * b->m + r];
        w[(size_t)f * l_in + l] = (float)acc;
      }
    int set = 0, all = 0;
    for (int g = 0; g < 6; ++g) b->nz_mask_pre[g] = 0;
    for (int g = 0; g < 6 && 4 * g < b->cfg.out_channels; ++g)
      for (int k = 0; k < l_in && k < 32; ++k) {
        bool nz = false;
        for (int c = 4 * g; c < 4 * g + 4 && c < b->cfg.out_channels; ++c)
          if (b->src_feed[c] >= 0 && w[(size_t)b->src_feed[c] * l_in + k] != 0.f) nz = true;
        ++all;
        if (nz) {
          ++set;
          b->nz_mask_pre[g] |= 1u << k;
        }
      }
    b->sparse_pre = 2 * set < all ? 1 : 0;
    HIPCHK(hipMalloc(&b->d_matrix_pre, sizeof(float) * w.size()));
    HIPCHK(hipMemcpy(b->d_matrix_pre, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
  }
  return IAMF_HIP_OK;
}

/* ---- demixer of scalable channel audio (reference src/iamf_dec/demixer.c) ---- */

int iamf_hip_batch_set_demixer(iamf_hip_batch *b, const iamf_hip_demix_config *c) {
  if (b && !on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  if (!b || !c || b->any_rendered || b->fir || b->d_pre) return IAMF_HIP_ERR_BAD_ARG;
  if (c->layout < 0 || c->layout > 8 || c->n_in != kLayoutCount[c->layout] || c->n_in != b->m ||
      c->n_gain < 0 || c->n_gain > 12)
    return IAMF_HIP_ERR_BAD_ARG;
  // which de-mix steps run: the reference reconstructs on demand (dmx_channel, demixer.c:379-424);
  // a step is skipped when its right-hand output is already there
  bool have[kChCount];
  for (int i = 0; i < kChCount; ++i) have[i] = false;
  for (int i = 0; i < c->n_in; ++i) {
    if (c->chs_in[i] <= 0 || c->chs_in[i] >= kChCount || have[c->chs_in[i]]) return IAMF_HIP_ERR_BAD_ARG;
    have[c->chs_in[i]] = true;
  }
  int steps = 0;
  bool ok = true;
  auto s2 = [&]() { if (!have[kChL2]) { ok = false; return; } if (have[kChR2]) return;
                    if (!have[kChMono]) { ok = false; return; } steps |= 1; have[kChR2] = true; };
  auto s3 = [&]() { if (have[kChR3]) return; s2(); if (!ok) return; if (!have[kChC]) { ok = false; return; }
                    steps |= 2; have[kChL3] = have[kChR3] = true; };
  auto s5 = [&]() { if (have[kChSR5]) return; s3(); if (!ok) return;
                    if (!have[kChL7] || !have[kChR7]) { ok = false; return; }
                    steps |= 4; have[kChSL5] = have[kChSR5] = true; };
  auto s7 = [&]() { if (have[kChBR7]) return; s5(); if (!ok) return;
                    if (!have[kChSL7] || !have[kChSR7]) { ok = false; return; }
                    steps |= 8; have[kChBL7] = have[kChBR7] = true; };
  auto h2 = [&]() { if (have[kChHR]) return; if (!have[kChTL] || !have[kChTR]) { ok = false; return; }
                    s5(); if (!ok) return; steps |= 16; have[kChHL] = have[kChHR] = true; };
  auto h4 = [&]() { if (have[kChHBR]) return; h2(); if (!ok) return;
                    if (!have[kChHFR] || !have[kChHFL]) { ok = false; return; }
                    steps |= 32; have[kChHBL] = have[kChHBR] = true; };
  int32_t tab[64];
  memset(tab, 0, sizeof(tab));
  for (int i = 0; i < c->n_in; ++i) {
    const int ch = kLayoutCh[c->layout][i];
    tab[i] = c->chs_in[i];
    tab[40 + c->chs_in[i]] = i;  // IAChannel -> decoded position (render_wide4.hpp)
    tab[12 + i] = ch;
    if (have[ch]) continue;
    switch (ch) {
      case kChR2: s2(); break;
      case kChL3: case kChR3: s3(); break;
      case kChSL5: case kChSR5: s5(); break;
      case kChBL7: case kChBR7: s7(); break;
      case kChHL: case kChHR: h2(); break;
      case kChHBL: case kChHBR: h4(); break;
      default: ok = false; break;
    }
    if (!ok || !have[ch]) return IAMF_HIP_ERR_BAD_ARG;
  }
  const int fs = b->cfg.frame_size;
  std::vector<float> ft(12 + 2 * (size_t)fs + 12, 0.f);
  float *gin = ft.data() + 12 + 2 * (size_t)fs;  // the same gains by decoded channel, for render_wide4.hpp
  for (int k = 0; k < 12; ++k) gin[k] = 1.f;
  int ng = 0, gmask = 0;
  bool gain_twice = false;
  for (int i = 0; i < c->n_gain; ++i) {  // dmx_gainup touches only channels that were decoded
    const int ch = c->gain_ch[i];
    int at = -1;
    for (int k = 0; k < c->n_in; ++k)
      if (c->chs_in[k] == ch) at = k;
    if (ch <= 0 || ch >= kChCount) return IAMF_HIP_ERR_BAD_ARG;
    if (at < 0) continue;
    tab[25 + ng] = ch;
    ft[ng] = c->gain[i];
    ++ng;
    gain_twice = gain_twice || ((gmask >> at) & 1);
    gmask |= 1 << at;
    gin[at] = c->gain[i];
  }
  tab[24] = ng;
  // demixer_open + demixer_set_frame_offset (demixer.c:476-567): Hann cross-fade of fs/16 samples
  // behind the first `skip` samples of every frame
  float *start = ft.data() + 12, *stop = start + fs;
  for (int i = 0; i < fs; ++i) {
    start[i] = 1;
    stop[i] = 0;
  }
  const int wl = fs / 8, ov = wl / 2, pre = (int)(c->frame_offset % (uint32_t)fs);
  if (pre + ov <= fs) {
    std::vector<float> hann((size_t)(wl > 0 ? wl : 1));
    for (int i = 0; i < wl; ++i) hann[i] = (float)(0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(wl - 1))));
    for (int i = 0; i < pre; ++i) {
      start[i] = 0;
      stop[i] = 1;
    }
    for (int i = pre, j = 0; j < ov; ++i, ++j) {
      start[i] = hann[j];
      stop[i] = hann[j + ov];
    }
  }
  (void)hipFree(b->d_demix_tab);
  (void)hipFree(b->d_demix_ftab);
  b->d_demix_tab = nullptr;
  b->d_demix_ftab = nullptr;
  HIPCHK(hipMalloc(&b->d_demix_tab, sizeof(tab)));
  HIPCHK(hipMemcpy(b->d_demix_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&b->d_demix_ftab, sizeof(float) * ft.size()));
  HIPCHK(hipMemcpy(b->d_demix_ftab, ft.data(), sizeof(float) * ft.size(), hipMemcpyHostToDevice));
  b->demix = true;
  b->demix_steps = steps;
  b->demix_skip = pre;
  b->demix_layout = c->layout;
  b->demix_gmask = gmask;
  // the in-register demixer keeps the first 192 entries of the cross-fade windows and reads them 4 at a time
  b->demix_w4 = (!gain_twice && fs >= 256 && pre + ov <= 192 && (pre & 3) == 0) ? 1 : 0;
  return IAMF_HIP_OK;
}

namespace {
// demixer.c:66-78 (double literals narrowed to float) and fixedp11_5.c:81-82
const struct { float alpha, beta, gamma, delta; int woff; } kDemixMat[7] = {
    {1.0, 1.0, (float)0.707, (float)0.707, -1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, -1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, -1}, {0, 0, 0, 0, 0},
    {1.0, 1.0, (float)0.707, (float)0.707, 1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, 1},
    {1.0, (float)0.866, (float)0.866, (float)0.866, 1}};
const float kDemixW[11] = {0.0, (float)0.0179, (float)0.0391, (float)0.0658, (float)0.1038, 0.25,
                           (float)0.3962, (float)0.4342, (float)0.4609, (float)0.4821, 0.5};
static void demix_factors(int mode, int w_idx, float out[5]) {
  out[0] = kDemixMat[mode].alpha;
  out[1] = kDemixMat[mode].beta;
  out[2] = kDemixMat[mode].gamma;
  out[3] = kDemixMat[mode].delta;
  out[4] = kDemixW[w_idx < 0 ? 0 : (w_idx > 10 ? 10 : w_idx)];
}
}  // namespace

void iamf_hip_demix_state_init(iamf_hip_demix_state *st) {
  memset(st, 0, sizeof(*st));
  for (int i = 0; i < 24; ++i) st->last_sfavg[i] = 1.0f;
}

int iamf_hip_demix_set_info(iamf_hip_demix_state *st, int mode, int w_idx) {
  if (!st || mode < 0 || mode == 3 || mode > 6) return IAMF_HIP_ERR_BAD_ARG;
  if (w_idx < 0 || w_idx > 10) {
    st->last_mode = st->mode;
    st->mode = mode;
    st->last_w_idx = st->w_idx;
    if (kDemixMat[mode].woff > 0)
      st->w_idx = st->last_w_idx + 1 < 10 ? st->last_w_idx + 1 : 10;
    else
      st->w_idx = st->last_w_idx - 1 > 0 ? st->last_w_idx - 1 : 0;
  } else {
    if (mode != st->mode) st->last_mode = st->mode = mode;
    if (st->w_idx != w_idx) st->last_w_idx = st->w_idx = w_idx;
  }
  return IAMF_HIP_OK;
}

void iamf_hip_demix_frame_fill(iamf_hip_demix_state *st, int n_recon, const int32_t *recon_ch,
                               const float *recon_gain, iamf_hip_demix_frame *out) {
  memset(out, 0, sizeof(*out));
  demix_factors(st->last_mode, st->last_w_idx, out->prev);
  demix_factors(st->mode, st->w_idx, out->cur);
  const float N = 7;
  out->n_recon = n_recon < 0 ? 0 : (n_recon > 12 ? 12 : n_recon);
  for (int i = 0; i < out->n_recon; ++i) {  // dmx_rms, demixer.c:447-478
    const int ch = recon_ch[i] > 0 && recon_ch[i] < 24 ? recon_ch[i] : 0;
    out->recon_ch[i] = ch;
    const float sf = recon_gain ? recon_gain[i] : 1.0f;
    const float sfavg = (2 / (N + 1)) * sf + (1 - 2 / (N + 1)) * st->last_sfavg[ch];
    out->recon_prev[i] = st->last_sfavg[ch];
    out->recon_cur[i] = sfavg;
    st->last_sfavg[ch] = sfavg;
  }
}

/* ---- down-mixer control plane (downmix_renderer.c:77-91,131-216; IAMF_utils.c:234-245;
 *      fixedp11_5.c:81-99) ---- */
int iamf_hip_dmx_layout_channels(int layout) { return layout >= 0 && layout < 9 ? kLayoutCount[layout] : 0; }

int iamf_hip_dmx_valid(int in, int out) {
  if (in == out || in < 0 || in >= 9 || out < 0 || out >= 9) return 0;  // binaural (9) is refused
  if (kLayoutTop[in] && !kLayoutTop[out]) return 0;
  return !(kLayoutSurround[in] < kLayoutSurround[out] || kLayoutTop[in] < kLayoutTop[out]);
}

void iamf_hip_dmx_state_init(iamf_hip_dmx_state *st) {
  memset(st, 0, sizeof(*st));
  st->mode = -1;
  st->w_idx = -1;
}

int iamf_hip_dmx_set_mode_weight(iamf_hip_dmx_state *st, int mode, int w_idx) {
  static const struct { float a, b, g, d; int woff; } mixf[7] = {
      {1.0, 1.0, (float)0.707, (float)0.707, -1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, -1},
      {1.0, (float)0.866, (float)0.866, (float)0.866, -1}, {0, 0, 0, 0, 0},
      {1.0, 1.0, (float)0.707, (float)0.707, 1}, {(float)0.707, (float)0.707, (float)0.707, (float)0.707, 1},
      {1.0, (float)0.866, (float)0.866, (float)0.866, 1}};
  static const float wtab[11] = {0.0, (float)0.0179, (float)0.0391, (float)0.0658, (float)0.1038, 0.25,
                                 (float)0.3962, (float)0.4342, (float)0.4609, (float)0.4821, 0.5};
  if (!st || mode < 0 || mode == 3 || mode >= 7) return IAMF_HIP_ERR_BAD_ARG;
  if (st->mode != mode) {
    st->mode = mode;
    st->alpha = mixf[mode].a;
    st->beta = mixf[mode].b;
    st->gamma = mixf[mode].g;
    st->delta = mixf[mode].d;
    st->w_idx_offset = mixf[mode].woff;
  }
  if (w_idx < 0 || w_idx > 10) {  // no explicit index: step the weight state one way
    int nw = st->w_idx_offset > 0 ? (st->w_idx + 1 < 10 ? st->w_idx + 1 : 10) : (st->w_idx - 1 > 0 ? st->w_idx - 1 : 0);
    st->w_idx = nw;
    st->gamma_w = st->gamma * wtab[nw];
  } else if (st->w_idx != w_idx) {
    st->w_idx = w_idx;
    st->gamma_w = st->gamma * wtab[w_idx];
  }
  return IAMF_HIP_OK;
}

void iamf_hip_dmx_coefficients(const iamf_hip_dmx_state *st, float out[5]) {
  out[0] = st->alpha;
  out[1] = st->beta;
  out[2] = st->gamma;
  out[3] = st->delta;
  out[4] = st->gamma_w;
}

int iamf_hip_batch_reset(iamf_hip_batch *b) {
  if (!b) return IAMF_HIP_ERR_BAD_ARG;
  if (!on_batch_device(b)) return IAMF_HIP_ERR_INVALID_STATE;
  HIPCHK(hipDeviceSynchronize());
  return reset_state(b);
}

}  // extern "C"
