// render_common.hpp — constants, parameter block and small device helpers shared by the render
// kernels (included only by iamf_render.hip, inside its anonymous namespace).
#pragma once

constexpr int kChunk = 256;      // samples per workgroup step = threads per workgroup
constexpr int kDelay = 240;      // limiter look-ahead (reference common/audio_defines.h:41)
constexpr int kRing = 512;       // LDS ring length (power of two >= kChunk + kDelay + 15)
constexpr int kSave = 256;       // samples of ring persisted per stream between calls
constexpr int kHead = 256;       // coefficient-table head kept in LDS
constexpr int kMaxOut = 24;      // reference MAX_OUTPUT_CHANNELS
constexpr int kMaxIn = 24;

// Opting a kernel into more than 64 KiB of dynamic LDS is a PER-DEVICE attribute: one flag per device
// ordinal, set only when every hipFuncSetAttribute of the group succeeded (a failed opt-in is retried
// and surfaces as a launch error instead of being remembered as done).  Host threads may drive batches on
// different devices at once: the begin..end bracket keeps its device and result per THREAD and only the
// done flags are shared (atomic; opting in twice is harmless).  Ordinals >= kMaxDevices opt in on every launch.
constexpr int kMaxDevices = 64;
struct OptIn {
  std::atomic<bool> done[kMaxDevices] = {};
  struct Bracket {
    int dev;
    bool ok;
  };
  static Bracket &cur() {
    static thread_local Bracket b{0, true};
    return b;
  }
  bool begin() {  // true if this device still has to opt in
    Bracket &b = cur();
    if (hipGetDevice(&b.dev) != hipSuccess || b.dev < 0) b.dev = kMaxDevices;
    b.ok = true;
    return b.dev >= kMaxDevices || !done[b.dev].load(std::memory_order_acquire);
  }
  void set(const void *fn, int bytes) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) cur().ok = false;
  }
  void end() {
    const Bracket &b = cur();
    if (b.dev < kMaxDevices && b.ok) done[b.dev].store(true, std::memory_order_release);
  }
};

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
  int32_t n_streams;        // streams of the batch (the per-stream arrays' extent)
  int32_t stream0, n_launch;  // the streams this launch renders: workgroup i takes stream stream0 + i
  int32_t n_feeds;
  int32_t out_ch;
  int32_t out_format;
  int32_t limiter_on;
  int32_t loudness_on;
  int32_t use_mfma;         // wide kernel: projection on v_mfma_f32_32x32x2_f32 instead of VALU
  int32_t n_atk, n_end;     // limiter table split points
  float thr;
  const int32_t *src_feed;  // device [out_ch]: output slot -> feed index, -1 = silent slot, -2 = LFE slot
                            // fed by the HOA LFE generator (render_lfe.hpp; generic and wide4 kernels)
  // wide4 VALU projection: bit m of nz_mask[g] = some output slot 4g..4g+3 has a non-zero weight for
  // input m; sparse = less than half of those bits are set (then all-zero weight batches are skipped)
  uint32_t nz_mask[6];
  int32_t sparse;
  // ---- optional extras (generic kernel only) ----
  const float *in2;         // second element (planar f32) or nullptr
  int64_t in2_stream_stride, in2_frame_stride;
  const float *matrix2;     // device, feed-major [n_feeds2][m2]
  const int32_t *src_feed2; // device [out_ch]
  const float *gains2;      // device [n_streams] element-2 constant gain
  int32_t m2;
  int32_t dmx_on;           // element 0 is rendered by the parametric down-mixer
  const float *elem_ramp, *elem2_ramp, *out_ramp;  // per-sample gains of this call or nullptr
  int64_t ramp_stream_stride;
  const iamf_hip_dmx_frame *dmx_frames;  // device [n_streams][frames of this call]
  int32_t dmx_n_in, dmx_n_out;
  int32_t dmx_in_layout, dmx_out_layout;  // IAChannelLayoutType ids (render_downmix.hpp)
  const int32_t *dmx_tab;   // device [24]: IAChannel ids of the inputs, then (from [12]) of the outputs
  // ---- ambisonics projection de-mapping in front of element 0 (generic kernel) ----
  const float *pre_matrix;  // device [pre_l][M] or nullptr
  int32_t pre_l;            // decoded channels per frame when pre_matrix is set
  // ---- demixer of scalable channel audio in front of element 0 (generic kernel) ----
  int32_t demix_on;
  int32_t demix_steps;      // bit 0 S1to2, 1 S2to3, 2 S3to5, 3 S5to7, 4 TF2toT2, 5 T2toT4
  int32_t demix_skip;       // samples at the start of every frame that use the previous mode
  int32_t demix_i0;         // frame position of the call's first sample (trimmed single-frame calls)
  const int32_t *demix_tab; // device: [0..12) chs_in, [12..24) chs_out, [24] n_gain, [25..37) gain_ch,
                            //         [40..64) decoded position of an IAChannel (0 if it is not decoded)
  const float *demix_ftab;  // device: [0..12) gains, start_window[frame_size], stop_window[frame_size],
                            //         then [0..12) the gain of every decoded channel (1 where none)
  const iamf_hip_demix_frame *demix_frames;  // device [n_streams][frames of this call]
  int32_t demix_layout;     // IAChannelLayoutType of the target layout (render_wide4_kernel<.., DMX>)
  int32_t demix_gmask;      // bit m: decoded channel m takes the output gain demix_ftab[12 + 2*frame_size + m]
  int32_t demix_w4;         // 1 if the in-register demixer of render_wide4.hpp covers this configuration
  // ---- render_wide4.hpp: where lanes that have nothing to emit send their 16-byte store, so that every
  //      chunk issues the same vector-memory instructions and the compiler can COUNT them (s_waitcnt vmcnt(N)
  //      instead of vmcnt(0) at the top of the chunk loop); device [n_streams][256 lanes][16 B] ----
  uint8_t *dump;
  // ---- HOA LFE generator (render_lfe.hpp): raw low-pass output of this call or nullptr ----
  const float *lfe;         // device, transposed by blocks of 64 streams: element lfe_index(s, k, lfe_t4) (render_lfe.hpp)
  int32_t lfe_t4;           // quads per stream in that buffer
  int32_t og_ch;            // output channels the OUTPUT gain multiplies (iamf_hip_batch_config::out_gain_channels; = out_ch: all)
  int32_t lfe_k0;           // the generator's output of the call's first sample sits at index lfe_k0 (trimmed frames: the
                            // filter also ran over the lfe_k0 samples cut off in front, iamf_hip_render_args)
  double lfe_div;           // sqrt(n) of h2m_rdr.c:1162; 0 = the `* 0.5` form (n <= 2)
  int32_t lfe_mask;         // bit c: output slot c is an LFE slot (src_feed[c] == -2), for render_wide4.hpp
  // ---- HRTF FIR renderer (render_fast_kernel<M, 2, true>): matrix = h[2][M][fir_taps] ----
  int32_t fir_taps;
  const float *fir_hist;    // device [n_streams][M][256] input history before this call
  float *fir_hist_next;     // device, same shape: history after this call
  const void *fir_h16;      // device: split-f16 filter tables [M][ear][hi/lo][shift 8][304] (render_fir16.hpp) or nullptr
  float fir_inv_scale;      // 1 / (filter scale * input scale) of those tables
  const float *fir_pq;      // device: spectra tables of the FFT stage [pairs][16][64] x 4 floats (render_fir_fft.hpp) or nullptr
  const float *fir_tw;      // device: its twiddles [16][64] + [16][4] complex
  const float *fir_zero;    // device: zero floats, M * frame size of them (what that stage loads for runs past the end of a
                            // call, at the input's channel stride)
  // the same history as fir_hist at the INPUT's channel stride, for fir_fft_kernel's two-base fetch (frame size a multiple
  // of 256 and >= 1024, M even): stream s, channel c, sample j of the last 256 at
  //   ((s / G) * M + c) * frame_size + (s % G) * 256 + j,   G = frame_size / 256   (G streams share the rows of a slab)
  const float *fir_pre;
  float *fir_pre_next;
  float *fir_y;             // device scratch [n_streams][2][total]: the FFT stage's output when it runs as a kernel of its own
  const float *fir_id_matrix;   // device: the 2 x 2 identity (feed-major) and its slot map, for the limiter / pack kernel behind it
  const int32_t *fir_id_feed;
  // ---- element 0 handed over as LPCM packets (render_fast_kernel<.., LP>, iamf_hip_batch_render_lpcm): 16-bit
  //      little-endian samples, one contiguous run per channel and frame.  Sample i of channel m, frame f, stream s:
  //      lpcm + s * lpcm_stream_stride + f * lpcm_frame_stride + lpcm_off[m] + 2 * i (bytes; every term a multiple of 8
  //      for i a multiple of 4).  `in` is not read then ----
  const uint8_t *lpcm;
  int64_t lpcm_stream_stride, lpcm_frame_stride;
  int32_t lpcm_off[16];
};

// IAChannel ids (reference IAMF_types.h:61-90; L5/R5 alias L7/R7)
enum {
  kChNone = 0, kChL7, kChR7, kChC, kChLFE, kChSL7, kChSR7, kChBL7, kChBR7, kChHFL, kChHFR, kChHBL,
  kChHBR, kChMono, kChL2, kChR2, kChTL, kChTR, kChL3, kChR3, kChSL5, kChSR5, kChHL, kChHR, kChCount
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

// Quotients that share a divisor (the demixer of scalable channel audio divides 8 numerators each by delta, beta and
// gamma of the frame: demixer.c:205-214,255-267,357-366).  With r = RN(1 / d) — ONE IEEE division per divisor —
//     q = n * r;  e = fma(-d, q, n);  q' = fma(e, r, q)
// is the correctly rounded n / d, bit for bit, for every f32 numerator with 2^-100 <= |n| < 2^126 and every divisor the
// demixing modes can produce (1, 0.707f, 0.866f: IAMF_utils.c:236-240); outside that range the residual e underflows or
// q overflows, and -0 comes out as +0.  Proven by an exhaustive sweep of all 2^32 numerators per divisor ON THE DEVICE
// (iamf_hip_selftest_shared_divisor, tests/test_gpu_wide4.py) and on the host.  w4_quot returns q' and clears *ok for a
// numerator outside the range; the caller then redoes its divisions the slow way (a wave-uniform, rare branch: digital
// silence takes it).  Three instructions per quotient instead of the ~10 of an IEEE division.
__device__ __forceinline__ float w4_quot(float n, float d, float r, bool &ok) {
  const float an = __builtin_fabsf(n);
  ok = ok && an >= 0x1p-100f && an < 0x1p126f;
  const float q = n * r;
  const float e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}

// which HRTF stage a FIR call runs (host): 3 = overlap-save FFT (default), 2 = split-f16 MFMA, 1 = f32 MFMA
// 4 = the FFT stage as a kernel of its own + the two-channel matrix kernel behind it (default); IAMF_HIP_FIR_FUSED=1 keeps
// the FFT stage inside render_fast_kernel<M, 2, 3> (one pass over HBM, but the hops of a stream run one pass after the other
// and the limiter stages at two workgroups per CU: 29 instead of the split's rate, NOTEBOOK.md 4.2c)
inline int fir_stage_choice(const RenderParams &p) {
  if (getenv("IAMF_HIP_FIR_F32")) return 1;
  if (getenv("IAMF_HIP_FIR_F16") && p.fir_h16) return 2;
  if (p.fir_pq && p.fir_tw && p.fir_zero && (p.frame_size & 63) == 0)   // its input runs of 64 must not straddle frames
    return (p.fir_y && p.fir_id_matrix && !getenv("IAMF_HIP_FIR_FUSED")) ? 4 : 3;
  return p.fir_h16 ? 2 : 1;
}
