// render_fast.hpp — the hot kernel: mono/stereo/binaural output layouts with the limiter on,
// aligned calls (see fast_path_ok() on the host).  One workgroup (4 waves) per stream, 1024-sample
// chunks, FOUR consecutive samples per lane:
//   * planar f32 input read as 16-byte loads (1 KiB per wave-instruction), the next chunk's loads
//     issued right after the projection so they fly under the limiter work;
//   * rendered samples kept only in an LDS ring (the limiter's delay line);
//   * 240-sample sliding maximum from per-16 suffix / prefix / block maxima (DPP quad ops + LDS);
//   * limiter gain: every lane first evaluates its gains under the hypothesis "no new trigger in
//     this chunk" (the gain is then a pure function of the sample index); only if some lane sees
//     peak*gain > threshold does wave 0 re-run the recurrence from that 64-sample block on:
//     ballot-driven speculation per block, and for runs of consecutive triggers (the limiter
//     holding a peak down) a wave-wide DPP shift chain that resolves one sample per 3-4 VALU ops;
//   * interleaved PCM written as 16-byte stores; input loads are non-temporal (read once).
// Same f32 operation order as render_generic.hpp and the reference: bit-exact.
#pragma once

constexpr int kFChunk = 1024;
constexpr int kFRing = 1280;   // >= chunk + look-ahead, multiple of 16 (not a power of two)
constexpr int kFWin = 1088;    // staged limiter-table window / head length (> chunk + 1), multiple of 64
constexpr int kBig = 0x7fffffff;

__device__ __forceinline__ float dpp_quad_bcast0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_bcast1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_bcast2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xAA, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_bcast3(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF, 0xf, 0xf, true));
}
// lane l receives lane l-1's value; lane 0 keeps its own (wave_shr:1, bound_ctrl off)
__device__ __forceinline__ float dpp_wave_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// 16-byte streaming load (read once: non-temporal, keeps the caches for what is re-read)
__device__ __forceinline__ float4 ld_stream4(const float *p) {
  using v4 = __attribute__((ext_vector_type(4))) float;
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Limiter recurrence for 64-sample blocks [b0, nblk) of the current chunk, run by ONE wave
// (lane = sample).  arr_p / arr_e hold the window maxima and thr/peak of the chunk; gains go to
// arr_g.  (n, gs, ge) is the state before the first sample of block b0 and is updated to the
// state after the last sample.  Restates audio_effect_peak_limiter.c:237-265 sample by sample:
// every accepted gain is produced by exactly the reference's f32 operations.
// With stop_clean the walk ends behind the first block after b0 in which nothing triggered (the caller
// then re-evaluates the rest of the chunk in parallel from the state reached); the return value is
// the first block NOT processed.
template <typename Lookup>
__device__ __forceinline__ int limiter_wave(const float *arr_p, float *arr_g,
                                            Lookup ctab, int b0, int nblk, int &n, float &gs,
                                            float &ge, float &g_last, float thr, int n_atk, int n_end,
                                            bool stop_clean = false) {
  const int lane = threadIdx.x & 63;
  // A dependent chain: whenever this wave has an instruction ready it should go first, ahead of the
  // co-resident workgroup's wave on the same SIMD (which has a whole chunk of independent work).
  __builtin_amdgcn_s_setprio(3);
  const float a1 = ctab(1);  // attack-curve value one step after a trigger
  float gacc = 1.0f;
  int b = b0;
  for (; b < nblk; ++b) {
    const float pk = arr_p[b * 64 + lane];
    const float e = thr / pk;  // targetEndGain if this sample triggers (IEEE division, as the reference)
    int l0 = 0;
    bool clean = true;
    while (true) {
      // speculate: no trigger in lanes l0..63 given the state before lane l0
      int n_pre = n + (lane - l0);
      n_pre = n_pre < n_end ? n_pre : n_end;
      n_pre = n_pre < 0 ? 0 : n_pre;
      const int ci = n_pre + 1 < n_end ? n_pre + 1 : n_end;
      const float c = ctab(ci);
      const float g = gain_at(n_pre, gs, ge, c, n_atk, n_end);
      const bool tr = lane >= l0 && (pk * g > thr);
      const unsigned long long mask = __ballot(tr);
      if (mask == 0ull) {
        if (lane >= l0) gacc = g;
        n = n + (64 - l0) < n_end ? n + (64 - l0) : n_end;
        break;
      }
      clean = false;
      const int f = __builtin_ctzll(mask);  // first trigger: lanes l0..f are settled
      if (lane >= l0 && lane <= f) gacc = g;
      gs = readlane_f(g, f);
      ge = readlane_f(e, f);
      n = 0;
      l0 = f + 1;
      if (l0 >= 64) break;

      // A trigger is usually followed by a run of triggers.  Assume every later lane's
      // predecessor triggered: g[l] = g[l-1] - a1*(g[l-1] - e[l-1]).  Jacobi sweeps with a
      // one-lane shift settle one more lane per sweep; lanes <= f are fixed points
      // (G = E' = gs there, so G - a1*(G - E') == G exactly).
      float G = gs;
      float ep = dpp_wave_shr1(e);
      ep = lane <= f ? gs : ep;
      // One sweep = G <- shr(G) - a1 * (shr(G) - ep) with the lane shift folded into the two
      // subtractions (DPP on src0; lane 0 has no source lane and keeps its value, it is a fixed
      // point anyway).  s_nop 1 = the two wait states a DPP read needs after a VALU write of the
      // same register.  Sweeps beyond the 63 - f needed ones leave every lane unchanged.
      for (int it = f + 1; it < 64; it += 8) {
        float tmp;
#define IAMF_SWEEP                                                              \
  "s_nop 1\n\t"                                                                 \
  "v_sub_f32_dpp %[t], %[g], %[ep] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"   \
  "v_mul_f32_e32 %[t], %[a1], %[t]\n\t"                                         \
  "v_sub_f32_dpp %[g], %[g], %[t] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP IAMF_SWEEP
                     : [g] "+v"(G), [t] "=&v"(tmp)
                     : [ep] "v"(ep), [a1] "v"(a1));
#undef IAMF_SWEEP
        // (leaving the loop as soon as the run is seen to have ended among the settled lanes costs more
        //  than it saves: 58 vs 61 Gsamples/s on a programme of short isolated peaks)
      }
      const bool tr2 = pk * G > thr;
      const unsigned long long stop = __ballot(lane > f && !tr2);
      if (stop == 0ull) {  // the run reaches the end of the block
        if (lane > f) gacc = G;
        gs = readlane_f(G, 63);
        ge = readlane_f(e, 63);
        n = 0;
        l0 = 64;
        break;
      }
      const int m = __builtin_ctzll(stop);  // first lane of the run that does NOT trigger
      if (lane > f && lane <= m) gacc = G;
      gs = readlane_f(G, m - 1);
      ge = readlane_f(e, m - 1);
      n = 1;  // lane m stepped once from the trigger at m-1 and did not re-trigger
      l0 = m + 1;
      if (l0 >= 64) break;
    }
    arr_g[b * 64 + lane] = gacc;
    if (stop_clean && clean && b > b0) {
      ++b;
      break;
    }
  }
  g_last = readlane_f(gacc, 63);
  __builtin_amdgcn_s_setprio(0);
  return b;
}

// Which of the workgroup's 4 waves runs the serial limiter recurrence.  Workgroups that share a CU
// tend to reach that phase together; if both ran it on "wave 0" the two dependent chains would
// sit on the same SIMD and each would get half its issue slots.  Every wave publishes its SIMD
// and its wave slot on that SIMD (HW_ID bits 5:4 and 3:0); co-resident workgroups hold different
// slots, so keying the choice on (simd - slot) spreads their chain waves over different SIMDs.
// slots4 = 4 LDS words; call before a __syncthreads(), read with chain_wave_pick() after it.
__device__ __forceinline__ void chain_wave_publish(float *slots4) {
  const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_ID[5:4]
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID[3:0]
  if ((threadIdx.x & 63) == 0) slots4[threadIdx.x >> 6] = __int_as_float((int)((simd - slot) & 3u));
}
__device__ __forceinline__ int chain_wave_pick(const float *slots4) {
  int best = 0, key = __float_as_int(slots4[0]);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const int k = __float_as_int(slots4[w]);
    if (k < key) {
      key = k;
      best = w;
    }
  }
  return best;
}

// LDS floats the fast kernel needs for an OC-channel layout, M inputs and a limiter table of
// `tab` entries (host and device use the same carve-up)
__host__ __device__ constexpr int fast_lds_floats(int oc, int m, int fir = 0) {
  // The limiter's curve table (9651 entries at 48 kHz) stays in global memory.  The matrix variant
  // stages, per chunk, the window of it the chunk can reach without a trigger plus the head that
  // follows a trigger (2 * kFWin floats): 37 KiB of LDS for a stereo layout, four workgroups per CU.
  // The MFMA HRTF variants read the table from global memory; the FFT one (fir == 3) stages the window like the
  // matrix variant: a hot programme walks the limiter chain in most chunks, and every table look-up of the walk
  // was an L2 round trip.
  return oc * kFRing + 2 * kFRing + 2 * (kFRing / 16) + 2 * kFChunk + ((fir && fir != 3) ? 0 : 2 * kFWin) + ((oc * m + 15) & ~15) + 16 +
         (fir == 3 ? kFftLdsFloats : (fir == 2 ? kF16LdsFloats : (fir ? kFirLdsFloats : 16 /* second element's matrix rows */)));
}

__device__ __forceinline__ int ring_wrap(int i) {  // i in [-R, 2R)
  i = i < 0 ? i + kFRing : i;
  return i >= kFRing ? i - kFRing : i;
}

// The f32-MFMA HRTF variant (FIR == 1) runs 512 threads: all eight waves work in fir_stage (ear x quarter of
// the channels), waves 0..3 alone (`act`) run the stages around it.  The split-f16 variant (FIR == 2) runs 256
// threads at two waves per SIMD: its stage wants the registers (render_fir16.hpp).
// DOWN: the element is rendered by the parametric down-mixer (render_downmix.hpp) instead of a matrix.
// IN2: the mixing variant.  A second element of at most kFIn2 channels (mono, stereo, first-order
//      ambisonics: a dialogue or commentary track next to the bed) is rendered by its own matrix and
//      mixed in (iamf_mixer_mix, IAMF_decoder.c:2702-2733), and / or the element and output gains are
//      per-sample ramps (animated mix-gain parameters, iamf_frame_gain with a gains[] array,
//      IAMF_decoder.c:1383-1408) instead of constants.
constexpr int kFIn2 = 4;
// FIR:  0 = gain matrix; 1 = HRTF stage on the f32 MFMA (render_fir.hpp); 2 = HRTF stage on the f16
//       MFMA with split operands (render_fir16.hpp); 3 = HRTF stage by overlap-save FFT on the VALU
//       (render_fir_fft.hpp: one pass per THREE chunks, a wave per 768-sample hop)
// LP:   element 0 arrives as 16-bit little-endian LPCM packets (RenderParams::lpcm) instead of planar f32: the
//       reference's LPCM "decoder" (pcm/IAMF_pcm_decoder.c:64-83: sample / 32768.f) runs where the samples are loaded,
//       8 bytes per lane and channel instead of 16, and the f32 copy of the element never exists in HBM.
// EARLY: (plain matrix variants) channel m of the NEXT chunk is requested as soon as channel m of this one has been
//       consumed by the projection, instead of all channels behind the projection.  Longer in flight: what a launch
//       of few workgroups per CU waits for (512 streams: f32 +4 %, LPCM +8 %); a launch that fills every wave slot is
//       bound by its instruction stream and loses to the per-channel issue (LPCM, 4096 streams: -6 %) — the host picks.
template <int M, int OC, int FIR = 0, bool DOWN = false, bool IN2 = false, bool LP = false, bool EARLY = true>
__global__ __launch_bounds__(FIR == 1 ? 512 : 256, FIR >= 2 ? 2 : ((FIR || (M <= 16 && !IN2)) ? 4 : 2)) void render_fast_kernel(const RenderParams p) {
  static_assert(!(FIR && DOWN), "one renderer");
  static_assert(!(LP && (FIR || DOWN || IN2)) && (!LP || M <= 16), "the LPCM input feeds the plain matrix variant");
  static_assert(!(IN2 && (FIR || DOWN)), "the second element joins a matrix-rendered first one");
  extern __shared__ float lds[];
  constexpr int R = kFRing;
  constexpr int NB = R / 16;
  const int n_atk = p.n_atk, n_end = p.n_end;
  float *ring_y = lds;                  // [OC][R]   rendered samples (limiter delay line)
  float *ring_pm = ring_y + OC * R;     // [R]       max |y| over channels
  float *ring_suf = ring_pm + R;        // [R]       suffix maxima of pm inside aligned 16-blocks
  float *ring_bm = ring_suf + R;        // [2][R/16] maxima of aligned 16-blocks, stored twice: entry b also at b + NB, so
                                        //           that the 14 blocks before any block are 14 CONSECUTIVE words
  float *arr_p = ring_bm + 2 * NB;      // [1024]    window maxima of the chunk
  float *arr_g = arr_p + kFChunk;       // [1024]    gains from the limiter wave
  float *win = arr_g + kFChunk;         // [kFWin]   ctab[min(n_st + 1 + i, n_end)] (not in the HRTF variant)
  float *head = win + kFWin;            // [kFWin]   ctab[i]                     (not in the HRTF variant)
  constexpr bool kTab = FIR == 0 || FIR == 3;   // the limiter-table window and head are staged in LDS
  // the matrix variant reads its weights four input channels at a time (one 16-byte LDS read per output slot and group)
  constexpr bool kWG = !FIR && !DOWN && (M % 4) == 0;
  constexpr bool kEarly = EARLY && !FIR && !DOWN && !IN2;
  float *mat = win + (kTab ? 2 * kFWin : 0);  // [OC*M]  feed-major matrix rows of the OC slots
  float *misc = mat + ((OC * M + 15) & ~15);  // [16]
  float *mat2 = misc + 16;              // [OC][kFIn2]  second element's matrix rows (IN2 only; aliases fir)
  float *fir = misc + 16;               // [kFirLdsFloats]  HRTF staging (FIR variant only)

  const int s = blockIdx.x + p.stream0;   // a launch covers streams [stream0, stream0 + n_launch) of the batch
  const bool act = FIR != 1 || threadIdx.x < 256;
  const int t = FIR == 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x;  // helper waves keep indices in range
  const int wave = t >> 6;
  const int lane = t & 63;
  const int q = t & 3;
  const int fs = p.frame_size;
  const float thr = p.thr;
  // ring position of the chunk's first sample: a multiple of 16 (the 16-blocks of maxima are aligned in the ring).  Where the
  // ring starts is this call's choice — everything below is relative to `base`, the persisted state is "the last 256
  // samples" — so a stream whose position is NOT a multiple of 16 (a first frame trimmed by 237 samples) runs here too,
  // from 240 samples on (before that the withheld look-ahead ends inside a lane's four samples: the general kernel)
  int base = (int)((p.pos0 & ~(int64_t)15) % R);

  // ---- stream state and constants -> LDS (persisted format is the generic kernel's) ----
  if (act) {
    const float *sy = p.ring_y + (int64_t)s * OC * kSave;
    const float *spm = p.ring_pm + (int64_t)s * kSave;
    const int rp = ring_wrap(base - kSave + t);  // saved entry t is sample pos0 - 256 + t
#pragma unroll
    for (int c = 0; c < OC; ++c) ring_y[c * R + rp] = sy[c * kSave + t];
    const float pm = spm[t];
    ring_pm[rp] = pm;
    float sfx = pm;  // pos0 % 16 == 0: 16-lane groups are aligned 16-blocks
    sfx = fmaxf(sfx, __shfl_down(sfx, 1, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 2, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 4, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 8, 16));
    // __shfl_down hands back the caller's own value past the end of the 16-lane segment, which
    // leaves the running maximum unchanged
    ring_suf[rp] = sfx;
    if ((t & 15) == 0) ring_bm[rp >> 4] = ring_bm[(rp >> 4) + NB] = sfx;
    if constexpr (kTab)
      for (int i = t; i < kFWin; i += 256) head[i] = p.ctab[i < n_end ? i : n_end];
    if (!FIR && !DOWN && t < OC * M) {
      const int c = t / M, m = t - c * M;
      const int f = p.src_feed[c];
      // LP: the LPCM decoder's "sample / 32768.f" (pcm/IAMF_pcm_decoder.c:64-83) is folded into the weight — (w * 2^-15) *
      // (float)s has the bits of w * (s * 2^-15): a power of two commutes with rounding in the normal range, which the host
      // has checked for this matrix (iamf_hip_batch::lp_scale_ok).  Two packed multiplies per channel and chunk less.
      mat[t] = (f >= 0 ? p.matrix[f * M + m] : 0.f) * (LP ? 1.0f / 32768.0f : 1.0f);
    }
    if (IN2 && p.in2 && t < OC * kFIn2) {
      const int c = t / kFIn2, m = t - c * kFIn2;
      const int f = p.src_feed2[c];
      mat2[t] = (f >= 0 && m < p.m2) ? p.matrix2[f * p.m2 + m] : 0.f;
    }
    chain_wave_publish(misc + 12);
  }
  LimState ls = p.lim[s];
  float g_cur = ls.g, gs = ls.gs, ge = ls.ge;
  int n_st = ls.n;
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);
  const float m_eg = eg_on ? eg : 1.f, m_og = og_on ? og : 1.f, m_lg = lg_on ? lg : 1.f;
  const bool any_gain = eg_on || og_on || lg_on;
  bool live[OC];
#pragma unroll
  for (int c = 0; c < OC; ++c) live[c] = FIR || DOWN || p.src_feed[c] >= 0;
  const float *fir_hist = FIR ? p.fir_hist + (int64_t)s * M * kFirHist : nullptr;
  FftTwiddles ftw;
  if constexpr (FIR == 3) fft_load_twiddles(p.fir_tw, t, fir + kFftLdsFloats - kFftTwFloats, ftw);
  // second element: constant gain (skipped by the reference when it is 1 or not positive)
  const float eg2 = (IN2 && p.in2) ? p.gains2[s] : 1.f;
  const float m_eg2 = (eg2 != 1.f && eg2 > 0.f) ? eg2 : 1.f;
  bool live2[OC];
#pragma unroll
  for (int c = 0; c < OC; ++c) live2[c] = IN2 && p.in2 && p.src_feed2[c] >= 0;
  const float *in2_s = IN2 ? p.in2 + (int64_t)s * p.in2_stream_stride : nullptr;
  float4 x2[IN2 ? kFIn2 : 1];
#pragma unroll
  for (int m = 0; m < (IN2 ? kFIn2 : 1); ++m) x2[m] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 rmp[IN2 ? 3 : 1];  // per-sample gains of element 0, element 1 and the output, where given
  auto load_x2 = [&](int f, int i) {  // the lane's 4 samples of the second element's channels (frame f, position i)
    if constexpr (IN2) {
      if (p.in2) {
        const float *src = in2_s + (int64_t)f * p.in2_frame_stride + i;
#pragma unroll
        for (int m = 0; m < kFIn2; ++m)
          x2[m] = m < p.m2 ? ld_stream4(src + (int64_t)m * fs) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int64_t ro = (int64_t)s * p.ramp_stream_stride + (int64_t)f * fs + i;  // sample index inside the call
      if (p.elem_ramp) rmp[0] = ld_stream4(p.elem_ramp + ro);
      if (p.elem2_ramp) rmp[1] = ld_stream4(p.elem2_ramp + ro);
      if (p.out_ramp) rmp[2] = ld_stream4(p.out_ramp + ro);
    }
  };
  if constexpr (IN2) {
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    rmp[0] = p.elem_ramp ? one : make_float4(m_eg, m_eg, m_eg, m_eg);
    rmp[1] = p.elem2_ramp ? one : make_float4(m_eg2, m_eg2, m_eg2, m_eg2);
    rmp[2] = p.out_ramp ? one : make_float4(m_og, m_og, m_og, m_og);
  }

  const int64_t out_base = p.pos0 > kDelay ? p.pos0 - kDelay : 0;
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;

  // input of the first chunk
  float4 x[FIR ? 1 : M];
  // LP: the prefetched packets' samples as they lie in memory (four 16-bit samples per channel), converted at the top of
  // the chunk that uses them: (float)sample * (1 / 32768), the expression of iamf_hip_lpcm_unpack and of the reference
  using lp_u2 = __attribute__((ext_vector_type(2))) unsigned;
  lp_u2 xr[LP ? M : 1];
  const uint8_t *lp_s = LP ? p.lpcm + (int64_t)s * p.lpcm_stream_stride : nullptr;
  // The element's input goes through BUFFER loads: one resource per stream (base = the stream's region), the lane's byte
  // offset in one register for all channels, the channel's offset as the instruction's scalar offset — no address
  // arithmetic per load (a 64-bit pointer per channel cost an add each and, for the packets' sixteen run offsets, sixteen
  // scalar register pairs that did not fit).  Offsets are 32-bit: the host sends longer calls elsewhere (fast_path_ok).
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(LP ? static_cast<const void *>(lp_s) : static_cast<const void *>(in_s)), 0, 0x7fffffff, 0x00020000);
  // Frames that are whole chunks (frame size a multiple of 1024 — the usual 1024): a chunk lies in ONE frame, its frame
  // number and position are workgroup-uniform counters; otherwise each lane divides.
  const bool fr_uni = (fs & (kFChunk - 1)) == 0;
  int fu = 0, iu = 0;   // frame / position of the first sample of the chunk the next prefetch fetches (fr_uni)
  auto frame_pos = [&](int kq, int &f, int &i) {
    if (fr_uni) {
      f = fu;
      i = iu + 4 * t;
    } else {
      f = kq / fs;
      i = kq - f * fs;
    }
  };
  // (a resource of zero records: every load through it is out of range — answered with zeros, no memory access.  The
  //  projection's requests for "the next chunk" go through it when there is none, instead of through a branch per channel)
  const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(LP ? static_cast<const void *>(lp_s) : static_cast<const void *>(in_s)), 0, 0, 0x00020000);
  auto load_lp_one = [&](int m, int f, int i, __amdgpu_buffer_rsrc_t rs) {
    if constexpr (LP) {
      const int vo = f * (int)p.lpcm_frame_stride + 2 * i;
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, p.lpcm_off[m], 2 /* nt */);
      xr[m] = lp_u2{v[0], v[1]};
    }
  };
  auto load_f32_one = [&](int m, int f, int i, __amdgpu_buffer_rsrc_t rs) {
    if constexpr (!LP && !FIR) {
      const int vo = 4 * (f * (int)p.in_frame_stride + i);
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 4 * m * fs, 2 /* nt */);
      x[m] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  };
  auto load_lp = [&](int f, int i) {
#pragma unroll
    for (int m = 0; m < (LP ? M : 0); ++m) load_lp_one(m, f, i, rs_in);
  };
  auto load_f32 = [&](int f, int i) {
#pragma unroll
    for (int m = 0; m < ((!LP && !FIR) ? M : 0); ++m) load_f32_one(m, f, i, rs_in);
  };
  float drec[DOWN ? 11 : 1];  // DOWN: the frame record of the lane's samples, fetched with them
  const int dmx_nfr = DOWN ? (p.total + fs - 1) / fs : 0;
  auto load_drec = [&](int f) {
    if constexpr (DOWN) {
      const float *d = reinterpret_cast<const float *>(p.dmx_frames + (int64_t)s * dmx_nfr + (f < dmx_nfr ? f : dmx_nfr - 1));
#pragma unroll
      for (int i = 0; i < 11; ++i) drec[i] = d[i];
    }
  };
  if constexpr (!FIR) {
    const int k = 4 * t;
    if (k < p.total) {
      int f, i;
      frame_pos(k, f, i);
      load_lp(f, i);
      load_f32(f, i);
      load_x2(f, i);
    } else {
#pragma unroll
      for (int m = 0; m < (LP ? M : 0); ++m) xr[m] = lp_u2{0u, 0u};
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < (IN2 ? kFIn2 : 0); ++m) x2[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_drec(k / fs);
  }
  __syncthreads();
  const int cw = chain_wave_pick(misc + 12);

  // Table window the next chunk can reach without a trigger: win[i] = ctab[min(n_st + 1 + i, n_end)] — the value sample i
  // of the chunk needs if nothing triggers before it (its step count n_st + i, the curve read one step on), so that a
  // lane's four consecutive samples read four consecutive, 16-byte aligned words.
  // Fetched BEFORE the chunk's PCM stores are issued and written to LDS BEFORE the next input
  // prefetch is issued: vector-memory operations retire in order, so a wait for these values at
  // any later point would drain the stores / the prefetch as well.
  float wv[kTab ? 5 : 1];
  auto fetch_window = [&](int n0) {
    if constexpr (kTab) {
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int i = n0 + 1 + t + 256 * r;
        wv[r] = 1.0f;
        if (n0 < n_end && t + 256 * r < kFWin) wv[r] = p.ctab[i < n_end ? i : n_end];
      }
    }
  };
  if constexpr (!FIR) fetch_window(n_st);

  for (int c0 = 0; c0 < p.total; c0 += kFChunk) {
    const int cnt = p.total - c0 < kFChunk ? p.total - c0 : kFChunk;  // multiple of 64
    const int k = c0 + 4 * t;
    const bool valid = act && 4 * t < cnt;
    const int64_t gk = p.pos0 + k;
    const int rp = ring_wrap(base + 4 * t);

    // kEarly: the table window goes to LDS first (its loads are older than anything still in flight), then the place of
    // the next chunk is worked out: the projection below requests it channel by channel
    int pf_f = 0, pf_i = 0;
    __amdgpu_buffer_rsrc_t rs_pf = rs_none;
    if constexpr (kEarly) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
        if (t + 256 * r < kFWin) win[t + 256 * r] = wv[r];
      const int kn = k + kFChunk;
      iu += kFChunk;
      if (iu >= fs) {
        iu -= fs;
        ++fu;
      }
      if (c0 + kFChunk < p.total) rs_pf = rs_in;   // workgroup-uniform: there is a next chunk
      frame_pos(kn, pf_f, pf_i);
      if (kn >= p.total) pf_f = pf_i = 0;   // a lane past the end of the call requests the stream's first samples (unused)
    }

    // ---- element renderer + gains (reference operation order) ----
    float4 y[OC];
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
    // The matrix variant's projection, input channel by input channel: every output slot still sums its products in
    // ascending channel order (the reference's order), and an LPCM channel is converted where it is consumed — the 64
    // registers of a converted element never exist beside the 32 of the packets (lanes past the end of the call
    // convert what they hold: never used).
    // (written on pairs of samples: two f32 products or sums per packed instruction, each rounded on its own)
    using f2 = __attribute__((ext_vector_type(2))) float;
    f2 prj[(!FIR && !DOWN) ? 2 * OC : 1];
    if constexpr (!FIR && !DOWN) {
#pragma unroll
      for (int c = 0; c < 2 * OC; ++c) prj[c] = f2{0.f, 0.f};
      float wg[OC][4];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        f2 xa, xb;
        if constexpr (kWG) {
          if ((m & 3) == 0) {
#pragma unroll
            for (int c = 0; c < OC; ++c) {
              const float4 w4 = *reinterpret_cast<const float4 *>(&mat[c * M + m]);
              wg[c][0] = w4.x, wg[c][1] = w4.y, wg[c][2] = w4.z, wg[c][3] = w4.w;
            }
          }
        }
        if constexpr (LP) {
          // (the packets of channel m are "produced" here, after channel m - 1 has been added up: left to itself the
          //  scheduler converts all sixteen channels first and spills what does not fit — 127 registers' worth)
          if constexpr (OC == 2)
            asm volatile("" : "+v"(xr[m].x), "+v"(xr[m].y), "+v"(prj[0]), "+v"(prj[1]), "+v"(prj[2]), "+v"(prj[3]));
          else
            asm volatile("" : "+v"(xr[m].x), "+v"(xr[m].y), "+v"(prj[0]), "+v"(prj[1]));
          const unsigned a = xr[m].x, b = xr[m].y;   // (the scale 2^-15 sits in the weights: see where `mat` is filled)
          xa = f2{(float)(int)(short)(a & 0xffffu), (float)((int)a >> 16)};
          xb = f2{(float)(int)(short)(b & 0xffffu), (float)((int)b >> 16)};
        } else {
          // (the same ordering for the f32 element: channel m is taken up when channel m - 1 has been added up, so that the
          //  weights of one group of four channels are all that is live beside the samples)
          if constexpr (OC == 2)
            asm volatile("" : "+v"(x[m].x), "+v"(x[m].y), "+v"(x[m].z), "+v"(x[m].w), "+v"(prj[0]), "+v"(prj[1]), "+v"(prj[2]), "+v"(prj[3]));
          else
            asm volatile("" : "+v"(x[m].x), "+v"(x[m].y), "+v"(x[m].z), "+v"(x[m].w), "+v"(prj[0]), "+v"(prj[1]));
          xa = f2{x[m].x, x[m].y};
          xb = f2{x[m].z, x[m].w};
        }
#pragma unroll
        for (int c = 0; c < OC; ++c) {
          const float w = kWG ? wg[c][m & 3] : mat[c * M + m];
          const f2 w2 = {w, w};
          prj[2 * c] = prj[2 * c] + w2 * xa;
          prj[2 * c + 1] = prj[2 * c + 1] + w2 * xb;
        }
        // channel m's registers are free: its samples of the NEXT chunk are requested now (kEarly), in flight under the
        // rest of the projection and everything behind it
        if constexpr (kEarly) {
          if constexpr (LP) load_lp_one(m, pf_f, pf_i, rs_pf);
          else load_f32_one(m, pf_f, pf_i, rs_pf);
        }
      }
    }
    if constexpr (FIR == 2) {  // both ears of this chunk AND the next three -> LDS (one pass per four chunks)
      if (((c0 >> 10) & 3) == 0) fir_stage16<M>(p, in_s, fir_hist, c0, fir, fir);
    }
    else if constexpr (FIR == 1) fir_stage<M>(p, in_s, fir_hist, c0, fir, fir);  // ... as four partial sums per ear
    else if constexpr (FIR == 3) {  // both ears of this chunk and the next two -> the waves' scratch areas
      if ((c0 >> 10) % 3 == 0) {
        fir_stage_fft<M>(p, in_s, fir_hist, c0, reinterpret_cast<fft_c32 *>(fir), ftw);
        __syncthreads();
      }
      fetch_window(n_st);   // this chunk's table window: in flight while the stage's output is read back (not held
                            // across the stage, whose registers are all spoken for)
    }
    float4 yd[DOWN ? OC : 1];
    if constexpr (DOWN) {
      float4 cf[5];
      downmix_factors(drec, k - (k / fs) * fs, cf);
      downmix4<M, OC>(x, yd, cf, p.dmx_in_layout, p.dmx_out_layout);
    }
    if (act)
#pragma unroll
    for (int c = 0; c < OC; ++c) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (FIR == 3) {
        // ear c of this chunk, left by the FFT stage's last pass (render_fir_fft.hpp)
        v = *reinterpret_cast<const float4 *>(fft_y_ptr(fir, c, 1024 * ((c0 >> 10) % 3) + 4 * t));
      } else if constexpr (FIR == 2) {
        // ear c of this chunk, left by the split-f16 stage's last pass
        const float *p0 = fir + (c * 4 + ((c0 >> 10) & 3)) * kF16Part;
        const int u = 4 * t + ((4 * t) >> 5);  // padded index; 4 consecutive samples stay in one 32-block
        v = make_float4(p0[u + 0], p0[u + 1], p0[u + 2], p0[u + 3]);
      } else if constexpr (FIR == 1) {
        // ear c: the partial sums of the four channel quarters (waves c, c+2, c+4, c+6), in that order
        const float *p0 = fir + c * (kFChunk + 32);
        const int u = 4 * t + ((4 * t) >> 5);  // padded index; 4 consecutive samples stay in one 32-block
        constexpr int PS = 2 * (kFChunk + 32);
        v.x = ((p0[u + 0] + p0[PS + u + 0]) + p0[2 * PS + u + 0]) + p0[3 * PS + u + 0];
        v.y = ((p0[u + 1] + p0[PS + u + 1]) + p0[2 * PS + u + 1]) + p0[3 * PS + u + 1];
        v.z = ((p0[u + 2] + p0[PS + u + 2]) + p0[2 * PS + u + 2]) + p0[3 * PS + u + 2];
        v.w = ((p0[u + 3] + p0[PS + u + 3]) + p0[2 * PS + u + 3]) + p0[3 * PS + u + 3];
      } else if constexpr (DOWN) {
        v = yd[c];
      } else if (live[c]) {
        if constexpr (!FIR && !DOWN) v = make_float4(prj[2 * c].x, prj[2 * c].y, prj[2 * c + 1].x, prj[2 * c + 1].y);
      }
      if constexpr (IN2) {
        // element gain, mixer (0 + y, then + y2), output and loudness gains in the reference's order;
        // a gain the reference skips is a multiplication by exactly 1
        float4 v2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live2[c]) {
#pragma unroll
          for (int m = 0; m < kFIn2; ++m) {
            if (m < p.m2) {
              const float w = mat2[c * kFIn2 + m];
              v2.x = v2.x + w * x2[m].x;
              v2.y = v2.y + w * x2[m].y;
              v2.z = v2.z + w * x2[m].z;
              v2.w = v2.w + w * x2[m].w;
            }
          }
        }
        // rmp[]: the call's per-sample gains where ramps are given, else the constant gains
        if (p.in2) {
          v.x = (((0.f + v.x * rmp[0].x) + v2.x * rmp[1].x) * rmp[2].x) * m_lg;
          v.y = (((0.f + v.y * rmp[0].y) + v2.y * rmp[1].y) * rmp[2].y) * m_lg;
          v.z = (((0.f + v.z * rmp[0].z) + v2.z * rmp[1].z) * rmp[2].z) * m_lg;
          v.w = (((0.f + v.w * rmp[0].w) + v2.w * rmp[1].w) * rmp[2].w) * m_lg;
        } else {
          v.x = ((0.f + v.x * rmp[0].x) * rmp[2].x) * m_lg;
          v.y = ((0.f + v.y * rmp[0].y) * rmp[2].y) * m_lg;
          v.z = ((0.f + v.z * rmp[0].z) * rmp[2].z) * m_lg;
          v.w = ((0.f + v.w * rmp[0].w) * rmp[2].w) * m_lg;
        }
      } else if (any_gain) {  // a skipped gain is a multiplication by exactly 1; the mixer's 0 + y only
                              // turns -0 into +0, which no output format can tell apart
        v.x = ((v.x * m_eg) * m_og) * m_lg;
        v.y = ((v.y * m_eg) * m_og) * m_lg;
        v.z = ((v.z * m_eg) * m_og) * m_lg;
        v.w = ((v.w * m_eg) * m_og) * m_lg;
      }
      y[c] = v;
      pm.x = fmaxf(pm.x, fabsf(v.x));
      pm.y = fmaxf(pm.y, fabsf(v.y));
      pm.z = fmaxf(pm.z, fabsf(v.z));
      pm.w = fmaxf(pm.w, fabsf(v.w));
    }
    if constexpr (FIR) __syncthreads();  // partials are consumed; the staging area is reused by the next chunk

    // ---- table window -> LDS, then the prefetch of the next chunk's input: the last vector-memory
    //      loads issued before the limiter work, so they stay in flight under everything below ----
    if constexpr (FIR == 3) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
        if (t + 256 * r < kFWin) win[t + 256 * r] = wv[r];
    }
    if constexpr (!FIR && !kEarly) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
        if (t + 256 * r < kFWin) win[t + 256 * r] = wv[r];
      const int kn = k + kFChunk;
      iu += kFChunk;   // (fr_uni) the next chunk's place: one frame on when this one ended its frame
      if (iu >= fs) {
        iu -= fs;
        ++fu;
      }
      if (kn < p.total) {
        int f, i;
        frame_pos(kn, f, i);
        load_lp(f, i);
        load_f32(f, i);
        load_drec(f);
        load_x2(f, i);
      }
    }

    // ---- per-16 prefix / suffix / block maxima: 4 lanes x 4 samples = one aligned block ----
    float4 pre_ex = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      const float i0 = pm.x, i1 = fmaxf(i0, pm.y), i2 = fmaxf(i1, pm.z), i3 = fmaxf(i2, pm.w);
      const float s3 = pm.w, s2 = fmaxf(pm.z, s3), s1 = fmaxf(pm.y, s2), s0 = fmaxf(pm.x, s1);
      const float qa = dpp_quad_bcast0(i3), qb = dpp_quad_bcast1(i3), qc = dpp_quad_bcast2(i3),
                  qd = dpp_quad_bcast3(i3);
      // maxima of the quad's lanes before / after this one (all values are >= 0: 0 is the identity); selects, no branches
      const float before = fmaxf(fmaxf(q >= 1 ? qa : 0.f, q >= 2 ? qb : 0.f), q >= 3 ? qc : 0.f);
      const float after = fmaxf(fmaxf(q <= 2 ? qd : 0.f, q <= 1 ? qc : 0.f), q <= 0 ? qb : 0.f);
      pre_ex = make_float4(before, fmaxf(before, i0), fmaxf(before, i1), fmaxf(before, i2));
      if (valid) {
#pragma unroll
        for (int c = 0; c < OC; ++c) *reinterpret_cast<float4 *>(&ring_y[c * R + rp]) = y[c];
        *reinterpret_cast<float4 *>(&ring_pm[rp]) = pm;
        *reinterpret_cast<float4 *>(&ring_suf[rp]) =
            make_float4(fmaxf(s0, after), fmaxf(s1, after), fmaxf(s2, after), fmaxf(s3, after));
        if (q == 0) ring_bm[rp >> 4] = ring_bm[(rp >> 4) + NB] = fmaxf(fmaxf(qa, qb), fmaxf(qc, qd));
      }
    }
    __syncthreads();

    // ---- 240-sample window maximum = tail of block b-15, blocks b-14..b-1, head of block b ----
    const int rd = ring_wrap(base + 4 * t - kDelay);  // ring position of sample gk - 240
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 pk4 = make_float4(0.f, 0.f, 0.f, 0.f);  // the lane's four window maxima
    if (act) {
      // blocks b-14 .. b-1: 14 consecutive words of the mirrored ring.  The four lanes of a quad belong to one block:
      // each reads four of the words (the last lane's overlap the third's), two quad exchanges give all four the maximum
      const float *bm = ring_bm + ((rp >> 4) + NB - 14 + (q < 3 ? 4 * q : 10));
      float w14 = fmaxf(fmaxf(bm[0], bm[1]), fmaxf(bm[2], bm[3]));
      w14 = fmaxf(w14, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w14), 0xB1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
      w14 = fmaxf(w14, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w14), 0x4E, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
      const float4 so = *reinterpret_cast<const float4 *>(&ring_suf[rd]);
      float4 pk;
      pk.x = fmaxf(fmaxf(so.x, w14), pre_ex.x);
      pk.y = fmaxf(fmaxf(so.y, w14), pre_ex.y);
      pk.z = fmaxf(fmaxf(so.z, w14), pre_ex.z);
      pk.w = fmaxf(fmaxf(so.w, w14), pre_ex.w);
      *reinterpret_cast<float4 *>(&arr_p[4 * t]) = pk;
      pk4 = pk;
    }
    // ---- limiter gains in rounds.  Every lane evaluates its gains under the hypothesis "no trigger
    //      from block bs on" (state = the one reached before block bs; round 0: the whole chunk, state
    //      of the chunk's start), a workgroup vote finds the first sample that contradicts it.  None:
    //      done.  Else the chain wave walks the recurrence from that block through the blocks that
    //      trigger and one more, and the next round re-evaluates what is left from the state it reached
    //      — so a chunk with a few isolated trigger runs costs a few short walks, not one walk to its end.
    const int n_chunk = n_st;  // what the staged table window is based on
    auto look = [&](int ci) {  // before the chunk's first trigger: the window; after one: the head
      if constexpr (!kTab) {
        return p.ctab[ci];
      } else {
        // (ci == n_chunk — d == -1 — only occurs past the head's range when ci is n_end itself, where gain_at returns 1
        //  whatever the coefficient: every step count after a trigger in this chunk stays below kFWin)
        const int d = ci - n_chunk - 1;
        return (d >= 0 && d < kFWin) ? win[d] : head[ci < kFWin ? ci : kFWin - 1];
      }
    };
    const int nblk = cnt >> 6;
    int bs = 0;
    while (true) {
      if (act) {
        int kfirst = kBig;
        const int o0 = 4 * t - 64 * bs;   // the lane's first sample, counted from the round's start state
        if (o0 >= 0) {
          // Hypothesis gains of the lane's four samples: sample j has made np = n_st + o0 + j steps since the state was
          // set and reads the curve at np + 1 (gain_at / look above, audio_effect_peak_limiter.c:241-255).  Below n_end no
          // clamp applies, at and past it the gain is 1 whatever is read.  The four coefficients are CONSECUTIVE words
          // of one staged table — round 0: the shifted window from its start; a later round: the head from n_st + 1 (a
          // walk has just ended, so n_st + 1 + o0 + 3 < kFWin) — one wave-uniform base, four reads, no index arithmetic.
          // Which formula applies is decided per ROUND where it can be (a chunk that starts idle; a chunk wholly inside
          // the 200 ms release — the two common cases), per sample otherwise.
          const int n0 = __builtin_amdgcn_readfirstlane(n_st);
          const int last = n0 + (cnt - 64 * bs) - 1;   // step count of the chunk's last sample under the hypothesis
          float gh[4] = {1.f, 1.f, 1.f, 1.f};
          if constexpr (kTab) {
            if (n0 < n_end) {
              const float *tb = (bs == 0 ? win : head + (n0 + 1)) + o0;
              const float c0v = tb[0], c1v = tb[1], c2v = tb[2], c3v = tb[3];
              const float cf[4] = {c0v, c1v, c2v, c3v};
              if (n0 >= n_atk && last < n_end) {   // release throughout: ge + c * (1 - ge)
                const float r1 = 1.0f - ge;
#pragma unroll
                for (int j = 0; j < 4; ++j) gh[j] = ge + cf[j] * r1;
              } else {
                const int nb = n0 + o0;
                const float a1 = gs - ge, r1 = 1.0f - ge;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float ga = gs - cf[j] * a1, gr = ge + cf[j] * r1;
                  gh[j] = nb + j < n_atk ? ga : (nb + j < n_end ? gr : 1.0f);
                }
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              int np = n_st + o0 + j;
              np = np < n_end ? np : n_end;
              const int ci = np + 1 < n_end ? np + 1 : n_end;
              gh[j] = gain_at(np, gs, ge, look(ci), n_atk, n_end);
            }
          }
          g = make_float4(gh[0], gh[1], gh[2], gh[3]);
          // the first sample that contradicts the hypothesis: sought only in a wave that has one
          const float px = pk4.x * g.x, py = pk4.y * g.y, pz = pk4.z * g.z, pw = pk4.w * g.w;
          const bool hit = valid && fmaxf(fmaxf(px, py), fmaxf(pz, pw)) > thr;
          if (__ballot(hit) != 0ull && hit) {
            if (pw > thr) kfirst = 4 * t + 3;
            if (pz > thr) kfirst = 4 * t + 2;
            if (py > thr) kfirst = 4 * t + 1;
            if (px > thr) kfirst = 4 * t + 0;
          }
          if (4 * t + 4 == cnt) misc[8] = g.w;  // gain of the chunk's last sample under the hypothesis
        }
        const unsigned long long any = __ballot(kfirst != kBig);
        if (lane == 0) misc[wave] = __int_as_float(any ? __builtin_amdgcn_readlane(kfirst, __builtin_ctzll(any)) : kBig);
      }
      __syncthreads();
      int kf = __float_as_int(misc[0]);
      kf = min(kf, __float_as_int(misc[1]));
      kf = min(kf, __float_as_int(misc[2]));
      kf = min(kf, __float_as_int(misc[3]));
      if (kf == kBig) {  // the hypothesis holds for the rest of the chunk
        g_cur = misc[8];
        n_st = n_st + (cnt - 64 * bs) < n_end ? n_st + (cnt - 64 * bs) : n_end;
        break;
      }
      const int b0 = kf >> 6;
      if (act && wave == cw) {
        int ln = n_st + 64 * (b0 - bs) < n_end ? n_st + 64 * (b0 - bs) : n_end;
        float lgs = gs, lge = ge, lgl = g_cur;
        const int be = limiter_wave(arr_p, arr_g, look, b0, nblk, ln, lgs, lge, lgl, thr, n_atk, n_end, true);
        if (lane == 0) {
          misc[4] = lgl;
          misc[5] = lgs;
          misc[6] = lge;
          misc[7] = __int_as_float(ln);
          misc[9] = __int_as_float(be);
        }
      }
      __syncthreads();
      const int be = __float_as_int(misc[9]);
      if (4 * t >= 64 * b0 && 4 * t < 64 * be) g = *reinterpret_cast<const float4 *>(&arr_g[4 * t]);
      g_cur = misc[4];
      gs = misc[5];
      ge = misc[6];
      n_st = __float_as_int(misc[7]);
      bs = be;
      if (bs >= nblk) break;
    }

    if constexpr (!FIR)
      if (c0 + kFChunk < p.total) fetch_window(n_st);  // for the next chunk, ahead of the stores

    // ---- emit 4 delayed samples * gain as interleaved PCM ----
    const int64_t j0 = gk - kDelay;
    if (valid && j0 >= 0) {
      float4 o[OC];
#pragma unroll
      for (int c = 0; c < OC; ++c) {
        const float4 d = *reinterpret_cast<const float4 *>(&ring_y[c * R + rd]);
        o[c] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      }
      uint8_t *dst = pcm + (j0 - out_base) * (int64_t)OC * bytes;
      if (p.out_format == IAMF_HIP_FMT_S16) {
        int v[OC][4];
#pragma unroll
        for (int c = 0; c < OC; ++c) {
          v[c][0] = (int)to_scaled(o[c].x, 32768.f, -32768.f, 32767.f);
          v[c][1] = (int)to_scaled(o[c].y, 32768.f, -32768.f, 32767.f);
          v[c][2] = (int)to_scaled(o[c].z, 32768.f, -32768.f, 32767.f);
          v[c][3] = (int)to_scaled(o[c].w, 32768.f, -32768.f, 32767.f);
        }
        if (OC == 2) {
          uint4 w;
          w.x = (uint32_t)(v[0][0] & 0xffff) | ((uint32_t)v[OC - 1][0] << 16);
          w.y = (uint32_t)(v[0][1] & 0xffff) | ((uint32_t)v[OC - 1][1] << 16);
          w.z = (uint32_t)(v[0][2] & 0xffff) | ((uint32_t)v[OC - 1][2] << 16);
          w.w = (uint32_t)(v[0][3] & 0xffff) | ((uint32_t)v[OC - 1][3] << 16);
          *reinterpret_cast<uint4 *>(dst) = w;
        } else {
          uint2 w;
          w.x = (uint32_t)(v[0][0] & 0xffff) | ((uint32_t)v[0][1] << 16);
          w.y = (uint32_t)(v[0][2] & 0xffff) | ((uint32_t)v[0][3] << 16);
          *reinterpret_cast<uint2 *>(dst) = w;
        }
      } else if (p.out_format == IAMF_HIP_FMT_S24) {
#pragma unroll
        for (int c = 0; c < OC; ++c) {
          const float ov[4] = {o[c].x, o[c].y, o[c].z, o[c].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int vv = (int)to_scaled(ov[j], 8388608.f, -8388608.f, 8388607.f);
            uint8_t *d3 = dst + (j * OC + c) * 3;
            d3[0] = (uint8_t)(vv & 0xff);
            d3[1] = (uint8_t)((vv >> 8) & 0xff);
            d3[2] = (uint8_t)(((vv >> 16) & 0x7f) | ((vv >> 24) & 0x80));
          }
        }
      } else if (p.out_format == IAMF_HIP_FMT_S32) {
        int32_t *d32 = reinterpret_cast<int32_t *>(dst);
#pragma unroll
        for (int c = 0; c < OC; ++c) {
          const float ov[4] = {o[c].x, o[c].y, o[c].z, o[c].w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            d32[j * OC + c] = (int32_t)(long long)to_scaled(ov[j], 2147483648.f, -2147483648.f, 2147483647.f);
        }
      } else {
        float *df = reinterpret_cast<float *>(dst);
#pragma unroll
        for (int c = 0; c < OC; ++c) {
          df[0 * OC + c] = o[c].x;
          df[1 * OC + c] = o[c].y;
          df[2 * OC + c] = o[c].z;
          df[3 * OC + c] = o[c].w;
        }
      }
    }
    base = base + cnt >= R ? base + cnt - R : base + cnt;
    __syncthreads();  // ring / arr slots are rewritten by the next chunk
  }

  if (FIR && act) {
    // input history for the next call: the last 256 samples of [old history | this call's input]
    float *hn = p.fir_hist_next + (int64_t)s * M * kFirHist;
    // (and its copy at the input's channel stride, where the batch keeps one for fir_fft_kernel<M, true>: every stage
    //  leaves both histories current, whichever stage takes the next call)
    const bool has_pre = p.fir_pre_next != nullptr;
    const int g256 = has_pre ? p.frame_size >> 8 : 1;
    const int64_t pre_off = has_pre ? ((int64_t)(s / g256) * M) * p.frame_size + (int64_t)(s % g256) * kFirHist : 0;
    for (int ch = 0; ch < M; ++ch) {
      const float x = fir_input(p, in_s, fir_hist, ch, p.total - kFirHist + t);
      hn[ch * kFirHist + t] = x;
      if (has_pre) p.fir_pre_next[pre_off + (int64_t)ch * p.frame_size + t] = x;
    }
  }
  // ---- persist stream state (same format as the generic kernel) ----
  if (act) {
    float *sy = p.ring_y + (int64_t)s * OC * kSave;
    float *spm = p.ring_pm + (int64_t)s * kSave;
    const int rp = ring_wrap(base - kSave + t);  // base = ring position of sample pos0 + total
#pragma unroll
    for (int c = 0; c < OC; ++c) sy[c * kSave + t] = ring_y[c * R + rp];
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
