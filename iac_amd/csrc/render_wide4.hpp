// render_wide4.hpp — multi-channel output layouts (even channel counts 4..24), 16-bit PCM, limiter
// on, calls of whole 1024-sample chunks (the last one may be shorter, down to 256 samples).  One workgroup (4 waves) per stream, FOUR consecutive
// samples per lane, and — unlike render_wide.hpp — the rendered samples never travel through an
// LDS ring:
//   * a lane keeps its 4 samples x C channels in registers and emits THEM itself once their gains
//     exist.  Sample j leaves the limiter with the gain computed at j + 240, so for the chunk's
//     first 784 samples (lanes 0..195) that gain is produced in the same chunk and only the GAIN
//     crosses lanes (one 16-byte LDS read); the last 240 samples (lanes 196..255) wait for the
//     next chunk in a lane-private LDS slot: each of those lanes swaps "my previous tail" for "my
//     current samples" and emits the previous tail.  Every lane emits exactly 4 sample-frames per
//     chunk as 8*C contiguous bytes;
//   * planar f32 input as 16-byte non-temporal loads, the next chunk's loads issued right after
//     the projection (the only vector-memory loads in flight across the limiter work besides the
//     table window, which is older in the in-order queue);
//   * projection on the VALU in the reference's operation order (packed f32 mul / add, no fma):
//     bit-exact.  Weights come from LDS as 16-byte broadcasts, 4 output slots x 4 samples at a time;
//   * 240-sample sliding maximum and the limiter exactly as in render_fast.hpp (no-trigger
//     hypothesis, wave-0 recurrence with ballot speculation and the DPP trigger-run chain), the
//     curve table staged per chunk as in render_wide.hpp (window reachable without a trigger +
//     the head that follows a trigger);
//   * two workgroup barriers per chunk on the no-trigger path.
#pragma once

// IAMF_W4_EXP: timing-only elimination builds (WRONG results; never the product, which is 0):
//   1 = no global PCM stores   2 = no pack / staging / stores   3 = no limiter gain rounds (gain 1)
//   4 = no projection (y = inputs)   6 = every PCM store redirected to the dump slot (no HBM writes)
//   7 = plain instead of non-temporal stores   8 = NO staging: each lane stores its own 8*C bytes (results stay RIGHT)
#ifndef IAMF_W4_EXP
#define IAMF_W4_EXP 0
#endif
constexpr int kW4Win = kFWin;       // staged table window / head length (render_fast.hpp)
constexpr int kW4TailLanes = 60;    // lanes holding the chunk's last 240 samples
constexpr int kW4FirstTail = 256 - kW4TailLanes;

// demixer variant: [2][5] frame records of 36 floats, the first 192 entries of the two cross-fade
// windows, playback position of every IAChannel
constexpr int kW4DmxRecs = 5, kW4DmxRec = 36, kW4DmxWin = 192;
constexpr int kW4DmxFloats = 2 * kW4DmxRecs * kW4DmxRec + 2 * kW4DmxWin + 24;  // records double-buffered

constexpr int kW4MixFloats = 4 * 24;  // mixing variant: [C][4] matrix rows of the second element
__host__ __device__ constexpr int wide4_base_floats(int c, int m, int extra) {
  return 4 * kW4TailLanes * c + 2 * kFRing + 2 * (kFRing / 16) + 2 * kFChunk + 2 * kW4Win + ((c + 3) & ~3) * m + 16 +
         extra;  // 0, kW4DmxFloats or kW4MixFloats
}
// Workgroups per CU the kernel is sized for.  The kernels are bound by (chunk latency) x (workgroups resident per
// CU) — NOTEBOOK.md 4.3 — so the light variants (no demixer / mixer, at most 12 output channels) are kept within a third
// of the CU: <= 52 KiB of LDS here, <= 168 VGPRs by themselves or (M, C <= 12: the 7.1.4 -> 7.1.4 VALU variant sits at
// 171) by __launch_bounds__; asking the same of the 16-input and down-mixer variants makes them spill.
__host__ __device__ constexpr int wide4_wgs(int c, int extra) { return (extra == 0 && c <= 12) ? 3 : 2; }
__host__ __device__ constexpr int wide4_budget_floats(int c, int extra) { return wide4_wgs(c, extra) == 3 ? 13312 : 20480; }
// PCM staging (per wave): a lane's 8*C output bytes = C/2 16-byte pieces, lane stride padded to an
// odd number of pieces (conflict-free).  All 64 lanes at once if that fits the budget, else 32 per round.
// The staging area starts ON arr_p (the chunk's 1024 window maxima, the last array of the base block): arr_p is
// written after barrier (1) and last read before the last barrier of the gain rounds, the staging area (and the
// demixer's scatter rows, which use it too) only between that barrier and the next chunk's barrier (1).
__host__ __device__ constexpr int wide4_stage_stride(int c) { return ((c / 2) & 1) ? c / 2 : c / 2 + 1; }
__host__ __device__ constexpr int wide4_stage_floats(int c, int lanes) { return 4 * lanes * wide4_stage_stride(c) * 4; }
__host__ __device__ constexpr int wide4_total_floats(int c, int m, int extra, int lanes) {
  return wide4_base_floats(c, m, extra) +
         (wide4_stage_floats(c, lanes) > kFChunk ? wide4_stage_floats(c, lanes) - kFChunk : 0);
}
__host__ __device__ constexpr int wide4_stage_lanes(int c, int m, int extra) {
  return wide4_total_floats(c, m, extra, 64) <= wide4_budget_floats(c, extra) ? 64 : 32;
}
__host__ __device__ constexpr int wide4_lds_floats(int c, int m, int extra) {
  return wide4_total_floats(c, m, extra, wide4_stage_lanes(c, m, extra));
}

using w4_f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float w4_comp(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ void w4_set(float4 &v, int i, float f) {
  if (i == 0) v.x = f;
  else if (i == 1) v.y = f;
  else if (i == 2) v.z = f;
  else v.w = f;
}
// 4x4 transpose across the wave's four 16-lane rows: on entry register a holds, in row b, element
// (a, b); on exit register a holds, in row b, element (b, a).  v_permlane32_swap exchanges rows 2,3
// of its first operand with rows 0,1 of its second; v_permlane16_swap rows 1,3 with rows 0,2.
__device__ __forceinline__ void w4_transpose_rows(float &r0, float &r1, float &r2, float &r3) {
  using u2 = __attribute__((ext_vector_type(2))) unsigned;
  u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
  u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  u2 c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  u2 d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  r0 = __uint_as_float(c[0]);
  r1 = __uint_as_float(c[1]);
  r2 = __uint_as_float(d[0]);
  r3 = __uint_as_float(d[1]);
}

// Demixer of scalable channel audio (reference demixer.c:636-664) on a lane's 4 consecutive samples:
// x[] arrives as the M decoded channels (bitstream order) and leaves as the target layout's channels
// in playback order.  The channel file, indexed by IAChannel, is a register array; scr = 64 * M (PAIR: 128 * M) floats
// of LDS private to the wave.
// Same operations in the same order as the generic kernel's demixer (render_generic.hpp).
template <int M, bool PAIR>
__device__ __forceinline__ void w4_demix(const RenderParams &p, float4 (&x)[M], 
                                         const float (&gin)[M], const int (&row)[kChCount], const float *dmx_rec,
                                         const float *dmx_ws,
                                         float *scr, int c0, int k4, int fs, int any_mask) {
  const int k = c0 + k4;
  const int f = k / fs;
  const int icur = k - f * fs;
  int j = f - c0 / fs;
  j = j < kW4DmxRecs - 1 ? j : kW4DmxRecs - 1;  // only lanes past the end of the call get clamped
  const float *rec = dmx_rec + kW4DmxRec * j;
  const int iw = icur + p.demix_i0;  // multiple of 4, like demix_skip: one mode for the lane's 4 samples
  const float *cf = rec + (iw < p.demix_skip ? 0 : 5);
  const float alpha = cf[0], beta = cf[1], gamma = cf[2], delta = cf[3], w = cf[4];
  float4 ws = make_float4(1.f, 1.f, 1.f, 1.f), wp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (iw < kW4DmxWin) {  // behind the cross-fade the windows are 1 and 0
    ws = *reinterpret_cast<const float4 *>(&dmx_ws[iw]);
    wp = *reinterpret_cast<const float4 *>(&dmx_ws[kW4DmxWin + iw]);
  }
  // Scatter: decoded channel m -> slot in_ch[m] of the channel file.  A register array cannot be
  // indexed by a run-time value, an LDS address can: every lane parks its M values in its own column
  // of the wave's scratch rows (row m = decoded channel m) and reads slot c back from row
  // demix_tab[40 + c] (row 0 for a slot no decoded channel lands in: such a slot holds a value no
  // planned step reads, iamf_hip_batch_set_demixer).  One sample position per pass; the data never
  // crosses lanes, so program order is all the synchronisation there is.
  float4 ch[kChCount];
  {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      // dmx_gainup (:426-435).  Unconditional: a channel without an output gain has g = 1.0f in the table
      // (iamf_hip_batch_set_demixer) and x * 1.0f is x, bit for bit — the test of the per-batch mask had been compiled
      // into a multiply AND four selects per channel (48 v_cndmask per lane and chunk).
      const float g = gin[m];
      x[m].x = x[m].x * g; x[m].y = x[m].y * g; x[m].z = x[m].z * g; x[m].w = x[m].w * g;
    }
    if constexpr (PAIR) {
      // two sample positions per pass (8-byte LDS operations, rows of 64 lanes x float2: row[c] = 128 * tab + 2 * lane):
      // 2 x (M + 23) LDS instructions per lane and chunk instead of 4 x (M + 23), where the wave's area holds 128 * M floats
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int m = 0; m < M; ++m)
          *reinterpret_cast<float2 *>(&scr[128 * m + 2 * lane]) = h == 0 ? make_float2(x[m].x, x[m].y) : make_float2(x[m].z, x[m].w);
#pragma unroll
        for (int c = 1; c < kChCount; ++c) {
          const float2 v = *reinterpret_cast<const float2 *>(&scr[row[c]]);
          if (h == 0) {
            ch[c].x = v.x;
            ch[c].y = v.y;
          } else {
            ch[c].z = v.x;
            ch[c].w = v.y;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) scr[64 * m + lane] = w4_comp(x[m], i);
#pragma unroll
        for (int c = 1; c < kChCount; ++c) w4_set(ch[c], i, scr[row[c]]);
      }
    }
  }
  const int steps = p.demix_steps;
  if (steps & 1) {  // S1to2 (:126-147)
    const float4 a = ch[kChMono], b = ch[kChL2];
    ch[kChR2] = make_float4(2 * a.x - b.x, 2 * a.y - b.y, 2 * a.z - b.z, 2 * a.w - b.w);
  }
  if (steps & 2) {  // S2to3 (:152-181): the reference's 0.707 literal makes this double arithmetic
    const float4 cc = ch[kChC], l = ch[kChL2], r = ch[kChR2];
    const double cx = 0.707 * (double)cc.x, cy = 0.707 * (double)cc.y, cz = 0.707 * (double)cc.z, cw = 0.707 * (double)cc.w;
    ch[kChL3] = make_float4((float)((double)l.x - cx), (float)((double)l.y - cy), (float)((double)l.z - cz), (float)((double)l.w - cw));
    ch[kChR3] = make_float4((float)((double)r.x - cx), (float)((double)r.y - cy), (float)((double)r.z - cz), (float)((double)r.w - cw));
  }
  // The 24 divisions of a lane and chunk have three divisors.  Eight numerators per divisor: quotients through the
  // divisor's reciprocal (w4_quot, render_common.hpp: bit-identical to the IEEE division in its proven range), and the
  // IEEE divisions themselves only if some numerator of the wave lies outside that range (wave-uniform branch).
  auto div8 = [](float4 &o0, float4 &o1, const float4 n0, const float4 n1, const float d) {
    const float r = 1.0f / d;
    bool ok = true;
    o0 = make_float4(w4_quot(n0.x, d, r, ok), w4_quot(n0.y, d, r, ok), w4_quot(n0.z, d, r, ok), w4_quot(n0.w, d, r, ok));
    o1 = make_float4(w4_quot(n1.x, d, r, ok), w4_quot(n1.y, d, r, ok), w4_quot(n1.z, d, r, ok), w4_quot(n1.w, d, r, ok));
    if (!__all(ok)) {
      o0 = make_float4(n0.x / d, n0.y / d, n0.z / d, n0.w / d);
      o1 = make_float4(n1.x / d, n1.y / d, n1.z / d, n1.w / d);
    }
  };
  if (steps & 4) {  // S3to5 (:186-230)
    const float4 a = ch[kChL3], b = ch[kChL7], c = ch[kChR3], d = ch[kChR7];
    div8(ch[kChSL5], ch[kChSR5], make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w),
         make_float4(c.x - d.x, c.y - d.y, c.z - d.z, c.w - d.w), delta);
  }
  if (steps & 8) {  // S5to7 (:236-284)
    const float4 a = ch[kChSL5], b = ch[kChSL7], c = ch[kChSR5], d = ch[kChSR7];
    div8(ch[kChBL7], ch[kChBR7], make_float4(a.x - b.x * alpha, a.y - b.y * alpha, a.z - b.z * alpha, a.w - b.w * alpha),
         make_float4(c.x - d.x * alpha, c.y - d.y * alpha, c.z - d.z * alpha, c.w - d.w * alpha), beta);
  }
  if (steps & 16) {  // TF2toT2 (:290-335)
    const float4 a = ch[kChTL], b = ch[kChSL5], c = ch[kChTR], d = ch[kChSR5];
    const float dw = delta * w;
    ch[kChHL] = make_float4(a.x - dw * b.x, a.y - dw * b.y, a.z - dw * b.z, a.w - dw * b.w);
    ch[kChHR] = make_float4(c.x - dw * d.x, c.y - dw * d.y, c.z - dw * d.z, c.w - dw * d.w);
  }
  if (steps & 32) {  // T2toT4 (:340-377)
    const float4 a = ch[kChHL], b = ch[kChHFL], c = ch[kChHR], d = ch[kChHFR];
    div8(ch[kChHBL], ch[kChHBR], make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w),
         make_float4(c.x - d.x, c.y - d.y, c.z - d.z, c.w - d.w), gamma);
  }
  // the target layout's channels in playback order; the layouts of M channels are the only candidates
  const int layout = p.demix_layout;
  if constexpr (M == 12) {  // 7.1.4 is the only one
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = ch[w4_layout_ch(7, m)];
  } else {
#pragma unroll
    for (int lt = 2; lt < 9; ++lt) {
      if (w4_layout_count(lt) == M && layout == lt) {
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = ch[w4_layout_ch(lt, m)];
      }
    }
  }
  // recon gain with its cross-fade from the last frame's smoothed value (dmx_rms, :447-478);
  // any_mask = the channels with a recon gain in any frame of this chunk (wave-uniform)
  const int mask = __float_as_int(rec[10]);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (!((any_mask >> m) & 1)) continue;
    const float2 g = *reinterpret_cast<const float2 *>(&rec[12 + 2 * m]);
    if ((mask >> m) & 1) {
      float4 v = x[m];
      v.x = v.x * (g.x * wp.x + g.y * ws.x);
      v.y = v.y * (g.x * wp.y + g.y * ws.y);
      v.z = v.z * (g.x * wp.z + g.y * ws.z);
      v.w = v.w * (g.x * wp.w + g.y * ws.w);
      x[m] = v;
    }
  }
}

// DOWN: the element is rendered by the parametric down-mixer (render_downmix.hpp) instead of a matrix.
// MIX:  the mixing variant, as in render_fast.hpp: a second element of at most kFIn2 channels rendered
//       by its own matrix and mixed in, and / or per-sample element / output gain ramps.
// LFE:  output slots marked in p.lfe_mask carry the HOA LFE generator's output (render_lfe.hpp, p.lfe) instead of a
//       matrix row (h2m_rdr.c:1154-1184), as in the generic kernel.
template <int M, int C, bool MFMA, bool DMX, bool DOWN = false, bool MIX = false, bool LFE = false>
__global__ __launch_bounds__(256, (M <= 12 && C <= 12 && !(DMX || DOWN || MIX)) ? 3 : 2) void render_wide4_kernel(const RenderParams p) {
  static_assert(!(LFE && (DMX || DOWN || MIX)), "the LFE generator belongs to a matrix-rendered ambisonics element");
  static_assert((C & 1) == 0 && C >= 4 && C <= 24, "even channel counts");
  static_assert(!(DMX && MFMA), "the demixer variant projects on the VALU");
  static_assert(!(DOWN && (MFMA || DMX)), "one renderer");
  static_assert(!(MIX && (DMX || DOWN)), "the second element joins a matrix-rendered first one");
  constexpr int kExtra = DMX ? kW4DmxFloats : (MIX ? kW4MixFloats : 0);
  extern __shared__ float lds[];
  constexpr int R = kFRing;
  constexpr int NB = R / 16;
  constexpr int C4 = (C + 3) & ~3;
  float4 *tail = reinterpret_cast<float4 *>(lds);   // [C][60]  lane-private: previous chunk's last 240 samples
  float *ring_pm = lds + 4 * kW4TailLanes * C;      // [R]     max |y| over channels
  float *ring_suf = ring_pm + R;                    // [R]     suffix maxima inside aligned 16-blocks
  float *ring_bm = ring_suf + R;                    // [2][R/16] maxima of aligned 16-blocks, stored twice so
                                                    //           that 'block b - j' needs no wrap
  float *arr_g = ring_bm + 2 * NB;                  // [1024]  limiter gains of the chunk
  float *win = arr_g + kFChunk;                     // [kW4Win] ctab[min(n_st + 1 + i, n_end)]
  float *head = win + kW4Win;                       // [kW4Win] ctab[i]
  float *mat = head + kW4Win;                       // [M][C4] weights, input-major
  float *misc = mat + C4 * M;                       // [16]
  float *dmx_rec = misc + 16;                            // DMX: [2][5][36] frame records of this / the next chunk
  float *dmx_ws = dmx_rec + 2 * kW4DmxRecs * kW4DmxRec;  // DMX: [192] start window, [192] stop window
  int *dmx_pos = reinterpret_cast<int *>(dmx_ws + 2 * kW4DmxWin);  // DMX: [24] playback position of an IAChannel
  float *mat2 = misc + 16;                           // MIX: [C][4] second element's matrix rows (aliases dmx_rec)
  float *arr_p = misc + 16 + kExtra;                // [1024]  window maxima of the chunk
  uint4 *stage = reinterpret_cast<uint4 *>(arr_p);  // [4 waves][LR][S] packed PCM, over arr_p (see wide4_stage_lanes)
  static_assert(wide4_lds_floats(C, M, kExtra) <= wide4_budget_floats(C, kExtra), "LDS per workgroup");

  // a launch covers streams [stream0, stream0 + n_launch) of the batch.  LFE: the generator's output lies transposed by
  // blocks of 64 streams (one 16-byte piece per stream and quad, render_lfe.hpp), so the 64 workgroups of a block read
  // the same 64-byte sectors — and workgroups go to the eight XCDs in turn, each with an L2 of its own: every XCD
  // fetched every sector (PMC: 108.9 B per sample-frame where 80 are needed).  Workgroup b therefore takes stream
  // (b mod 8) * n / 8 + b / 8: the workgroups of one XCD = a contiguous eighth of the streams = whole blocks.
  int wg = blockIdx.x;
  if constexpr (LFE) {
    if ((p.n_launch & 511) == 0) wg = (wg & 7) * (p.n_launch >> 3) + (wg >> 3);
  }
  const int s = wg + p.stream0;
  const int t = threadIdx.x;
  const int wave = t >> 6;
  const int lane = t & 63;
  const int fs = p.frame_size;
  const float thr = p.thr;
  const int n_atk = p.n_atk, n_end = p.n_end;
  const bool is_tail = t >= kW4FirstTail;
  const int tl = is_tail ? t - kW4FirstTail : 0;
  int base = (int)((p.pos0 & ~(int64_t)15) % R);  // ring position of the chunk's first sample: a multiple of 16 (render_fast.hpp)

  // ---- stream state and constants -> LDS (persisted format is the generic kernel's) ----
  {
    const float *sy = p.ring_y + (int64_t)s * C * kSave;
    const float *spm = p.ring_pm + (int64_t)s * kSave;
    const int rp = ring_wrap(base - kSave + t);  // saved entry t is sample pos0 - 256 + t
    if (is_tail) {                               // tail lane tl: samples pos0 - 240 + 4*tl .. +3
#pragma unroll
      for (int c = 0; c < C; ++c)
        tail[c * kW4TailLanes + tl] = *reinterpret_cast<const float4 *>(&sy[c * kSave + 16 + 4 * tl]);
    }
    const float pm = spm[t];
    ring_pm[rp] = pm;
    float sfx = pm;
    sfx = fmaxf(sfx, __shfl_down(sfx, 1, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 2, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 4, 16));
    sfx = fmaxf(sfx, __shfl_down(sfx, 8, 16));
    ring_suf[rp] = sfx;
    if ((t & 15) == 0) ring_bm[rp >> 4] = ring_bm[(rp >> 4) + NB] = sfx;
    for (int i = t; i < kW4Win; i += 256) head[i] = p.ctab[i < n_end ? i : n_end];
    for (int i = t; i < C4 * M; i += 256) {
      const int m = i / C4, c = i - m * C4;
      const int f = c < C ? p.src_feed[c] : -1;
      mat[i] = f >= 0 ? p.matrix[f * M + m] : 0.f;
    }
    if constexpr (MIX) {
      if (p.in2)
        for (int i = t; i < 4 * C; i += 256) {
          const int c = i >> 2, m = i & 3;
          const int f = p.src_feed2[c];
          mat2[i] = (f >= 0 && m < p.m2) ? p.matrix2[f * p.m2 + m] : 0.f;
        }
    }
    chain_wave_publish(misc + 12);
    if constexpr (DMX) {
      for (int i = t; i < kW4DmxWin; i += 256) {
        dmx_ws[i] = p.demix_ftab[12 + i];
        dmx_ws[kW4DmxWin + i] = p.demix_ftab[12 + fs + i];
      }
      if (t < 24) {
        int pos = -1;
        for (int m = 0; m < M; ++m)
          if (p.demix_tab[12 + m] == t) pos = m;
        dmx_pos[t] = pos;
      }
    }
  }
  // Demixer, control side.  A chunk touches at most 5 frames (frame_size >= 256).  Their records
  // (iamf_hip_demix_frame: 47 dwords, recon gains keyed by IAChannel) are re-keyed by playback
  // position into LDS: [0..5) factors of the previous mode, [5..10) of the current one, [10] bit m =
  // output channel m has a recon gain, [12 + 2m] its smoothed gain of the last frame, [13 + 2m] of
  // this frame.  16 threads per record fetch 5 dwords each next to the table window (ahead of the
  // PCM stores and of the input prefetch in the in-order vector-memory queue) and write them with the
  // window, into the buffer the current chunk does not read.
  const int dmx_nfr = DMX ? (p.total + fs - 1) / fs : 0;
  float dv[DMX ? 5 : 1];
  auto fetch_rec = [&](int cbase, int tt) {
    if constexpr (DMX) {
      // every thread loads (those past the 80 that write repeat record 4): straight-line code, so
      // that nothing has to wait for these loads before put_rec does
      const int j = (tt >> 4) < kW4DmxRecs ? (tt >> 4) : kW4DmxRecs - 1, r = tt & 15;
      int fi = cbase / fs + j;
      fi = fi < dmx_nfr ? fi : dmx_nfr - 1;
      const float *d = reinterpret_cast<const float *>(p.demix_frames + (int64_t)s * dmx_nfr + fi);
      // r < 12: n_recon, recon_ch[r], recon_prev[r], recon_cur[r];  r = 12: prev[0..5);  r > 12: cur[0..5)
      const int o0 = r < 12 ? 10 : (r == 12 ? 0 : 5), st = r < 12 ? 12 : 1, o1 = r < 12 ? 11 + r : o0 + 1;
      dv[0] = d[o0];
      dv[1] = d[o1];
      dv[2] = d[o1 + st];
      dv[3] = d[o1 + 2 * st];
      dv[4] = d[r < 12 ? 10 : o0 + 4];
    }
  };
  auto put_rec = [&](int tt, int buf) {
    if constexpr (DMX) {
      if (tt < 16 * kW4DmxRecs) {
        const int j = tt >> 4, r = tt & 15;
        float *rec = dmx_rec + kW4DmxRec * (kW4DmxRecs * buf + j);
        int bit = 0;
        if (r < 12) {
          const int n = __float_as_int(dv[0]), ch = __float_as_int(dv[1]);
          if (r < n && ch > 0 && ch < 24) {
            const int m = dmx_pos[ch];
            if (m >= 0) {
              rec[12 + 2 * m] = dv[2];
              rec[13 + 2 * m] = dv[3];
              bit = 1 << m;
            }
          }
        }
        bit |= __shfl_xor(bit, 1, 16);
        bit |= __shfl_xor(bit, 2, 16);
        bit |= __shfl_xor(bit, 4, 16);
        bit |= __shfl_xor(bit, 8, 16);
        if (r == 12) rec[10] = __int_as_float(bit);
        if (r == 12 || r == 13) {
#pragma unroll
          for (int i = 0; i < 5; ++i) rec[(r == 12 ? 0 : 5) + i] = dv[i];
        }
      }
    }
  };
  // the demixer's scatter rows lie in the wave's PCM staging area: two sample positions per pass where 128 * M floats fit
  constexpr bool kDmxPair = DMX && wide4_stage_lanes(C, M, kW4DmxFloats) * wide4_stage_stride(C) * 4 >= 128 * M;
  float dmx_gin[DMX ? M : 1];  // output gain of every decoded channel (wave-uniform)
  int dmx_row[DMX ? kChCount : 1];  // where the lane finds IAChannel c in the wave's scatter rows (w4_demix)
  if constexpr (DMX) {
#pragma unroll
    for (int c = 1; c < kChCount; ++c) dmx_row[c] = kDmxPair ? 128 * p.demix_tab[40 + c] + 2 * lane : 64 * p.demix_tab[40 + c] + lane;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      dmx_gin[m] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__float_as_int(p.demix_ftab[12 + 2 * fs + m])));
    }
    fetch_rec(0, t);
    __syncthreads();  // dmx_pos visible
    put_rec(t, 0);
    fetch_rec(kFChunk, t);  // written to LDS in the first chunk
  }
  LimState ls = p.lim[s];
  float g_cur = ls.g, gs = ls.gs, ge = ls.ge;
  int n_st = ls.n;
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);
  // a gain the reference would skip is a multiplication by exactly 1; the mixer's 0 + y only turns
  // -0 into +0, which no output format can tell apart
  const float m_eg = eg_on ? eg : 1.f, m_og = og_on ? og : 1.f, m_lg = lg_on ? lg : 1.f;
  const bool any_gain = eg_on || og_on || lg_on;
  // MIX: the second element's channels of the lane's samples; the gains of element 0, element 1 and the
  // output as the call's per-sample ramps where given, else the constant gains (a gain the reference
  // skips is a multiplication by exactly 1)
  float4 x2[MIX ? kFIn2 : 1], rmp[MIX ? 3 : 1];
  if constexpr (MIX) {
    const float eg2 = p.in2 ? p.gains2[s] : 1.f;
    const float m_eg2 = (eg2 != 1.f && eg2 > 0.f) ? eg2 : 1.f;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    rmp[0] = p.elem_ramp ? one : make_float4(m_eg, m_eg, m_eg, m_eg);
    rmp[1] = p.elem2_ramp ? one : make_float4(m_eg2, m_eg2, m_eg2, m_eg2);
    rmp[2] = p.out_ramp ? one : make_float4(m_og, m_og, m_og, m_og);
#pragma unroll
    for (int m = 0; m < kFIn2; ++m) x2[m] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // gains (and, MIX, the mixer) on the 4 samples of output slot c, in the reference's order
  auto gains4 = [&](int c, float4 v) -> float4 {
    if constexpr (MIX) {
      if (p.in2) {
        const float4 w = *reinterpret_cast<const float4 *>(&mat2[4 * c]);  // zeros beyond m2 and for a silent slot
        float4 v2 = make_float4(0.f, 0.f, 0.f, 0.f);
        v2.x = v2.x + w.x * x2[0].x; v2.y = v2.y + w.x * x2[0].y; v2.z = v2.z + w.x * x2[0].z; v2.w = v2.w + w.x * x2[0].w;
        v2.x = v2.x + w.y * x2[1].x; v2.y = v2.y + w.y * x2[1].y; v2.z = v2.z + w.y * x2[1].z; v2.w = v2.w + w.y * x2[1].w;
        v2.x = v2.x + w.z * x2[2].x; v2.y = v2.y + w.z * x2[2].y; v2.z = v2.z + w.z * x2[2].z; v2.w = v2.w + w.z * x2[2].w;
        v2.x = v2.x + w.w * x2[3].x; v2.y = v2.y + w.w * x2[3].y; v2.z = v2.z + w.w * x2[3].z; v2.w = v2.w + w.w * x2[3].w;
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
    } else if (any_gain) {
      v.x = ((v.x * m_eg) * m_og) * m_lg;
      v.y = ((v.y * m_eg) * m_og) * m_lg;
      v.z = ((v.z * m_eg) * m_og) * m_lg;
      v.w = ((v.w * m_eg) * m_og) * m_lg;
    }
    return v;
  };

  const int lead = p.pos0 > kDelay ? kDelay : (int)p.pos0;  // pos0 - (first sample of the output buffer)
  const int pos_small = p.pos0 > (1 << 20) ? (1 << 20) : (int)p.pos0;
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;

  // Input registers.
  //   VALU variant: x[m] = this lane's 4 samples of channel m.
  //   MFMA variant (v_mfma_f32_16x16x4_f32, D[channel][sample] = W[channel][input] * X[input][sample]):
  //   the B operand of lane (j = lane & 15, kg = lane >> 4) at k-step ks for sample column j of
  //   column group cg is X[4*ks + kg][...], so the lane loads, instead of all channels of its own
  //   samples, the channels = kg (mod 4) of the samples owned by lanes 16*cg + j, cg = 0..3:
  //   x[4*ks + cg] — the same number of 16-byte loads, already in operand layout.
  constexpr int KS = (M + 3) / 4, RT = (C + 15) / 16;
  constexpr int NX = MFMA ? 4 * KS : M;
  float4 x[NX];
  float4 lq = make_float4(0.f, 0.f, 0.f, 0.f);  // LFE: the generator's output for the lane's 4 samples, fetched with them
  float drec[DOWN ? 11 : 1];  // DOWN: the frame record (iamf_hip_dmx_frame) of the lane's samples, fetched with them
  const int down_nfr = DOWN ? (p.total + fs - 1) / fs : 0;
  auto load_x = [&](int cbase, int tt) {
    if constexpr (LFE) {  // transposed by blocks of 64 streams (lfe_index, render_lfe.hpp); unconditional like the rest
      const int k0 = cbase + 4 * tt;
      const int k = k0 < p.total ? k0 : p.total - 4;   // past the end: the call's last quad (total >= 256, a multiple of 64)
      // (default cache policy, not the streaming one of the element PCM: the 64-byte sector this piece lies in is read by
      //  three neighbouring streams' workgroups too)
      lq = *reinterpret_cast<const float4 *>(p.lfe + (((((int64_t)(s >> 6) * p.lfe_t4 + (k >> 2)) * 64 + (s & 63)) << 2)));
    }
    if constexpr (MIX) {
      // the lane's own 4 samples (lanes past the end of a short last chunk re-read the call's last quad)
      int k = cbase + 4 * tt;
      k = k < p.total ? k : p.total - 4;
      const int f = k / fs;
      const int i = k - f * fs;
      if (p.in2) {
        const float *src = p.in2 + (int64_t)s * p.in2_stream_stride + (int64_t)f * p.in2_frame_stride + i;
#pragma unroll
        for (int m = 0; m < kFIn2; ++m)
          x2[m] = m < p.m2 ? ld_stream4(src + (int64_t)m * fs) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int64_t ro = (int64_t)s * p.ramp_stream_stride + k;
      if (p.elem_ramp) rmp[0] = ld_stream4(p.elem_ramp + ro);
      if (p.elem2_ramp) rmp[1] = ld_stream4(p.elem2_ramp + ro);
      if (p.out_ramp) rmp[2] = ld_stream4(p.out_ramp + ro);
    }
    if constexpr (DOWN) {
      const int f = (cbase + 4 * tt) / fs;
      const float *d = reinterpret_cast<const float *>(p.dmx_frames + (int64_t)s * down_nfr + (f < down_nfr ? f : down_nfr - 1));
#pragma unroll
      for (int i = 0; i < 11; ++i) drec[i] = d[i];
    }
    if constexpr (MFMA) {
      const int j = tt & 15, kg = (tt >> 4) & 3, wv_ = tt >> 6;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        // every lane loads on every call: the same vector-memory instructions in every chunk, see the note at
        // the PCM stores.  Lanes past the end of the call read the call's first samples instead; nothing such
        // a lane computes is ever stored (ring, PCM and state writes are all guarded by `valid` / `emit`)
        const int k0 = cbase + 256 * wv_ + 4 * (16 * cg + j);
        const int k = k0 < p.total ? k0 : 4 * (16 * cg + j);
        const int f = k / fs;
        const int i = k - f * fs;
        const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int m = 4 * ks + kg;
          x[4 * ks + cg] = (M % 4 == 0 || m < M) ? ld_stream4(src + (int64_t)m * fs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else if constexpr (DMX) {
      // lanes past the end of a short last chunk load the call's last quad instead of zeros: nothing
      // they compute is ever stored, and unconditional loads keep the prefetch free of branches
      int k = cbase + 4 * tt;
      k = k < p.total ? k : p.total - 4;
      const int f = k / fs;
      const int i = k - f * fs;
      const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = ld_stream4(src + (int64_t)m * fs);
    } else {
      // unconditional loads, as in the MFMA variant; a lane past the end of the call re-reads the call's last quad
      // (4 * tt would lie past a call shorter than one chunk: total is only known to be >= 256 and a multiple of 64)
      const int k0 = cbase + 4 * tt;
      const int k = k0 < p.total ? k0 : p.total - 4;
      const int f = k / fs;
      const int i = k - f * fs;
      const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = ld_stream4(src + (int64_t)m * fs);
    }
  };
  load_x(0, t);
  __syncthreads();
  const int cw = chain_wave_pick(misc + 12);

  // Table window the next chunk can reach without a trigger: win[i] = ctab[min(n_st + 1 + i, n_end)] (what sample i of
  // the chunk reads if nothing triggers before it: a lane's four samples are four consecutive, 16-byte aligned words).
  // Fetched BEFORE the chunk's PCM stores are issued: vector-memory operations retire in order, so
  // a load issued after the stores could only be waited for by draining the stores as well.
  float wv[5];
  auto fetch_window = [&](int n0, int tt) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {  // five loads, always (ctab[n_end] is 1: the idle limiter)
      const int i = n0 + 1 + tt + 256 * r;
      wv[r] = p.ctab[i < n_end ? i : n_end];
    }
  };
  fetch_window(n_st, t);

  // One drain before the loop (once per call): with nothing pending on entry, the waits inside the loop are
  // the ones the loop body itself implies.
  if constexpr (!(DMX || DOWN || MIX)) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  float4 y[C];
  for (int c0 = 0; c0 < p.total; c0 += kFChunk) {
    // opaque per-chunk copy of the thread index: address arithmetic derived from it is recomputed
    // per chunk (a few dozen VALU ops) instead of being hoisted into ~60 loop-invariant registers
    int tv = threadIdx.x;
    asm volatile("" : "+v"(tv));
    const int lane = tv & 63, q = tv & 3;
    const bool is_tail = tv >= kW4FirstTail;
    const int tl = is_tail ? tv - kW4FirstTail : 0;
    const int cnt = p.total - c0 < kFChunk ? p.total - c0 : kFChunk;  // only the call's last chunk may be short:
                                                                      // a multiple of 64, at least 256
    const bool valid = 4 * tv < cnt;
    const int rp = ring_wrap(base + 4 * tv);

    float4 xw[NX];
    float4(&X)[NX] = DMX ? xw : x;  // what the projection reads
    if constexpr (DMX) {
      // the prefetched 16-byte tuples stay what the loop carries; the demixer works on copies, so that
      // no later use of a component can pull a copy (and with it a wait) up to where the load is issued
#pragma unroll
      for (int m = 0; m < M; ++m) {
        xw[m] = x[m];
        asm volatile("" : "+v"(xw[m].x), "+v"(xw[m].y), "+v"(xw[m].z), "+v"(xw[m].w));
      }
      const float *recs = dmx_rec + kW4DmxRec * kW4DmxRecs * ((c0 >> 10) & 1);
      int any_mask = 0;
#pragma unroll
      for (int j = 0; j < kW4DmxRecs; ++j) any_mask |= __float_as_int(recs[kW4DmxRec * j + 10]);
      static_assert(wide4_stage_lanes(C, M, kW4DmxFloats) * wide4_stage_stride(C) * 4 >= 64 * M, "scatter rows fit the wave's staging area");
      float *scr = reinterpret_cast<float *>(stage + (tv >> 6) * (wide4_stage_lanes(C, M, kW4DmxFloats) * wide4_stage_stride(C)));
      w4_demix<M, kDmxPair>(p, xw, dmx_gin, dmx_row, recs, dmx_ws, scr, c0, 4 * tv, fs, __builtin_amdgcn_readfirstlane(any_mask));
    }

    // ---- element renderer + gains (reference operation order), 4 slots x 4 samples at a time ----
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 lv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LFE) {  // `* 0.5` or `/ sqrt(n)`: double expressions narrowed by the store (h2m_rdr.c:1162)
      const double dv = p.lfe_div;
      if (dv == 0.0)
        lv = make_float4((float)((double)lq.x * 0.5), (float)((double)lq.y * 0.5), (float)((double)lq.z * 0.5), (float)((double)lq.w * 0.5));
      else
        lv = make_float4((float)((double)lq.x / dv), (float)((double)lq.y / dv), (float)((double)lq.z / dv), (float)((double)lq.w / dv));
    }
    auto slot_value = [&](int c, const float4 v) -> float4 {  // what slot c carries before the gains
      if constexpr (LFE) return ((p.lfe_mask >> c) & 1) ? lv : v;
      return v;
    };
    if constexpr (DOWN) {
      float4 cf[5];
      const int kk = c0 + 4 * tv;
      downmix_factors(drec, kk - (kk / fs) * fs, cf);
      downmix4<M, C>(x, y, cf, p.dmx_in_layout, p.dmx_out_layout);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float4 v = gains4(c, y[c]);
        y[c] = v;
        pm.x = fmaxf(pm.x, fabsf(v.x));
        pm.y = fmaxf(pm.y, fabsf(v.y));
        pm.z = fmaxf(pm.z, fabsf(v.z));
        pm.w = fmaxf(pm.w, fabsf(v.w));
      }
    } else if constexpr (MFMA) {
      // One sample position i of every lane's quad at a time: 4 column groups x RT row tiles.  Lane
      // (j, g) receives channels 16*rt + 4*g + r of the sample owned by lane 16*cg + j; the row
      // transpose hands every lane all channels of its own sample.  Exact f32 products in a
      // k-ordered fma chain: <= 1 ulp per term away from the reference's separately rounded
      // multiply and add (tests/test_gpu_mfma.py: +-1 LSB of the PCM).
      // MFMA A operand: lane (i = lane & 15, kg = lane >> 4) holds W[slot 16*rt + i][input 4*ks + kg], read
      // per chunk from the LDS copy of the matrix (mat[input][slot], zero for silent slots): eight registers
      // that are not live through the limiter and pack phases
      float aw[RT * KS];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int slot = 16 * rt + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int m = 4 * ks + (lane >> 4);
          aw[rt * KS + ks] = (slot < C4 && m < M) ? mat[m * C4 + slot] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w4_f32x4 acc[4][RT];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[cg][rt] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[cg][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[rt * KS + ks], w4_comp(x[4 * ks + cg], i), acc[cg][rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float r0 = acc[0][rt][r], r1 = acc[1][rt][r], r2 = acc[2][rt][r], r3 = acc[3][rt][r];
            w4_transpose_rows(r0, r1, r2, r3);
            const float rr[4] = {r0, r1, r2, r3};
#pragma unroll
            for (int g = 0; g < 4; ++g)
              if (16 * rt + 4 * g + r < C) w4_set(y[16 * rt + 4 * g + r], i, rr[g]);
          }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float4 v = gains4(c, slot_value(c, y[c]));
        y[c] = v;
        pm.x = fmaxf(pm.x, fabsf(v.x));
        pm.y = fmaxf(pm.y, fabsf(v.y));
        pm.z = fmaxf(pm.z, fabsf(v.z));
        pm.w = fmaxf(pm.w, fabsf(v.w));
      }
    } else {
      constexpr int MB = (C >= 24 || (M & 1)) ? 1 : 2, NB_M = M / MB, G = C4 / 4;
      static_assert(M % MB == 0, "inputs per batch");
      constexpr float4 z4 = {0.f, 0.f, 0.f, 0.f};
      auto mac = [&](const float4 w, const float4 xv, int g, float4 &a0, float4 &a1, float4 &a2, float4 &a3) {
        a0.x = a0.x + w.x * xv.x; a0.y = a0.y + w.x * xv.y; a0.z = a0.z + w.x * xv.z; a0.w = a0.w + w.x * xv.w;
        a1.x = a1.x + w.y * xv.x; a1.y = a1.y + w.y * xv.y; a1.z = a1.z + w.y * xv.z; a1.w = a1.w + w.y * xv.w;
        if (4 * g + 2 < C) {
          a2.x = a2.x + w.z * xv.x; a2.y = a2.y + w.z * xv.y; a2.z = a2.z + w.z * xv.z; a2.w = a2.w + w.z * xv.w;
          a3.x = a3.x + w.w * xv.x; a3.y = a3.y + w.w * xv.y; a3.z = a3.z + w.w * xv.z; a3.w = a3.w + w.w * xv.w;
        }
      };
      auto finish = [&](int g, const float4 (&acc)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (4 * g + i < C) {
            const float4 v = gains4(4 * g + i, slot_value(4 * g + i, acc[i]));
            y[4 * g + i] = v;
            pm.x = fmaxf(pm.x, fabsf(v.x));
            pm.y = fmaxf(pm.y, fabsf(v.y));
            pm.z = fmaxf(pm.z, fabsf(v.z));
            pm.w = fmaxf(pm.w, fabsf(v.w));
          }
        }
      };
      if (p.sparse) {
        // Sparse matrices (the 7.1.4 -> J matrix of cfg2 is the identity, down-mix matrices have a
        // few entries per row): a batch of inputs whose weights are zero for all four slots of the
        // group adds 0 * x = +-0 to every accumulator, which leaves them unchanged for every FINITE
        // input, so it is skipped (wave-uniform branch on the host-computed mask).  Non-finite
        // inputs — which an LPCM decoder cannot produce — would give NaN in the reference.
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float4 a0 = z4, a1 = z4, a2 = z4, a3 = z4;
          const uint32_t nz = p.nz_mask[g];
#pragma unroll
          for (int b = 0; b < NB_M; ++b) {
            if (nz & (((1u << MB) - 1u) << (b * MB))) {
#pragma unroll
              for (int j = 0; j < MB; ++j)
                mac(*reinterpret_cast<const float4 *>(&mat[(b * MB + j) * C4 + 4 * g]), X[b * MB + j], g, a0, a1, a2, a3);
            }
          }
          const float4 acc[4] = {a0, a1, a2, a3};
          finish(g, acc);
        }
      } else {
        // weights of MB inputs x 4 slots per LDS batch, the next batch fetched while this one is
        // multiplied (explicit double buffer: left alone the scheduler hoists a whole group's
        // weight reads and runs out of registers)
        float4 wc[MB], wn[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j) wc[j] = *reinterpret_cast<const float4 *>(&mat[j * C4]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float4 a0 = z4, a1 = z4, a2 = z4, a3 = z4;
#pragma unroll
          for (int b = 0; b < NB_M; ++b) {
            const int nb = b + 1 < NB_M ? b + 1 : 0, ng = b + 1 < NB_M ? g : g + 1;
            if (ng < G) {
#pragma unroll
              for (int j = 0; j < MB; ++j) wn[j] = *reinterpret_cast<const float4 *>(&mat[(nb * MB + j) * C4 + 4 * ng]);
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) mac(wc[j], X[b * MB + j], g, a0, a1, a2, a3);
#pragma unroll
            for (int j = 0; j < MB; ++j) wc[j] = wn[j];
            __builtin_amdgcn_sched_barrier(0);
          }
          const float4 acc[4] = {a0, a1, a2, a3};
          finish(g, acc);
        }
      }
    }

#if IAMF_W4_EXP == 4
#pragma unroll
    for (int c = 0; c < C; ++c) y[c] = X[c % NX];
#endif
    // ---- table window -> LDS, THEN the prefetch of the next chunk's input.  Vector-memory operations
    //      retire in order: waiting for the window values (fetched before the previous chunk's stores)
    //      after the prefetch had been issued would drain the prefetch on the spot ----
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (tv + 256 * r < kW4Win) win[tv + 256 * r] = wv[r];
    if constexpr (DMX || DOWN || MIX) {
      if (c0 + kFChunk < p.total) {
        put_rec(tv, ((c0 >> 10) + 1) & 1);  // the next chunk's frame records
        load_x(c0 + kFChunk, tv);
      }
    } else {
      load_x(c0 + kFChunk, tv);  // past the end of the call: loaded from its start and dropped (see load_x)
    }

    // ---- per-16 prefix / suffix / block maxima: 4 lanes x 4 samples = one aligned block ----
    const float i0 = pm.x, i1 = fmaxf(i0, pm.y), i2 = fmaxf(i1, pm.z), i3 = fmaxf(i2, pm.w);
    const float s3 = pm.w, s2 = fmaxf(pm.z, s3), s1 = fmaxf(pm.y, s2), s0 = fmaxf(pm.x, s1);
    const float qa = dpp_quad_bcast0(i3), qb = dpp_quad_bcast1(i3), qc = dpp_quad_bcast2(i3),
                qd = dpp_quad_bcast3(i3);
    // maxima of the quad's lanes before / after this one (all values are >= 0: 0 is the identity); selects, no branches
    const float before = fmaxf(fmaxf(q >= 1 ? qa : 0.f, q >= 2 ? qb : 0.f), q >= 3 ? qc : 0.f);
    const float after = fmaxf(fmaxf(q <= 2 ? qd : 0.f, q <= 1 ? qc : 0.f), q <= 0 ? qb : 0.f);
    const float4 pre_ex = make_float4(before, fmaxf(before, i0), fmaxf(before, i1), fmaxf(before, i2));
    if (valid) {
      *reinterpret_cast<float4 *>(&ring_pm[rp]) = pm;
      *reinterpret_cast<float4 *>(&ring_suf[rp]) =
          make_float4(fmaxf(s0, after), fmaxf(s1, after), fmaxf(s2, after), fmaxf(s3, after));
      if (q == 0) ring_bm[rp >> 4] = ring_bm[(rp >> 4) + NB] = fmaxf(fmaxf(qa, qb), fmaxf(qc, qd));
    }
    __syncthreads();  // (1) maxima and table window visible

    // ---- 240-sample window maximum = tail of block b-15, blocks b-14..b-1, head of block b ----
    // blocks b-14 .. b-1 are 14 consecutive words of the mirrored ring; the four lanes of a quad (one block) read four
    // of them each (the last lane's overlap the third's) and exchange their maxima
    const float *bmp = ring_bm + ((rp >> 4) + NB - 14 + (q < 3 ? 4 * q : 10));
    float w14 = fmaxf(fmaxf(bmp[0], bmp[1]), fmaxf(bmp[2], bmp[3]));
    w14 = fmaxf(w14, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w14), 0xB1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
    w14 = fmaxf(w14, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w14), 0x4E, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
    const int rd = ring_wrap(base + 4 * tv - kDelay);
    const float4 so = *reinterpret_cast<const float4 *>(&ring_suf[rd]);
    float4 pk;
    pk.x = fmaxf(fmaxf(so.x, w14), pre_ex.x);
    pk.y = fmaxf(fmaxf(so.y, w14), pre_ex.y);
    pk.z = fmaxf(fmaxf(so.z, w14), pre_ex.z);
    pk.w = fmaxf(fmaxf(so.w, w14), pre_ex.w);
    *reinterpret_cast<float4 *>(&arr_p[4 * tv]) = pk;
    // ---- limiter gains in rounds, as in render_fast.hpp: hypothesis "no trigger from block bs on" for
    //      every sample not yet settled -> vote -> the chain wave walks the blocks that trigger and one
    //      more -> the next round re-evaluates the rest from the state reached.  win[i] = ctab[min(n_chunk
    //      + i, n_end)] serves round 0 directly, `look` every later one.
    {
      const int n_chunk = n_st;
      auto look = [win, head, n_chunk](int ci) {
        const int d = ci - n_chunk - 1;   // (the window starts one step on: render_fast.hpp)
        return (d >= 0 && d < kW4Win) ? win[d] : head[ci < kW4Win ? ci : kW4Win - 1];
      };
      const int nblk = cnt >> 6;
      int bs = 0;
      while (IAMF_W4_EXP != 3 && IAMF_W4_EXP != 5) {
        int kfirst = kBig;
        const int o0 = 4 * tv - 64 * bs;   // the lane's first sample, counted from the round's start state
        if (o0 >= 0) {
          // hypothesis gains, as in render_fast.hpp: four consecutive words of one staged table (round 0: the window;
          // later: the head from n_st + 1), the formula chosen per round where the whole round is idle or in release
          const int n0 = __builtin_amdgcn_readfirstlane(n_st);
          const int last = n0 + (cnt - 64 * bs) - 1;
          float gh[4] = {1.f, 1.f, 1.f, 1.f};
          if (n0 < n_end) {
            const float *tb = (bs == 0 ? win : head + (n0 + 1)) + o0;
            const float cf[4] = {tb[0], tb[1], tb[2], tb[3]};
            if (n0 >= n_atk && last < n_end) {
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
          const float4 g = make_float4(gh[0], gh[1], gh[2], gh[3]);
          const float4 pq = *reinterpret_cast<const float4 *>(&arr_p[4 * tv]);  // the lane's own maxima, back from LDS
          const float px = pq.x * g.x, py = pq.y * g.y, pz = pq.z * g.z, pw = pq.w * g.w;
          const bool hit = valid && fmaxf(fmaxf(px, py), fmaxf(pz, pw)) > thr;
          if (__ballot(hit) != 0ull && hit) {
            if (pw > thr) kfirst = 4 * tv + 3;
            if (pz > thr) kfirst = 4 * tv + 2;
            if (py > thr) kfirst = 4 * tv + 1;
            if (px > thr) kfirst = 4 * tv + 0;
          }
          *reinterpret_cast<float4 *>(&arr_g[4 * tv]) = g;
          if (4 * tv + 4 == cnt) misc[8] = g.w;  // gain of the chunk's last sample under the hypothesis
        }
        {
          const unsigned long long any = __ballot(kfirst != kBig);
          if (lane == 0) misc[wave] = __int_as_float(any ? __builtin_amdgcn_readlane(kfirst, __builtin_ctzll(any)) : kBig);
        }
        __syncthreads();  // (2) vote, gains and window maxima visible
        int kf = __float_as_int(misc[0]);
        kf = min(kf, __float_as_int(misc[1]));
        kf = min(kf, __float_as_int(misc[2]));
        kf = min(kf, __float_as_int(misc[3]));
        if (kf == kBig) {
          g_cur = misc[8];
          n_st = n_st + (cnt - 64 * bs) < n_end ? n_st + (cnt - 64 * bs) : n_end;
          break;
        }
        const int b0 = kf >> 6;
        if (wave == cw) {
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
        __syncthreads();  // (3) recurrence gains visible
        g_cur = misc[4];
        gs = misc[5];
        ge = misc[6];
        n_st = __float_as_int(misc[7]);
        bs = __float_as_int(misc[9]);
        if (bs >= nblk) break;
      }
    }

    // ---- emit 4 sample-frames per lane: lanes < 196 their own (gains at +240 in this chunk),
    //      tail lanes the previous chunk's (gains at 4t - 784), leaving their own in the slot ----
    if constexpr (DMX || DOWN || MIX) {
      if (c0 + kFChunk < p.total) fetch_window(n_st, tv);  // for the next chunk, ahead of the stores
    } else {
      fetch_window(n_st, tv);
    }
    if (c0 + 2 * kFChunk < p.total) fetch_rec(c0 + 2 * kFChunk, tv);  // written to LDS in the next chunk
    if constexpr (DMX) __builtin_amdgcn_sched_barrier(0);  // keep both fetches ahead of the stores
    float4 gq = *reinterpret_cast<const float4 *>(&arr_g[is_tail ? 4 * tv - (kFChunk - kDelay) : 4 * tv + kDelay]);
#if IAMF_W4_EXP == 3 || IAMF_W4_EXP == 5
    gq = make_float4(1.f, 1.f, 1.f, 1.f);
#endif
    const float gs4[4] = {gq.x * 32768.f, gq.y * 32768.f, gq.z * 32768.f, gq.w * 32768.f};  // exact scaling
    if (c0 + kFChunk >= p.total) {  // wave-uniform
      if (valid && 4 * tv >= cnt - kSave) {
        // last chunk of the call: its last 256 rendered samples are the stream state the next call
        // (any kernel) starts from.  Written here so that y is dead once it has been packed below.
        float *sy = p.ring_y + (int64_t)s * C * kSave;
#pragma unroll
        for (int c = 0; c < C; ++c) *reinterpret_cast<float4 *>(&sy[c * kSave + 4 * tv - (cnt - kSave)]) = y[c];
      }
      // These stores sit in a conditional block; left pending they would make the loop-top wait for the
      // prefetched input a vmcnt(0) on EVERY chunk (see the note at the PCM stores).  Drained here, once per
      // call, this path reaches the loop header with nothing in flight and the steady path keeps its count.
      if constexpr (!(DMX || DOWN || MIX)) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (is_tail) {  // swap in place: y <- previous tail, slot <- this chunk's samples
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float4 *sl = &tail[c * kW4TailLanes + tl];
        const float4 old = *sl;
        *sl = y[c];
        y[c] = old;
      }
    }
#if IAMF_W4_EXP == 2
    {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) acc += y[c].x * gs4[0] + y[c].y * gs4[1] + y[c].z * gs4[2] + y[c].w * gs4[3];
      if (acc == 123.456f && p.total < 0) pcm[tv] = 1;
    }
#else
    {
      // Pack to s16 (rint then saturate == the reference's clamp then lrintf: the bounds are
      // integers).  Piece k of a lane = dwords 4k..4k+3 of its 8*C bytes; dword d = sample
      // d / (C/2), channels 2*(d % (C/2)) and +1.
      constexpr int H2 = C / 2, S = wide4_stage_stride(C), LR = wide4_stage_lanes(C, M, kExtra);
      uint32_t od[4 * H2];  // filled channel pair by channel pair so that y dies as od is born
#pragma unroll
      for (int c = 0; c < C; c += 2) {
        const float4 ya = y[c], yb = y[c + 1];
        od[0 * H2 + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.x * gs4[0]), (int)rintf(yb.x * gs4[0])));
        od[1 * H2 + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.y * gs4[1]), (int)rintf(yb.y * gs4[1])));
        od[2 * H2 + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.z * gs4[2]), (int)rintf(yb.z * gs4[2])));
        od[3 * H2 + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.w * gs4[3]), (int)rintf(yb.w * gs4[3])));
      }
      __builtin_amdgcn_sched_barrier(0);
      // Through the wave's staging area so that every store instruction writes one contiguous run:
      // in output order the chunk is [tail lanes 196..255 | lanes 0..195], so the lanes of a round
      // cover at most two contiguous stretches of the PCM stream.
#if IAMF_W4_EXP == 8
      // experiment: no staging — every lane stores its own 8*C contiguous bytes, piece by piece (each store
      // instruction then writes 64 pieces 8*C bytes apart, left to the L2 to merge)
      if constexpr (!(DMX || DOWN || MIX)) {
        const bool ts8 = tv >= kW4FirstTail;
        const int rel8 = c0 + 4 * tv - (ts8 ? kFChunk : 0);
        const bool emit8 = ts8 ? rel8 + pos_small >= 0 : 4 * tv < cnt - kDelay;
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
#pragma unroll
        for (int k = 0; k < H2; ++k) {
          uint8_t *to = emit8 ? pcm + (uint32_t)((rel8 + lead) * (C * 2) + k * 16) : p.dump + (uint32_t)((s * 256 + tv) * 16);
          __builtin_nontemporal_store(u4{od[4 * k], od[4 * k + 1], od[4 * k + 2], od[4 * k + 3]}, reinterpret_cast<u4 *>(to));
        }
      } else
#endif
      {
      uint4 *stg = stage + wave * (LR * S);
      const int lane_e = lane;
#pragma unroll
      for (int r = 0; r < 64 / LR; ++r) {
        if (LR == 64 || (lane_e >> 5) == r) {
#pragma unroll
          for (int k = 0; k < H2; ++k)
            stg[(lane_e & (LR - 1)) * S + k] = make_uint4(od[4 * k], od[4 * k + 1], od[4 * k + 2], od[4 * k + 3]);
        }
        // other lanes of this wave read what was just written: LDS executes a wave's accesses in
        // order, the fence keeps the compiler from reordering them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < (LR * H2 + 63) / 64; ++i) {
          const int j = i * 64 + lane_e;
          if ((LR * H2) % 64 == 0 || j < LR * H2) {
            const int l2 = j / H2, k = j - l2 * H2;
            const uint4 v = stg[l2 * S + k];
            const int t2 = wave * 64 + r * LR + l2;
            // tail-slot lanes emit the previous chunk's last 240 samples; the others their own, if the
            // gain 240 samples on exists in this chunk (a short last chunk leaves the rest to the
            // persisted state)
            const bool ts = t2 >= kW4FirstTail;
            const int rel = c0 + 4 * t2 - (ts ? kFChunk : 0);  // sample, relative to pos0
#if IAMF_W4_EXP == 6
            const bool emit = false;  // every store goes to the lane's dump slot: instruction issue without HBM writes
#else
            const bool emit = ts ? rel + pos_small >= 0 : 4 * t2 < cnt - kDelay;
#endif
            using u4 = __attribute__((ext_vector_type(4))) unsigned;
            if constexpr (DMX || DOWN || MIX) {
              if (emit)  // written once, never read back by this GPU: streaming store
                __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w},
                                            reinterpret_cast<u4 *>(pcm + (uint32_t)((rel + lead) * (C * 2) + k * 16)));
            } else {
              // EVERY lane stores in EVERY chunk: a lane with nothing to emit (no previous tail yet at the start
              // of a stream, the end of a short last chunk) sends its 16 bytes to its slot of the dump buffer.
              // With the loads above that makes the chunk's vector-memory instructions the same on every path,
              // which is what lets the compiler wait for the prefetched input with a COUNTED s_waitcnt
              // vmcnt(5 + stores) at the top of the loop instead of vmcnt(0): vector-memory operations retire in
              // order, and vmcnt(0) there drained this chunk's PCM stores — a store round trip (~2 us) exposed
              // on every chunk, 25-35 % of cfg2 / cfg3 (tools/debug/w4_exp.sh).
              // (the dump address is rebuilt from the per-chunk thread index: two registers that do not stay live)
              uint8_t *to = emit ? pcm + (uint32_t)((rel + lead) * (C * 2) + k * 16)
                                 : p.dump + (uint32_t)((s * 256 + tv) * 16);
#if IAMF_W4_EXP == 1
              if (v.x == 0x12345678u && p.total < 0)
#endif
#if IAMF_W4_EXP == 7
              *reinterpret_cast<u4 *>(to) = u4{v.x, v.y, v.z, v.w};   // plain instead of non-temporal
#else
              __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4 *>(to));
#endif
              __builtin_amdgcn_sched_barrier(0);  // piece by piece: without the branches the scheduler would
                                                  // hoist a round's LDS reads above its stores (24 registers)
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round overwrites the staging area
        __builtin_amdgcn_wave_barrier();
      }
      }
    }
#endif
    base = base + cnt >= R ? base + cnt - R : base + cnt;
    // no barrier here: the next chunk writes ring_* / win before its barrier (1), whose readers
    // all finished before barrier (2)/(3) of this chunk; arr_* / misc are written after (1)
  }

  // ---- persist the rest of the stream state (same format as the generic kernel) ----
  {
    float *spm = p.ring_pm + (int64_t)s * kSave;
    const int rp = ring_wrap(base - kSave + t);  // base = ring position of sample pos0 + total
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
