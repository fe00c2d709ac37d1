// render_wide4.hpp — multi-channel output layouts (even channel counts 4..24), 16-bit PCM, limiter
// on, calls of whole 1024-sample chunks.  One workgroup (4 waves) per stream, FOUR consecutive
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

constexpr int kW4Win = 1088;        // staged table window / head length (> chunk + 1), multiple of 64
constexpr int kW4TailLanes = 60;    // lanes holding the chunk's last 240 samples
constexpr int kW4FirstTail = 256 - kW4TailLanes;

__host__ __device__ constexpr int wide4_lds_floats(int c, int m) {
  return 4 * kW4TailLanes * c + 2 * kFRing + kFRing / 16 + 2 * kFChunk + 2 * kW4Win + ((c + 3) & ~3) * m + 16;
}

template <int M, int C>
__global__ __launch_bounds__(256, 2) void render_wide4_kernel(const RenderParams p) {
  static_assert((C & 1) == 0 && C >= 4 && C <= 24, "even channel counts");
  extern __shared__ float lds[];
  constexpr int R = kFRing;
  constexpr int NB = R / 16;
  constexpr int C4 = (C + 3) & ~3;
  float4 *tail = reinterpret_cast<float4 *>(lds);   // [C][60]  lane-private: previous chunk's last 240 samples
  float *ring_pm = lds + 4 * kW4TailLanes * C;      // [R]     max |y| over channels
  float *ring_suf = ring_pm + R;                    // [R]     suffix maxima inside aligned 16-blocks
  float *ring_bm = ring_suf + R;                    // [R/16]  maxima of aligned 16-blocks
  float *arr_p = ring_bm + NB;                      // [1024]  window maxima of the chunk
  float *arr_g = arr_p + kFChunk;                   // [1024]  limiter gains of the chunk
  float *win = arr_g + kFChunk;                     // [kW4Win] ctab[min(n_st + i, n_end)]
  float *head = win + kW4Win;                       // [kW4Win] ctab[i]
  float *mat = head + kW4Win;                       // [M][C4] weights, input-major
  float *misc = mat + C4 * M;                       // [16]

  const int s = blockIdx.x;
  const int t = threadIdx.x;
  const int wave = t >> 6;
  const int lane = t & 63;
  const int q = t & 3;
  const int fs = p.frame_size;
  const float thr = p.thr;
  const int n_atk = p.n_atk, n_end = p.n_end;
  const bool is_tail = t >= kW4FirstTail;
  const int tl = is_tail ? t - kW4FirstTail : 0;
  int base = (int)(p.pos0 % R);  // ring position of the chunk's first sample; multiple of 16

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
    if ((t & 15) == 0) ring_bm[rp >> 4] = sfx;
    for (int i = t; i < kW4Win; i += 256) head[i] = p.ctab[i < n_end ? i : n_end];
    for (int i = t; i < C4 * M; i += 256) {
      const int m = i / C4, c = i - m * C4;
      const int f = c < C ? p.src_feed[c] : -1;
      mat[i] = f >= 0 ? p.matrix[f * M + m] : 0.f;
    }
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

  const int64_t out_base = p.pos0 > kDelay ? p.pos0 - kDelay : 0;
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;

  float4 x[M];
  {
    const int k = 4 * t;  // total >= 1024
    const int f = k / fs;
    const int i = k - f * fs;
    const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = ld_stream4(src + (int64_t)m * fs);
  }
  __syncthreads();

  float4 y[C];
  for (int c0 = 0; c0 < p.total; c0 += kFChunk) {
    const int k = c0 + 4 * t;
    const int64_t gk = p.pos0 + k;
    const int rp = ring_wrap(base + 4 * t);

    // table window this chunk can reach without a trigger (older than the prefetch in the in-order
    // vmcnt queue, so waiting for it does not drain the prefetch)
    float wv[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int i = n_st + t + 256 * r;
      wv[r] = 1.0f;
      if (n_st < n_end && t + 256 * r < kW4Win) wv[r] = p.ctab[i < n_end ? i : n_end];
    }

    // ---- element renderer + gains (reference operation order), 4 slots x 4 samples at a time ----
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.dbg & 2) {
#pragma unroll
      for (int c = 0; c < C; ++c) { y[c] = x[c % M]; pm.x = fmaxf(pm.x, fabsf(y[c].x)); pm.y = fmaxf(pm.y, fabsf(y[c].y)); pm.z = fmaxf(pm.z, fabsf(y[c].z)); pm.w = fmaxf(pm.w, fabsf(y[c].w)); }
    } else
#pragma unroll
    for (int cb = 0; cb < C4; cb += 4) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float4 w = *reinterpret_cast<const float4 *>(&mat[m * C4 + cb]);
        a0.x = a0.x + w.x * x[m].x; a0.y = a0.y + w.x * x[m].y; a0.z = a0.z + w.x * x[m].z; a0.w = a0.w + w.x * x[m].w;
        a1.x = a1.x + w.y * x[m].x; a1.y = a1.y + w.y * x[m].y; a1.z = a1.z + w.y * x[m].z; a1.w = a1.w + w.y * x[m].w;
        if (cb + 2 < C) {
          a2.x = a2.x + w.z * x[m].x; a2.y = a2.y + w.z * x[m].y; a2.z = a2.z + w.z * x[m].z; a2.w = a2.w + w.z * x[m].w;
          a3.x = a3.x + w.w * x[m].x; a3.y = a3.y + w.w * x[m].y; a3.z = a3.z + w.w * x[m].z; a3.w = a3.w + w.w * x[m].w;
        }
      }
      const float4 acc[4] = {a0, a1, a2, a3};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (cb + i < C) {
          float4 v = acc[i];
          if (any_gain) {
            v.x = ((v.x * m_eg) * m_og) * m_lg;
            v.y = ((v.y * m_eg) * m_og) * m_lg;
            v.z = ((v.z * m_eg) * m_og) * m_lg;
            v.w = ((v.w * m_eg) * m_og) * m_lg;
          }
          y[cb + i] = v;
          pm.x = fmaxf(pm.x, fabsf(v.x));
          pm.y = fmaxf(pm.y, fabsf(v.y));
          pm.z = fmaxf(pm.z, fabsf(v.z));
          pm.w = fmaxf(pm.w, fabsf(v.w));
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the weight loads of later groups from piling up in registers
    }

    // ---- prefetch the next chunk's input ----
    {
      const int kn = k + kFChunk;
      if (kn < p.total) {
        const int f = kn / fs;
        const int i = kn - f * fs;
        const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = ld_stream4(src + (int64_t)m * fs);
      }
    }

    // ---- per-16 prefix / suffix / block maxima: 4 lanes x 4 samples = one aligned block ----
    const float i0 = pm.x, i1 = fmaxf(i0, pm.y), i2 = fmaxf(i1, pm.z), i3 = fmaxf(i2, pm.w);
    const float s3 = pm.w, s2 = fmaxf(pm.z, s3), s1 = fmaxf(pm.y, s2), s0 = fmaxf(pm.x, s1);
    const float qa = dpp_quad_bcast0(i3), qb = dpp_quad_bcast1(i3), qc = dpp_quad_bcast2(i3),
                qd = dpp_quad_bcast3(i3);
    const float before = q == 0 ? 0.f : (q == 1 ? qa : (q == 2 ? fmaxf(qa, qb) : fmaxf(fmaxf(qa, qb), qc)));
    const float after = q == 3 ? 0.f : (q == 2 ? qd : (q == 1 ? fmaxf(qc, qd) : fmaxf(fmaxf(qb, qc), qd)));
    const float4 pre_ex = make_float4(before, fmaxf(before, i0), fmaxf(before, i1), fmaxf(before, i2));
    *reinterpret_cast<float4 *>(&ring_pm[rp]) = pm;
    *reinterpret_cast<float4 *>(&ring_suf[rp]) =
        make_float4(fmaxf(s0, after), fmaxf(s1, after), fmaxf(s2, after), fmaxf(s3, after));
    if (q == 0) ring_bm[rp >> 4] = fmaxf(fmaxf(qa, qb), fmaxf(qc, qd));
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (t + 256 * r < kW4Win) win[t + 256 * r] = wv[r];
    __syncthreads();  // (1) maxima and table window visible

    // ---- 240-sample window maximum = tail of block b-15, blocks b-14..b-1, head of block b ----
    const int bpos = rp >> 4;
    float w14 = 0.f;
#pragma unroll
    for (int j = 1; j <= 14; ++j) {
      int bi = bpos - j;
      bi = bi < 0 ? bi + NB : bi;
      w14 = fmaxf(w14, ring_bm[bi]);
    }
    const int rd = ring_wrap(base + 4 * t - kDelay);
    const float4 so = *reinterpret_cast<const float4 *>(&ring_suf[rd]);
    float4 pk;
    pk.x = fmaxf(fmaxf(so.x, w14), pre_ex.x);
    pk.y = fmaxf(fmaxf(so.y, w14), pre_ex.y);
    pk.z = fmaxf(fmaxf(so.z, w14), pre_ex.z);
    pk.w = fmaxf(fmaxf(so.w, w14), pre_ex.w);
    // ---- gains under the hypothesis "no trigger in this chunk": win[i] = ctab[min(n_st+i, n_end)] ----
    float gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int np = n_st + 4 * t + j;
      np = np < n_end ? np : n_end;
      gh[j] = gain_at(np, gs, ge, win[4 * t + j + 1], n_atk, n_end);
    }
    const float4 g = make_float4(gh[0], gh[1], gh[2], gh[3]);
    int kfirst = kBig;
    if (pk.w * g.w > thr) kfirst = 4 * t + 3;
    if (pk.z * g.z > thr) kfirst = 4 * t + 2;
    if (pk.y * g.y > thr) kfirst = 4 * t + 1;
    if (pk.x * g.x > thr) kfirst = 4 * t + 0;
    *reinterpret_cast<float4 *>(&arr_p[4 * t]) = pk;
    *reinterpret_cast<float4 *>(&arr_g[4 * t]) = g;
    {
      const unsigned long long any = __ballot(kfirst != kBig);
      if (lane == 0) misc[wave] = __int_as_float(any ? __builtin_amdgcn_readlane(kfirst, __builtin_ctzll(any)) : kBig);
      if (t == 255) misc[8] = g.w;  // gain of the chunk's last sample under the hypothesis
    }
    __syncthreads();  // (2) vote, gains and window maxima visible
    int kf = __float_as_int(misc[0]);
    kf = min(kf, __float_as_int(misc[1]));
    kf = min(kf, __float_as_int(misc[2]));
    kf = min(kf, __float_as_int(misc[3]));
    if (kf == kBig) {
      g_cur = misc[8];
      n_st = n_st + kFChunk < n_end ? n_st + kFChunk : n_end;
    } else {
      const int b0 = kf >> 6;
      if (wave == 0) {
        const int n_chunk = n_st;
        int ln = n_st + 64 * b0 < n_end ? n_st + 64 * b0 : n_end;
        float lgs = gs, lge = ge, lgl = g_cur;
        auto look = [win, head, n_chunk](int ci) {
          const int d = ci - n_chunk;
          return (d >= 0 && d < kW4Win) ? win[d] : head[ci < kW4Win ? ci : kW4Win - 1];
        };
        limiter_wave(arr_p, arr_g, look, b0, kFChunk >> 6, ln, lgs, lge, lgl, thr, n_atk, n_end);
        if (lane == 0) {
          misc[4] = lgl;
          misc[5] = lgs;
          misc[6] = lge;
          misc[7] = __int_as_float(ln);
        }
      }
      __syncthreads();  // (3) recurrence gains visible
      g_cur = misc[4];
      gs = misc[5];
      ge = misc[6];
      n_st = __float_as_int(misc[7]);
    }

    // ---- emit 4 sample-frames per lane: lanes < 196 their own (gains at +240 in this chunk),
    //      tail lanes the previous chunk's (gains at 4t - 784), leaving their own in the slot ----
    float4 gq = *reinterpret_cast<const float4 *>(&arr_g[is_tail ? 4 * t - (kFChunk - kDelay) : 4 * t + kDelay]);
    gq = make_float4(gq.x * 32768.f, gq.y * 32768.f, gq.z * 32768.f, gq.w * 32768.f);  // exact scaling
    const int64_t j0 = is_tail ? gk - kFChunk : gk;
    uint32_t od[2 * C];
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      float4 ya = y[c], yb = y[c + 1];
      if (is_tail) {
        float4 *sa = &tail[c * kW4TailLanes + tl], *sb = &tail[(c + 1) * kW4TailLanes + tl];
        const float4 ta = *sa, tb = *sb;
        *sa = ya;
        *sb = yb;
        ya = ta;
        yb = tb;
      }
      // rint then saturate == the reference's clamp then lrintf (the bounds are integers)
      od[0 * (C / 2) + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.x * gq.x), (int)rintf(yb.x * gq.x)));
      od[1 * (C / 2) + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.y * gq.y), (int)rintf(yb.y * gq.y)));
      od[2 * (C / 2) + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.z * gq.z), (int)rintf(yb.z * gq.z)));
      od[3 * (C / 2) + c / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16((int)rintf(ya.w * gq.w), (int)rintf(yb.w * gq.w)));
    }
    if (j0 >= 0 && !((p.dbg & 1) && od[0] != 0x12345u)) {
      uint4 *dst = reinterpret_cast<uint4 *>(pcm + (j0 - out_base) * (int64_t)C * 2);
#pragma unroll
      for (int i = 0; i < C / 2; ++i) dst[i] = make_uint4(od[4 * i], od[4 * i + 1], od[4 * i + 2], od[4 * i + 3]);
    }
    base = base + kFChunk >= R ? base + kFChunk - R : base + kFChunk;
    // no barrier here: the next chunk writes ring_* / win before its barrier (1), whose readers
    // all finished before barrier (2)/(3) of this chunk; arr_* / misc are written after (1)
  }

  // ---- persist stream state (same format as the generic kernel): y of the last 256 samples sits
  //      in the registers of lanes 192..255 ----
  {
    float *sy = p.ring_y + (int64_t)s * C * kSave;
    float *spm = p.ring_pm + (int64_t)s * kSave;
    if (t >= 192) {
#pragma unroll
      for (int c = 0; c < C; ++c) *reinterpret_cast<float4 *>(&sy[c * kSave + 4 * (t - 192)]) = y[c];
    }
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
