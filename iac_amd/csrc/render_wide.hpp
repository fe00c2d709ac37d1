// render_wide.hpp — multi-channel output layouts (3..24 channels) with the limiter on, aligned
// calls.  One workgroup (4 waves) per stream, 256-sample chunks; everything after the
// projection runs lane = sample.
//   * rendered samples live in an LDS ring laid out exactly like the interleaved output
//     ([position][channel], no padding), so the emit stage is a linear copy: 32-byte LDS reads,
//     gain, round, 16-byte coalesced global stores;
//   * projection, two variants:
//       MFMA = false : VALU in the reference's operation order (bit-exact), four output slots at
//                      a time, weights fetched as one 16-byte LDS broadcast per input channel;
//       MFMA = true  : v_mfma_f32_32x32x2_f32, D[slot][sample] = W[slot][in] * X[in][sample]; the
//                      weights sit in KS registers for the whole call, the planar input is loaded
//                      straight in the B-operand layout (2 channel rows x 128 B per instruction),
//                      each lane ends up with 4-slot groups it writes to the ring as 16-byte
//                      stores.  Exact f32 products accumulated as a k-ordered fma chain: differs
//                      from the reference's separate mul/add roundings by <= 1 ulp per term
//                      (within +-1 LSB of the PCM, see tests/test_gpu_mfma.py);
//   * the next chunk's input loads are issued right after the projection;
//   * 240-sample sliding maximum from DPP row scans (a 16-lane row = one aligned 16-block);
//   * limiter gains as in render_fast.hpp: no-trigger hypothesis for the whole chunk, otherwise
//     wave 0 re-runs the recurrence (speculation + DPP trigger-run chain).  The curve table is too
//     big to sit in LDS next to a 24-channel ring, so each chunk stages the 320-entry window it
//     can reach without a trigger plus the table head that follows a trigger.
#pragma once

constexpr int kWChunk = 256;
constexpr int kWPos = 512;   // ring positions (power of two >= chunk + look-ahead + 15)
constexpr int kWWin = 320;   // staged table window / head length (> chunk + 1)

using f32x16 = __attribute__((ext_vector_type(16))) float;

__host__ __device__ constexpr int wide_lds_floats(int c, int m) {
  return kWPos * c + 2 * kWPos + kWPos / 16 + 3 * kWChunk + 2 * kWWin + ((c + 3) & ~3) * m + 16;
}

// DPP helpers inside a 16-lane row; lanes shifted in from outside the row read 0
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {  // lane i <- lane i+N
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) {  // lane i <- lane i-N
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true));
}

template <int M, bool MFMA>
__global__ __launch_bounds__(256) void render_wide_kernel(const RenderParams p) {
  extern __shared__ float lds[];
  constexpr int KS = (M + 1) / 2;  // k-steps of the 32x32x2 MFMA
  const int C = p.out_ch;
  const int C4 = (C + 3) & ~3;
  float *ring = lds;                      // [kWPos][C]  rendered samples, interleaved like the output
  float *ring_pm = ring + kWPos * C;      // [kWPos]     max |y| over channels
  float *ring_suf = ring_pm + kWPos;      // [kWPos]     suffix maxima of pm inside aligned 16-blocks
  float *ring_bm = ring_suf + kWPos;      // [kWPos/16]  maxima of aligned 16-blocks
  float *arr_p = ring_bm + kWPos / 16;    // [256]
  float *arr_e = arr_p + kWChunk;         // [256]
  float *arr_g = arr_e + kWChunk;         // [256]
  float *win = arr_g + kWChunk;           // [kWWin]     ctab[n_st + i] for this chunk
  float *head = win + kWWin;              // [kWWin]     ctab[i]
  float *mat = head + kWWin;              // [M][C4]     weights, input-major (VALU variant)
  float *misc = mat + C4 * M;             // [16]

  const int s = blockIdx.x + p.stream0;   // a launch covers streams [stream0, stream0 + n_launch) of the batch
  const int t = threadIdx.x;
  const int wave = t >> 6;
  const int lane = t & 63;
  const int fs = p.frame_size;
  const float thr = p.thr;
  const int n_atk = p.n_atk, n_end = p.n_end;

  // ---- stream state and constants -> LDS ----
  {
    const float *sy = p.ring_y + (int64_t)s * C * kSave;
    const float *spm = p.ring_pm + (int64_t)s * kSave;
    const int rp = (int)((p.pos0 - kSave + t) & (kWPos - 1));  // pos0 % 16 == 0
    for (int c = 0; c < C; ++c) ring[rp * C + c] = sy[c * kSave + t];
    const float pm = spm[t];
    ring_pm[rp] = pm;
    float sfx = pm;
    sfx = fmaxf(sfx, dpp_row_shl<1>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<2>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<4>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<8>(sfx));
    ring_suf[rp] = sfx;
    if ((t & 15) == 0) ring_bm[rp >> 4] = sfx;
    for (int i = t; i < kWWin; i += 256) head[i] = p.ctab[i < n_end ? i : n_end];
    if (!MFMA) {
      for (int i = t; i < C4 * M; i += 256) {
        const int m = i / C4, c = i - m * C4;
        const int f = c < C ? p.src_feed[c] : -1;
        mat[i] = f >= 0 ? p.matrix[f * M + m] : 0.f;
      }
    }
    chain_wave_publish(misc + 12);
  }
  // MFMA A operand: lane l holds W[slot = l & 31][in = 2*ks + (l >> 5)]
  float aw[KS];
  if (MFMA) {
    const int slot = lane & 31;
    const int f = slot < C ? p.src_feed[slot] : -1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int m = 2 * ks + (lane >> 5);
      aw[ks] = (f >= 0 && m < M) ? p.matrix[f * M + m] : 0.f;
    }
  }
  LimState ls = p.lim[s];
  float g_cur = ls.g, gs = ls.gs, ge = ls.ge;
  int n_st = ls.n;
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);
  // a gain the reference would skip is a multiplication by exactly 1 (x * 1.0f == x); the
  // reference's  0 + y  of the mixer only turns -0 into +0, which no output format can tell apart
  const float m_eg = eg_on ? eg : 1.f, m_og = og_on ? og : 1.f, m_lg = lg_on ? lg : 1.f;
  const bool any_gain = eg_on || og_on || lg_on;

  const int64_t out_base = p.pos0 > kDelay ? p.pos0 - kDelay : 0;
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;
  // flat element index -> sample: srel = (f * inv_c) >> 20 is exact for f < 256*24 and c <= 24
  const uint32_t inv_c = ((1u << 20) + C - 1) / C;

  // input registers: VALU variant x[m] = channel m of this lane's sample;
  // MFMA variant x[tt*KS + ks] = B operand of tile tt, k-step ks
  constexpr int NX = MFMA ? 2 * KS : M;
  float x[NX];
  auto load_chunk = [&](int cbase) {
    if (MFMA) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int k = cbase + 64 * wave + 32 * tt + (lane & 31);
        const int f = k / fs;
        const int i = k - f * fs;
        const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int m = 2 * ks + (lane >> 5);
          x[tt * KS + ks] = (k < p.total && m < M) ? src[(int64_t)m * fs] : 0.f;
        }
      }
    } else {
      const int k = cbase + t;
      if (k < p.total) {
        const int f = k / fs;
        const int i = k - f * fs;
        const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = src[(int64_t)m * fs];
      } else {
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = 0.f;
      }
    }
  };
  load_chunk(0);
  __syncthreads();
  const int cw = chain_wave_pick(misc + 12);

  for (int c0 = 0; c0 < p.total; c0 += kWChunk) {
    const int cnt = p.total - c0 < kWChunk ? p.total - c0 : kWChunk;  // multiple of 64
    const int k = c0 + t;
    const bool valid = t < cnt;
    const int64_t gk = p.pos0 + k;
    const int rp = (int)(gk & (kWPos - 1));

    // table window this chunk can reach without a trigger (older than the prefetch in the
    // in-order vmcnt queue, so waiting for it does not drain the prefetch)
    float wv0 = 1.0f, wv1 = 1.0f;
    if (n_st < n_end) {
      const int i0 = n_st + t, i1 = n_st + t + 256;
      wv0 = p.ctab[i0 < n_end ? i0 : n_end];
      if (t < kWWin - 256) wv1 = p.ctab[i1 < n_end ? i1 : n_end];
    }

    // ---- element renderer + gains -> ring; pm = max |y| of this lane's sample ----
    float pm = 0.f;
    if (MFMA) {
      float pmt[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ks], x[tt * KS + ks], acc, 0, 0, 0);
        // lane l: column = sample 32*tt + (l & 31); register r: slot (r&3) + 8*(r>>2) + 4*(l>>5)
        const int ksmp = 64 * wave + 32 * tt + (lane & 31);
        const int pos = (int)((p.pos0 + c0 + ksmp) & (kWPos - 1));
        if (any_gain) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] * m_eg) * m_og) * m_lg;
        }
        float pmv = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int slot0 = 8 * gq + 4 * (lane >> 5);
          const float y0 = acc[4 * gq], y1 = acc[4 * gq + 1], y2 = acc[4 * gq + 2], y3 = acc[4 * gq + 3];
          if ((C & 3) == 0) {
            if (slot0 < C) {
              pmv = fmaxf(fmaxf(pmv, fabsf(y0)), fmaxf(fabsf(y1), fmaxf(fabsf(y2), fabsf(y3))));
              if (ksmp < cnt) *reinterpret_cast<float4 *>(&ring[pos * C + slot0]) = make_float4(y0, y1, y2, y3);
            }
          } else {
            const float y[4] = {y0, y1, y2, y3};
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (slot0 + i < C) {
                pmv = fmaxf(pmv, fabsf(y[i]));
                if (ksmp < cnt) ring[pos * C + slot0 + i] = y[i];
              }
          }
        }
        pmt[tt] = fmaxf(pmv, __shfl_xor(pmv, 32));  // the sample's other 12 slots live in lane ^ 32
      }
      pm = lane < 32 ? pmt[0] : pmt[1];
    } else {
      for (int cb = 0; cb < C4; cb += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float4 w = *reinterpret_cast<const float4 *>(&mat[m * C4 + cb]);
          a0 = a0 + w.x * x[m];
          a1 = a1 + w.y * x[m];
          a2 = a2 + w.z * x[m];
          a3 = a3 + w.w * x[m];
        }
        float y[4] = {a0, a1, a2, a3};
        if (any_gain) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = ((y[i] * m_eg) * m_og) * m_lg;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (cb + i < C) pm = fmaxf(pm, fabsf(y[i]));
        if (valid) {
          if ((C & 3) == 0) {
            *reinterpret_cast<float4 *>(&ring[rp * C + cb]) = make_float4(y[0], y[1], y[2], y[3]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (cb + i < C) ring[rp * C + cb + i] = y[i];
          }
        }
      }
    }

    // ---- table window -> LDS first (waiting for it after the prefetch had been issued would drain the
    //      prefetch: vector-memory operations retire in order), then the next chunk's input ----
    win[t] = wv0;
    if (t < kWWin - 256) win[t + 256] = wv1;
    if (c0 + kWChunk < p.total) load_chunk(c0 + kWChunk);

    // ---- per-16 suffix / exclusive prefix / block maxima by DPP row scans ----
    float sfx = pm;
    sfx = fmaxf(sfx, dpp_row_shl<1>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<2>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<4>(sfx));
    sfx = fmaxf(sfx, dpp_row_shl<8>(sfx));
    float pre = dpp_row_shr<1>(pm);
    pre = fmaxf(pre, dpp_row_shr<1>(pre));
    pre = fmaxf(pre, dpp_row_shr<2>(pre));
    pre = fmaxf(pre, dpp_row_shr<4>(pre));
    pre = fmaxf(pre, dpp_row_shr<8>(pre));
    if (valid) {
      ring_pm[rp] = pm;
      ring_suf[rp] = sfx;
      if ((t & 15) == 0) ring_bm[rp >> 4] = sfx;
    }
    __syncthreads();

    // ---- 240-sample window maximum = tail of block b-15, blocks b-14..b-1, head of block b ----
    const int bpos = rp >> 4;
    float w14 = 0.f;
#pragma unroll
    for (int j = 1; j <= 14; ++j) w14 = fmaxf(w14, ring_bm[(bpos - j) & (kWPos / 16 - 1)]);
    const float pk = fmaxf(fmaxf(ring_suf[(int)((gk - kDelay) & (kWPos - 1))], w14), pre);
    // ---- gains under the no-trigger hypothesis ----
    int n_pre = n_st + t;
    n_pre = n_pre < n_end ? n_pre : n_end;
    float g = gain_at(n_pre, gs, ge, win[t + 1], n_atk, n_end);
    const bool trig = valid && (pk * g > thr);
    arr_p[t] = pk;
    {
      const unsigned long long any = __ballot(trig);
      if (lane == 0) misc[wave] = __int_as_float(any ? 64 * wave + (int)__builtin_ctzll(any) : kBig);
      if (t + 1 == cnt) misc[8] = g;
    }
    __syncthreads();
    int kf = __float_as_int(misc[0]);
    kf = min(kf, __float_as_int(misc[1]));
    kf = min(kf, __float_as_int(misc[2]));
    kf = min(kf, __float_as_int(misc[3]));
    if (kf == kBig) {
      arr_g[t] = g * 32768.f;  // exact power-of-two scaling, folded into the emit multiply
      g_cur = misc[8];
      n_st = n_st + cnt < n_end ? n_st + cnt : n_end;
    } else {
      const int b0 = kf >> 6;
      if (wave == cw) {
        const int n_chunk = n_st;
        int ln = n_st + 64 * b0 < n_end ? n_st + 64 * b0 : n_end;
        float lgs = gs, lge = ge, lgl = g_cur;
        auto look = [win, head, n_chunk](int ci) {
          const int d = ci - n_chunk;
          return (d >= 0 && d < kWWin) ? win[d] : head[ci < kWWin ? ci : kWWin - 1];
        };
        limiter_wave(arr_p, arr_g, look, b0, cnt >> 6, ln, lgs, lge, lgl, thr, n_atk, n_end);
        if (lane == 0) {
          misc[4] = lgl;
          misc[5] = lgs;
          misc[6] = lge;
          misc[7] = __int_as_float(ln);
        }
      }
      __syncthreads();
      arr_g[t] = (t < 64 * b0 ? g : arr_g[t]) * 32768.f;
      g_cur = misc[4];
      gs = misc[5];
      ge = misc[6];
      n_st = __float_as_int(misc[7]);
    }
    __syncthreads();  // arr_g (gain * 2^15) complete for the emit stage

    // ---- emit: the chunk's 256 delayed samples are one contiguous run of the ring ----
    const int64_t jc = p.pos0 + c0 - kDelay;  // first emitted sample of the chunk (may be < 0)
    const int epos = (int)(jc & (kWPos - 1));
    if (p.out_format == IAMF_HIP_FMT_S16) {
      const int np = (cnt * C) >> 3;  // 8-element pieces (cnt % 64 == 0)
      const int ring_flat = kWPos * C;
      for (int q = t; q < np; q += 256) {
        const int f0 = 8 * q;
        int srel = (int)(((uint32_t)f0 * inv_c) >> 20);
        int r = f0 - srel * C;
        if (jc + srel < 0) continue;  // withheld look-ahead samples; 240*C is a multiple of 8
        int rf = epos * C + f0;
        rf = rf >= ring_flat ? rf - ring_flat : rf;
        const float4 v0 = *reinterpret_cast<const float4 *>(&ring[rf]);
        const float4 v1 = *reinterpret_cast<const float4 *>(&ring[rf + 4]);
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        int o[8];
        // rint then saturate == the reference's clamp then lrintf (the bounds are integers)
        if (C >= 8) {  // a piece of 8 spans at most two sample-frames
          const float g0 = arr_g[srel];
          const float g1 = arr_g[srel + 1 < kWChunk ? srel + 1 : kWChunk - 1];
          const int cross = C - r;  // elements of the piece that belong to the first sample-frame
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (int)rintf(vv[i] * (i < cross ? g0 : g1));
        } else {
          float gq = arr_g[srel];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            o[i] = (int)rintf(vv[i] * gq);
            ++r;
            if (r == C) {  // next sample-frame: next gain
              r = 0;
              ++srel;
              gq = arr_g[srel < kWChunk ? srel : kWChunk - 1];
            }
          }
        }
        uint4 w;
        w.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(o[0], o[1]));
        w.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(o[2], o[3]));
        w.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(o[4], o[5]));
        w.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(o[6], o[7]));
        *reinterpret_cast<uint4 *>(pcm + ((jc - out_base) * (int64_t)C + f0) * 2) = w;
      }
    } else {
      // s24 / s32 / f32: as above, a lane takes PIECES of the chunk's contiguous run — four elements: 12 or 16 bytes — so
      // that a store instruction writes one contiguous stretch.  (Until round 4 a lane stored the C elements of its own
      // sample-frame one by one, C * bytes apart from its neighbour's: 7.1.4 -> J 15 Gsamples/s in s24 and 34 in s32 where
      // s16 runs at 65 on this kernel's sibling, tools/debug/format_cliff_probe.py.)  Same products, same roundings.
      const int np = (cnt * C) >> 2;  // 4-element pieces (cnt % 64 == 0)
      const int ring_flat = kWPos * C;
      for (int q = t; q < np; q += 256) {
        const int f0 = 4 * q;
        int srel = (int)(((uint32_t)f0 * inv_c) >> 20);
        int r = f0 - srel * C;
        if (jc + srel < 0) continue;  // withheld look-ahead samples; 240*C is a multiple of 4
        int rf = epos * C + f0;
        rf = rf >= ring_flat ? rf - ring_flat : rf;
        const float4 v0 = *reinterpret_cast<const float4 *>(&ring[rf]);
        const float vv[4] = {v0.x, v0.y, v0.z, v0.w};
        float xg[4];
        float gq = arr_g[srel] * (1.0f / 32768.f);  // exact: undo the power-of-two scaling
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xg[i] = vv[i] * gq;
          ++r;
          if (r == C) {  // next sample-frame: next gain
            r = 0;
            ++srel;
            gq = arr_g[srel < kWChunk ? srel : kWChunk - 1] * (1.0f / 32768.f);
          }
        }
        uint8_t *dst = pcm + ((jc - out_base) * (int64_t)C + f0) * bytes;   // 12- or 16-byte pieces from a 16-byte aligned base
        if (p.out_format == IAMF_HIP_FMT_S24) {
          uint32_t u[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = (int)to_scaled(xg[i], 8388608.f, -8388608.f, 8388607.f);
            u[i] = (uint32_t)(v & 0xffff) | ((uint32_t)(((v >> 16) & 0x7f) | ((v >> 24) & 0x80)) << 16);
          }
          using u3 = __attribute__((ext_vector_type(3))) unsigned;
          *reinterpret_cast<u3 *>(dst) = u3{u[0] | (u[1] << 24), (u[1] >> 8) | (u[2] << 16), (u[2] >> 16) | (u[3] << 8)};
        } else if (p.out_format == IAMF_HIP_FMT_S32) {
          int4 w;
          w.x = (int32_t)(long long)to_scaled(xg[0], 2147483648.f, -2147483648.f, 2147483647.f);
          w.y = (int32_t)(long long)to_scaled(xg[1], 2147483648.f, -2147483648.f, 2147483647.f);
          w.z = (int32_t)(long long)to_scaled(xg[2], 2147483648.f, -2147483648.f, 2147483647.f);
          w.w = (int32_t)(long long)to_scaled(xg[3], 2147483648.f, -2147483648.f, 2147483647.f);
          *reinterpret_cast<int4 *>(dst) = w;
        } else {
          *reinterpret_cast<float4 *>(dst) = make_float4(xg[0], xg[1], xg[2], xg[3]);
        }
      }
    }
    __syncthreads();  // ring / arr slots are rewritten by the next chunk
  }

  // ---- persist stream state (same format as the generic kernel) ----
  {
    float *sy = p.ring_y + (int64_t)s * C * kSave;
    float *spm = p.ring_pm + (int64_t)s * kSave;
    const int64_t end = p.pos0 + p.total;
    const int rp = (int)((end - kSave + t) & (kWPos - 1));
    for (int c = 0; c < C; ++c) sy[c * kSave + t] = ring[rp * C + c];
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
