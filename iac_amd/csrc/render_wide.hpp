// render_wide.hpp — multi-channel output layouts (3..24 channels) with the limiter on, aligned
// calls.  One workgroup (4 waves) per stream, 256-sample chunks, lane = sample.
//   * rendered samples live in an LDS ring laid out exactly like the interleaved output
//     ([position][channel], no padding), so the emit stage is a linear copy: 32-byte LDS reads,
//     gain, round, 16-byte coalesced global stores — no 2-byte scattered stores;
//   * projection on VALU in the reference's operation order (bit-exact), four output slots at a
//     time with their weights fetched as one 16-byte LDS broadcast per input channel;
//   * the next chunk's input loads are issued right after the projection;
//   * limiter gains as in render_fast.hpp: no-trigger hypothesis for the whole chunk, otherwise
//     wave 0 re-runs the recurrence (speculation + DPP trigger-run chain).  The curve table is too
//     big to sit in LDS next to a 24-channel ring, so each chunk stages the 320-entry window it
//     can reach without a trigger plus the table head that follows a trigger.
#pragma once

constexpr int kWChunk = 256;
constexpr int kWPos = 512;   // ring positions (power of two >= chunk + look-ahead + 15)
constexpr int kWWin = 320;   // staged table window / head length (> chunk + 1)

__host__ __device__ constexpr int wide_lds_floats(int c, int m) {
  return kWPos * c + 2 * kWPos + 3 * kWChunk + 2 * kWWin + ((c + 3) & ~3) * m + 16;
}

template <int M>
__global__ __launch_bounds__(256) void render_wide_kernel(const RenderParams p) {
  extern __shared__ float lds[];
  const int C = p.out_ch;
  const int C4 = (C + 3) & ~3;
  float *ring = lds;                      // [kWPos][C]  rendered samples, interleaved like the output
  float *ring_pm = ring + kWPos * C;      // [kWPos]     max |y| over channels
  float *ring_b16 = ring_pm + kWPos;      // [kWPos]     max of pm over the trailing 16 samples
  float *arr_p = ring_b16 + kWPos;        // [256]
  float *arr_e = arr_p + kWChunk;         // [256]
  float *arr_g = arr_e + kWChunk;         // [256]
  float *win = arr_g + kWChunk;           // [kWWin]     ctab[n_st + i] for this chunk
  float *head = win + kWWin;              // [kWWin]     ctab[i]
  float *mat = head + kWWin;              // [M][C4]     weights, input-major
  float *misc = mat + C4 * M;             // [16]

  const int s = blockIdx.x;
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
    const int rp = (int)((p.pos0 - kSave + t) & (kWPos - 1));
    for (int c = 0; c < C; ++c) ring[rp * C + c] = sy[c * kSave + t];
    ring_pm[rp] = spm[t];
    for (int i = t; i < kWWin; i += 256) head[i] = p.ctab[i < n_end ? i : n_end];
    for (int i = t; i < C4 * M; i += 256) {
      const int m = i / C4, c = i - m * C4;
      const int f = c < C ? p.src_feed[c] : -1;
      mat[i] = f >= 0 ? p.matrix[f * M + m] : 0.f;
    }
  }
  __syncthreads();
  {
    const int64_t gk = p.pos0 - kSave + t;
    float b = 0.f;
    for (int j = 0; j < 16; ++j)
      if (t - j >= 0) b = fmaxf(b, ring_pm[(int)((gk - j) & (kWPos - 1))]);
    ring_b16[(int)(gk & (kWPos - 1))] = b;
  }
  LimState ls = p.lim[s];
  float g_cur = ls.g, gs = ls.gs, ge = ls.ge;
  int n_st = ls.n;
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);

  const int64_t out_base = p.pos0 > kDelay ? p.pos0 - kDelay : 0;
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  uint8_t *pcm = p.pcm + (int64_t)s * p.pcm_stream_stride;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;

  float x[M];
  {
    if (t < p.total) {
      const int f = t / fs;
      const int i = t - f * fs;
      const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = src[(int64_t)m * fs];
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = 0.f;
    }
  }
  __syncthreads();

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

    // ---- element renderer + gains, four output slots at a time ----
    float pm = 0.f;
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
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = y[i];
        if (eg_on) v = v * eg;
        v = 0.f + v;  // mixer: 0 += frame
        if (og_on) v = v * og;
        if (lg_on) v = v * lg;
        y[i] = v;
        if (cb + i < C) pm = fmaxf(pm, fabsf(v));
      }
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

    // ---- prefetch the next chunk's input ----
    {
      const int kn = k + kWChunk;
      if (kn < p.total) {
        const int f = kn / fs;
        const int i = kn - f * fs;
        const float *src = in_s + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = src[(int64_t)m * fs];
      }
    }

    if (valid) ring_pm[rp] = pm;
    win[t] = wv0;
    if (t < kWWin - 256) win[t + 256] = wv1;
    __syncthreads();
    float b = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) b = fmaxf(b, ring_pm[(int)((gk - j) & (kWPos - 1))]);
    if (valid) ring_b16[rp] = b;
    __syncthreads();
    float pk = 0.f;
#pragma unroll
    for (int j = 0; j < 15; ++j) pk = fmaxf(pk, ring_b16[(int)((gk - 1 - 16 * j) & (kWPos - 1))]);
    const float e = thr / pk;

    // ---- gains under the no-trigger hypothesis ----
    int n_pre = n_st + t;
    n_pre = n_pre < n_end ? n_pre : n_end;
    float g = gain_at(n_pre, gs, ge, win[t + 1], n_atk, n_end);
    const bool trig = valid && (pk * g > thr);
    arr_p[t] = pk;
    arr_e[t] = e;
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
      arr_g[t] = g;
      g_cur = misc[8];
      n_st = n_st + cnt < n_end ? n_st + cnt : n_end;
    } else {
      const int b0 = kf >> 6;
      if (t < 64 * b0) arr_g[t] = g;
      if (wave == 0) {
        const int n_chunk = n_st;
        int ln = n_st + 64 * b0 < n_end ? n_st + 64 * b0 : n_end;
        float lgs = gs, lge = ge, lgl = g_cur;
        auto look = [win, head, n_chunk](int ci) {
          const int d = ci - n_chunk;
          return (d >= 0 && d < kWWin) ? win[d] : head[ci < kWWin ? ci : kWWin - 1];
        };
        limiter_wave(arr_p, arr_e, arr_g, look, b0, cnt >> 6, ln, lgs, lge, lgl, thr, n_atk, n_end);
        if (lane == 0) {
          misc[4] = lgl;
          misc[5] = lgs;
          misc[6] = lge;
          misc[7] = __int_as_float(ln);
        }
      }
      __syncthreads();
      g_cur = misc[4];
      gs = misc[5];
      ge = misc[6];
      n_st = __float_as_int(misc[7]);
    }
    __syncthreads();  // arr_g complete for the emit stage

    // ---- emit: the chunk's 256 delayed samples are one contiguous run of the ring ----
    const int64_t jc = p.pos0 + c0 - kDelay;  // first emitted sample of the chunk (may be < 0)
    const int epos = (int)(jc & (kWPos - 1));
    if (p.out_format == IAMF_HIP_FMT_S16) {
      const int np = (cnt * C) >> 3;  // 8-element pieces (cnt % 64 == 0)
      const int ring_flat = kWPos * C;
      for (int q = t; q < np; q += 256) {
        const int f0 = 8 * q;
        int srel = f0 / C;
        int r = f0 - srel * C;
        if (jc + srel < 0) continue;  // withheld look-ahead samples; 240*C is a multiple of 8
        int rf = epos * C + f0;
        rf = rf >= ring_flat ? rf - ring_flat : rf;
        const float4 v0 = *reinterpret_cast<const float4 *>(&ring[rf]);
        const float4 v1 = *reinterpret_cast<const float4 *>(&ring[rf + 4]);
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        int o[8];
        float gq = arr_g[srel];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = (int)to_scaled(vv[i] * gq, 32768.f, -32768.f, 32767.f);
          ++r;
          if (r == C) {
            r = 0;
            ++srel;
            gq = arr_g[srel < kWChunk ? srel : kWChunk - 1];
          }
        }
        uint4 w;
        w.x = (uint32_t)(o[0] & 0xffff) | ((uint32_t)o[1] << 16);
        w.y = (uint32_t)(o[2] & 0xffff) | ((uint32_t)o[3] << 16);
        w.z = (uint32_t)(o[4] & 0xffff) | ((uint32_t)o[5] << 16);
        w.w = (uint32_t)(o[6] & 0xffff) | ((uint32_t)o[7] << 16);
        *reinterpret_cast<uint4 *>(pcm + ((jc - out_base) * (int64_t)C + f0) * 2) = w;
      }
    } else {
      const int64_t j = gk - kDelay;
      if (valid && j >= 0) {
        const int rd = (int)(j & (kWPos - 1));
        const float gq = arr_g[t];
        uint8_t *dst = pcm + (j - out_base) * (int64_t)C * bytes;
        if (p.out_format == IAMF_HIP_FMT_S24) {
          for (int c = 0; c < C; ++c) {
            const int v = (int)to_scaled(ring[rd * C + c] * gq, 8388608.f, -8388608.f, 8388607.f);
            dst[c * 3 + 0] = (uint8_t)(v & 0xff);
            dst[c * 3 + 1] = (uint8_t)((v >> 8) & 0xff);
            dst[c * 3 + 2] = (uint8_t)(((v >> 16) & 0x7f) | ((v >> 24) & 0x80));
          }
        } else if (p.out_format == IAMF_HIP_FMT_S32) {
          int32_t *d32 = reinterpret_cast<int32_t *>(dst);
          for (int c = 0; c < C; ++c)
            d32[c] = (int32_t)(long long)to_scaled(ring[rd * C + c] * gq, 2147483648.f, -2147483648.f, 2147483647.f);
        } else {
          float *df = reinterpret_cast<float *>(dst);
          for (int c = 0; c < C; ++c) df[c] = ring[rd * C + c] * gq;
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
