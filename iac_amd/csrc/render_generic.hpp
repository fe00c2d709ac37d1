// render_generic.hpp — the general render kernel: any channel counts, any frame size, any call
// length, limiter on or off, every output format.  One sample per lane, 256-sample chunks.
// The fast kernels (render_fast.hpp) cover the aligned common case; this one is the exact
// fallback for everything else (still HIP: there is no CPU path).
#pragma once

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
  float *dmx_ch = st + 4;                     // [kChCount][kChunk] per-thread channel file (down-mixer only)

  const int s = blockIdx.x + p.stream0;   // a launch covers streams [stream0, stream0 + n_launch) of the batch
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
    if (valid && p.in && p.pre_matrix) {
      // projection de-mapping (IAMF_core_decoder.c:116-130): x[r] = 0; x[r] += in[l] * P[l][r]
      const int f = k / fs;
      const int i = k - f * fs;
      const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = 0.f;
      for (int l = 0; l < p.pre_l; ++l) {
        const float v = src[(int64_t)l * fs];
        const float *row = p.pre_matrix + l * M;
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = x[m] + v * row[m];
      }
    } else if (valid && p.in) {
      const int f = k / fs;
      const int i = k - f * fs;
      const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = src[(int64_t)m * fs];
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = 0.f;
    }

    const int fcur = valid ? k / fs : 0;
    const int icur = k - fcur * fs;
    if (p.demix_on && valid && p.in) {
      // Demixer of scalable channel audio (demixer.c:636-664), one sample per thread; the thread's
      // channel file (indexed by IAChannel) lives in its own LDS column.  x[] arrives in decoded
      // order and leaves in the target layout's playback order.
      const iamf_hip_demix_frame *fr = p.demix_frames + (int64_t)s * ((p.total + fs - 1) / fs) + fcur;
      const int iw = icur + p.demix_i0;  // position inside the frame (windows, previous-mode prefix)
      const bool use_prev = iw < p.demix_skip;
      float cf[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) cf[j] = use_prev ? fr->prev[j] : fr->cur[j];
      const float alpha = cf[0], beta = cf[1], gamma = cf[2], delta = cf[3], w = cf[4];
      auto CH = [&](int c) -> float & { return dmx_ch[c * kChunk + t]; };
#pragma unroll
      for (int m = 0; m < M; ++m) CH(p.demix_tab[m]) = x[m];
      const int n_gain = p.demix_tab[24];
      for (int g = 0; g < n_gain; ++g) {  // dmx_gainup (:426-435)
        float &v = CH(p.demix_tab[25 + g]);
        v = v * p.demix_ftab[g];
      }
      const int steps = p.demix_steps;
      if (steps & 1) CH(kChR2) = 2 * CH(kChMono) - CH(kChL2);                       // S1to2 (:126-147)
      if (steps & 2) {  // S2to3 (:152-181): the reference's 0.707 literal makes this double arithmetic
        const double c = 0.707 * (double)CH(kChC);
        CH(kChL3) = (float)((double)CH(kChL2) - c);
        CH(kChR3) = (float)((double)CH(kChR2) - c);
      }
      if (steps & 4) {  // S3to5 (:186-230)
        CH(kChSL5) = (CH(kChL3) - CH(kChL7)) / delta;
        CH(kChSR5) = (CH(kChR3) - CH(kChR7)) / delta;
      }
      if (steps & 8) {  // S5to7 (:236-284)
        CH(kChBL7) = (CH(kChSL5) - CH(kChSL7) * alpha) / beta;
        CH(kChBR7) = (CH(kChSR5) - CH(kChSR7) * alpha) / beta;
      }
      if (steps & 16) {  // TF2toT2 (:290-335)
        CH(kChHL) = CH(kChTL) - delta * w * CH(kChSL5);
        CH(kChHR) = CH(kChTR) - delta * w * CH(kChSR5);
      }
      if (steps & 32) {  // T2toT4 (:340-377)
        CH(kChHBL) = (CH(kChHL) - CH(kChHFL)) / gamma;
        CH(kChHBR) = (CH(kChHR) - CH(kChHFR)) / gamma;
      }
      const int n_recon = fr->n_recon;
      const float wstart = p.demix_ftab[12 + iw], wstop = p.demix_ftab[12 + fs + iw];
      for (int r = 0; r < n_recon; ++r) {  // dmx_rms (:447-478)
        const float filt = fr->recon_prev[r] * wstop + fr->recon_cur[r] * wstart;
        float &v = CH(fr->recon_ch[r]);
        v = v * filt;
      }
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = CH(p.demix_tab[12 + m]);
    }

    // ---- element renderer(s) + gains; the rendered sample goes to the LDS delay ring ----
    float er = 1.f, er2 = 1.f, orr = 1.f;  // per-sample mix gains of this call, when given
    if (valid) {
      if (p.elem_ramp) er = p.elem_ramp[(int64_t)s * p.ramp_stream_stride + k];
      if (p.elem2_ramp) er2 = p.elem2_ramp[(int64_t)s * p.ramp_stream_stride + k];
      if (p.out_ramp) orr = p.out_ramp[(int64_t)s * p.ramp_stream_stride + k];
    }
    if (p.dmx_on) {
      // parametric down-mixer (downmix_renderer.c:115-129,218-242): every derived channel is the
      // f32 sum (from 0) of two scaled sources; input channels are taken as they are.  The
      // thread's channel file lives in its own LDS column.
      float cf[5] = {1.f, 1.f, 1.f, 1.f, 0.f};
      if (valid && p.dmx_frames) {
        const iamf_hip_dmx_frame *fr = p.dmx_frames + (int64_t)s * ((p.total + fs - 1) / fs) + fcur;
        const bool use_prev = icur < fr->offset;
#pragma unroll
        for (int j = 0; j < 5; ++j) cf[j] = use_prev ? fr->prev[j] : fr->cur[j];
      }
      for (int c = 0; c < kChCount; ++c) dmx_ch[c * kChunk + t] = 0.f;
      bool is_in[kChCount];
#pragma unroll
      for (int c = 0; c < kChCount; ++c) is_in[c] = false;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int ch = p.dmx_tab[m];
        dmx_ch[ch * kChunk + t] = x[m];
#pragma unroll
        for (int c = 0; c < kChCount; ++c) is_in[c] = is_in[c] || (ch == c);
      }
      auto CH = [&](int c) -> float & { return dmx_ch[c * kChunk + t]; };
      auto rule = [&](int dst, int s0, float k0, int s1, float k1) {
        if (!is_in[dst]) {
          float sum = 0.f;
          sum = sum + CH(s0) * k0;
          sum = sum + CH(s1) * k1;
          CH(dst) = sum;
        }
      };
      const float c707 = (float)0.707;
      rule(kChSL5, kChSL7, cf[0], kChBL7, cf[1]);
      rule(kChSR5, kChSR7, cf[0], kChBR7, cf[1]);
      rule(kChHL, kChHFL, 1.f, kChHBL, cf[2]);
      rule(kChHR, kChHFR, 1.f, kChHBR, cf[2]);
      rule(kChL3, kChL7, 1.f, kChSL5, cf[3]);
      rule(kChR3, kChR7, 1.f, kChSR5, cf[3]);
      rule(kChTL, kChHL, 1.f, kChSL5, cf[4]);
      rule(kChTR, kChHR, 1.f, kChSR5, cf[4]);
      rule(kChL2, kChL3, 1.f, kChC, c707);
      rule(kChR2, kChR3, 1.f, kChC, c707);
      rule(kChMono, kChR2, 0.5f, kChL2, 0.5f);
    }
    const float *src2 = nullptr;
    if (p.in2 && valid) src2 = p.in2 + (int64_t)s * p.in2_stream_stride + (int64_t)fcur * p.in2_frame_stride + icur;
    const float eg2 = p.in2 ? p.gains2[s] : 1.f;
    float pm = 0.f;
    for (int c = 0; c < out_ch; ++c) {
      float y = 0.f;
      if (p.dmx_on) {
        y = dmx_ch[p.dmx_tab[12 + c] * kChunk + t];
      } else {
        const int f = p.src_feed[c];
        if (f >= 0) {
          const float *row = p.matrix + f * M;  // wave-uniform -> scalar loads
          float acc = 0.f;
#pragma unroll
          for (int m = 0; m < M; ++m) acc = acc + row[m] * x[m];
          y = acc;
        } else if (f == -2 && p.lfe && valid) {
          // LFE slot (h2m_rdr.c:1154-1184): the generator's output `* 0.5` or `/ sqrt(n_size)` — double
          // expressions narrowed by the store; slot lfe2 repeats slot lfe1
          const int kl = k + p.lfe_k0;
          const float o = p.lfe[((((int64_t)(s >> 6) * p.lfe_t4 + (kl >> 2)) * 64 + (s & 63)) << 2) + (kl & 3)];
          y = p.lfe_div == 0.0 ? (float)((double)o * 0.5) : (float)((double)o / p.lfe_div);
        }
      }
      if (p.elem_ramp) {
        y = y * er;  // iamf_frame_gain with a gains[] array: unconditional (IAMF_decoder.c:1401-1405)
      } else if (eg_on) {
        y = y * eg;
      }
      y = 0.f + y;  // iamf_mixer_mix: memset 0 then += (IAMF_decoder.c:2719-2730)
      if (p.in2) {
        float y2 = 0.f;
        const int f2 = p.src_feed2[c];
        if (f2 >= 0 && src2) {
          const float *row2 = p.matrix2 + f2 * p.m2;
          float acc = 0.f;
          for (int m = 0; m < p.m2; ++m) acc = acc + row2[m] * src2[(int64_t)m * fs];
          y2 = acc;
        }
        if (p.elem2_ramp) {
          y2 = y2 * er2;
        } else if (eg2 != 1.f && eg2 > 0.f) {
          y2 = y2 * eg2;
        }
        y = y + y2;
      }
      if (c < p.og_ch) {   // (all channels, unless the batch was told otherwise: RenderParams::og_ch)
        if (p.out_ramp) {
          y = y * orr;
        } else if (og_on) {
          y = y * og;
        }
      }
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
        // at least one trigger: wave 0 runs the recurrence over the chunk
        arr_p[t] = pk;
        arr_e[t] = e;
        __syncthreads();
        if (t < 64) {
          // whole 64-sample blocks: the wave-level recurrence of the fast kernels (render_fast.hpp);
          // what is left of a ragged chunk: sample by sample
          float lgc = g_cur, lgs = gs, lge = ge;
          int ln = n_st;
          const int nfull = cnt >> 6;
          if (nfull > 0)
            limiter_wave(arr_p, arr_g, [head, &p](int ci) { return ci < kHead ? head[ci] : p.ctab[ci]; }, 0, nfull,
                         ln, lgs, lge, lgc, thr, n_atk, n_end);
          for (int i = 64 * nfull; t == 0 && i < cnt; ++i) {
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
          if (t == 0) {
            st[0] = lgc;
            st[1] = lgs;
            st[2] = lge;
            st[3] = __int_as_float(ln);
          }
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

