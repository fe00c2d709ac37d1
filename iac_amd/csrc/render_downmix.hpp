// render_downmix.hpp — the parametric down-mixer (reference downmix_renderer.c:115-129,218-242) on a
// lane's 4 consecutive samples, for the 4-samples-per-lane kernels (render_fast.hpp, render_wide4.hpp).
// Same operations in the same order as the generic kernel's down-mixer (render_generic.hpp): every
// channel the input layout lacks is the f32 sum (from 0) of two scaled sources, evaluated in the
// fixed order SL5 SR5 HL HR L3 R3 TL TR L2 R2 Mono; channels the input carries are taken as they are.
// Both layouts are per-batch facts: the input layout is one of the (at most two) layouts with M
// channels, the output layout one of those with OC channels, picked by wave-uniform branches, so
// that every index into the channel file is a constant and the file lives in registers.
#pragma once

// playback order of the scalable layouts (IAChannelLayoutType; reference IAMF_utils.c:117-133)
__host__ __device__ constexpr int w4_layout_count(int layout) {
  constexpr int n[9] = {1, 2, 6, 8, 10, 8, 10, 12, 6};
  return n[layout];
}
__host__ __device__ constexpr int w4_layout_ch(int layout, int i) {
  constexpr int ch[9][12] = {
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
  return ch[layout][i];
}


__host__ __device__ constexpr bool w4_layout_has(int layout, int ch) {
  for (int i = 0; i < w4_layout_count(layout); ++i)
    if (w4_layout_ch(layout, i) == ch) return true;
  return false;
}

__device__ __forceinline__ float4 dm_rule(const float4 a, const float4 ka, const float4 b, const float4 kb) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  s.x = s.x + a.x * ka.x; s.y = s.y + a.y * ka.y; s.z = s.z + a.z * ka.z; s.w = s.w + a.w * ka.w;
  s.x = s.x + b.x * kb.x; s.y = s.y + b.y * kb.y; s.z = s.z + b.z * kb.z; s.w = s.w + b.w * kb.w;
  return s;
}

template <int LIN, int M, int OC>
__device__ __forceinline__ void downmix4_from(const float4 (&x)[M], float4 (&y)[OC], const float4 (&cf)[5], int out_layout) {
  float4 ch[kChCount];
#pragma unroll
  for (int c = 0; c < kChCount; ++c) ch[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < M; ++m) ch[w4_layout_ch(LIN, m)] = x[m];
  const float c707s = (float)0.707;
  const float4 c707 = make_float4(c707s, c707s, c707s, c707s), one = make_float4(1.f, 1.f, 1.f, 1.f),
               half = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
#define DM_RULE(dst, s0, k0, s1, k1) \
  if constexpr (!w4_layout_has(LIN, dst)) ch[dst] = dm_rule(ch[s0], k0, ch[s1], k1);
  DM_RULE(kChSL5, kChSL7, cf[0], kChBL7, cf[1])
  DM_RULE(kChSR5, kChSR7, cf[0], kChBR7, cf[1])
  DM_RULE(kChHL, kChHFL, one, kChHBL, cf[2])
  DM_RULE(kChHR, kChHFR, one, kChHBR, cf[2])
  DM_RULE(kChL3, kChL7, one, kChSL5, cf[3])
  DM_RULE(kChR3, kChR7, one, kChSR5, cf[3])
  DM_RULE(kChTL, kChHL, one, kChSL5, cf[4])
  DM_RULE(kChTR, kChHR, one, kChSR5, cf[4])
  DM_RULE(kChL2, kChL3, one, kChC, c707)
  DM_RULE(kChR2, kChR3, one, kChC, c707)
  DM_RULE(kChMono, kChR2, half, kChL2, half)
#undef DM_RULE
#pragma unroll
  for (int lo = 0; lo < 9; ++lo) {
    if (w4_layout_count(lo) == OC && out_layout == lo) {
#pragma unroll
      for (int c = 0; c < OC; ++c) y[c] = ch[w4_layout_ch(lo, c)];
    }
  }
}

// x: the lane's 4 samples of the M input channels (playback order of the input layout); y: the OC
// channels of the output layout; cf: alpha, beta, gamma, delta, gamma*w for each of the 4 samples
template <int M, int OC>
__device__ __forceinline__ void downmix4(const float4 (&x)[M], float4 (&y)[OC], const float4 (&cf)[5], int in_layout,
                                         int out_layout) {
#pragma unroll
  for (int c = 0; c < OC; ++c) y[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (M == 12) downmix4_from<7, M, OC>(x, y, cf, out_layout);
  if constexpr (M == 10) {
    if (in_layout == 4) downmix4_from<4, M, OC>(x, y, cf, out_layout);
    else downmix4_from<6, M, OC>(x, y, cf, out_layout);
  }
  if constexpr (M == 8) {
    if (in_layout == 3) downmix4_from<3, M, OC>(x, y, cf, out_layout);
    else downmix4_from<5, M, OC>(x, y, cf, out_layout);
  }
  if constexpr (M == 6) {
    if (in_layout == 2) downmix4_from<2, M, OC>(x, y, cf, out_layout);
    else downmix4_from<8, M, OC>(x, y, cf, out_layout);
  }
  if constexpr (M == 2) downmix4_from<1, M, OC>(x, y, cf, out_layout);
}

// The lane's frame record (iamf_hip_dmx_frame: offset, prev[5], cur[5]) is prefetched with the
// input; this picks the factors for the lane's 4 samples, the first of which sits at position icur
// of its frame (a frame's first `offset` samples use the previous mode).
__device__ __forceinline__ void downmix_factors(const float (&rec)[11], int icur, float4 (&cf)[5]) {
  const int off = __float_as_int(rec[0]);
  const bool p0 = icur < off, p1 = icur + 1 < off, p2 = icur + 2 < off, p3 = icur + 3 < off;
#pragma unroll
  for (int j = 0; j < 5; ++j)
    cf[j] = make_float4(p0 ? rec[1 + j] : rec[6 + j], p1 ? rec[1 + j] : rec[6 + j], p2 ? rec[1 + j] : rec[6 + j],
                        p3 ? rec[1 + j] : rec[6 + j]);
}
