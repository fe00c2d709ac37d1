// render_nolim.hpp — the element renderer WITHOUT the peak limiter (IAMF_decoder_peak_limiter_enable(handle, 0), the
// player's -disable_limiter; also the first of the three launches of a rate-converting stream: render -> iamf_resample ->
// limiter, IAMF_decoder.c:3459-3500, and a second element's batch).  With the limiter off there is no delay line and no
// state: every (stream, sample) is independent.  The general kernel serves it one sample per lane with element-wise
// stores (TOA -> stereo 45, 7.1.4 -> J 28 Gsamples/s, tools/debug/shape_cliff_probe.py); this one takes the plain case —
// one matrix-rendered element, constant gains — four consecutive samples per lane (16-byte loads), the workgroup's 1024
// sample-frames packed in LDS in the output format and written as one contiguous run of 16-byte pieces.
// The same operations in the same order as render_kernel<M> (render_generic.hpp): acc from 0 over the inputs in order,
// element gain, `0.f + y` (iamf_mixer_mix), output gain, loudness gain, the format's conversion.
#pragma once

constexpr int kNlChunk = 1024;                       // sample-frames per workgroup
constexpr int kNlMaxFrameBytes = 60;                 // out_ch * bytes per sample-frame the LDS tile takes (60 KiB)

// what the kernel's addressing needs; the caller has checked that the call is of the plain kind
__host__ inline bool nolim_shape_ok(const RenderParams &p) {
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  if ((p.frame_size & 3) || (p.total & 3) || p.total <= 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.in) & 15) || (p.in_stream_stride & 3) || (p.in_frame_stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(p.pcm) & 15) || (p.pcm_stream_stride & 15)) return false;
  if (((p.out_ch * bytes) & 3) || p.out_ch * bytes > kNlMaxFrameBytes) return false;  // a lane's 4 sample-frames = whole 16-byte pieces
  return true;
}

template <int M>
__global__ __launch_bounds__(256) void render_nolim_kernel(const RenderParams p) {
  extern __shared__ float nl_lds[];
  uint8_t *tile = reinterpret_cast<uint8_t *>(nl_lds);   // [1024 sample-frames][out_ch * bytes]
  const int s = blockIdx.x + p.stream0;
  const int t = threadIdx.x;
  const int fs = p.frame_size, oc = p.out_ch;
  const int bytes = p.out_format == IAMF_HIP_FMT_S16 ? 2 : (p.out_format == IAMF_HIP_FMT_S24 ? 3 : 4);
  const int c0 = (int)blockIdx.y * kNlChunk;
  const int k = c0 + 4 * t;          // this lane's first sample of the call
  const bool valid = k < p.total;    // total % 4 == 0: all four or none
  const float eg = p.gains[s], og = p.gains[p.n_streams + s], lg = p.gains[2 * p.n_streams + s];
  const bool eg_on = (eg != 1.f && eg > 0.f);
  const bool og_on = (og != 1.f && og > 0.f);
  const bool lg_on = p.loudness_on && (lg != 1.0f);
  float4 x[M];
  if (valid) {
    const int f = k / fs, i = k - f * fs;   // fs % 4 == 0: the four samples lie in one frame
    const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i;
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = *reinterpret_cast<const float4 *>(src + (int64_t)m * fs);
    uint8_t *mine = tile + (size_t)(4 * t) * oc * bytes;
    for (int c = 0; c < oc; ++c) {
      const int fd = p.src_feed[c];
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      if (fd >= 0) {
        const float *row = p.matrix + fd * M;  // wave-uniform -> scalar loads
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc.x = acc.x + row[m] * x[m].x;
          acc.y = acc.y + row[m] * x[m].y;
          acc.z = acc.z + row[m] * x[m].z;
          acc.w = acc.w + row[m] * x[m].w;
        }
        y = acc;
      }
      float v[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float z = v[q];
        if (eg_on) z = z * eg;
        z = 0.f + z;  // iamf_mixer_mix: memset 0 then += (IAMF_decoder.c:2719-2730)
        if (og_on) z = z * og;
        if (lg_on) z = z * lg;
        uint8_t *d = mine + (size_t)(q * oc + c) * bytes;
        if (p.out_format == IAMF_HIP_FMT_S16) {
          *reinterpret_cast<int16_t *>(d) = (int16_t)(int)to_scaled(z, 32768.f, -32768.f, 32767.f);
        } else if (p.out_format == IAMF_HIP_FMT_S24) {
          const int w = (int)to_scaled(z, 8388608.f, -8388608.f, 8388607.f);
          d[0] = (uint8_t)(w & 0xff);
          d[1] = (uint8_t)((w >> 8) & 0xff);
          d[2] = (uint8_t)(((w >> 16) & 0x7f) | ((w >> 24) & 0x80));
        } else if (p.out_format == IAMF_HIP_FMT_S32) {
          // +full scale wraps to INT32_MIN as in the reference (IAMF_decoder.c:114-119)
          *reinterpret_cast<int32_t *>(d) = (int32_t)(long long)to_scaled(z, 2147483648.f, -2147483648.f, 2147483647.f);
        } else {
          *reinterpret_cast<float *>(d) = z;
        }
      }
    }
  }
  __syncthreads();
  // the tile's valid part -> PCM, 16 bytes per lane and round: one contiguous run
  const int n_here = p.total - c0 < kNlChunk ? p.total - c0 : kNlChunk;
  const int pieces = (n_here * oc * bytes) >> 4;   // (n_here % 4 == 0 and out_ch * bytes % 4 == 0: whole pieces)
  uint8_t *dst = p.pcm + (int64_t)s * p.pcm_stream_stride + (int64_t)c0 * oc * bytes;
  using u4 = __attribute__((ext_vector_type(4))) unsigned;
  for (int q = t; q < pieces; q += 256) *reinterpret_cast<u4 *>(dst + (size_t)q * 16) = *reinterpret_cast<const u4 *>(tile + (size_t)q * 16);
}
