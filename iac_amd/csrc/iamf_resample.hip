// iamf_resample.hip — the decoder's sample-rate converter (reference src/iamf_dec/resample.c, a
// speexdsp derivative at quality 4; glue iamf_resample, IAMF_decoder.c:3223-3248) for a batch of
// streams on the GPU.
//
// The reference walks each channel serially with a phase accumulator.  Output k of a call is a
// pure function of the call's start state: pos_k = last_sample + floor((frac + k*num)/den),
// phase_k = (frac + k*num) mod den, so every (stream, output, channel) is one independent thread
// doing the reference's 64-tap f32 dot product in the reference's order (bit-exact); the tiled
// kernel stages a workgroup's stretch of input and the filter table in LDS first.  Buffers are
// interleaved f32 ([sample][channel]) — what the render kernels emit with IAMF_HIP_FMT_F32 and
// what a frame_size-1 batch consumes — so the resampled path is: render(F32, limiter off) ->
// resample -> render(identity matrix, loudness, limiter, PCM pack).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/iamf_hip.h"
#include "../data/resample_window_q4.h"

namespace {

struct RsParams {
  const float *in;        // [stream][ns][ch] or nullptr (zeros)
  int64_t in_stream_stride;
  const float *hist;      // [stream][N-1][ch]
  float *hist_next;       // [stream][N-1][ch]
  float *out;             // [stream][cap][ch]
  int64_t out_stream_stride;
  const float *table;
  int32_t ch, ns, n_out, consumed;
  int32_t N, oversample, direct;
  uint32_t num, den, int_adv;
  int32_t ls0;
  uint32_t fr0;
  const float *table4;    // interpolated mode: [oversample][N][4] = the four table values a tap multiplies, per offset
  int32_t win_cap;        // resample_tile_kernel: samples of [hist | in] one workgroup stages
  int32_t s0;             // first stream of this launch (blockIdx.y + s0): a range of streams that share one phase
};

// resample.c:246-256
__device__ __forceinline__ void cubic_coef(float frac, float interp[4]) {
  interp[0] = -0.16667f * frac + 0.16667f * frac * frac * frac;
  interp[1] = frac + 0.5f * frac * frac - 0.5f * frac * frac * frac;
  interp[3] = -0.33333f * frac + 0.5f * frac * frac - 0.16667f * frac * frac * frac;
  interp[2] = (float)(1. - interp[0] - interp[1] - interp[3]);
}

__global__ __launch_bounds__(256) void resample_kernel(const RsParams p) {
  const int s = blockIdx.y + p.s0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // flat (output, channel)
  const int hist_len = p.N - 1;
  const float *in = p.in ? p.in + (int64_t)s * p.in_stream_stride : nullptr;
  const float *hist = p.hist + (int64_t)s * hist_len * p.ch;

  // history for the next call: [hist | in] shifted by the consumed samples (resample.c:801-809)
  if (e < (int64_t)hist_len * p.ch) {
    const int j = (int)(e / p.ch), c = (int)(e - (int64_t)j * p.ch);
    const int src = j + p.consumed;  // index into [hist | in]
    float v = 0.f;
    if (src < hist_len)
      v = hist[src * p.ch + c];
    else if (in)
      v = in[(int64_t)(src - hist_len) * p.ch + c];
    p.hist_next[((int64_t)s * hist_len + j) * p.ch + c] = v;
  }
  if (e >= (int64_t)p.n_out * p.ch) return;
  const int k = (int)(e / p.ch), c = (int)(e - (int64_t)k * p.ch);
  const uint64_t tot = (uint64_t)p.fr0 + (uint64_t)k * p.num;  // int_adv*den + frac_adv == num
  const int pos = p.ls0 + (int)(tot / p.den);
  const uint32_t frac = (uint32_t)(tot % p.den);
  const int N = p.N;

  // sample j of the window = position pos + j of [hist | in]
  auto tap = [&](int j) -> float {
    const int idx = pos + j;
    if (idx < hist_len) return hist[idx * p.ch + c];
    return in ? in[(int64_t)(idx - hist_len) * p.ch + c] : 0.f;
  };

  float sum;
  if (p.direct) {  // resample.c:273-281
    const float *sinct = p.table + (size_t)frac * N;
    sum = 0.f;
    for (int j = 0; j < N; ++j) sum = sum + sinct[j] * tap(j);
  } else {  // resample.c:372-400
    const int offset = (int)(frac * (uint32_t)p.oversample / p.den);
    const float fr = ((float)((frac * (uint32_t)p.oversample) % p.den)) / (float)p.den;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < N; ++j) {
      const float cur = tap(j);
      const float *t = p.table + 4 + (j + 1) * p.oversample - offset;
      a0 = a0 + cur * t[-2];
      a1 = a1 + cur * t[-1];
      a2 = a2 + cur * t[0];
      a3 = a3 + cur * t[1];
    }
    float interp[4];
    cubic_coef(fr, interp);
    sum = interp[0] * a0 + interp[1] * a1 + interp[2] * a2 + interp[3] * a3;
  }
  sum = sum < -1.0f ? -1.0f : (sum > 1.0f ? 1.0f : sum);  // FLTADJUST, resample.c:84,959
  p.out[(int64_t)s * p.out_stream_stride + e] = sum;
}

// Both filter modes (interpolated: every rate pair but the small-denominator ones, e.g. 44.1 <-> 48
// kHz; direct: 2:1, 3:2 ...) with the operands in LDS.  A workgroup owns 1024 consecutive (output, channel) elements of one stream: the
// stretch of [hist | in] they read is staged once (interleaved as in memory: lanes read consecutive
// floats), and so is the filter table, re-laid per interpolation offset so that the four values
// a tap multiplies are one aligned 16-byte read.  The same f32 operations in the same order as
// resample_kernel (resample.c:372-400): bit-exact.
constexpr int kRsTile = 1024;  // (output, channel) elements per workgroup: 4 per thread
__global__ __launch_bounds__(256) void resample_tile_kernel(const RsParams p) {
  extern __shared__ float rs_lds[];
  const int s = blockIdx.y + p.s0;
  const int64_t e0 = (int64_t)blockIdx.x * kRsTile;
  const int hist_len = p.N - 1;
  const int N = p.N, ch = p.ch;
  const float *in = p.in ? p.in + (int64_t)s * p.in_stream_stride : nullptr;
  const float *hist = p.hist + (int64_t)s * hist_len * ch;
  float *win = rs_lds;                                  // [win_cap][ch]
  float4 *tab = reinterpret_cast<float4 *>(rs_lds + (((size_t)p.win_cap * ch + 3) & ~(size_t)3));  // [oversample][N + 1]

  // history for the next call: [hist | in] shifted by the consumed samples (resample.c:801-809)
  for (int64_t e = e0 + threadIdx.x; e < e0 + kRsTile && e < (int64_t)hist_len * ch; e += 256) {
    const int j = (int)(e / ch), c = (int)(e - (int64_t)j * ch);
    const int src = j + p.consumed;  // index into [hist | in]
    float v = 0.f;
    if (src < hist_len)
      v = hist[src * ch + c];
    else if (in)
      v = in[(int64_t)(src - hist_len) * ch + c];
    p.hist_next[((int64_t)s * hist_len + j) * ch + c] = v;
  }
  // first window position of the workgroup's first output (positions grow with the output index)
  const int64_t k_first = e0 / ch;
  const int base = p.ls0 + (int)(((uint64_t)p.fr0 + (uint64_t)k_first * p.num) / p.den);
  for (int i = threadIdx.x; i < p.win_cap * ch; i += 256) {
    const int idx = base + i / ch, c = i - (i / ch) * ch;
    float v = 0.f;
    if (idx < hist_len)
      v = hist[idx * ch + c];
    else if (in && idx - hist_len < p.ns)
      v = in[(int64_t)(idx - hist_len) * ch + c];
    win[i] = v;
  }
  // one row per offset (interpolated mode) or per phase (direct mode), rows padded by 16 bytes: the
  // lanes of a wave sit on several rows at the same tap, and a row is a multiple of the LDS bank span
  const int N4 = N >> 2;  // direct mode: N is a multiple of 8, a row = N / 4 float4 of consecutive taps
  if (p.direct) {
    const float4 *t4 = reinterpret_cast<const float4 *>(p.table);
    for (int i = threadIdx.x; i < (int)p.den * N4; i += 256) tab[i + i / N4] = t4[i];
  } else {
    const float4 *t4 = reinterpret_cast<const float4 *>(p.table4);
    for (int i = threadIdx.x; i < p.oversample * N; i += 256) tab[i + i / N] = t4[i];
  }
  __syncthreads();
  for (int64_t e = e0 + threadIdx.x; e < e0 + kRsTile && e < (int64_t)p.n_out * ch; e += 256) {
    const int k = (int)(e / ch), c = (int)(e - (int64_t)k * ch);
    const uint64_t tot = (uint64_t)p.fr0 + (uint64_t)k * p.num;
    const int pos = p.ls0 + (int)(tot / p.den);
    const uint32_t frac = (uint32_t)(tot % p.den);
    const int offset = (int)(frac * (uint32_t)p.oversample / p.den);
    const float fr = ((float)((frac * (uint32_t)p.oversample) % p.den)) / (float)p.den;
    const float *wp = win + (size_t)(pos - base) * ch + c;
    float sum;
    if (p.direct) {  // resample.c:273-281
      const float4 *tp = tab + (size_t)frac * (N4 + 1);
      sum = 0.f;
      for (int j = 0; j < N4; ++j) {
        const float4 w = tp[j];
        sum = sum + w.x * wp[(size_t)(4 * j + 0) * ch];
        sum = sum + w.y * wp[(size_t)(4 * j + 1) * ch];
        sum = sum + w.z * wp[(size_t)(4 * j + 2) * ch];
        sum = sum + w.w * wp[(size_t)(4 * j + 3) * ch];
      }
    } else {  // resample.c:372-400
      const float4 *tp = tab + (size_t)offset * (N + 1);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int j = 0; j < N; ++j) {
        const float cur = wp[(size_t)j * ch];
        const float4 w = tp[j];
        a0 = a0 + cur * w.x;
        a1 = a1 + cur * w.y;
        a2 = a2 + cur * w.z;
        a3 = a3 + cur * w.w;
      }
      float interp[4];
      cubic_coef(fr, interp);
      sum = interp[0] * a0 + interp[1] * a1 + interp[2] * a2 + interp[3] * a3;
    }
    sum = sum < -1.0f ? -1.0f : (sum > 1.0f ? 1.0f : sum);  // FLTADJUST, resample.c:84,959
    p.out[(int64_t)s * p.out_stream_stride + e] = sum;
  }
}

// floats between two samples of the staged window: C, padded so that CS / 2 is odd (8, 12, 16, 24 channels put the lanes'
// 8-byte reads 8 / 4 / 16 / 8 bank pairs apart: 2- to 8-way conflicts; measured with 8 channels: 0.5 of the tiled kernel)
__host__ __device__ constexpr int rs_cs(int c) { return c == 1 ? 1 : (((c / 2) & 1) ? c : c + 2); }

// Register-blocked form of resample_tile_kernel for the INTERPOLATED mode (44.1 <-> 48 kHz ...; round 4, second half).
// In the tiled kernel a thread = one (output, channel) element reads, per tap, 16 bytes of table and 4 bytes of input for 4 multiply-adds: 5 B of LDS traffic per MAC.  Here a thread owns R outputs of ONE
// phase — k, k + den G, ..., k + (R - 1) den G: frac, hence the interpolation offset / the table row, repeats with period
// den, and the window moves on by exactly num samples per den outputs — and ALL C channels of them, so a tap's four table
// values are read once for 4 R C multiply-adds and the C channels of a sample are one contiguous read (stereo, R = 4: 1.5
// B per MAC).  What bounds it then is the vector ALU: the reference rounds product and sum separately (no fma), a wave
// issues v_pk_mul_f32 / v_pk_add_f32 every 6 cycles (tools/debug/pk_rate_probe.hip) = 26 T MAC/s chip-wide, of which this
// kernel reaches 18-19 (stereo 44.1 -> 48 kHz: 25.2 -> 37.0 G output sample-frames/s, 6 / 8 / 12 channels 9.3 -> 11.6 / 7.1 ->
// 8.6 / 5.0 -> 7.1; tools/debug/resample_probe.py, profiles/r04_resample_probe.txt).  The staged window's sample stride is
// padded (rs_cs): with 8 or 12 floats between samples the lanes' reads collide on LDS banks (8 channels unpadded: 0.57 of
// the tiled kernel).  The direct mode has a kernel of its own below.  Every (output, channel) still runs the reference's
// operations in the reference's order (resample.c:372-400): bit-exact, and bit-equal to the two kernels above.
// Workgroup = den * G threads (rounded up to whole waves), thread u taking outputs u + r * den * G of the tile of
// den * R * G consecutive outputs of one stream: neighbouring lanes read neighbouring windows.
template <int C, int R>
__global__ __launch_bounds__(512) void resample_block_kernel(const RsParams p, int G) {
  extern __shared__ float rs_lds[];
  const int s = blockIdx.y + p.s0;
  const int hist_len = p.N - 1;
  const int N = p.N;
  const int den = (int)p.den;
  const int T = den * R * G;
  const float *in = p.in ? p.in + (int64_t)s * p.in_stream_stride : nullptr;
  const float *hist = p.hist + (int64_t)s * hist_len * C;
  constexpr int CS = rs_cs(C);
  float *win = rs_lds;                                  // [win_cap][CS]
  float4 *tab = reinterpret_cast<float4 *>(rs_lds + (((size_t)p.win_cap * CS + 3) & ~(size_t)3));

  // history for the next call: [hist | in] shifted by the consumed samples (resample.c:801-809); workgroup 0 of the stream
  if (blockIdx.x == 0) {
    for (int e = threadIdx.x; e < hist_len * C; e += blockDim.x) {
      const int j = e / C, c = e - j * C;
      const int src = j + p.consumed;  // index into [hist | in]
      float v = 0.f;
      if (src < hist_len)
        v = hist[src * C + c];
      else if (in)
        v = in[(int64_t)(src - hist_len) * C + c];
      p.hist_next[((int64_t)s * hist_len + j) * C + c] = v;
    }
  }
  const int64_t k_first = (int64_t)blockIdx.x * T;
  const int base = p.ls0 + (int)(((uint64_t)p.fr0 + (uint64_t)k_first * p.num) / p.den);
  for (int i = threadIdx.x; i < p.win_cap * C; i += blockDim.x) {
    const int idx = base + i / C, c = i - (i / C) * C;
    float v = 0.f;
    if (idx < hist_len)
      v = hist[idx * C + c];
    else if (in && idx - hist_len < p.ns)
      v = in[(int64_t)(idx - hist_len) * C + c];
    win[(i / C) * CS + c] = v;
  }
  {
    const float4 *t4 = reinterpret_cast<const float4 *>(p.table4);
    for (int i = threadIdx.x; i < p.oversample * N; i += blockDim.x) tab[i + i / N] = t4[i];
  }
  __syncthreads();
  const int u = threadIdx.x;
  if (u >= den * G) return;
  const int64_t k0 = k_first + u;   // this thread's outputs: k0 + r * den * G (neighbouring lanes: neighbouring outputs)
  if (k0 >= p.n_out) return;
  const uint64_t tot = (uint64_t)p.fr0 + (uint64_t)k0 * p.num;
  const int pos = p.ls0 + (int)(tot / p.den);
  const uint32_t frac = (uint32_t)(tot % p.den);
  const float *wp = win + (size_t)(pos - base) * CS;
  const int step = (int)p.num * G * CS;   // floats between the windows of outputs k and k + den * G
  float res[R][C];
  {  // resample.c:372-400
    const int offset = (int)(frac * (uint32_t)p.oversample / p.den);
    const float fr = ((float)((frac * (uint32_t)p.oversample) % p.den)) / (float)p.den;
    const float4 *tp = tab + (size_t)offset * (N + 1);
    float4 acc[R][C];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[r][c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
      const float4 w = tp[j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float *x = wp + r * step + (size_t)j * CS;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float cur = x[c];
          acc[r][c].x = acc[r][c].x + cur * w.x;
          acc[r][c].y = acc[r][c].y + cur * w.y;
          acc[r][c].z = acc[r][c].z + cur * w.z;
          acc[r][c].w = acc[r][c].w + cur * w.w;
        }
      }
    }
    float interp[4];
    cubic_coef(fr, interp);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c)
        res[r][c] = interp[0] * acc[r][c].x + interp[1] * acc[r][c].y + interp[2] * acc[r][c].z + interp[3] * acc[r][c].w;
  }
  float *out = p.out + (int64_t)s * p.out_stream_stride;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t k = k0 + (int64_t)r * den * G;
    if (k < p.n_out) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float sum = res[r][c];
        sum = sum < -1.0f ? -1.0f : (sum > 1.0f ? 1.0f : sum);  // FLTADJUST, resample.c:84,959
        out[k * C + c] = sum;
      }
    }
  }
}

// R for a channel count: 4 R C accumulators per thread in interpolated mode
constexpr int rs_block_r(int c) { return c <= 2 ? 4 : (c <= 8 ? 2 : 1); }
template <int C, int R>
void rs_block_launch_cr(const RsParams &p, int G, dim3 grid, unsigned block, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((resample_block_kernel<C, R>), grid, dim3(block), lds, st, p, G);
}
// mono, stereo / binaural, 5.1 / 3.1.2, 7.1 / 5.1.2; `r` is one of 1, 2, 4 and at most rs_block_r(ch).  false: no instantiation
bool rs_block_launch(int ch, int r, const RsParams &p, int G, dim3 grid, unsigned block, size_t lds, hipStream_t st) {
#define RS_CASE(C_)                                                                   \
  case C_:                                                                            \
    if (r >= 4 && rs_block_r(C_) >= 4) rs_block_launch_cr<C_, rs_block_r(C_) >= 4 ? 4 : 1>(p, G, grid, block, lds, st);      \
    else if (r >= 2 && rs_block_r(C_) >= 2) rs_block_launch_cr<C_, rs_block_r(C_) >= 2 ? 2 : 1>(p, G, grid, block, lds, st); \
    else rs_block_launch_cr<C_, 1>(p, G, grid, block, lds, st);                        \
    return true;
  switch (ch) {
    RS_CASE(1) RS_CASE(2) RS_CASE(6) RS_CASE(8) RS_CASE(10) RS_CASE(12) RS_CASE(14) RS_CASE(24)
    default: return false;
  }
#undef RS_CASE
}

// DIRECT mode (small denominators: 2:1, 3:1, 1:2, 1:3, 2:3, 3:2 ...; resample.c:273-281) with the phase's filter row in
// REGISTERS.  The tiled kernel reads 8 B of LDS per multiply-add there (4 B of table + 4 B of input per tap), and its lanes'
// windows start num / den * C floats apart — for 2:1 an even stride, 2-way bank conflicts on top.  Here a thread owns one
// PHASE (outputs k, k + den G, ...: R of them, one after the other) and all C channels: the row's N taps are loaded once
// into N registers, and a tap costs one contiguous read of the sample's C channels: 4 B per MAC, no table traffic.
// Whole-ratio down-sampling (den == 1, NUMP = num): the staged window is de-interleaved by residue mod num — sample i at
// plane i % num, slot i / num — so that the lanes, whose windows start num samples apart, read CONSECUTIVE slots of one
// plane for a tap (the tile's first output sits at window position 0, so plane and slot offset of tap j are the
// compile-time j % num, j / num).  Other ratios (NUMP = 1) keep the plain layout: their lanes advance by less than two
// samples.  The same products and sums in the same order as resample_kernel: bit-exact.
// acc + w * x for a pair of channels, w = half H of a register PAIR of taps: v_pk_mul_f32 takes the same half of its first
// source for both results (op_sel / op_sel_hi), so the row stays N registers — written out by the compiler the broadcast
// (w, w) is a register pair of its own per tap: 2 N registers.  Product and sum are rounded separately, as in the reference.
using rs_v2 = float __attribute__((ext_vector_type(2)));
template <int H>
__device__ __forceinline__ rs_v2 rs_mac2(rs_v2 acc, rs_v2 wpair, rs_v2 x) {
  rs_v2 t;
  if constexpr (H == 0)
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %0" : "=&v"(t), "+v"(acc) : "v"(wpair), "v"(x));
  else
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_add_f32 %1, %1, %0" : "=&v"(t), "+v"(acc) : "v"(wpair), "v"(x));
  return acc;
}

template <int C, int N, int NUMP, int R>
__global__ __launch_bounds__(256, 2) void resample_direct_kernel(const RsParams p, int G) {
  extern __shared__ float rs_lds[];
  constexpr int CS = rs_cs(C);
  const int s = blockIdx.y + p.s0;
  const int hist_len = N - 1;
  const int den = (int)p.den;
  const int T = den * R * G;
  const float *in = p.in ? p.in + (int64_t)s * p.in_stream_stride : nullptr;
  const float *hist = p.hist + (int64_t)s * hist_len * C;
  const int slots = (p.win_cap + NUMP - 1) / NUMP + 1;   // per plane
  float *win = rs_lds;                                  // [NUMP][slots][CS]
  float *tab = rs_lds + (((size_t)NUMP * slots * CS + 3) & ~(size_t)3);   // [den][N]

  if (blockIdx.x == 0) {  // history for the next call (resample.c:801-809)
    for (int e = threadIdx.x; e < hist_len * C; e += blockDim.x) {
      const int j = e / C, c = e - j * C;
      const int src = j + p.consumed;
      float v = 0.f;
      if (src < hist_len)
        v = hist[src * C + c];
      else if (in)
        v = in[(int64_t)(src - hist_len) * C + c];
      p.hist_next[((int64_t)s * hist_len + j) * C + c] = v;
    }
  }
  const int64_t k_first = (int64_t)blockIdx.x * T;
  const int base = p.ls0 + (int)(((uint64_t)p.fr0 + (uint64_t)k_first * p.num) / p.den);
  for (int i = threadIdx.x; i < p.win_cap * C; i += blockDim.x) {
    const int w = i / C, c = i - w * C;
    const int idx = base + w;
    float v = 0.f;
    if (idx < hist_len)
      v = hist[idx * C + c];
    else if (in && idx - hist_len < p.ns)
      v = in[(int64_t)(idx - hist_len) * C + c];
    win[((w % NUMP) * slots + w / NUMP) * CS + c] = v;
  }
  for (int i = threadIdx.x; i < den * N; i += blockDim.x) tab[i] = p.table[i];
  __syncthreads();
  const int u = threadIdx.x;
  if (u >= den * G) return;
  const int64_t k0 = k_first + u;   // this thread's outputs: k0 + r * den * G, their windows num * G samples apart
  if (k0 >= p.n_out) return;
  const uint64_t tot = (uint64_t)p.fr0 + (uint64_t)k0 * p.num;
  const uint32_t frac = (uint32_t)(tot % p.den);
  const int pos = p.ls0 + (int)(tot / p.den) - base;   // window position of output k0: a multiple of NUMP
  rs_v2 w[N / 2];   // taps 2 i, 2 i + 1
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    const float4 t4 = *reinterpret_cast<const float4 *>(&tab[(size_t)frac * N + j]);
    w[j / 2] = rs_v2{t4.x, t4.y};
    w[j / 2 + 1] = rs_v2{t4.z, t4.w};
  }
  const float *wp = win + (size_t)(pos / NUMP) * CS;
  const int step = ((int)p.num * G / NUMP) * CS;   // floats between the windows of outputs r and r + 1 (num * G is a multiple of NUMP)
  // R outputs at once: R * C / 2 independent chains of (product, sum) per lane — one output's N dependent sums alone
  // leave a wave waiting for its own previous instruction
  rs_v2 acc2[R][(C + 1) / 2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < (C + 1) / 2; ++c) acc2[r][c] = rs_v2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {   // taps 2 i and 2 i + 1: the two halves of w[i]
    // (fully unrolled for the register row; without this the scheduler issues all N reads first)
    if ((i & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    const int o0 = (((2 * i) % NUMP) * slots + (2 * i) / NUMP) * CS;
    const int o1 = (((2 * i + 1) % NUMP) * slots + (2 * i + 1) / NUMP) * CS;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float *x0 = wp + r * step + o0;
      const float *x1 = wp + r * step + o1;
      if constexpr ((C & 1) == 0) {
#pragma unroll
        for (int c = 0; c < C; c += 2) acc2[r][c / 2] = rs_mac2<0>(acc2[r][c / 2], w[i], *reinterpret_cast<const rs_v2 *>(x0 + c));
#pragma unroll
        for (int c = 0; c < C; c += 2) acc2[r][c / 2] = rs_mac2<1>(acc2[r][c / 2], w[i], *reinterpret_cast<const rs_v2 *>(x1 + c));
      } else {
        static_assert(C == 1 || (C & 1) == 0, "one channel, or pairs");
        acc2[r][0].x = acc2[r][0].x + w[i].x * x0[0];
        acc2[r][0].x = acc2[r][0].x + w[i].y * x1[0];
      }
    }
  }
  float *out = p.out + (int64_t)s * p.out_stream_stride;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t k = k0 + (int64_t)r * den * G;
    if (k < p.n_out) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float sum = (c & 1) ? acc2[r][c / 2].y : acc2[r][c / 2].x;
        sum = sum < -1.0f ? -1.0f : (sum > 1.0f ? 1.0f : sum);  // FLTADJUST, resample.c:84,959
        out[k * C + c] = sum;
      }
    }
  }
}

// outputs a thread works on at once
constexpr int rs_direct_r(int c) { return c <= 2 ? 4 : (c <= 8 ? 2 : 1); }
template <int C>
bool rs_direct_launch_c(int n, int nump, const RsParams &p, int G, dim3 grid, unsigned block, size_t lds, hipStream_t st) {
  constexpr int R = rs_direct_r(C);
  if (n == 64 && nump == 1) hipLaunchKernelGGL((resample_direct_kernel<C, 64, 1, R>), grid, dim3(block), lds, st, p, G);
  else if (n == 96 && nump == 1) hipLaunchKernelGGL((resample_direct_kernel<C, 96, 1, R>), grid, dim3(block), lds, st, p, G);
  else if (n == 128 && nump == 2) hipLaunchKernelGGL((resample_direct_kernel<C, 128, 2, R>), grid, dim3(block), lds, st, p, G);
  else if (n == 192 && nump == 3 && (C == 6 || C >= 10))   // (192 taps + accumulators: two waves per SIMD; 1, 2 and 8 channels measured 0.8 of the tiled kernel)
    hipLaunchKernelGGL((resample_direct_kernel<C, 192, 3, R>), grid, dim3(block), lds, st, p, G);
  else return false;
  return true;
}
// filter lengths of quality 4: 64 (up-sampling), 96 (3:2), 128 (2:1), 192 (3:1).  false: no instantiation
bool rs_direct_launch(int ch, int n, int nump, const RsParams &p, int G, dim3 grid, unsigned block, size_t lds, hipStream_t st) {
  switch (ch) {
    case 1: return rs_direct_launch_c<1>(n, nump, p, G, grid, block, lds, st);
    case 2: return rs_direct_launch_c<2>(n, nump, p, G, grid, block, lds, st);
    case 6: return rs_direct_launch_c<6>(n, nump, p, G, grid, block, lds, st);
    case 8: return rs_direct_launch_c<8>(n, nump, p, G, grid, block, lds, st);
    case 10: return rs_direct_launch_c<10>(n, nump, p, G, grid, block, lds, st);
    case 12: return rs_direct_launch_c<12>(n, nump, p, G, grid, block, lds, st);
    default: return false;
  }
}

// ---- filter design on the host: resample.c:194-231 (window, sinc) and :527-611 ----
double window_at(float x) {
  float y, frac;
  double interp[4];
  int ind;
  y = x * IAMF_RS_Q4_WINDOW_OVERSAMPLE;
  ind = (int)floor(y);
  frac = (y - ind);
  interp[3] = -0.1666666667 * frac + 0.1666666667 * (frac * frac * frac);
  interp[2] = frac + 0.5 * (frac * frac) - 0.5 * (frac * frac * frac);
  interp[0] = -0.3333333333 * frac + 0.5 * (frac * frac) - 0.1666666667 * (frac * frac * frac);
  interp[1] = 1.f - interp[3] - interp[2] - interp[0];
  return interp[0] * iamf_rs_q4_window[ind] + interp[1] * iamf_rs_q4_window[ind + 1] +
         interp[2] * iamf_rs_q4_window[ind + 2] + interp[3] * iamf_rs_q4_window[ind + 3];
}

float sinc_at(float cutoff, float x, int N) {
  float xx = x * cutoff;
  if (fabs(x) < 1e-6)
    return cutoff;
  else if (fabs(x) > .5 * N)
    return 0;
  return (float)(cutoff * sin(M_PI * xx) / (M_PI * xx) * window_at((float)fabs(2. * x / N)));
}

unsigned gcd_u(unsigned a, unsigned b) {
  while (b) {
    unsigned t = a;
    a = b;
    b = t % b;
  }
  return a;
}

#define RS_HIPCHK(expr)                                                                \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      fprintf(stderr, "iamf_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
              __FILE__, __LINE__);                                                     \
      return IAMF_HIP_ERR_DEVICE;                                                      \
    }                                                                                  \
  } while (0)

}  // namespace

struct iamf_hip_resampler {
  int n_streams = 0, ch = 0, in_rate = 0, out_rate = 0;
  unsigned num = 0, den = 0, filt_len = 0, oversample = 0, int_adv = 0, frac_adv = 0;
  int direct = 0;
  float cutoff = 0.f;
  // per stream (round 4: a group of decoder handles resamples streams that do not advance in step): the phase the
  // stream's next call starts from and which of the two history buffers holds its past
  std::vector<int> last_sample;
  std::vector<unsigned> frac;
  std::vector<uint8_t> cur;
  float *d_table = nullptr, *d_table4 = nullptr, *d_hist[2] = {nullptr, nullptr};
};

namespace {

// streams [s0, s0 + cnt): they must share one state (phase and history buffer) — one launch, uniform parameters
int rs_run(iamf_hip_resampler *r, const float *d_in, int64_t in_stride, int ns, float *d_out,
           int64_t out_stride, int out_len, void *stream, int s0, int cnt) {
  if (s0 < 0 || cnt <= 0 || s0 + cnt > r->n_streams) return IAMF_HIP_ERR_BAD_ARG;
  for (int i = s0 + 1; i < s0 + cnt; ++i)
    if (r->last_sample[(size_t)i] != r->last_sample[(size_t)s0] || r->frac[(size_t)i] != r->frac[(size_t)s0] ||
        r->cur[(size_t)i] != r->cur[(size_t)s0])
      return IAMF_HIP_ERR_INVALID_STATE;
  // host replica of the phase walk (resample.c:269-303): how many outputs this call yields
  int ls = r->last_sample[(size_t)s0];
  unsigned fr = r->frac[(size_t)s0];
  const int cur = r->cur[(size_t)s0];
  int n_out = 0;
  while (!(ls >= ns || n_out >= out_len)) {
    ++n_out;
    ls += (int)r->int_adv;
    fr += r->frac_adv;
    if (fr >= r->den) {
      fr -= r->den;
      ls++;
    }
  }
  const int consumed = ls < ns ? ls : ns;  // resample.c:801-804
  RsParams p;
  memset(&p, 0, sizeof(p));
  p.in = d_in;
  p.in_stream_stride = in_stride;
  p.hist = r->d_hist[cur];
  p.hist_next = r->d_hist[cur ^ 1];
  p.s0 = s0;
  p.out = d_out;
  p.out_stream_stride = out_stride;
  p.table = r->d_table;
  p.ch = r->ch;
  p.ns = ns;
  p.n_out = n_out;
  p.consumed = consumed;
  p.N = (int)r->filt_len;
  p.oversample = (int)r->oversample;
  p.direct = r->direct;
  p.num = r->num;
  p.den = r->den;
  p.int_adv = r->int_adv;
  p.ls0 = r->last_sample[(size_t)s0];
  p.fr0 = r->frac[(size_t)s0];
  const int64_t work = (int64_t)(n_out > (int)r->filt_len - 1 ? n_out : (int)r->filt_len - 1) * r->ch;
  dim3 grid((unsigned)((work + 255) / 256), (unsigned)cnt);
  // the LDS-tiled kernel where its operands fit: 256 / ch outputs advance num / den samples each, plus the taps
  const int outs = kRsTile / r->ch + 2;
  const int64_t win_cap = ((int64_t)outs * r->num + r->den - 1) / r->den + (int64_t)r->filt_len + 2;
  const size_t tab_floats = r->direct ? (size_t)r->den * (r->filt_len + 4) : (size_t)4 * r->oversample * (r->filt_len + 1);
  const size_t lds = sizeof(float) * ((((size_t)win_cap * r->ch + 3) & ~(size_t)3) + tab_floats);
  bool blocked = false;
  // direct mode: a thread = one phase with its filter row in registers (resample_direct_kernel)
  if (r->direct && r->den <= 16 && !getenv("IAMF_HIP_RESAMPLE_PLAIN") && !getenv("IAMF_HIP_RESAMPLE_TILE")) {
    const int nump = r->den == 1 ? (int)r->num : 1;
    const int R = rs_direct_r(r->ch);
    int G = 256 / (int)r->den;
    if (cnt < 256) G = G > 4 ? G / (cnt < 64 ? 4 : 2) : G;   // few streams: more, smaller tiles
    for (; G >= 1; G >>= 1) {
      const int64_t T = (int64_t)r->den * R * G;
      const int64_t wcap = (T * r->num + r->den - 1) / r->den + (int64_t)r->filt_len + 2;
      const int64_t slots = (wcap + nump - 1) / nump + 1;
      const size_t dlds = sizeof(float) * ((((size_t)nump * slots * rs_cs(r->ch) + 3) & ~(size_t)3) + (size_t)r->den * r->filt_len);
      if (dlds <= 60 * 1024) {
        p.win_cap = (int)wcap;
        const unsigned blk = (unsigned)(((int)r->den * G + 63) & ~63);
        dim3 dgrid((unsigned)(n_out > 0 ? (n_out + T - 1) / T : 1), (unsigned)cnt);
        blocked = rs_direct_launch(r->ch, (int)r->filt_len, nump, p, G, dgrid, blk, dlds, static_cast<hipStream_t>(stream));
        break;
      }
    }
  }
  // the register-blocked kernel: a thread = R outputs of one phase x all channels (resample_block_kernel)
  if (!blocked && !r->direct && r->d_table4 != nullptr && r->den <= 512 && !getenv("IAMF_HIP_RESAMPLE_PLAIN") &&
      !getenv("IAMF_HIP_RESAMPLE_TILE")) {
    // G groups of den threads: the fullest whole waves within 512 threads, nearest to 256 among equals
    int G = 1, best_num = 0, best_blk = 64;
    for (int g2 = 1; (int)r->den * g2 <= 512; ++g2) {
      const int used = (int)r->den * g2, blk = (used + 63) & ~63;
      if ((int64_t)used * best_blk > (int64_t)best_num * blk ||
          ((int64_t)used * best_blk == (int64_t)best_num * blk && abs(blk - 256) < abs(best_blk - 256))) {
        G = g2;
        best_num = used;
        best_blk = blk;
      }
    }
    int R = cnt >= 256 ? 4 : (cnt >= 64 ? 2 : 1);   // few streams: more, smaller tiles
    while (R > rs_block_r(r->ch)) R >>= 1;
    for (;;) {
      const int64_t T = (int64_t)r->den * R * G;
      const int64_t wcap = (T * r->num + r->den - 1) / r->den + (int64_t)r->filt_len + 2;
      const size_t blds = sizeof(float) * ((((size_t)wcap * rs_cs(r->ch) + 3) & ~(size_t)3) + tab_floats);
      if (blds <= 60 * 1024) {
        p.table4 = r->d_table4;
        p.win_cap = (int)wcap;
        const unsigned blk = (unsigned)(((int)r->den * G + 63) & ~63);
        dim3 bgrid((unsigned)(n_out > 0 ? (n_out + T - 1) / T : 1), (unsigned)cnt);
        blocked = rs_block_launch(r->ch, R, p, G, bgrid, blk, blds, static_cast<hipStream_t>(stream));
        break;
      }
      if (R > 1) R >>= 1;
      else if (G > 1) --G;
      else break;
    }
  }
  if (blocked) {
  } else if ((r->direct ? (r->filt_len & 7) == 0 : r->d_table4 != nullptr) && lds <= 48 * 1024 && !getenv("IAMF_HIP_RESAMPLE_PLAIN")) {
    p.table4 = r->d_table4;
    p.win_cap = (int)win_cap;
    dim3 tgrid((unsigned)((work + kRsTile - 1) / kRsTile), (unsigned)cnt);
    hipLaunchKernelGGL(resample_tile_kernel, tgrid, dim3(256), lds, static_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
  }
  RS_HIPCHK(hipGetLastError());
  for (int i = s0; i < s0 + cnt; ++i) {
    r->last_sample[(size_t)i] = ls - consumed;
    r->frac[(size_t)i] = fr;
    r->cur[(size_t)i] = (uint8_t)(cur ^ 1);
  }
  return n_out;
}

}  // namespace

extern "C" {

int iamf_hip_resampler_create(int n_streams, int channels, int in_rate, int out_rate,
                              iamf_hip_resampler **out) {
  if (!out || n_streams <= 0 || channels <= 0 || channels > 24 || in_rate <= 0 || out_rate <= 0)
    return IAMF_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  RS_HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return IAMF_HIP_ERR_DEVICE;
  iamf_hip_resampler *r = new (std::nothrow) iamf_hip_resampler();
  if (!r) return IAMF_HIP_ERR_ALLOC_FAIL;
  r->n_streams = n_streams;
  r->ch = channels;
  r->in_rate = in_rate;
  r->out_rate = out_rate;
  const unsigned g = gcd_u((unsigned)in_rate, (unsigned)out_rate);
  r->num = (unsigned)in_rate / g;
  r->den = (unsigned)out_rate / g;
  r->int_adv = r->num / r->den;
  r->frac_adv = r->num % r->den;
  r->oversample = IAMF_RS_Q4_OVERSAMPLE;
  r->filt_len = IAMF_RS_Q4_BASE_LENGTH;
  if (r->num > r->den) {  // down-sampling: longer, narrower filter (resample.c:539-552)
    r->cutoff = IAMF_RS_Q4_DOWN_BW * r->den / r->num;
    r->filt_len = (unsigned)((unsigned long long)r->filt_len * r->num / r->den);
    r->filt_len = ((r->filt_len - 1) & (~0x7U)) + 8;
    if (2 * r->den < r->num) r->oversample >>= 1;
    if (4 * r->den < r->num) r->oversample >>= 1;
    if (8 * r->den < r->num) r->oversample >>= 1;
    if (16 * r->den < r->num) r->oversample >>= 1;
    if (r->oversample < 1) r->oversample = 1;
  } else {
    r->cutoff = IAMF_RS_Q4_UP_BW;
  }
  r->direct = r->filt_len * r->den <= r->filt_len * r->oversample + 8;
  std::vector<float> tab;
  if (r->direct) {
    tab.resize((size_t)r->filt_len * r->den);
    for (unsigned i = 0; i < r->den; i++)
      for (int j = 0; j < (int)r->filt_len; j++)
        tab[(size_t)i * r->filt_len + j] =
            sinc_at(r->cutoff, ((j - (int)r->filt_len / 2 + 1) - ((float)i) / r->den), (int)r->filt_len);
  } else {
    tab.resize((size_t)r->filt_len * r->oversample + 8);
    for (int i = -4; i < (int)(r->oversample * r->filt_len + 4); i++)
      tab[i + 4] = sinc_at(r->cutoff, (i / (float)r->oversample - r->filt_len / 2), (int)r->filt_len);
  }
  std::vector<float> tab4;
  if (!r->direct) {  // resample_tile_kernel: the four values tap j multiplies at interpolation offset o, contiguous
    tab4.resize((size_t)4 * r->oversample * r->filt_len);
    for (unsigned o = 0; o < r->oversample; ++o)
      for (unsigned j = 0; j < r->filt_len; ++j)
        for (int q = 0; q < 4; ++q)
          tab4[((size_t)o * r->filt_len + j) * 4 + q] = tab[4 + (size_t)(j + 1) * r->oversample - o + (q - 2)];
  }
  r->last_sample.assign((size_t)n_streams, (int)(r->filt_len / 2));  // speex_resampler_skip_zeros (IAMF_decoder.c:1902)
  r->frac.assign((size_t)n_streams, 0u);
  r->cur.assign((size_t)n_streams, 0);
  const size_t hist_bytes = sizeof(float) * (size_t)n_streams * (r->filt_len - 1) * channels;
  if (hipMalloc(&r->d_table, sizeof(float) * tab.size()) != hipSuccess ||
      hipMalloc(&r->d_hist[0], hist_bytes) != hipSuccess || hipMalloc(&r->d_hist[1], hist_bytes) != hipSuccess ||
      hipMemcpy(r->d_table, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) != hipSuccess ||
      (!tab4.empty() && (hipMalloc(&r->d_table4, sizeof(float) * tab4.size()) != hipSuccess ||
                         hipMemcpy(r->d_table4, tab4.data(), sizeof(float) * tab4.size(), hipMemcpyHostToDevice) != hipSuccess)) ||
      hipMemset(r->d_hist[0], 0, hist_bytes) != hipSuccess || hipMemset(r->d_hist[1], 0, hist_bytes) != hipSuccess) {
    iamf_hip_resampler_destroy(r);
    return IAMF_HIP_ERR_DEVICE;
  }
  *out = r;
  return IAMF_HIP_OK;
}

void iamf_hip_resampler_destroy(iamf_hip_resampler *r) {
  if (!r) return;
  (void)hipFree(r->d_table);
  (void)hipFree(r->d_table4);
  (void)hipFree(r->d_hist[0]);
  (void)hipFree(r->d_hist[1]);
  delete r;
}

int iamf_hip_resampler_out_capacity(const iamf_hip_resampler *r, int ns) {
  return r ? ns * (r->out_rate / r->in_rate + 1) : 0;
}

int iamf_hip_resampler_flush_capacity(const iamf_hip_resampler *r) {
  return r ? (int)(((r->filt_len / 2) * r->den + (r->num >> 1)) / r->num) : 0;
}

int iamf_hip_resampler_process(iamf_hip_resampler *r, const float *d_in, int64_t in_stream_stride, int ns,
                               float *d_out, int64_t out_stream_stride, void *stream) {
  if (!r || !d_in || !d_out || ns < 0) return IAMF_HIP_ERR_BAD_ARG;
  const int cap = iamf_hip_resampler_out_capacity(r, ns);
  if (r->n_streams > 1 && out_stream_stride < (int64_t)cap * r->ch) return IAMF_HIP_ERR_BUFFER_TOO_SMALL;
  return rs_run(r, d_in, in_stream_stride, ns, d_out, out_stream_stride, cap, stream, 0, r->n_streams);
}

int iamf_hip_resampler_flush(iamf_hip_resampler *r, float *d_out, int64_t out_stream_stride, void *stream) {
  if (!r || !d_out) return IAMF_HIP_ERR_BAD_ARG;
  const int cap = iamf_hip_resampler_flush_capacity(r);
  return rs_run(r, nullptr, 0, (int)(r->filt_len / 2), d_out, out_stream_stride, cap, stream, 0, r->n_streams);
}

// the same for the streams [stream0, stream0 + n_streams) only; buffers and strides are indexed by the stream's number
int iamf_hip_resampler_process_range(iamf_hip_resampler *r, const float *d_in, int64_t in_stream_stride, int ns, float *d_out,
                                     int64_t out_stream_stride, void *stream, int32_t stream0, int32_t n_streams) {
  if (!r || !d_in || !d_out || ns < 0) return IAMF_HIP_ERR_BAD_ARG;
  const int cap = iamf_hip_resampler_out_capacity(r, ns);
  if (r->n_streams > 1 && out_stream_stride < (int64_t)cap * r->ch) return IAMF_HIP_ERR_BUFFER_TOO_SMALL;
  return rs_run(r, d_in, in_stream_stride, ns, d_out, out_stream_stride, cap, stream, stream0, n_streams);
}

int iamf_hip_resampler_flush_range(iamf_hip_resampler *r, float *d_out, int64_t out_stream_stride, void *stream, int32_t stream0,
                                   int32_t n_streams) {
  if (!r || !d_out) return IAMF_HIP_ERR_BAD_ARG;
  const int cap = iamf_hip_resampler_flush_capacity(r);
  return rs_run(r, nullptr, 0, (int)(r->filt_len / 2), d_out, out_stream_stride, cap, stream, stream0, n_streams);
}

int iamf_hip_resampler_same_state(const iamf_hip_resampler *r, int32_t a, int32_t b) {
  if (!r || a < 0 || b < 0 || a >= r->n_streams || b >= r->n_streams) return 0;
  return r->last_sample[(size_t)a] == r->last_sample[(size_t)b] && r->frac[(size_t)a] == r->frac[(size_t)b] &&
         r->cur[(size_t)a] == r->cur[(size_t)b];
}

}  // extern "C"
