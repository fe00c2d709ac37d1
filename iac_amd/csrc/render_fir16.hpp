// render_fir16.hpp — the binaural HRTF stage of render_fir.hpp on the f16 matrix cores
// (v_mfma_f32_16x16x32_f16, 16x the rate of the f32 MFMA) without giving up f32 accuracy: every
// operand is split into two halves,  v * scale = hi + lo * 2^-11  (hi, lo in f16), and
//     h * x  =  hi_h*hi_x + (hi_h*lo_x + lo_h*hi_x) * 2^-11          [ + lo*lo * 2^-22, dropped ]
// costs three MFMAs whose products are exact in the f32 accumulators.  What is dropped is 2^-22
// of a product, the same order as the f32 rounding of the sum itself (tests: 2^-17 absolute against
// a float64 convolution, like the f32 stage).  PARITY UNPINNED, see render_fir.hpp.
//
// Mapping (as render_fir.hpp: t = 16T + i, D[i][T] = sum_m' hp[m' + i] * x[16T + 15 - m'], hp[j] =
// h[j - 15]) with K = 32 taps per MFMA.  Lane (i or T = lane & 15, g = lane >> 4) needs 8
// consecutive k of each operand, one 16-byte LDS read each:
//   B (input): x[16T + 15 - 32s - 8g - j], j = 0..7  -> the slice is stored REVERSED, position
//      q = 1279 - u for slice sample u, so the run starts at q0 = 1008 - 256ct - 16T + 32s + 8g, a
//      multiple of 8 halves; the wave's 64 reads cover one contiguous stretch (conflict-free);
//   A (filter): hp[32s + 8g + i + j]: the phase i puts the run at any 2-byte boundary.  gfx950 reads
//      16 bytes from LDS at any alignment (unaligned access mode; tools/lds_unaligned_probe.hip:
//      correct data, same latency), so ONE copy of each table is enough — 2.5 KB per channel, and
//      the tables of all 16 channels stay in LDS for the whole kernel (loaded once per workgroup).
// All eight waves work on ONE channel at a time: wave w = (ear w & 1, column tile w >> 1) keeps two
// accumulators (hi*hi and the cross terms) for its 256 samples of its ear across the channel loop,
// so nothing is summed across waves.  Per channel only the input slice is staged (f32 -> hi/lo f16,
// fetched one channel ahead), and the operands of a step are read while the previous step's MFMAs run.
#pragma once

constexpr int kF16Taps = 320;                 // padded hp table, halves (largest index read: 302)
constexpr int kF16Slice = 1312;               // reversed slice: 1280 samples + 32 zeros of padding
constexpr int kF16HBytes = 2 * 2 * kF16Taps * 2;   // [ear][hi/lo][taps] halves = 2560 B per channel
constexpr int kF16XBytes = 2 * kF16Slice * 2;      // slice hi + lo: 5248 B
__host__ __device__ constexpr int f16_lds_floats(int m) { return (m * kF16HBytes + 2 * kF16XBytes) / 4; }
constexpr float kF16InScale = 64.f;           // input scale 2^6: |x| < 1023 stays finite, -120 dB stays normal

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// 16 bytes from LDS at any 2-byte boundary: a plain vector load through a pointer the compiler takes
// to be 16-byte aligned, so that it emits one ds_read_b128 (and keeps track of it for s_waitcnt).
__device__ __forceinline__ f16x8 fir16_read(const _Float16 *p) { return *reinterpret_cast<const f16x8 *>(p); }

// The filter tables of all channels -> LDS, once per workgroup (all 512 threads; the caller's next
// barrier publishes them).  tab = this kernel's fir scratch.
template <int M>
__device__ __forceinline__ void fir16_load_tables(const RenderParams &p, float *fir) {
  const uint4 *src = static_cast<const uint4 *>(p.fir_h16);
  uint4 *dst = reinterpret_cast<uint4 *>(fir);
  for (int i = threadIdx.x; i < M * kF16HBytes / 16; i += 512) dst[i] = src[i];
}

// y[e][c0 .. c0+1024) for both ears into part ([2][1024 + 32] floats, padded by one per 32; aliases the
// slice buffers, which are dead by then).  All 512 threads must call it.
template <int M>
__device__ __forceinline__ void fir_stage16(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                            float *fir) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int t = threadIdx.x;  // 0..511
  const int w = t >> 6, lane = t & 63;
  const int ear = w & 1, ct = w >> 1;
  const int col = lane & 15, g = lane >> 4;
  const int L = p.fir_taps;
  const int KS = (L + 15 + 31) >> 5;  // steps of 32 taps over m' in [0, L + 14]; <= 9
  unsigned char *hb = reinterpret_cast<unsigned char *>(fir);  // [M][ear][hi/lo][320] halves
  unsigned char *xb = hb + M * kF16HBytes;                      // [2 buffers][hi/lo][1312] halves
  float *part = reinterpret_cast<float *>(xb);

  // where this thread's 4 slice samples come from (the same for every channel): slice position
  // u = 4 (t - 8), sample n = c0 - 256 + u; threads 0..7 write the 32 halves of zero padding
  const bool xs_on = t < 8 + 320;
  const int u = 4 * (t - 8);
  const int n = c0 - kFirHist + u;
  int xoff = -1;  // >= 0: offset in the channel's plane of the call's input; -1: zeros; <= -2: history
  if (t >= 8 && xs_on) {
    if (n < 0) {
      xoff = -2 - (kFirHist + n);
    } else if (n < p.total) {
      const int f = n / p.frame_size;
      xoff = (int)(f * p.in_frame_stride) + (n - f * p.frame_size);
    }
  }
  float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int ch) {  // global -> registers
    if (xs_on) {
      xr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xoff >= 0) xr = *reinterpret_cast<const float4 *>(in_s + (int64_t)ch * p.frame_size + xoff);
      else if (xoff <= -2) xr = *reinterpret_cast<const float4 *>(hist + ch * kFirHist + (-2 - xoff));
    }
  };
  auto stash = [&](int b) {  // registers -> slice buffer b
    if (xs_on) {
      _Float16 *xh = reinterpret_cast<_Float16 *>(xb + b * kF16XBytes), *xl = xh + kF16Slice;
      const float v[4] = {xr.x * kF16InScale, xr.y * kF16InScale, xr.z * kF16InScale, xr.w * kF16InScale};
      f16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // slice sample u + k sits at q = 1279 - u - k: reversed inside the quad
        const _Float16 h = (_Float16)v[k];
        hi[3 - k] = h;
        lo[3 - k] = (_Float16)((v[k] - (float)h) * 2048.f);
      }
      const int q = 1276 - u;  // = 1279 - u - 3; u = -32 .. 1276 -> q = 1308 .. 0
      *reinterpret_cast<f16x4 *>(xh + q) = hi;
      *reinterpret_cast<f16x4 *>(xl + q) = lo;
    }
  };

  f32x4 acc_hh = {0.f, 0.f, 0.f, 0.f}, acc_x = {0.f, 0.f, 0.f, 0.f};
  // the lane's operand runs at step 0 of channel 0 / buffer 0
  const _Float16 *a0 = reinterpret_cast<const _Float16 *>(hb) + ear * 2 * kF16Taps + 8 * g + col;
  const _Float16 *b0 = reinterpret_cast<const _Float16 *>(xb) + (1008 - 256 * ct - 16 * col + 8 * g);
  fetch(0);
  stash(0);
  __syncthreads();
  for (int ch = 0; ch < M; ++ch) {
    if (ch + 1 < M) fetch(ch + 1);
    {
      const _Float16 *aa = a0 + ch * (kF16HBytes / 2), *bb = b0 + (ch & 1) * (kF16XBytes / 2);
      f16x8 a_hi = fir16_read(aa), a_lo = fir16_read(aa + kF16Taps);
      f16x8 b_hi = fir16_read(bb), b_lo = fir16_read(bb + kF16Slice);
      for (int s = 0; s < KS; ++s) {
        const f16x8 ah = a_hi, al = a_lo, bh = b_hi, bl = b_lo;
        if (s + 1 < KS) {  // the next step's operands, read while this step's MFMAs run
          a_hi = fir16_read(aa + 32 * (s + 1));
          a_lo = fir16_read(aa + kF16Taps + 32 * (s + 1));
          b_hi = fir16_read(bb + 32 * (s + 1));
          b_lo = fir16_read(bb + kF16Slice + 32 * (s + 1));
        }
        acc_hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc_hh, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_x, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_x, 0, 0, 0);
      }
    }
    if (ch + 1 < M) stash((ch + 1) & 1);  // the other buffer: its readers finished before the last barrier
    __syncthreads();
  }
  // D[row = phase][col = block]: lane holds block col of its tile, phases 4g + r: four consecutive samples
  float *pw = part + ear * (kFirChunk + 32);
  const int nn = 256 * ct + 16 * col + 4 * g;
  const int uu = nn + (nn >> 5);
  const float sc = p.fir_inv_scale;
#pragma unroll
  for (int r = 0; r < 4; ++r) pw[uu + r] = (acc_hh[r] + acc_x[r] * (1.f / 2048.f)) * sc;
  __syncthreads();
}
