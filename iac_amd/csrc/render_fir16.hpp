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
// consecutive k of each operand as ONE 16-byte LDS read:
//   B (input): x[16T + 15 - 32s - 8g - j], j = 0..7  -> the slice is stored REVERSED, position
//      q = 1279 - u for slice sample u, so the run starts at q0 = 1008 - 256ct - 16T + 32s + 8g, a
//      multiple of 8 halves; the wave's 64 reads cover one contiguous stretch (conflict-free);
//   A (filter): hp[32s + 8g + i + j]: the phase i misaligns it, so the table is kept in 8 copies
//      shifted by r = i & 7 (built once per batch on the host, staged per channel by plain copies).
// All eight waves work on ONE channel at a time; a wave takes a 2 x 2 block of output tiles (both ears
// x two 256-sample column tiles: 8 operand reads feed 12 MFMAs).  Slice and tables of the next channel
// are fetched into registers during the current channel's MFMAs and stored to LDS between two barriers.
//
// Measured (16 channels x 256 taps, MI355X): 15.4 Gsamples/s against 6.2 for the f32 stage.  How it got
// there: the first version (one tile per wave, one chunk per pass, tables double-buffered) ran at 9.1
// and its time was NOT in the matrix cores (no MFMA loop at all: 10.7) but in moving the shifted tables
// from L2 into LDS, 19.5 KB per channel and chunk (no loads: 24.7; no LDS stores either: 92).  2 x 2
// tiles per wave (LDS reads per MFMA halved): 9.3; two chunks per pass: 12.5; four chunks per pass with
// single-buffered tables (what fits 80 KB): 15.4.  Tried: ONE table copy per channel read at its
// 2-byte boundary, all 16 channels resident in LDS (no table traffic at all) — correct, but a
// ds_read_b128 that is not 16-byte aligned is served one lane per cycle (tools/lds_unaligned_probe.hip:
// 64 instead of 23 cycles per wave-read at 2, 4 or 8 bytes off): 3.7; with the reads forced aligned
// (wrong taps) 13.5 at one chunk per pass.
#pragma once

constexpr int kF16Taps = 304;                 // padded hp table, halves (see render_fir.hpp)
constexpr int kF16Span = 4096;                // samples per pass of the stage: FOUR chunks (see below)
constexpr int kF16Slice = kF16Span + 256 + 32;  // reversed slice: history + samples + 32 zeros of padding
constexpr int kF16HBytes = 2 * 2 * 8 * kF16Taps * 2;   // [ear][hi/lo][shift][taps] halves = 19456 B per channel
constexpr int kF16XBytes = 2 * kF16Slice * 2;          // slice hi + lo: 17536 B
constexpr int kF16LdsFloats = (kF16XBytes + kF16HBytes) / 4;  // slice + tables: 9248 floats
constexpr int kF16Part = kFirChunk + 32;      // one (ear, chunk) row of sums
static_assert(8 * kF16Part <= kF16LdsFloats, "the sums alias the staging area");
constexpr float kF16InScale = 64.f;           // input scale 2^6: |x| < 1023 stays finite, -120 dB stays normal

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// What the stage costs is moving the shifted tables (19.5 KB per channel) from L2 into LDS, so one pass
// covers FOUR chunks: the caller runs it on every fourth chunk and reads the other chunks' sums from
// the same place in the following iterations.  Wave w takes column tiles 2w and 2w + 1 of the 16, both
// ears, all K steps (tiles past the end of the call are skipped).  y[e][c0 .. c0+4096) goes to part
// ([2 ears][4 chunks][1024 + 32] floats, padded by one per 32; aliases the staging area, which is dead
// by then).  All 512 threads must call it.
template <int M>
__device__ __forceinline__ void fir_stage16(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                            float *fir, float *part) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int t = threadIdx.x;  // 0..511
  const int w = t >> 6, lane = t & 63;
  const int ctp = w;  // column tiles 2 ctp, 2 ctp + 1 of the 16
  const bool work = c0 + 512 * ctp < p.total;  // else both tiles lie past the end of the call
  const int col = lane & 15, g = lane >> 4;
  const int L = p.fir_taps;
  const int KS = (L + 15 + 31) >> 5;  // steps of 32 taps over m' in [0, L + 14]; <= 9
  unsigned char *xbuf = reinterpret_cast<unsigned char *>(fir);  // slice: [hi/lo][kF16Slice] halves
  unsigned char *hbuf = xbuf + kF16XBytes;                       // tables: [kF16HBytes]

  // where this thread's slice quads come from (the same for every channel): quad j = t + 512 r covers
  // slice positions u = 4 (j - 8) .. + 3, sample n = c0 - 256 + u; quads 0..7 are the zero padding
  constexpr int kNoQuad = -(1 << 30);
  int xoff[3];  // >= 0: offset in the channel's plane of the call's input; -1: zeros; <= -2: history; kNoQuad: none
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int j = t + 512 * r;
    const int n = c0 - kFirHist + 4 * (j - 8);
    xoff[r] = j < 8 + (kF16Span + 256) / 4 ? -1 : kNoQuad;
    if (j >= 8 && xoff[r] == -1) {
      if (n < 0) {
        xoff[r] = -2 - (kFirHist + n);
      } else if (n < p.total) {
        const int f = n / p.frame_size;
        xoff[r] = (int)(f * p.in_frame_stride) + (n - f * p.frame_size);
      }
    }
  }
  float4 xr[3];
  uint4 hr[3];
  auto fetch = [&](int ch) {  // global -> registers
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      xr[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xoff[r] >= 0) {  // streamed once: non-temporal, so that it does not push the shared filter tables out of L2
        using v4 = __attribute__((ext_vector_type(4))) float;
        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(in_s + (int64_t)ch * p.frame_size + xoff[r]));
        xr[r] = make_float4(v.x, v.y, v.z, v.w);
      }
      else if (xoff[r] <= -2 && xoff[r] != kNoQuad) xr[r] = *reinterpret_cast<const float4 *>(hist + ch * kFirHist + (-2 - xoff[r]));
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const unsigned char *>(p.fir_h16) + (size_t)ch * kF16HBytes);
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (t + 512 * r < kF16HBytes / 16) hr[r] = src[t + 512 * r];
  };
  auto stash_h = [&]() {  // tables: registers -> LDS
    uint4 *dst = reinterpret_cast<uint4 *>(hbuf);
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (t + 512 * r < kF16HBytes / 16) dst[t + 512 * r] = hr[r];
  };
  auto stash_x = [&]() {  // slice: registers -> LDS, f32 -> hi/lo f16, reversed
    _Float16 *xh = reinterpret_cast<_Float16 *>(xbuf), *xl = xh + kF16Slice;
    // |x| >= 1023.5 (60 dB over full scale) saturates instead of becoming an f16 infinity
    auto sat = [](float a) { return fminf(fmaxf(a * kF16InScale, -65504.f), 65504.f); };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (xoff[r] == kNoQuad) continue;
      const float v[4] = {sat(xr[r].x), sat(xr[r].y), sat(xr[r].z), sat(xr[r].w)};
      f16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // slice sample u + k sits at q = (kF16Span + 255) - u - k: reversed inside the quad
        const _Float16 h = (_Float16)v[k];
        hi[3 - k] = h;
        lo[3 - k] = (_Float16)((v[k] - (float)h) * 2048.f);
      }
      const int u = 4 * (t + 512 * r - 8);
      const int q = (kF16Span + 255) - u - 3;  // u = -32 .. kF16Span + 252 -> q = kF16Slice - 4 .. 0
      *reinterpret_cast<f16x4 *>(xh + q) = hi;
      *reinterpret_cast<f16x4 *>(xl + q) = lo;
    }
  };

  f32x4 acc_hh[2][2], acc_x[2][2];  // [ear][tile of the pair]: hi*hi and the cross terms
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc_hh[e][c] = acc_x[e][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  stash_h();
  stash_x();
  __syncthreads();
  for (int ch = 0; ch < M; ++ch) {
    if (ch + 1 < M) fetch(ch + 1);
    if (work) {
      const _Float16 *xh = reinterpret_cast<const _Float16 *>(xbuf), *xl = xh + kF16Slice;
      const _Float16 *hb = reinterpret_cast<const _Float16 *>(hbuf);
      // filter: [ear][hi/lo][shift r = col & 7][taps]; the lane starts at 8g + (col & 8)
      const _Float16 *a0 = hb + (col & 7) * kF16Taps + 8 * g + (col & 8);
      // slice sample u = 256 + 256 tile + 16 col + 15 - m' sits at q = kF16Span + 255 - u
      const int q0 = (kF16Span - 16) - 512 * ctp - 16 * col + 8 * g;  // first tile of the pair; the second: - 256
      for (int s = 0; s < KS; ++s) {
        f16x8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          a_hi[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 0) * 8 * kF16Taps + 32 * s);
          a_lo[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 1) * 8 * kF16Taps + 32 * s);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          b_hi[c] = *reinterpret_cast<const f16x8 *>(xh + q0 - 256 * c + 32 * s);
          b_lo[c] = *reinterpret_cast<const f16x8 *>(xl + q0 - 256 * c + 32 * s);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            acc_hh[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[e], b_hi[c], acc_hh[e][c], 0, 0, 0);
            acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[e], b_lo[c], acc_x[e][c], 0, 0, 0);
            acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[e], b_hi[c], acc_x[e][c], 0, 0, 0);
          }
      }
    }
    __syncthreads();  // everybody has read this channel's slice and tables
    if (ch + 1 < M) {
      stash_h();
      stash_x();
      __syncthreads();
    }
  }
  // D[row = phase][col = block]: lane holds block col of a tile, phases 4g + r: four consecutive samples
  const float sc = p.fir_inv_scale;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int tile = 2 * ctp + c;  // chunk tile >> 2, 256-sample tile (tile & 3) inside it
      float *pw = part + (e * 4 + (tile >> 2)) * kF16Part;
      const int nn = 256 * (tile & 3) + 16 * col + 4 * g;
      const int uu = nn + (nn >> 5);
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[uu + r] = (acc_hh[e][c][r] + acc_x[e][c][r] * (1.f / 2048.f)) * sc;
    }
  __syncthreads();
}
