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
// All four waves work on ONE channel at a time; a wave takes a 2 x 4 block of output tiles (both ears
// x four 256-sample column tiles: 12 operand reads feed 24 MFMAs).  Slice and tables of the next channel
// are fetched into registers during the current channel's MFMAs and stored to LDS between two barriers.
//
// Measured (16 channels x 256 taps, MI355X): 15.4 Gsamples/s against 6.2 for the f32 stage.  How it got
// there: the first version (one tile per wave, one chunk per pass, tables double-buffered) ran at 9.1
// and its time was NOT in the matrix cores (no MFMA loop at all: 10.7) but in moving the shifted tables
// from L2 into LDS, 19.5 KB per channel and chunk (no loads: 24.7; no LDS stores either: 92).  2 x 2
// tiles per wave (LDS reads per MFMA halved): 9.3; two chunks per pass: 12.5; four chunks per pass with
// single-buffered tables (what fits 80 KB): 15.4.  Tried: ONE table copy per channel read at its
// 2-byte boundary, all 16 channels resident in LDS (no table traffic at all) — correct, but a
// ds_read_b128 that is not 16-byte aligned is served one lane per cycle (tools/debug/lds_unaligned_probe.hip:
// 64 instead of 23 cycles per wave-read at 2, 4 or 8 bytes off): 3.7; with the reads forced aligned
// (wrong taps) 13.5 at one chunk per pass.
// Round 2: four waves x (2 ears x 4 tiles), two operand sets, branch-free prefetch, no scratch: 18.2-18.8
// (see fir_stage16 below); from there on the kernel is power-limited (NOTEBOOK.md 4.2, profiles/r02_fir16/).
#pragma once

// IAMF_F16_EXP: timing-only elimination builds (WRONG results; the product is 0):
//   1 = slice and tables staged for channel 0 only (no per-channel fetch / LDS stores / second barrier)
//   3 = no MFMAs (reads kept)
//   4 = no K loop at all (staging and barriers only)   5 = the stage returns at once (the rest of the kernel)
#ifndef IAMF_F16_EXP
#define IAMF_F16_EXP 0
#endif
// hooks of tools/debug/fir16_stage_probe.hip (per-phase s_memtime stamps); nothing in the product
#ifndef IAMF_F16_STAMP
#define IAMF_F16_STAMP_DECL
#define IAMF_F16_STAMP(k)
#define IAMF_F16_STAMP_END
#endif
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

// One pass covers FOUR chunks (the caller runs it on every fourth chunk and reads the other chunks' sums from
// the same place in the following iterations): the shifted tables of a channel (19.5 KB from L2) are staged
// once per 4096 samples.  The workgroup is 4 waves, one per SIMD, two workgroups per CU: 256 VGPRs per wave.
// Wave w takes the four 256-sample column tiles of chunk w of the span, both ears: per K step 4 filter + 8
// slice reads (16 bytes per lane each) feed 24 MFMAs, and the reads of step s + 1 are issued before the MFMAs
// of step s (two operand sets).  [The first version ran 8 waves x (2 ears x 2 tiles) at 128 VGPRs: the table
// prefetch was spilled to scratch behind a vmcnt(0) and the K loop waited for every LDS read where it was
// issued — tools/debug/f16_exp.sh, profiles/r02_fir16/.]
// y[e][c0 .. c0+4096) goes to part ([2 ears][4 chunks][1024 + 32] floats, padded by one per 32; aliases the
// staging area, which is dead by then).  All 256 threads must call it.
template <int M>
__device__ __forceinline__ void fir_stage16(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                            float *fir, float *part) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  const int t = threadIdx.x;  // 0..255
  const int w = t >> 6, lane = t & 63;
  if (IAMF_F16_EXP == 5) return;
  const bool work = IAMF_F16_EXP != 4 && c0 + kFirChunk * w < p.total;  // else the wave's chunk lies past the end of the call
  const int col = lane & 15, g = lane >> 4;
  const int L = p.fir_taps;
  const int KS = (L + 15 + 31) >> 5;  // steps of 32 taps over m' in [0, L + 14]; 1..9
  unsigned char *xbuf = reinterpret_cast<unsigned char *>(fir);  // slice: [hi/lo][kF16Slice] halves
  unsigned char *hbuf = xbuf + kF16XBytes;                       // tables: [kF16HBytes]

  // where this thread's slice quads come from (the same for every channel): quad j = t + 256 r covers
  // slice positions u = 4 (j - 8) .. + 3, sample n = c0 - 256 + u; quads 0..7 are the zero padding
  constexpr int kQuads = 8 + (kF16Span + 256) / 4, NQ = (kQuads + 255) / 256;  // 1096 quads: 5 per thread
  constexpr int kHVec = kF16HBytes / 16, NH = (kHVec + 255) / 256;             // 1216 x 16 bytes: 5 per thread
  constexpr int kNoQuad = -(1 << 30);
  int xoff[NQ];  // >= 0: offset in the channel's plane of the call's input; -1: zeros; <= -2: history; kNoQuad: none
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    const int j = t + 256 * r;
    const int n = c0 - kFirHist + 4 * (j - 8);
    xoff[r] = j < kQuads ? -1 : kNoQuad;
    if (j >= 8 && xoff[r] == -1) {
      if (n < 0) {
        xoff[r] = -2 - (kFirHist + n);
      } else if (n < p.total) {
        const int f = n / p.frame_size;
        xoff[r] = (int)(f * p.in_frame_stride) + (n - f * p.frame_size);
      }
    }
  }
  // (native vector types: arrays of HIP's float4 / uint4 structs indexed like this end up in scratch memory)
  using v4 = __attribute__((ext_vector_type(4))) float;
  using u4 = __attribute__((ext_vector_type(4))) unsigned;
  v4 xr[NQ];
  u4 hr[NH];
  // Every thread issues the same ten loads for every channel, no branches: with the loads in conditional blocks
  // the compiler separated them by s_waitcnt vmcnt(0) (a quarter of the kernel's time went to ISSUING them —
  // tools/debug/fir16_stage_probe.hip).  A quad that is padding or lies past the end of the call loads the channel's
  // first quad instead and is zeroed when it is stored to LDS.
  auto fetch = [&](int ch) {  // global -> registers
    const float *in_c = in_s + (int64_t)ch * p.frame_size;
    const float *hist_c = hist + ch * kFirHist;
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      const float *src = xoff[r] >= 0 ? in_c + xoff[r] : ((xoff[r] <= -2 && xoff[r] != kNoQuad) ? hist_c + (-2 - xoff[r]) : in_c);
      // streamed once: non-temporal, so that it does not push the shared filter tables out of L2
      xr[r] = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(src));
    }
    const u4 *tab = reinterpret_cast<const u4 *>(static_cast<const unsigned char *>(p.fir_h16) + (size_t)ch * kF16HBytes);
#pragma unroll
    for (int r = 0; r < NH; ++r) hr[r] = tab[t + 256 * r < kHVec ? t + 256 * r : kHVec - 1];
  };
  auto stash_h = [&]() {  // tables: registers -> LDS
    u4 *dst = reinterpret_cast<u4 *>(hbuf);
#pragma unroll
    for (int r = 0; r < NH; ++r)
      if (t + 256 * r < kHVec) dst[t + 256 * r] = hr[r];
  };
  auto stash_x = [&]() {  // slice: registers -> LDS, f32 -> hi/lo f16, reversed
    _Float16 *xh = reinterpret_cast<_Float16 *>(xbuf), *xl = xh + kF16Slice;
    // |x| >= 1023.5 (60 dB over full scale) saturates instead of becoming an f16 infinity
    auto sat = [](float a) { return fminf(fmaxf(a * kF16InScale, -65504.f), 65504.f); };
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      if (xoff[r] == kNoQuad) continue;
      const v4 xv = xoff[r] == -1 ? v4{0.f, 0.f, 0.f, 0.f} : xr[r];
      const float v[4] = {sat(xv.x), sat(xv.y), sat(xv.z), sat(xv.w)};
      f16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // slice sample u + k sits at q = (kF16Span + 255) - u - k: reversed inside the quad
        const _Float16 h = (_Float16)v[k];
        hi[3 - k] = h;
        lo[3 - k] = (_Float16)((v[k] - (float)h) * 2048.f);
      }
      const int u = 4 * (t + 256 * r - 8);
      const int q = (kF16Span + 255) - u - 3;  // u = -32 .. kF16Span + 252 -> q = kF16Slice - 4 .. 0
      *reinterpret_cast<f16x4 *>(xh + q) = hi;
      *reinterpret_cast<f16x4 *>(xl + q) = lo;
    }
  };

  f32x4 acc_hh[2][4], acc_x[2][4];  // [ear][tile of the chunk]: hi*hi and the cross terms
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc_hh[e][c] = acc_x[e][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Ops {
    f16x8 a_hi[2], a_lo[2], b_hi[4], b_lo[4];
  };
  const _Float16 *xh = reinterpret_cast<const _Float16 *>(xbuf), *xl = xh + kF16Slice;
  // filter: [ear][hi/lo][shift r = col & 7][taps]; the lane starts at 8g + (col & 8)
  const _Float16 *a0 = reinterpret_cast<const _Float16 *>(hbuf) + (col & 7) * kF16Taps + 8 * g + (col & 8);
  // slice sample u = 256 + 256 tile + 16 col + 15 - m' sits at q = kF16Span + 255 - u; tile = 4w + c
  const int q0 = (kF16Span - 16) - kFirChunk * w - 16 * col + 8 * g;
  auto ld = [&](int s, Ops &o) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      o.a_hi[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 0) * 8 * kF16Taps + 32 * s);
      o.a_lo[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 1) * 8 * kF16Taps + 32 * s);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o.b_hi[c] = *reinterpret_cast<const f16x8 *>(xh + q0 - 256 * c + 32 * s);
      o.b_lo[c] = *reinterpret_cast<const f16x8 *>(xl + q0 - 256 * c + 32 * s);
    }
  };
  auto mm = [&](const Ops &o) {
#if IAMF_F16_EXP == 3
    asm volatile("" ::"v"(o.a_hi[0]), "v"(o.a_lo[0]), "v"(o.a_hi[1]), "v"(o.a_lo[1]));
    asm volatile("" ::"v"(o.b_hi[0]), "v"(o.b_lo[0]), "v"(o.b_hi[1]), "v"(o.b_lo[1]), "v"(o.b_hi[2]), "v"(o.b_lo[2]), "v"(o.b_hi[3]), "v"(o.b_lo[3]));
#else
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc_hh[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[e], o.b_hi[c], acc_hh[e][c], 0, 0, 0);
        acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[e], o.b_lo[c], acc_x[e][c], 0, 0, 0);
        acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_lo[e], o.b_hi[c], acc_x[e][c], 0, 0, 0);
      }
#endif
  };
  IAMF_F16_STAMP_DECL
  fetch(0);
  stash_h();
  stash_x();
  __syncthreads();
  IAMF_F16_STAMP(0)  // prologue: first channel's fetch, stores and barrier
  for (int ch = 0; ch < M; ++ch) {
    if (IAMF_F16_EXP != 1 && ch + 1 < M) fetch(ch + 1);
    IAMF_F16_STAMP(1)  // issuing the next channel's loads
    if (work) {
      // two operand sets: the reads of the next step are in flight while this step's MFMAs issue (the last
      // step re-reads itself instead of branching)
      Ops o0, o1;
      ld(0, o0);
      int s = 0;
      for (; s + 2 <= KS; s += 2) {
        ld(s + 1, o1);
        mm(o0);
        ld(s + 2 < KS ? s + 2 : KS - 1, o0);
        mm(o1);
      }
      if (s < KS) mm(o0);
    }
    IAMF_F16_STAMP(2)  // K loop
    __syncthreads();  // everybody has read this channel's slice and tables
    IAMF_F16_STAMP(3)  // barrier A
    if (IAMF_F16_EXP != 1 && ch + 1 < M) {
      stash_h();
      stash_x();
      IAMF_F16_STAMP(4)  // LDS stores of the next channel (incl. the wait for its loads)
      __syncthreads();
      IAMF_F16_STAMP(5)  // barrier B
    }
  }
  // D[row = phase][col = block]: lane holds block col of a tile, phases 4g + r: four consecutive samples
  const float sc = p.fir_inv_scale;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float *pw = part + (e * 4 + w) * kF16Part;  // chunk w of the span, 256-sample tile c inside it
      const int nn = 256 * c + 16 * col + 4 * g;
      const int uu = nn + (nn >> 5);
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[uu + r] = (acc_hh[e][c][r] + acc_x[e][c][r] * (1.f / 2048.f)) * sc;
    }
  __syncthreads();
  IAMF_F16_STAMP(6)  // sums -> LDS
  IAMF_F16_STAMP_END
}
