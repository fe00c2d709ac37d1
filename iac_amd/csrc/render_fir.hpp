// render_fir.hpp — binaural HRTF stage for scene-based (HOA) elements: direct-form FIR
//     y[e][t] = sum_{c < M} sum_{k < L} h[e][c][k] * x[c][t - k]          (e = 0,1; L <= 256)
// on the f32 MFMA (v_mfma_f32_32x32x2_f32), used by render_fast_kernel<M, 2, true>.
//
// PARITY UNPINNED: the reference delegates this to Resonance Audio / BEAR, whose sources are not
// in the reference tree (h2b_rdr.c:44-130, m2b_rdr.c:44-121; compiled out by DISABLE_BINAURALIZER
// 1, ae_rdr.h:67-69).  The arithmetic above is this repo's specification; tests check it against
// a float64 convolution.
//
// Mapping to the matrix core.  Split t = 32*T + i (T = 32-sample block, i = phase) and substitute
// m = k - i:   y[e][32T + i] = sum_c sum_m h[e][c][m + i] * x[c][32T - m],  m in [-31, L-1].
// That is a GEMM  D[(e,i)][T] = A[(e,i)][(c,m)] * B[(c,m)][T]  with a Toeplitz A (only h itself is
// stored: the A operand of lane (i, kk) at step s is hp[2s + kk + i]) and B a stride-32 view of
// the input (B operand of lane (T, kk) at step s is x[c0 + 32T + 31 - 2s - kk]).  One chunk of
// 1024 samples is exactly one 32-column tile; 89 % of the issued MACs are useful (L / (L+31)).
// Wave w owns ear (w & 1) and half of the channels (w >> 1); the two halves are added at the end.
// Per channel the input slice (1280 samples incl. 256 of history, behind 32 zeros that absorb the
// steps of the padded tail) and both ears' filters are staged in LDS (register double-buffered
// global loads, one barrier per channel).  The slice is padded by one float per 32, which makes
// the stride-32 B reads bank-conflict free AND keeps every operand address of a step equal to a
// per-lane base plus a wave-uniform offset, so the inner loop is two ds_read_b32 + one MFMA.
#pragma once

constexpr int kFirChunk = 1024;  // = kFChunk of render_fast.hpp
constexpr int kFirHist = 256;    // history kept per channel = maximum taps
constexpr int kFirLead = 32;     // zeros in front of the slice
constexpr int kFirXs = 1360;     // padded slice: (32 + 1280) * 33 / 32, rounded up
constexpr int kFirHp = 336;      // padded filter: hp[j] = h[j - 31], zeros elsewhere
constexpr int kFirLdsFloats = 2 * 2 * kFirXs + 2 * 2 * 2 * kFirHp;  // [half][buf] xs + [half][buf][ear] hp

// sample n (relative to the start of this call) of channel ch of stream s; history for n < 0
__device__ __forceinline__ float fir_input(const RenderParams &p, const float *in_s, const float *hist, int ch,
                                           int n) {
  if (n < 0) return hist[ch * kFirHist + kFirHist + n];
  if (n >= p.total) return 0.f;
  const int f = n / p.frame_size;
  const int i = n - f * p.frame_size;
  return in_s[(int64_t)f * p.in_frame_stride + (int64_t)ch * p.frame_size + i];
}

// Computes y[e][c0 .. c0+1024) for both ears into `part` ([4][1024 + 32], padded by one per 32):
// the caller adds part[e] + part[e + 2].  All 256 threads must call it.  fir = LDS scratch of
// kFirLdsFloats floats; part aliases it (it is dead once the last channel has been multiplied).
template <int M>
__device__ __forceinline__ void fir_stage(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                          float *fir, float *part) {
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int MH = (M + 1) / 2;  // channel iterations (half 0 takes the extra one when M is odd)
  constexpr int U = 8;             // MFMA steps per unrolled block
  const int t = threadIdx.x;
  const int w = t >> 6, lane = t & 63;
  const int ear = w & 1, half = w >> 1;
  const int th = t & 127;  // thread index inside the half
  const int L = p.fir_taps;
  const int KS = (L + 32) >> 1;
  const int KSP = (KS + U - 1) & ~(U - 1);  // the padded steps multiply zeros of hp
  const int col = lane & 31, kk = lane >> 5;
  const int my_n = half == 0 ? MH : M / 2;  // channels this half multiplies
  const int ch0 = half == 0 ? 0 : MH;
  float *xs = fir;                   // [half][buf][kFirXs]
  float *hp = fir + 2 * 2 * kFirXs;  // [half][buf][ear][kFirHp]

  float xr[10], hr[6];
  auto fetch = [&](int ci) {  // global -> registers for channel ci of this half
    const int ch = ch0 + (ci < my_n ? ci : 0);
#pragma unroll
    for (int r = 0; r < 10; ++r) xr[r] = fir_input(p, in_s, hist, ch, c0 - kFirHist + th + 128 * r);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int j = th + 128 * r;  // 0..767 >= 2 * kFirHp = [ear][kFirHp]
      const int e2 = j >= kFirHp ? 1 : 0;
      const int tap = j - e2 * kFirHp - 31;
      hr[r] = (j < 2 * kFirHp && tap >= 0 && tap < L) ? p.matrix[((int64_t)e2 * M + ch) * L + tap] : 0.f;
    }
  };
  auto stash = [&](int buf) {  // registers -> LDS
    float *xb = xs + (half * 2 + buf) * kFirXs;
    float *hb = hp + (half * 2 + buf) * 2 * kFirHp;
    if (th < kFirLead) xb[th] = 0.f;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const int q = kFirLead + th + 128 * r;
      xb[q + (q >> 5)] = xr[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (th + 128 * r < 2 * kFirHp) hb[th + 128 * r] = hr[r];
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int ci = 0; ci < MH; ++ci) {
    const int buf = ci & 1;
    if (ci + 1 < MH) fetch(ci + 1);
    if (ci < my_n) {
      // step s: A operand hp[2s + kk + col]; B operand = slice sample 32*col + 319 - kk - 2s, whose
      // padded position is (33*col - kk) + o + (o >> 5) with the wave-uniform o = 319 - 2s
      const float *ha = hp + ((half * 2 + buf) * 2 + ear) * kFirHp + (kk + col);
      const float *xl = xs + (half * 2 + buf) * kFirXs + (33 * col - kk);
      for (int s0 = 0; s0 < KSP; s0 += U) {
        float a[U], b[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int o = 319 - 2 * (s0 + j);
          a[j] = ha[2 * (s0 + j)];
          b[j] = xl[o + (o >> 5)];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
      }
    }
    if (ci + 1 < MH) stash(buf ^ 1);
    __syncthreads();
  }
  // D[row = phase i][col = block T]: lane holds col = lane & 31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  float *pw = part + w * (kFirChunk + 32);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    pw[33 * col + row] = acc[r];  // sample 32*col + row, padded by one per 32
  }
  __syncthreads();
}
