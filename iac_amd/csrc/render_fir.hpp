// render_fir.hpp — binaural HRTF stage for scene-based (HOA) elements: direct-form FIR
//     y[e][t] = sum_{c < M} sum_{k < L} h[e][c][k] * x[c][t - k]          (e = 0,1; L <= 256)
// on the f32 MFMA (v_mfma_f32_16x16x4_f32), used by render_fast_kernel<M, 2, true>.
//
// PARITY UNPINNED: the reference delegates this to Resonance Audio / BEAR, whose sources are not
// in the reference tree (h2b_rdr.c:44-130, m2b_rdr.c:44-121; compiled out by DISABLE_BINAURALIZER
// 1, ae_rdr.h:67-69).  The arithmetic above is this repo's specification; tests check it against
// a float64 convolution.
//
// Mapping to the matrix core.  Split t = 16*T + i (T = 16-sample block, i = phase) and substitute
// m = k - i:   y[e][16T + i] = sum_c sum_m h[e][c][m + i] * x[c][16T - m],  m in [-15, L-1].
// That is a GEMM  D[i][T] = A[i][(c,m)] * B[(c,m)][T]  with a Toeplitz A (only h itself is stored:
// with m = 4s + kk - 15 the A operand of lane (i, kk) at step s is hp[4s + kk + i], hp[j] = h[j-15])
// and B a stride-16 view of the input (B operand of lane (T, kk) at step s is x[16T + 15 - 4s - kk]).
// A 16x16 tile covers 256 samples; a wave carries the chunk's FOUR column tiles as four independent
// accumulators, so one A read feeds four MFMAs and back-to-back MFMAs never depend on each other;
// 94 % of the issued MACs are useful (L / (L + 15)).  The stage runs with EIGHT waves: wave w owns
// ear (w & 1) and a quarter of the channels (w >> 1); the caller adds the four quarters.  With two
// workgroups per CU that puts four waves on every SIMD, so the matrix pipe finds a wave with MFMAs
// ready while others stage, wait at a barrier or run the limiter stages.  Per channel the input
// slice (1280 samples incl. 256 of history) and both ears' filters are staged in LDS by the two
// waves of the quarter (global loads into registers one channel ahead; single LDS buffer, two
// barriers per channel).  The slice is padded by one float per 16, which makes the stride-16 B reads
// (nearly) bank-conflict free AND keeps every operand address of a step equal to a per-lane base
// plus a wave-uniform offset, so a step is five ds_read_b32 and four MFMAs.
// Tried and dropped: wave-private staging without barriers (every wave then loads the slice itself
// and the extra address arithmetic costs more issue slots than the barriers did); operand
// double-buffering in registers; a lane stride of 20 floats (no measurable change).
#pragma once

constexpr int kFirChunk = 1024;  // = kFChunk of render_fast.hpp
constexpr int kFirHist = 256;    // history kept per channel = maximum taps
constexpr int kFirXs = 1360;     // padded slice: 1280 * 17 / 16
constexpr int kFirHp = 304;      // padded filter: hp[j] = h[j - 15], zeros elsewhere (4 * 68 + 16 + slack)
constexpr int kFirStage = 4 * kFirXs + 4 * 2 * kFirHp;  // [quarter] xs + [quarter][ear] hp
constexpr int kFirPart = 8 * (kFirChunk + 32);         // the eight waves' partial sums (alias the staging area)
constexpr int kFirLdsFloats = kFirPart > kFirStage ? kFirPart : kFirStage;

// sample n (relative to the start of this call) of channel ch of stream s; history for n < 0
__device__ __forceinline__ float fir_input(const RenderParams &p, const float *in_s, const float *hist, int ch,
                                           int n) {
  if (n < 0) return hist[ch * kFirHist + kFirHist + n];
  if (n >= p.total) return 0.f;
  const int f = n / p.frame_size;
  const int i = n - f * p.frame_size;
  return in_s[(int64_t)f * p.in_frame_stride + (int64_t)ch * p.frame_size + i];
}

// Computes y[e][c0 .. c0+1024) for both ears into `part` ([8][1024 + 32], padded by one per 32):
// wave w writes part[w]; ear e = part[e] + part[e+2] + part[e+4] + part[e+6].  All 512 threads must
// call it.  fir = LDS scratch of kFirLdsFloats floats; part aliases it (the staging area is dead
// once the last channel has been multiplied).
template <int M>
__device__ __forceinline__ void fir_stage(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                          float *fir, float *part) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int MQ = (M + 3) / 4;  // channel iterations (the first M % 4 quarters take one more)
  constexpr int U = 4;             // steps per unrolled block (16 MFMAs)
  const int t = threadIdx.x;       // 0..511
  const int w = t >> 6, lane = t & 63;
  const int ear = w & 1, quarter = w >> 1;
  const int th = t & 127;  // thread index inside the quarter
  const int L = p.fir_taps;
  const int KS = (L + 15 + 3) >> 2;         // steps of 4 taps over m' = m + 15 in [0, L + 14]
  const int KSP = (KS + U - 1) & ~(U - 1);  // the padded steps multiply zeros of hp; <= 68
  const int col = lane & 15, kk = lane >> 4;
  const int my_n = M / 4 + (quarter < M % 4 ? 1 : 0);  // channels this quarter multiplies
  const int ch0 = quarter * (M / 4) + (quarter < M % 4 ? quarter : M % 4);
  float *xb = fir + quarter * kFirXs;                   // [kFirXs]
  float *hb = fir + 4 * kFirXs + quarter * 2 * kFirHp;  // [ear][kFirHp]

  // where this thread's 10 slice samples come from does not depend on the channel: >= 0 = offset in
  // the channel's plane of the call's input, -1 = past the end of the call (zero), <= -2 = history
  int off[10];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const int n = c0 - kFirHist + th + 128 * r;
    if (n < 0) {
      off[r] = -2 - (kFirHist + n);
    } else if (n >= p.total) {
      off[r] = -1;
    } else {
      const int f = n / p.frame_size;
      off[r] = (int)(f * p.in_frame_stride) + (n - f * p.frame_size);
    }
  }
  float xr[10], hr[5];
  auto fetch = [&](int ci) {  // global -> registers for channel ci of this quarter
    const int ch = ch0 + (ci < my_n ? ci : 0);
    const float *plane = in_s + (int64_t)ch * p.frame_size;
    const float *hch = hist + ch * kFirHist;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const int o = off[r];
      xr[r] = o >= 0 ? plane[o] : (o == -1 ? 0.f : hch[-2 - o]);
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int j = th + 128 * r;  // 0..639 >= 2 * kFirHp = [ear][kFirHp]
      const int e2 = j >= kFirHp ? 1 : 0;
      const int tap = j - e2 * kFirHp - 15;
      hr[r] = (j < 2 * kFirHp && tap >= 0 && tap < L) ? p.matrix[((int64_t)e2 * M + ch) * L + tap] : 0.f;
    }
  };
  auto stash = [&]() {  // registers -> LDS
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const int q = th + 128 * r;  // slice position: sample c0 - 256 + q
      xb[q + (q >> 4)] = xr[r];
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (th + 128 * r < 2 * kFirHp) hb[th + 128 * r] = hr[r];
  };

  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  stash();
  __syncthreads();
  for (int ci = 0; ci < MQ; ++ci) {
    if (ci + 1 < MQ) fetch(ci + 1);
    if (ci < my_n) {
      // step s: A operand hp[4s + kk + col]; B operand of column tile ct = slice sample
      // 256*ct + 16*col + 271 - kk - 4s, whose padded position is (17*col - kk) + 272*ct + o + (o >> 4)
      // with the wave-uniform o = 271 - 4s (o % 16 is 15, 11, 7 or 3 >= kk, so -kk never crosses a pad)
      const float *ha = hb + ear * kFirHp + (kk + col);
      const float *xl = xb + (17 * col - kk);
      for (int s0 = 0; s0 < KSP; s0 += U) {
        float a[U], b[U][4];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int o = 271 - 4 * (s0 + j);
          const int po = o + (o >> 4);
          a[j] = ha[4 * (s0 + j)];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) b[j][ct] = xl[272 * ct + po];
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][ct], acc[ct], 0, 0, 0);
      }
    }
    __syncthreads();  // everybody has read this channel's slice
    if (ci + 1 < MQ) {
      stash();
      __syncthreads();
    }
  }
  // D[row = phase][col = block]: lane holds block col of tile ct, phases 4*kk + r: four consecutive samples
  float *pw = part + w * (kFirChunk + 32);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = 256 * ct + 16 * col + 4 * kk;  // first of the lane's 4 samples; they share a 32-block
    const int u = n + (n >> 5);
    pw[u + 0] = acc[ct][0];
    pw[u + 1] = acc[ct][1];
    pw[u + 2] = acc[ct][2];
    pw[u + 3] = acc[ct][3];
  }
  __syncthreads();
}
