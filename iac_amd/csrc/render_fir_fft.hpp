// render_fir_fft.hpp — the binaural HRTF stage of render_fir.hpp by OVERLAP-SAVE in the frequency domain,
// on the VALU (render_fast_kernel<M, 2, 3>).
//
//     y[e][t] = sum_{c < M} sum_{k < L} h[e][c][k] * x[c][t - k]          (e = 0,1; L <= 256)
//
// PARITY UNPINNED, as the other two HRTF stages: the reference delegates binaural rendering to Resonance Audio /
// BEAR, whose sources are not in its tree (h2b_rdr.c:109-130, m2b_rdr.c:103-121, call sites IAMF_decoder.c:2568-2570,
// 2608-2610); the formula above is this library's specification, checked against float64 (tests/test_gpu_fir.py).
//
// Why.  The direct form costs 16 384 flop per sample-frame at 16 channels x 256 taps; on the f16 matrix cores with
// split operands that is 49 152 issued flop and the kernel ends up power-limited at 0.12 of the f16 peak (18.5
// Gsamples/s, NOTEBOOK.md 4.2).  Overlap-save with N = 1024, hop 768 (N - hop = 256 >= L) costs ~770 flop per
// sample-frame on the vector ALU and moves the stage towards the HBM bound of the whole kernel (68 B per sample-frame).
//
// Algebra.  Two real channels a, b ride one complex FFT, z_p = x_a + i x_b, and both ears ride one inverse FFT,
// w = y_left + i y_right.  With g_c = h_left,c + i h_right,c and G_c = FFT(g_c):
//     W[k] = sum_p  P_p[k] Z_p[k] + Q_p[k] conj(Z_p[N - k]),   P_p = (G_a - i G_b) / 2,  Q_p = (G_a + i G_b) / 2.
// The mirrored term is not taken per pair: U[k] += P_p[k] Z_p[k] and V[k] += conj(Q_p[N - k]) Z_p[k] accumulate in
// the lane's own bins, and W[k] = U[k] + conj(V[N - k]) needs ONE mirror exchange per hop.  The tables hold
// P_p / N and conj(Q_p[N - k]) / N in the FFT's own (permuted) bin order, so no reordering pass exists at all:
// the forward FFT is decimation in frequency (natural in, permuted out), the inverse is its exact transpose
// (permuted in, natural out).
//
// One WAVE runs one hop from input to output — 8 forward FFTs (M = 16), the accumulation, the mirror, one inverse —
// so the stage has no workgroup barrier inside; the four waves of the workgroup take four consecutive hops = 3072
// samples = three chunks per pass.  A 1024-point FFT = 16 x 16 x 4 with 16 complex points per lane:
//     n = 64 n1 + 4 n2 + n3,  k = k1 + 16 k2 + 256 k3
//     A: DFT16 over n1 (registers), x W_1024^{l k1} (l = 4 n2 + n3 = lane)      -> exchange 1 (LDS)
//     B: DFT16 over n2 (registers), x W_64^{n3 k2}                              -> exchange 2 (LDS, inside quads)
//     C: DFT4 over n3 (registers)
// LDS layouts are padded / skewed so that every ds_write_b64 (banks mod 32, 16-lane groups) and ds_read_b64 (banks
// mod 64, 32-lane groups) is conflict-free (MI355X_MICROARCH.md, LDS): element (k1, n2, n3) of exchange 1 sits at
// 68 k1 + 4 n2 + n3; element (k1, k2 = 4 g + j, n3) of exchange 2 at 68 k1 + 16 g + 4 n3 + ((n3 + j) & 3).
//
// The arithmetic core below is __host__ __device__: tests/fft_host/ runs it lane by lane on the CPU against a
// float64 DFT and a direct convolution (-m "not gpu"), so the index algebra is checked without a GPU.
#pragma once

#ifndef FFT_HD
#define FFT_HD __host__ __device__ __forceinline__
#endif

#ifndef IAMF_FFT_NT
#define IAMF_FFT_NT 1                // 0: the sample fetch with the default cache policy throughout (A/B builds)
#endif
#ifndef IAMF_FFT_X2_LDS
#define IAMF_FFT_X2_LDS 0            // 1: exchange 2 through LDS (the first form; kept for A/B builds, tools/debug/fft_exp.sh)
#endif
constexpr int kFftN = 1024;          // FFT size
constexpr int kFftHop = 768;         // new samples per hop; kFftN - kFftHop = 256 >= taps
constexpr int kFftHops = 4;          // hops per pass = waves per workgroup
constexpr int kFftSpan = kFftHop * kFftHops;   // 3072 samples = three 1024-sample chunks per pass
constexpr int kFftRow = 68;          // exchange stride of k1 in complex elements (64 + 4 of padding)
constexpr int kFftScratch = 16 * kFftRow;      // complex elements of LDS one wave needs (1088 >= 1024 for the mirror)

typedef float fft_c32 __attribute__((ext_vector_type(2)));   // (re, im)

FFT_HD fft_c32 fft_mk(float re, float im) {
  fft_c32 r;
  r.x = re;
  r.y = im;
  return r;
}
// Complex products.  On the device every one is exactly TWO packed instructions: v_pk_mul_f32 / v_pk_fma_f32 pick the
// halves of their 64-bit operands per result half (op_sel for the low result, op_sel_hi for the high one) and negate
// them (neg_lo / neg_hi), so neither the broadcast of a.x / a.y nor the swap (-b.y, b.x) costs an instruction or a
// register.  Written as inline asm because the compiler builds the broadcast pairs and the swapped operand as values of
// their own (for loop-invariant twiddles it even hoists them: 60 registers, which it then spills).
#if defined(__HIP_DEVICE_COMPILE__)
// (Both instructions of a product in ONE asm statement: between two statements the compiler's hazard recogniser puts an
// s_nop — it takes op_sel_hi of a packed-f32 instruction for the 16-bit "dst_sel" forwarding hazard of gfx940 — which cost
// 70 issue slots per pair of channels.)
// a * b = (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
__device__ __forceinline__ fft_c32 fft_cmul(fft_c32 a, fft_c32 b) {
  fft_c32 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"                                              // (a.x b.x, a.x b.y)
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"               // + (-a.y b.y, a.y b.x)
      : "=&v"(d)
      : "v"(a), "v"(b));
  return d;
}
// a * conj(b) = (a.x b.x + a.y b.y, -a.x b.y + a.y b.x)
__device__ __forceinline__ fft_c32 fft_cmul_conj(fft_c32 a, fft_c32 b) {
  fft_c32 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"                                 // (a.x b.x, -a.x b.y)
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                              // + (a.y b.y, a.y b.x)
      : "=&v"(d)
      : "v"(a), "v"(b));
  return d;
}
// acc + a * b
__device__ __forceinline__ fft_c32 fft_cmac(fft_c32 acc, fft_c32 a, fft_c32 b) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
      : "+v"(acc)
      : "v"(a), "v"(b));
  return acc;
}
// a + w d and a - w d with w = -i (INV = false) or +i (INV = true):  -i d = (d.y, -d.x)
template <bool INV>
__device__ __forceinline__ fft_c32 fft_add_rot(fft_c32 a, fft_c32 d) {
  fft_c32 r;
  if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));   // (a.x - d.y, a.y + d.x)
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));       // (a.x + d.y, a.y - d.x)
  return r;
}
template <bool INV>
__device__ __forceinline__ fft_c32 fft_sub_rot(fft_c32 a, fft_c32 d) { return fft_add_rot<!INV>(a, d); }
// w a
template <bool INV>
__device__ __forceinline__ fft_c32 fft_rot(fft_c32 a) { return fft_add_rot<INV>(fft_mk(0.f, 0.f), a); }
#else
FFT_HD fft_c32 fft_cmul(fft_c32 a, fft_c32 b) {
  const fft_c32 bx = fft_mk(b.x, b.x), by = fft_mk(b.y, b.y), ar = fft_mk(-a.y, a.x);
  return __builtin_elementwise_fma(by, ar, bx * a);
}
FFT_HD fft_c32 fft_cmul_conj(fft_c32 a, fft_c32 b) {
  const fft_c32 bx = fft_mk(b.x, b.x), by = fft_mk(b.y, b.y), ar = fft_mk(a.y, -a.x);
  return __builtin_elementwise_fma(by, ar, bx * a);
}
FFT_HD fft_c32 fft_cmac(fft_c32 acc, fft_c32 a, fft_c32 b) {
  const fft_c32 bx = fft_mk(b.x, b.x), by = fft_mk(b.y, b.y), ar = fft_mk(-a.y, a.x);
  return __builtin_elementwise_fma(by, ar, __builtin_elementwise_fma(bx, a, acc));
}
template <bool INV>
FFT_HD fft_c32 fft_rot(fft_c32 a) {   // -i a (INV = false) or +i a (INV = true)
  return INV ? fft_mk(-a.y, a.x) : fft_mk(a.y, -a.x);
}
template <bool INV>
FFT_HD fft_c32 fft_add_rot(fft_c32 a, fft_c32 d) { return a + fft_rot<INV>(d); }
template <bool INV>
FFT_HD fft_c32 fft_sub_rot(fft_c32 a, fft_c32 d) { return a - fft_rot<INV>(d); }
#endif

// 4-point DFT in place; INV = false: W_4 = -i, INV = true: W_4 = +i (unscaled).  Eight packed additions.
template <bool INV>
FFT_HD void fft_dft4(fft_c32 &a0, fft_c32 &a1, fft_c32 &a2, fft_c32 &a3) {
  const fft_c32 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = fft_add_rot<INV>(t1, d);
  a3 = fft_sub_rot<INV>(t1, d);
}

// 16-point DFT in place, natural order in and out: X[k] = sum_n x[n] W_16^{+-nk}
template <bool INV>
FFT_HD void fft_dft16(fft_c32 (&z)[16]) {
  // n = 4 n1 + n2, k = k1 + 4 k2: DFT4 over n1, twiddle W_16^{n2 k1}, DFT4 over n2
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) fft_dft4<INV>(z[n2], z[n2 + 4], z[n2 + 8], z[n2 + 12]);   // z[n2 + 4 k1]
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  const float sg = INV ? 1.f : -1.f;   // forward: e^{-i phi}
  const fft_c32 w1 = fft_mk(c1, sg * s1), w2 = fft_mk(h, sg * h), w3 = fft_mk(s1, sg * c1), w6 = fft_mk(-h, sg * h),
                w9 = fft_mk(-c1, -sg * s1);
  // (n2, k1): 1,1 -> W^1   1,2 -> W^2   1,3 -> W^3   2,1 -> W^2   2,2 -> W^4 = -+i   2,3 -> W^6   3,1 -> W^3   3,2 -> W^6   3,3 -> W^9
  z[1 + 4] = fft_cmul(z[1 + 4], w1);
  z[1 + 8] = fft_cmul(z[1 + 8], w2);
  z[1 + 12] = fft_cmul(z[1 + 12], w3);
  z[2 + 4] = fft_cmul(z[2 + 4], w2);
  z[2 + 8] = fft_rot<INV>(z[2 + 8]);
  z[2 + 12] = fft_cmul(z[2 + 12], w6);
  z[3 + 4] = fft_cmul(z[3 + 4], w3);
  z[3 + 8] = fft_cmul(z[3 + 8], w6);
  z[3 + 12] = fft_cmul(z[3 + 12], w9);
  // DFT4 over n2 for every k1: inputs z[n2 + 4 k1], outputs X[k1 + 4 k2]
  fft_c32 o[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    fft_c32 a0 = z[0 + 4 * k1], a1 = z[1 + 4 * k1], a2 = z[2 + 4 * k1], a3 = z[3 + 4 * k1];
    fft_dft4<INV>(a0, a1, a2, a3);
    o[k1] = a0;
    o[k1 + 4] = a1;
    o[k1 + 8] = a2;
    o[k1 + 12] = a3;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) z[k] = o[k];
}

// ---- the three register stages and the two exchanges.  `S` is the wave's LDS scratch (kFftScratch elements). ----
// Layouts (lane, register):  L0 (4 n2 + n3, n1)   L1 (4 k1 + n3, n2)   L2 (4 k1 + j, 4 g + n3 or 4 g + k3), k2 = 4 g + j
FFT_HD int fft_e1(int k1, int n2, int n3) { return kFftRow * k1 + 4 * n2 + n3; }
FFT_HD int fft_e2(int k1, int g, int n3, int j) { return kFftRow * k1 + 16 * g + 4 * n3 + ((n3 + j) & 3); }
// bin held by register r of lane `lane` after the forward transform / before the inverse (layout L2)
FFT_HD int fft_bin(int lane, int r) { return (lane >> 2) + 16 * ((lane & 3) + 4 * (r >> 2)) + 256 * (r & 3); }

// tw1[k1] = W_1024^{lane k1}, tw2[k2] = W_64^{(lane & 3) k2}  (forward sign; the inverse conjugates)
FFT_HD void fft_fwd_a(fft_c32 (&z)[16], const fft_c32 (&tw1)[16]) {
  fft_dft16<false>(z);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) z[k1] = fft_cmul(z[k1], tw1[k1]);
}
template <typename P>
FFT_HD void fft_x1_write(const fft_c32 (&z)[16], int lane, P S) {   // from L0 (forward) : z[k1] of lane l
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) S[kFftRow * k1 + lane] = z[k1];
}
template <typename P>
FFT_HD void fft_x1_read(fft_c32 (&z)[16], int lane, P S) {          // into L1: z[n2]
  const int b = fft_e1(lane >> 2, 0, lane & 3);
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) z[n2] = S[b + 4 * n2];
}
FFT_HD void fft_fwd_b(fft_c32 (&z)[16], const fft_c32 (&tw2)[16]) {
  fft_dft16<false>(z);
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) z[k2] = fft_cmul(z[k2], tw2[k2]);
}
// the same with the twiddles read where they are used: tw1s[64 k1] (a table [k1][lane] offset by the lane) and
// tw2s[4 k2] (a table [k2][n3] offset by the lane's n3)
// (the table reads are issued BEFORE the 16-point transform that hides their latency and pinned there: left alone, the
// scheduler sinks every read to its use to save registers and the wave waits for each of them in turn)
FFT_HD void fft_pin() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}
FFT_HD void fft_fwd_a_tab(fft_c32 (&z)[16], const fft_c32 *tw1s) {
  fft_c32 w[16];
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) w[k1] = tw1s[64 * k1];
  fft_pin();
  fft_dft16<false>(z);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) z[k1] = fft_cmul(z[k1], w[k1]);
}
FFT_HD void fft_inv_a_tab(fft_c32 (&z)[16], const fft_c32 *tw1s) {
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) z[k1] = fft_cmul_conj(z[k1], tw1s[64 * k1]);
  fft_dft16<true>(z);
}
FFT_HD void fft_fwd_b_tab(fft_c32 (&z)[16], const fft_c32 *tw2s) {
  fft_c32 w[16];
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) w[k2] = tw2s[4 * k2];
  fft_pin();
  fft_dft16<false>(z);
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) z[k2] = fft_cmul(z[k2], w[k2]);
}
FFT_HD void fft_inv_b_tab(fft_c32 (&z)[16], const fft_c32 *tw2s) {
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) z[k2] = fft_cmul_conj(z[k2], tw2s[4 * k2]);
  fft_dft16<true>(z);
}
template <typename P>
FFT_HD void fft_x2_write(const fft_c32 (&z)[16], int lane, P S) {   // from L1: z[k2]
  const int k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) S[fft_e2(k1, k2 >> 2, n3, k2 & 3)] = z[k2];
}
template <typename P>
FFT_HD void fft_x2_read(fft_c32 (&z)[16], int lane, P S) {          // into L2: z[4 g + n3]
  const int k1 = lane >> 2, j = lane & 3;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = S[fft_e2(k1, r >> 2, r & 3, j)];
}
#if defined(__HIP_DEVICE_COMPILE__)
// Exchange 2 without LDS.  It is a 4 x 4 transpose inside every quad of lanes — lane (k1, n3) holds z[4 g + j], lane
// (k1, j) wants z[4 g + n3] — i.e. two swaps of a lane-index bit with a register-index bit, and a swap costs ONE
// instruction per register: v_cndmask_b32 with a DPP quad_perm on its first source ("my own value, or my neighbour's other
// register").  64 VALU instructions per transform instead of 16 ds_write_b64 + 16 ds_read_b64 and a round trip through the
// LDS queue: the stage kernel ran its VALU 49 % of the time with the LDS store path (ds_write_b64: ~85 B/clk per CU) as
// the second limit, so the transform trades a resource it is short of twice for one it has to spare between the waits.
// D = VCC ? src1 : dpp(src0).  Inline asm: the masks go through VCC (the DPP encoding has no SGPR-pair operand), and a DPP
// source must not have been written by the two instructions before it (the assembler adds no wait states inside asm): the
// order below keeps three instructions between any write and its DPP read, s_nop covers the code in front.
__device__ __forceinline__ void fft_quad_transpose2(float &a0, float &a1, float &a2, float &a3, float &b0, float &b1, float &b2,
                                                    float &b3) {
  float s0, s1, s2, s3, t0, t1, t2, t3;
  const uint64_t me = 0x5555555555555555ull, mo = 0xaaaaaaaaaaaaaaaaull, ml = 0x3333333333333333ull, mh = 0xccccccccccccccccull;
  asm volatile(
      "s_nop 1\n\t"
      "s_mov_b64 vcc, %[me]\n\t"
      "v_cndmask_b32_dpp %[s0], %[a1], %[a0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[s2], %[a3], %[a2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t0], %[b1], %[b0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t2], %[b3], %[b2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mo]\n\t"
      "v_cndmask_b32_dpp %[s1], %[a0], %[a1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[s3], %[a2], %[a3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t1], %[b0], %[b1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[t3], %[b2], %[b3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[ml]\n\t"
      "v_cndmask_b32_dpp %[a0], %[s2], %[s0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b0], %[t2], %[t0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a1], %[s3], %[s1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b1], %[t3], %[t1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 vcc, %[mh]\n\t"
      "v_cndmask_b32_dpp %[a2], %[s0], %[s2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b2], %[t0], %[t2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[a3], %[s1], %[s3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %[b3], %[t1], %[t3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3),
        [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
      : [me] "s"(me), [mo] "s"(mo), [ml] "s"(ml), [mh] "s"(mh)
      : "vcc");
}
__device__ __forceinline__ void fft_x2_quads(fft_c32 (&z)[16]) {   // both directions: the transpose is its own inverse
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float a0 = z[4 * g].x, a1 = z[4 * g + 1].x, a2 = z[4 * g + 2].x, a3 = z[4 * g + 3].x;
    float b0 = z[4 * g].y, b1 = z[4 * g + 1].y, b2 = z[4 * g + 2].y, b3 = z[4 * g + 3].y;
    fft_quad_transpose2(a0, a1, a2, a3, b0, b1, b2, b3);
    z[4 * g] = fft_mk(a0, b0);
    z[4 * g + 1] = fft_mk(a1, b1);
    z[4 * g + 2] = fft_mk(a2, b2);
    z[4 * g + 3] = fft_mk(a3, b3);
  }
}
#elif defined(__HIPCC__)
__device__ void fft_x2_quads(fft_c32 (&z)[16]);   // (the host pass of a .hip file only parses the device functions that call it)
#endif
FFT_HD void fft_fwd_c(fft_c32 (&z)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) fft_dft4<false>(z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]);
}
// the inverse: the same graph transposed, twiddles conjugated (unscaled: the tables carry 1 / N)
FFT_HD void fft_inv_c(fft_c32 (&z)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) fft_dft4<true>(z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]);
}
template <typename P>
FFT_HD void fft_x2_write_back(const fft_c32 (&z)[16], int lane, P S) {   // from L2: z[4 g + n3]
  const int k1 = lane >> 2, j = lane & 3;
#pragma unroll
  for (int r = 0; r < 16; ++r) S[fft_e2(k1, r >> 2, r & 3, j)] = z[r];
}
template <typename P>
FFT_HD void fft_x2_read_back(fft_c32 (&z)[16], int lane, P S) {          // into L1: z[k2]
  const int k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) z[k2] = S[fft_e2(k1, k2 >> 2, n3, k2 & 3)];
}
FFT_HD void fft_inv_b(fft_c32 (&z)[16], const fft_c32 (&tw2)[16]) {
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) z[k2] = fft_cmul_conj(z[k2], tw2[k2]);
  fft_dft16<true>(z);
}
template <typename P>
FFT_HD void fft_x1_write_back(const fft_c32 (&z)[16], int lane, P S) {   // from L1: z[n2]
  const int b = fft_e1(lane >> 2, 0, lane & 3);
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) S[b + 4 * n2] = z[n2];
}
template <typename P>
FFT_HD void fft_x1_read_back(fft_c32 (&z)[16], int lane, P S) {          // into L0: z[k1]
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) z[k1] = S[kFftRow * k1 + lane];
}
FFT_HD void fft_inv_a(fft_c32 (&z)[16], const fft_c32 (&tw1)[16]) {
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) z[k1] = fft_cmul_conj(z[k1], tw1[k1]);
  fft_dft16<true>(z);
}
// W[k] = U[k] + conj(V[N - k]).  Register r of lane `lane` holds bin L + 64 q, L = (lane >> 2) + 16 (lane & 3), q = (r >> 2)
// + 4 (r & 3); V goes through the scratch at POSITION lane + 64 q (contiguous lanes: conflict-free).  The mirror bin
// 1024 - L - 64 q is (64 - L) + 64 (15 - q) for L > 0, held by lane 67 - lane (lanes 4..63) or 4 - lane (lanes 1..3) in its
// register of q' = 15 - q; for L = 0 (lane 0) it is 64 (16 - q): lane 0 itself, q' = 16 - q, and so that q = 0 needs no
// wrap lane 0 also stores its bin 0 at position 1024.  Every read address is a per-lane base + a compile-time offset.
FFT_HD int fft_mirror_q(int r) { return (r >> 2) + 4 * (r & 3); }
FFT_HD int fft_mirror_base(int lane) { return lane == 0 ? 64 : (lane < 4 ? 4 - lane : 67 - lane); }
FFT_HD fft_c32 fft_add_conj(fft_c32 a, fft_c32 m) {   // a + conj(m)
#if defined(__HIP_DEVICE_COMPILE__)
  fft_c32 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(m));
  return r;
#else
  return fft_mk(a.x + m.x, a.y - m.y);
#endif
}
template <typename P>
FFT_HD void fft_mirror_write(const fft_c32 (&v)[16], int lane, P S) {
#pragma unroll
  for (int r = 0; r < 16; ++r) S[lane + 64 * fft_mirror_q(r)] = v[r];
  if (lane == 0) S[1024] = v[0];
}
template <typename P>
FFT_HD void fft_mirror_read_add(fft_c32 (&u)[16], int lane, P S) {
  const int b = fft_mirror_base(lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) u[r] = fft_add_conj(u[r], S[b + 64 * (15 - fft_mirror_q(r))]);
}

// ---- host side: the tables of one HRIR set ----
// hrir: [2 ears][m][taps] floats.  pq: [pairs][16 registers][64 lanes] x (P'.re, P'.im, Q'.re, Q'.im), P' = P / N,
// Q'[k] = conj(Q[N - k]) / N, in the layout-L2 bin order; tw: [16][64] (W_1024^{lane k1}) then [16][4] (W_64^{n3 k2})
// complex.  Returns the number of pairs.  Everything is evaluated in double and rounded once.
// (compiled where IAMF_FFT_HOST_TABLES is defined; the includer provides <math.h> and <vector>)
#if defined(IAMF_FFT_HOST_TABLES)
inline int fft_build_tables(const float *hrir, int m, int taps, std::vector<float> &pq, std::vector<float> &tw) {
  const int pairs = (m + 1) / 2;
  const double two_pi = 6.283185307179586476925286766559;
  std::vector<double> cr(kFftN), ci(kFftN);
  for (int k = 0; k < kFftN; ++k) {
    cr[k] = cos(two_pi * k / kFftN);
    ci[k] = -sin(two_pi * k / kFftN);
  }
  // G_c[k] = sum_n (hl[n] + i hr[n]) W^{nk}
  std::vector<double> gr((size_t)m * kFftN), gi((size_t)m * kFftN);
  for (int c = 0; c < m; ++c)
    for (int k = 0; k < kFftN; ++k) {
      double ar = 0, ai = 0;
      for (int n = 0; n < taps; ++n) {
        const double hl = hrir[((size_t)0 * m + c) * taps + n], hr = hrir[((size_t)1 * m + c) * taps + n];
        const int e = (n * k) & (kFftN - 1);
        ar += hl * cr[e] - hr * ci[e];
        ai += hl * ci[e] + hr * cr[e];
      }
      gr[(size_t)c * kFftN + k] = ar;
      gi[(size_t)c * kFftN + k] = ai;
    }
  pq.assign((size_t)pairs * 16 * 64 * 4, 0.f);
  for (int p = 0; p < pairs; ++p) {
    const int a = 2 * p, b = 2 * p + 1;
    auto G = [&](int c, int k, double &re, double &im) {
      re = c < m ? gr[(size_t)c * kFftN + k] : 0.0;
      im = c < m ? gi[(size_t)c * kFftN + k] : 0.0;
    };
    for (int r = 0; r < 16; ++r)
      for (int lane = 0; lane < 64; ++lane) {
        const int k = fft_bin(lane, r), km = (kFftN - k) & (kFftN - 1);
        double ar, ai, br, bi;
        G(a, k, ar, ai);
        G(b, k, br, bi);
        // P = (Ga - i Gb) / 2 = ((ar + bi) + i (ai - br)) / 2
        const double pr = 0.5 * (ar + bi), pi = 0.5 * (ai - br);
        G(a, km, ar, ai);
        G(b, km, br, bi);
        // Q[km] = (Ga + i Gb) / 2 = ((ar - bi) + i (ai + br)) / 2 ; Q' = conj(Q[km])
        const double qr = 0.5 * (ar - bi), qi = -0.5 * (ai + br);
        float *o = &pq[(((size_t)p * 16 + r) * 64 + lane) * 4];
        o[0] = (float)(pr / kFftN);
        o[1] = (float)(pi / kFftN);
        o[2] = (float)(qr / kFftN);
        o[3] = (float)(qi / kFftN);
      }
  }
  tw.assign((size_t)(16 * 64 + 16 * 4) * 2, 0.f);
  for (int k1 = 0; k1 < 16; ++k1)
    for (int lane = 0; lane < 64; ++lane) {
      const int e = (lane * k1) & (kFftN - 1);
      tw[((size_t)k1 * 64 + lane) * 2 + 0] = (float)cr[e];
      tw[((size_t)k1 * 64 + lane) * 2 + 1] = (float)ci[e];
    }
  for (int k2 = 0; k2 < 16; ++k2)
    for (int n3 = 0; n3 < 4; ++n3) {
      const int e = (16 * n3 * k2) & (kFftN - 1);   // W_64^{n3 k2} = W_1024^{16 n3 k2}
      tw[((size_t)16 * 64 + k2 * 4 + n3) * 2 + 0] = (float)cr[e];
      tw[((size_t)16 * 64 + k2 * 4 + n3) * 2 + 1] = (float)ci[e];
    }
  return pairs;
}
#endif

#if defined(__HIPCC__)
// IAMF_FFT_EXP: timing-only elimination builds (WRONG results), tools/debug/fft_exp.sh:
//   1 = no spectra-table loads (the accumulation multiplies by a constant)   2 = the next pair's samples are not fetched
//   3 = no LDS exchanges (the register stages run on whatever they hold)     4 = the stage returns at once
//   5 = no accumulation at all (transforms only)
#ifndef IAMF_FFT_EXP
#define IAMF_FFT_EXP 0
#endif
// ------------------------------------------------------------------------------------------------------
// device stage.  LDS of the variant: [4 waves][kFftScratch] complex scratch + the twiddle tables.  A wave leaves its
// hop's 768 outputs per ear in ITS OWN scratch ([2 ears][768] floats: the scratch is idle between two passes), where the
// limiter stages of the following three chunks read them (fft_y_ptr).
// ------------------------------------------------------------------------------------------------------
constexpr int kFftLdsFloats = kFftHops * kFftScratch * 2 + (16 * 64 + 16 * 4) * 2;   // scratch + the twiddle tables

// Lane-constant twiddles, both read from LDS where they are used: tw1[k1] = W_1024^{lane k1} (a [16][64] table) and tw2[k2] =
// W_64^{(lane & 3) k2} ([16][4]).  In registers they would be 60 of the 256 a wave has at two waves per SIMD, and the
// accumulators, the transform in flight and the two prefetches (next pair's samples, this pair's spectra) need those.
constexpr int kFftTwFloats = (16 * 64 + 16 * 4) * 2;
struct FftTwiddles {
  const fft_c32 *tw1;   // LDS: [k1][64], already offset by the lane
  const fft_c32 *tw2;   // LDS: [k2][4], already offset by the lane's n3
};
__device__ __forceinline__ void fft_load_twiddles(const float *tw, int t, float *tw_lds, FftTwiddles &o) {
  const fft_c32 *g = reinterpret_cast<const fft_c32 *>(tw);
  fft_c32 *l = reinterpret_cast<fft_c32 *>(tw_lds);
  for (int i = t; i < 16 * 64 + 16 * 4; i += 256) l[i] = g[i];   // visible after the kernel's first barrier
  o.tw1 = l + (t & 63);
  o.tw2 = l + 16 * 64 + (t & 3);
}

__device__ __forceinline__ void fft_wave_sync() {   // LDS accesses of one wave execute in order; keep the compiler from reordering them
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The 16 runs of channel pair `pr` -> z (re = channel 2 pr, im = channel 2 pr + 1 or zeros).  Lane n1 < 16 of (ra_lo,
// ra_hi) holds run n1's address for channel 0 (fir_stage_fft); v_readlane brings it to scalar registers where it is used,
// so every load is (scalar base) + 4 lane and the 16 runs cost two vector registers instead of 32 scalar ones; which runs
// are history / zeros is one bit each of two scalar masks.
template <int M, int N1A = 0, int N1B = 16>
__device__ __forceinline__ void fft_fetch(int pr, fft_c32 (&z)[16], unsigned ra_lo, unsigned ra_hi, unsigned zmask,
                                          unsigned hmask, unsigned cs_in, uint64_t a_zero, unsigned lo) {
  const int a = 2 * pr, b = 2 * pr + 1;
#pragma unroll
  for (int n1 = N1A; n1 < N1B; ++n1) {
    const uint64_t r0 = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)ra_hi, n1) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)ra_lo, n1);
    // channel stride of the run in bytes: the frame size in the call's input, 256 samples in the history, 0 in the zeros
    const uint64_t cs = ((zmask >> n1) & 1u) ? 0u : (((hmask >> n1) & 1u) ? 4u * kFirHist : cs_in);
    // (addresses built from integers: say that they are GLOBAL ones, or the loads become flat loads)
    typedef const float __attribute__((address_space(1))) *gptr;
    const gptr sa = (gptr)(r0 + (uint64_t)a * cs);
    const gptr sb = (gptr)(b < M ? r0 + (uint64_t)b * cs : a_zero);
    float va, vb;
    if (IAMF_FFT_NT && n1 >= 4 && n1 < 12) {   // read by this wave alone: streamed past the caches
      va = __builtin_nontemporal_load(sa + lo);
      vb = __builtin_nontemporal_load(sb + lo);
    } else {                    // the hop's first and last 256 samples are also the neighbouring hops' overlap
      va = sa[lo];
      vb = sb[lo];
    }
    z[n1] = fft_mk(va, vb);
  }
}
// A use of all 16 registers that costs nothing: the compiler puts the wait for the loads that fill them in front of it.
__device__ __forceinline__ void fft_touch(fft_c32 (&z)[16]) {
  asm volatile(""
               : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]),
                 "+v"(z[9]), "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15]));
}

// where the stage left sample m (0 .. 3071, a multiple of 4) of the pass for ear e: four consecutive samples lie together
__device__ __forceinline__ const float *fft_y_ptr(const float *scratch_all, int e, int m) {
  const int hop = m / kFftHop;
  return scratch_all + hop * (kFftScratch * 2) + e * kFftHop + (m - hop * kFftHop);
}

// y[e][c0 .. c0 + 3072) of both ears -> the waves' scratch areas (fft_y_ptr).
// Wave w takes hop w = samples [c0 + 768 w, + 768).  All 256 threads call it; the caller synchronises afterwards.
// gy != nullptr: the outputs go to global memory instead, planar [2 ears][p.total] for this stream (fir_fft_kernel).
// The same fetch where the 1024-sample window of a hop consists of at most TWO stretches that are contiguous per channel
// and share one channel stride (frame size a multiple of 256 and >= 1024: the window crosses at most one frame boundary,
// and "before the call" / "past its end" begin at one; the history is kept at the input's stride for this, RenderParams::
// fir_pre; the zeros too).  Run n1 of channel c is then  base + 256 n1 + c * stride  with base one of two wave-uniform
// addresses: a scalar compare + select per run, the run's 256 n1 in the instruction's offset field and the channel in the
// lane's offset register (two additions per pair).  The general form above spends per LOAD two v_readlane, the wait states
// between a VALU-written SGPR and its use as an address, and a 64-bit scalar multiply-add — 240 of the ~660 instruction
// slots of a pair step; without its sample fetch the stage kernel took 1.06 instead of 1.51 ms (tools/debug/fft_exp.sh).
template <int M, int N1A, int N1B>
__device__ __forceinline__ void fft_fetch2(int pr, fft_c32 (&z)[16], uint64_t base1, uint64_t base2p, int kseg, unsigned cs4,
                                           unsigned lo4) {
  typedef const char __attribute__((address_space(1))) *gbptr;
  typedef const float __attribute__((address_space(1))) *gptr;
  int k = kseg;
  asm volatile("" : "+s"(k));   // per step: sixteen selected bases hoisted out of the pair loop would not fit the scalar registers
  const unsigned oa = lo4 + (unsigned)(2 * pr) * cs4, ob = oa + cs4;
#pragma unroll
  for (int n1 = N1A; n1 < N1B; ++n1) {
    const gbptr base = (gbptr)(n1 < k ? base1 : base2p);
    const gptr sa = (gptr)(base + 256 * n1 + oa), sb = (gptr)(base + 256 * n1 + ob);
    float va, vb;
    if (IAMF_FFT_NT && n1 >= 4 && n1 < 12) {   // read by this wave alone: streamed past the caches
      va = __builtin_nontemporal_load(sa);
      vb = __builtin_nontemporal_load(sb);
    } else {                    // the hop's first and last 256 samples are also the neighbouring hops' overlap
      va = *sa;
      vb = *sb;
    }
    z[n1] = fft_mk(va, vb);
  }
}
template <int M, bool FAST2 = false>
__device__ __forceinline__ void fir_stage_fft(const RenderParams &p, const float *in_s, const float *hist, int c0,
                                              fft_c32 *scratch_all, const FftTwiddles &tw, float *gy = nullptr,
                                              const float *pre = nullptr) {
  const int t = threadIdx.x, lane = t & 63, lane_ = lane;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform, and known to be: what follows stays in scalar registers
  const int b0 = c0 + kFftHop * w;   // the hop's first new sample, relative to the call
  if (b0 >= p.total) return;         // wave-uniform: the hop lies past the end of the call
  if (IAMF_FFT_EXP == 4) return;
  fft_c32 *S = scratch_all + w * kFftScratch;
  constexpr int kPairs = (M + 1) / 2;
  // Where the 64-sample runs n = b0 - 256 + 64 n1 + [0, 64) of a channel come from (the same for every channel).
  // b0 is a multiple of 256, so samples before the call exist only in the call's first hop and there they are exactly the
  // runs n1 < 4 (history samples 64 n1 ..); the call's length is a multiple of 64, so "past the end" is decided per run;
  // and the frame size is a multiple of 64 (the host takes this stage only then: fir_stage_choice), so a run never
  // straddles a frame.  Run n1 of channel c starts at  base + c * cstride  with base / cstride = a place in the call's
  // input / the frame size, in the history / 256, or — past the end of the call — a block of zeros / 0.  LANE n1 works
  // that out for run n1 (one division per hop, in parallel) and keeps it; fft_fetch reads it across with v_readlane.
  unsigned ra_lo, ra_hi, zmask, hmask;
  const uint64_t a_zero = reinterpret_cast<uint64_t>(p.fir_zero);
  {
    const int fs = p.frame_size;
    const int n1 = lane & 15;
    const int n = b0 - (kFftN - kFftHop) + 64 * n1;
    const int nn = n < 0 ? 0 : n;
    const int f = nn / fs, i = nn - f * fs;
    uint64_t ad = reinterpret_cast<uint64_t>(in_s) + 4 * (uint64_t)((int64_t)f * p.in_frame_stride + i);
    unsigned cs = 4u * (unsigned)fs;
    if (n < 0) {            // history sample 256 + n
      ad = reinterpret_cast<uint64_t>(hist) + 4 * (uint64_t)(kFirHist + n);
      cs = 4u * kFirHist;
    } else if (n >= p.total) {
      ad = a_zero;
      cs = 0;
    }
    ra_lo = (unsigned)ad;
    ra_hi = (unsigned)(ad >> 32);
    // bit n1: run n1 is history / lies past the end of the call (the 16 flags of lanes 0 .. 15, wave-uniform)
    hmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(__ballot(n < 0) & 0xffffu));
    zmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(__ballot(n >= 0 && n >= p.total) & 0xffffu));
    (void)cs;
  }
  const unsigned cs_in = 4u * (unsigned)p.frame_size;
  const unsigned lo = (unsigned)lane;
  // FAST2: the two stretches of the window (wave-uniform; see fft_fetch2)
  uint64_t base1 = 0, base2p = 0;
  int kseg = 16;
  if constexpr (FAST2) {
    const int fs = p.frame_size;
    const int n0 = b0 - (kFftN - kFftHop);   // the window's first sample: -256 in a call's first hop, else >= 0
    int64_t nb = 0;                          // first sample of the second stretch
    if (n0 < 0) {
      base1 = reinterpret_cast<uint64_t>(pre);   // `pre` = the stream's sample -256 of channel 0
    } else {
      const int f0 = n0 / fs, i0 = n0 - f0 * fs;
      base1 = reinterpret_cast<uint64_t>(in_s + (int64_t)f0 * p.in_frame_stride + i0);
      nb = (int64_t)(f0 + 1) * fs;
    }
    kseg = (int)((nb - n0) >> 6);            // runs of the first stretch (>= 16: the window lies inside one frame)
    const uint64_t b2 = nb >= p.total ? a_zero : reinterpret_cast<uint64_t>(in_s + (nb / fs) * p.in_frame_stride);
    base2p = b2 - 256ull * (uint64_t)kseg;   // run n1 >= kseg: base2p + 256 n1
    base1 = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base1 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base1);
    base2p = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base2p >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base2p);
    kseg = __builtin_amdgcn_readfirstlane(kseg);
  }
  const unsigned lo4 = 4u * lo;
#define FFT_FETCH(pr, z)                                                          \
  do {                                                                            \
    if constexpr (FAST2) fft_fetch2<M, 0, 16>(pr, z, base1, base2p, kseg, cs_in, lo4); \
    else fft_fetch<M>(pr, z, ra_lo, ra_hi, zmask, hmask, cs_in, a_zero, lo);      \
  } while (0)
#define FFT_FETCH_HALF(pr, z, A, B)                                               \
  do {                                                                            \
    if constexpr (FAST2) fft_fetch2<M, A, B>(pr, z, base1, base2p, kseg, cs_in, lo4); \
    else fft_fetch<M, A, B>(pr, z, ra_lo, ra_hi, zmask, hmask, cs_in, a_zero, lo); \
  } while (0)
  fft_c32 u[16], v[16], z[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) u[r] = v[r] = fft_mk(0.f, 0.f);
  const float4 *pq = reinterpret_cast<const float4 *>(p.fir_pq);   // uniform base; the lane's share is added per load
  fft_c32 zb[16];
  // One pair: transform za (fetched a whole pair earlier), accumulate, then fetch the pair after next into the same
  // registers.  Vector-memory loads return IN ORDER per wave, so a short load issued behind a long one waits for it: with
  // the sample fetch (HBM, microseconds under load) at the top of the step, the second half of the spectra table (L2)
  // queued behind it and every pair paid an HBM round trip (tools/debug/fft_exp.sh: removing the fetch halved the kernel's
  // time).  Order now: table half 1 -> transform -> accumulate half 1 / table half 2 -> accumulate half 2 -> samples of
  // pair p + 2.  The table reads only ever queue behind a fetch that is a whole step old.
  auto pair_step = [&](int pr, fft_c32 (&za)[16]) {
    // an opaque copy of the lane index per step: the dozen LDS base addresses derived from it (exchange layouts, twiddle
    // rows) are recomputed here — a few instructions — instead of living in registers across the loop, where they were
    // the ones spilled (and a reload from scratch waits with vmcnt(0), i.e. for the sample fetch in flight)
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const fft_c32 *tw1 = tw.tw1 - lane_ + lane, *tw2 = tw.tw2 - (lane_ & 3) + (lane & 3);
    const float4 *tq = pq + pr * 16 * 64;
    float4 c[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) c[r] = IAMF_FFT_EXP == 1 ? make_float4(0.5f, 0.25f, 0.125f, 0.0625f) : (tq + r * 64)[lo];
    fft_pin();   // issued here, ahead of the transform, and stays here
    fft_fwd_a_tab(za, tw1);
    if (IAMF_FFT_EXP != 3) {
      fft_x1_write(za, lane, S);
      fft_wave_sync();
      fft_x1_read(za, lane, S);
    }
    fft_fwd_b_tab(za, tw2);
    if (IAMF_FFT_EXP != 3) {
#if IAMF_FFT_X2_LDS
      fft_wave_sync();
      fft_x2_write(za, lane, S);
      fft_wave_sync();
      fft_x2_read(za, lane, S);
      fft_wave_sync();
#else
      fft_wave_sync();   // exchange 1 has been read back: its LDS rows may be overwritten by the next pair
      fft_x2_quads(za);
#endif
    }
    fft_fwd_c(za);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      u[r] = fft_cmac(u[r], za[r], fft_mk(c[r].x, c[r].y));
      v[r] = fft_cmac(v[r], za[r], fft_mk(c[r].z, c[r].w));
      c[r] = IAMF_FFT_EXP == 1 ? make_float4(0.5f, 0.25f, 0.125f, 0.0625f) : (tq + (8 + r) * 64)[lo];
    }
    fft_pin();
    // za[0 .. 7] are free: the first half of the samples of the pair after next (past the last pair: the last one again,
    // dropped; no branch) — issued HERE, between the two halves of the accumulation: the second half of the table was
    // requested a few instructions ago and the L2 round trip it needs is what issuing these sixteen loads takes
    const int prn = pr + 2 < kPairs ? pr + 2 : kPairs - 1;
    if (IAMF_FFT_EXP != 2) FFT_FETCH_HALF(prn, za, 0, 8);
    fft_pin();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      u[8 + r] = fft_cmac(u[8 + r], za[8 + r], fft_mk(c[r].x, c[r].y));
      v[8 + r] = fft_cmac(v[8 + r], za[8 + r], fft_mk(c[r].z, c[r].w));
    }
    fft_pin();
    if (IAMF_FFT_EXP != 2) FFT_FETCH_HALF(prn, za, 8, 16);
    fft_pin();
  };
  FFT_FETCH(0, z);
  FFT_FETCH(kPairs > 1 ? 1 : 0, zb);
  // Enter the loop in the state its back edge has — one fetch (32 loads) in flight, the step's own registers arrived:
  // with both fetches pending (64 loads, more than the wait counter counts) the compiler put s_waitcnt vmcnt(0) at the
  // loop's head, which in steady state waits for the fetch issued a moment earlier: an HBM round trip per pair.
  fft_touch(z);
  fft_touch(zb);   // (both: a fetch still pending at the loop's head from the prologue — into registers the allocator
                   //  gave it there — makes the head's wait a vmcnt(0) for every trip; this one is paid once per hop)
  // two pairs per trip, the sample registers alternating: no copies between pairs
#pragma unroll 1
  for (int pr = 0; pr + 1 < kPairs; pr += 2) {
    pair_step(pr, z);
    pair_step(pr + 1, zb);
  }
  if constexpr (kPairs & 1) pair_step(kPairs - 1, z);
  fft_mirror_write(v, lane, S);
  fft_wave_sync();
  fft_mirror_read_add(u, lane, S);
  fft_wave_sync();
  fft_inv_c(u);
#if IAMF_FFT_X2_LDS
  fft_x2_write_back(u, lane, S);
  fft_wave_sync();
  fft_x2_read_back(u, lane, S);
#else
  fft_x2_quads(u);
#endif
  fft_inv_b_tab(u, tw.tw2);
  fft_wave_sync();
  fft_x1_write_back(u, lane, S);
  fft_wave_sync();
  fft_x1_read_back(u, lane, S);
  fft_inv_a_tab(u, tw.tw1);
  // block samples 256 .. 1023 are the hop's 768 outputs: re = left, im = right.  They go to the wave's own scratch, which
  // nobody else touches and the wave itself is done with ([2][768] floats; the last exchange has been read back above).
  fft_wave_sync();
  if (gy) {   // (wave-uniform) 64 consecutive floats per store instruction; the call's length is a multiple of 64
#pragma unroll
    for (int n1 = 4; n1 < 16; ++n1) {
      const int m = b0 + 64 * (n1 - 4);
      if (m < p.total) {
        __builtin_nontemporal_store(u[n1].x, gy + m + lane);
        __builtin_nontemporal_store(u[n1].y, gy + p.total + m + lane);
      }
    }
  } else {
    float *y = reinterpret_cast<float *>(S);
#pragma unroll
    for (int n1 = 4; n1 < 16; ++n1) {
      y[64 * (n1 - 4) + lane] = u[n1].x;
      y[kFftHop + 64 * (n1 - 4) + lane] = u[n1].y;
    }
  }
#undef FFT_FETCH
#undef FFT_FETCH_HALF
}

// The stage as a kernel of its own (round 3, what the batch launches for kind FIR): overlap-save blocks are independent —
// a hop needs its 1024 input samples and nothing of its neighbours' results — so there is no reason to run them behind one
// another inside a per-stream workgroup.  Grid (passes of 3072 samples, streams): a workgroup = four waves = four hops; y goes
// to HBM as planar f32 [stream][2][total], and the limiter / pack of the SAME call is the plain two-channel matrix kernel
// (render_fast_kernel<2, 2>, identity matrix, four workgroups per CU) over that.  Costs 16 B per sample-frame of extra HBM
// traffic (y written and read once) and wins back what the fused kernel lost by running the limiter stages at two
// workgroups per CU and the hops of a stream one pass after the other.
#ifndef IAMF_FFT_OCC
#define IAMF_FFT_OCC 2   // workgroups per CU the stage kernel is compiled for (tools/debug/fft_exp.sh: 3 = 168 registers)
#endif
template <int M, bool FAST2 = false>
__global__ __launch_bounds__(256, IAMF_FFT_OCC) void fir_fft_kernel(const RenderParams p, float *gy, int64_t gy_stream_stride) {
  extern __shared__ float fft_lds[];
  const int s = blockIdx.y + p.stream0, t = threadIdx.x;
  FftTwiddles tw;
  fft_load_twiddles(p.fir_tw, t, fft_lds + kFftHops * kFftScratch * 2, tw);
  __syncthreads();
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;
  const float *hist = p.fir_hist + (int64_t)s * M * kFirHist;
  // (FAST2) the stream's rows of the history kept at the input's channel stride: G = frame size / 256 streams per slab
  // Both instantiations keep BOTH copies of the history up to date: a call that ends inside a frame runs the
  // general fetch, the whole-frame call after it the two-base fetch — which reads `fir_pre` of the call before it.
  const bool has_pre = p.fir_pre_next != nullptr;   // the batch keeps the copy: frame size a multiple of 256, >= 1024
  const int g256 = has_pre ? p.frame_size >> 8 : 1;
  const int64_t pre_off = has_pre ? ((int64_t)(s / g256) * M) * p.frame_size + (int64_t)(s % g256) * kFirHist : 0;
  fir_stage_fft<M, FAST2>(p, in_s, hist, kFftSpan * (int)blockIdx.x, reinterpret_cast<fft_c32 *>(fft_lds), tw,
                          gy + (int64_t)s * gy_stream_stride, FAST2 ? p.fir_pre + pre_off : nullptr);
  if (blockIdx.x == 0) {   // input history for the next call: the last 256 samples of [old history | this call's input]
    float *hn = p.fir_hist_next + (int64_t)s * M * kFirHist;
    for (int ch = 0; ch < M; ++ch) {
      const float x = fir_input(p, in_s, hist, ch, p.total - kFirHist + t);
      hn[ch * kFirHist + t] = x;
      if (has_pre) p.fir_pre_next[pre_off + (int64_t)ch * p.frame_size + t] = x;
    }
  }
}
#endif  // __HIPCC__
