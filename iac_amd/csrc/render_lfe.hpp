// render_lfe.hpp — HOA LFE generator (reference h2m_rdr.c:1151-1239, compiled in by the
// reference's switch -DDISABLE_LFE_HOA=0, call site IAMF_decoder.c:2625-2636): a 2nd-order Butterworth
// low-pass at 120 Hz over ambisonics channel 0 (W) whose scaled output fills the layout's LFE slot(s).
//
//   y[j] = a1*w[j] + a2*w[j-1] + a3*w[j-2] - b1*y[j-1] - b2*y[j-2]      (f32, left to right)
//
// Every rounding of the reference is kept, so the feedback part is a serial recurrence per stream:
// three dependent f32 operations per sample (b1*y1 -> u - p -> - q), nothing to scan.  What IS
// parallel is split off:
//   lfe_ff_kernel     u[j] = (a1*w[j] + a2*w[j-1]) + a3*w[j-2], one lane per 4 samples, written through
//                     an LDS transpose into [block of 64 streams][quad][stream] order;
//   lfe_chain_kernel  lane = stream, 64 streams per wave: y = (u - b1*y1) - b2*y2 over the whole call,
//                     u read as coalesced 1 KiB rows prefetched two blocks ahead, y written IN PLACE over u
//                     (first version: y stored per stream, 64 scattered 16-byte transactions per store
//                     instruction in front of the prefetch loads in the in-order memory pipe — 53 cycles per
//                     step, 1.44 ms per 64-frame call; in place: see NOTEBOOK.md 4.6);
//   the scaling (y * 0.5 or y / sqrt(n), both double expressions in the reference, h2m_rdr.c:1157-1163)
//   is done by the render kernel where it reads the slot.
// State per stream between calls: w[-1], w[-2], y[-1], y[-2] (lfe_filter_t's two histories).
#pragma once

#include "lfe_chain_asm.inc"

constexpr int kLfeTileQ = 16;   // quads (of 4 samples) per ff tile

struct LfeParams {
  const float *in;            // element PCM (planar per frame, as RenderParams::in)
  int64_t in_stream_stride, in_frame_stride;
  const float *pre_matrix;    // projection de-mapping [pre_l][M] or nullptr: w = sum_l in[l] * P[l][0]
  int32_t pre_l, pre_m;
  int32_t frame_size, n_streams, total, t4;   // t4 = quads per stream in the transposed buffer
  float a1, a2, a3, b1, b2;
  float *state;               // [n_streams][4]: w1, w2, y1, y2
  float *state_next;          // [n_streams][2]: w1, w2 after this call (adopted by the chain kernel)
  float4 *u_t;                // [n_blocks64][t4][64] quads: u from the ff kernel, y after the chain kernel
  int32_t s_first, s_count;   // the streams this call renders (iamf_hip_batch_render_range): the others keep their state
};

// where the render kernel finds the generator's output for sample k of stream s (floats into u_t)
__device__ __forceinline__ int64_t lfe_index(int s, int k, int t4) {
  return ((((int64_t)(s >> 6) * t4 + (k >> 2)) * 64 + (s & 63)) << 2) + (k & 3);
}

__device__ __forceinline__ float lfe_w_at(const LfeParams &p, int s, int k) {
  const int f = k / p.frame_size, i = k - f * p.frame_size;
  const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i;
  if (!p.pre_matrix) return src[0];
  float w = 0.f;  // IAMF_core_decoder.c:116-130: x[0] = 0; x[0] += in[l] * P[l][0]
  for (int l = 0; l < p.pre_l; ++l) w = w + src[(int64_t)l * p.frame_size] * p.pre_matrix[l * p.pre_m];
  return w;
}

__global__ __launch_bounds__(256) void lfe_ff_kernel(const LfeParams p) {
  __shared__ float4 tile[kLfeTileQ * 65];
  const int t = threadIdx.x;
  const int q_base = blockIdx.x * kLfeTileQ, sb = blockIdx.y;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int q = t & 15, sl = (t >> 4) + 16 * pass;
    const int s = sb * 64 + sl, k0 = 4 * (q_base + q);
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= p.s_first && s < p.s_first + p.s_count && k0 < p.total) {
      const float h1 = p.state[4 * s + 0], h2 = p.state[4 * s + 1];
      float wm1, wm2, w[4], uu[4];
      const bool quick = !p.pre_matrix && ((p.frame_size | p.in_stream_stride | p.in_frame_stride) & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && k0 + 4 <= p.total;
      if (quick) {   // the usual case: one division per quad instead of six (the quad lies inside one frame)
        const int f = k0 / p.frame_size, i0 = k0 - f * p.frame_size;
        const float *src = p.in + (int64_t)s * p.in_stream_stride + (int64_t)f * p.in_frame_stride + i0;
        const float4 q = *reinterpret_cast<const float4 *>(src);
        w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
        if (i0 >= 4) {
          wm1 = src[-1];
          wm2 = src[-2];
        } else if (k0 == 0) {
          wm1 = h1;
          wm2 = h2;
        } else {     // the frame before: its last two samples
          const float *prv = p.in + (int64_t)s * p.in_stream_stride + (int64_t)(f - 1) * p.in_frame_stride + p.frame_size;
          wm1 = prv[-1];
          wm2 = prv[-2];
        }
      } else {
        wm1 = k0 >= 1 ? lfe_w_at(p, s, k0 - 1) : h1;
        wm2 = k0 >= 2 ? lfe_w_at(p, s, k0 - 2) : (k0 == 1 ? h1 : h2);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!quick) w[i] = k0 + i < p.total ? lfe_w_at(p, s, k0 + i) : 0.f;
        const float a = i >= 1 ? w[i - 1] : wm1, b = i >= 2 ? w[i - 2] : (i == 1 ? wm1 : wm2);
        uu[i] = (p.a1 * w[i] + p.a2 * a) + p.a3 * b;
      }
      u = make_float4(uu[0], uu[1], uu[2], uu[3]);
      if (k0 + 4 >= p.total) {  // the quad with the call's last sample leaves the next call's input history
        const int il = p.total - 1 - k0;
        p.state_next[2 * s + 0] = w[il];
        p.state_next[2 * s + 1] = il >= 1 ? w[il - 1] : wm1;
      }
    }
    tile[q * 65 + sl] = u;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int q = (t >> 6) + 4 * pass, sl = t & 63;
    if (q_base + q < p.t4) p.u_t[((int64_t)sb * p.t4 + q_base + q) * 64 + sl] = tile[q * 65 + sl];
  }
}

__global__ __launch_bounds__(64) void lfe_chain_kernel(const LfeParams p) {
  const int lane = threadIdx.x, sb = blockIdx.x;
  const int s = sb * 64 + lane;
  const bool active = s >= p.s_first && s < p.s_first + p.s_count;
  const int sc = active ? s : p.s_first;
  float y1 = p.state[4 * sc + 2], y2 = p.state[4 * sc + 3];
  const float b1 = p.b1, b2 = p.b2;
  float4 *row = p.u_t + (int64_t)sb * p.t4 * 64 + lane;   // this lane's quad q sits at row[64 q]
  const int nfull = p.total >> 2;                          // quads whose four samples all belong to the call
  // One step: y = (u - b1*y1) - b2*y2, three dependent f32 operations (used for what the main loop leaves).
  auto step4 = [&](const float4 u) {
    float4 o;
    o.x = (u.x - b1 * y1) - b2 * y2;
    o.y = (u.y - b1 * o.x) - b2 * y1;
    o.z = (u.z - b1 * o.y) - b2 * o.x;
    o.w = (u.w - b1 * o.z) - b2 * o.y;
    y2 = o.z;
    y1 = o.w;
    return o;
  };
  // Main loop: whole rotations of IAMF_LFE_RING quads as one asm statement (lfe_chain_asm.inc, written by
  // tools/gen_lfe_chain_asm.py): per step v_sub, v_sub and ONE v_pk_mul_f32 that makes b1*y and b2*y from y where it
  // lies in the output quad; registers by hand, loads a rotation ahead, stores in place.  (As C++ the same loop is
  // 4.65 instruction slots per step: the compiler pairs the two products too and pays with copies into aligned pairs
  // and into the store's quad, NOTEBOOK.md 4.6.)
  int q_done = 0;
  {
    const int nrot = nfull / IAMF_LFE_RING;
    if (nrot > 0) {
      const float4 *base = p.u_t + (int64_t)sb * p.t4 * 64;   // wave-uniform: the block's first row
      const unsigned voff = (unsigned)lane * 16u;
      asm volatile(IAMF_LFE_ASM_BODY
                   : [y1] "+v"(y1), [y2] "+v"(y2)
                   : [base] "s"(base), [nrot] "s"(nrot), [voff] "v"(voff), [b1] "v"(b1), [b2] "v"(b2)
                   : IAMF_LFE_ASM_CLOBBERS);
      q_done = nrot * IAMF_LFE_RING;
    }
  }
  // what is left: fewer than a rotation of whole quads — all their loads first, then the steps (wave-uniform guards)
  {
    const int nrem = nfull - q_done;
    float4 rem[IAMF_LFE_RING];
#pragma unroll
    for (int i = 0; i < IAMF_LFE_RING; ++i)
      if (i < nrem) rem[i] = row[(int64_t)(q_done + i) * 64];
#pragma unroll
    for (int i = 0; i < IAMF_LFE_RING; ++i)
      if (i < nrem) row[(int64_t)(q_done + i) * 64] = step4(rem[i]);
  }
  const int left = p.total - 4 * nfull;  // 0..3 samples in the last quad
  if (left > 0) {
    const float4 u = row[(int64_t)nfull * 64];
    float4 o;
    o.x = (u.x - b1 * y1) - b2 * y2;
    o.y = (u.y - b1 * o.x) - b2 * y1;
    o.z = (u.z - b1 * o.y) - b2 * o.x;
    o.w = 0.f;
    row[(int64_t)nfull * 64] = o;
    const float r1 = left == 3 ? o.z : (left == 2 ? o.y : o.x);
    const float r2 = left == 3 ? o.y : (left == 2 ? o.x : y1);
    y2 = r2;
    y1 = r1;
  }
  if (active) {
    p.state[4 * s + 0] = p.state_next[2 * s + 0];
    p.state[4 * s + 1] = p.state_next[2 * s + 1];
    p.state[4 * s + 2] = y1;
    p.state[4 * s + 3] = y2;
  }
}
