// iamf_probe.hip — diagnostic entry point: the render kernels' HBM traffic SHAPE with no compute, so that a
// measured rate can be put next to what the memory system delivers for that shape (bench.py:
// roofline.same_traffic_no_compute).  One 256-thread workgroup per stream; per 1024-sample chunk every lane
// reads `rows` x 16 B (rows 4 KiB apart inside a frame of rows x 4 KiB, prefetched one chunk ahead, non-temporal)
// and writes `pieces` x 16 B, each store instruction covering 1 KiB contiguous per wave — the geometry of
// render_fast.hpp (rows 16, pieces 1) and render_wide4.hpp (cfg2: 12 / 6, cfg3: 16 / 12).
// Round-2 findings: read-only this pattern runs at 7.1 TB/s; with a share of writes at 6.2-6.5 TB/s (headline shape)
// when input and output lie in memory regions of different kinds and at 5.45 TB/s when they lie in regions of the
// same kind (NOTEBOOK.md 3, tools/debug/placement_va_probe.hip) — iamf_hip_pick_buffer_pair below finds a fast pair among
// candidates.  Nothing here is used by the render path.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/iamf_hip.h"

namespace {

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <int ROWS>
__global__ __launch_bounds__(256, 2) void traffic_probe_kernel(const v4 *in, int64_t in_stream_stride4, u4 *out,
                                                               int64_t out_stream_stride4, int chunks, int pieces) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (int64_t)s * in_stream_stride4;
  u4 *dst = out + (int64_t)s * out_stream_stride4;
  v4 x[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * 256 + t);
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    const int cn = c + 1 < chunks ? c + 1 : c;  // unconditional prefetch (the last one re-reads)
#pragma unroll
    for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + ((int64_t)cn * ROWS + m) * 256 + t);
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
    for (int k = 0; k < pieces; ++k)
      __builtin_nontemporal_store(w, dst + ((int64_t)c * pieces * 4 + wave * pieces + k) * 64 + lane);
  }
}

}  // namespace

extern "C" int iamf_hip_probe_traffic(int n_streams, int chunks, int rows, int pieces, const void *d_in,
                                      int64_t in_stream_stride_bytes, void *d_out, int64_t out_stream_stride_bytes,
                                      void *stream) {
  if (n_streams <= 0 || chunks <= 0 || pieces < 0 || pieces > 64 || !d_in || !d_out ||
      in_stream_stride_bytes < (int64_t)chunks * rows * 4096 || out_stream_stride_bytes < (int64_t)chunks * pieces * 4096 ||
      (in_stream_stride_bytes & 15) || (out_stream_stride_bytes & 15))
    return IAMF_HIP_ERR_BAD_ARG;
  const v4 *in = static_cast<const v4 *>(d_in);
  u4 *out = static_cast<u4 *>(d_out);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_streams), block(256);
  const int64_t is4 = in_stream_stride_bytes / 16, os4 = out_stream_stride_bytes / 16;
  switch (rows) {
#define CASE_R(R) \
  case R: hipLaunchKernelGGL(traffic_probe_kernel<R>, grid, block, 0, st, in, is4, out, os4, chunks, pieces); break;
    CASE_R(1) CASE_R(2) CASE_R(4) CASE_R(6) CASE_R(8) CASE_R(9) CASE_R(10) CASE_R(12) CASE_R(16)
#undef CASE_R
    default: return IAMF_HIP_ERR_UNIMPLEMENTED;
  }
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}

// Times the traffic kernel on every (input candidate, output candidate) pair — 1 + reps launches each, the median of
// the reps — and reports the fastest pair: what a host with long-lived stream buffers does once at start-up
// (INTEGRATION.md 5).  Synchronous; the candidates' contents are read / overwritten.
extern "C" int iamf_hip_pick_buffer_pair(int n_streams, int chunks, int rows, int pieces, const void *const *d_in_candidates,
                                         int n_in, int64_t in_stream_stride_bytes, void *const *d_out_candidates, int n_out,
                                         int64_t out_stream_stride_bytes, void *stream, int *best_in, int *best_out,
                                         float *ms_out) {
  if (!d_in_candidates || !d_out_candidates || n_in <= 0 || n_out <= 0 || n_in > 256 || n_out > 256 || !best_in || !best_out)
    return IAMF_HIP_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    return IAMF_HIP_ERR_DEVICE;
  }
  constexpr int kReps = 3;
  int rc = IAMF_HIP_OK;
  float best = 0.f;
  *best_in = *best_out = 0;
  for (int i = 0; i < n_in && rc == IAMF_HIP_OK; ++i)
    for (int j = 0; j < n_out && rc == IAMF_HIP_OK; ++j) {
      float t[kReps];
      for (int r = -1; r < kReps && rc == IAMF_HIP_OK; ++r) {  // r = -1: untimed first touch
        if (hipEventRecord(e0, st) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
        if (rc == IAMF_HIP_OK)
          rc = iamf_hip_probe_traffic(n_streams, chunks, rows, pieces, d_in_candidates[i], in_stream_stride_bytes,
                                      d_out_candidates[j], out_stream_stride_bytes, stream);
        if (rc == IAMF_HIP_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = IAMF_HIP_ERR_DEVICE;
        float ms = 0.f;
        if (rc == IAMF_HIP_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
        if (r >= 0) t[r] = ms;
      }
      if (rc != IAMF_HIP_OK) break;
      float a = t[0], b = t[1], c = t[2];   // median of three
      const float med = a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
      if (ms_out) ms_out[i * n_out + j] = med;
      if ((i == 0 && j == 0) || med < best) {
        best = med;
        *best_in = i;
        *best_out = j;
      }
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// ---- self-test of w4_quot (render_common.hpp): every f32 numerator against the IEEE division, on the device ----
namespace {
__device__ __forceinline__ float st_quot(float n, float d, float r, bool &ok) {   // the same three operations
  const float an = __builtin_fabsf(n);
  ok = ok && an >= 0x1p-100f && an < 0x1p126f;
  const float q = n * r;
  const float e = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e, r, q);
}
__global__ __launch_bounds__(256) void shared_divisor_sweep_kernel(float d, unsigned long long *out) {
  const float r = 1.0f / d;
  unsigned long long in_range = 0, bad_in = 0, bad_out = 0;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;   // 2^20 threads x 2^12 numerators each
  for (uint32_t i = 0; i < 4096u; ++i) {
    const uint32_t u = (i << 20) | t;
    const float n = __uint_as_float(u);
    bool ok = true;
    const float f = st_quot(n, d, r, ok), g = n / d;
    const bool same = __float_as_uint(f) == __float_as_uint(g);
    in_range += ok;
    bad_in += ok && !same;
    bad_out += !ok && !same && !(g != g && f != f);
  }
  atomicAdd(&out[0], in_range);
  atomicAdd(&out[1], bad_in);
  atomicAdd(&out[2], bad_out);
}
}  // namespace

extern "C" int iamf_hip_selftest_shared_divisor(float divisor, uint64_t counts[3]) {
  if (!counts || !(divisor > 0.f)) return IAMF_HIP_ERR_BAD_ARG;
  unsigned long long *d_out = nullptr;
  if (hipMalloc(&d_out, 3 * sizeof(unsigned long long)) != hipSuccess) return IAMF_HIP_ERR_DEVICE;
  int rc = IAMF_HIP_OK;
  if (hipMemset(d_out, 0, 3 * sizeof(unsigned long long)) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
  if (rc == IAMF_HIP_OK) {
    hipLaunchKernelGGL(shared_divisor_sweep_kernel, dim3(4096), dim3(256), 0, nullptr, divisor, d_out);
    unsigned long long h[3] = {0, 0, 0};
    if (hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
    counts[0] = h[0];
    counts[1] = h[1];
    counts[2] = h[2];
  }
  (void)hipFree(d_out);
  return rc;
}
