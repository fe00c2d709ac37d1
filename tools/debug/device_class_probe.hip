// device_class_probe.hip — what distinguishes the "fast" and "slow" MI355X devices of NOTEBOOK.md 5?  Plain grid-stride
// read / write / copy of 2 GiB against the render kernels' one-workgroup-per-stream traffic shapes, in one run.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/device_class_probe.hip -o /tmp/dcp && /tmp/dcp
#include <hip/hip_runtime.h>
#include <cstdio>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ __launch_bounds__(256) void k_read(const v4 *in, size_t n, float *sink) {
  v4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += __builtin_nontemporal_load(in + i);
  if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;
}
__global__ __launch_bounds__(256) void k_write(u4 *out, size_t n) {
  const u4 w = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(w, out + i);
}
__global__ __launch_bounds__(256) void k_copy(const v4 *in, v4 *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
template <int ROWS, int PIECES>
__global__ __launch_bounds__(256, 2) void k_stream(const v4 *in, u4 *out, int chunks, long in_stride4) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (long)s * in_stride4;
  u4 *dst = out + (long)s * chunks * (PIECES ? PIECES : 1) * 256;
  v4 x[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * 256 + t);
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    if (c + 1 < chunks) {
#pragma unroll
      for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + ((long)(c + 1) * ROWS + m) * 256 + t);
    }
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
#pragma unroll
    for (int k = 0; k < PIECES; ++k) __builtin_nontemporal_store(w, dst + ((long)c * PIECES * 4 + wave * PIECES + k) * 64 + lane);
    if (PIECES == 0 && a == 123.456f) dst[t] = w;
  }
}

template <typename F>
float best_ms(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const int S = 512, chunks = 64;
  const size_t in_bytes = (size_t)S * (chunks * 16 * 4096 + 4096), out_bytes = (size_t)S * chunks * 12 * 4096;
  v4 *in;
  u4 *out;
  float *sink;
  (void)hipMalloc(&in, in_bytes);
  (void)hipMalloc(&out, out_bytes);
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(in, 0, in_bytes);
  (void)hipMemset(out, 0, out_bytes);
  const size_t n = (size_t)S * chunks * 16 * 256;  // 16-byte elements in 2 GiB
  const long stride4 = (long)chunks * 16 * 256 + 256;  // the bench's 4 KiB stagger
  float ms;
  ms = best_ms([&] { k_read<<<4096, 256>>>(in, n, sink); });
  printf("grid-stride read   2 GiB            %.3f ms  %5.0f GB/s\n", ms, n * 16 / ms / 1e6);
  ms = best_ms([&] { k_write<<<4096, 256>>>(out, n / 2); });
  printf("grid-stride write  1 GiB            %.3f ms  %5.0f GB/s\n", ms, n * 8 / ms / 1e6);
  ms = best_ms([&] { k_copy<<<4096, 256>>>(in, reinterpret_cast<v4 *>(out), n / 2); });
  printf("grid-stride copy   1 GiB -> 1 GiB   %.3f ms  %5.0f GB/s (read + written)\n", ms, n * 16 / ms / 1e6);
  ms = best_ms([&] { k_stream<16, 0><<<S, 256>>>(in, out, chunks, stride4); });
  printf("per-stream pattern, read only       %.3f ms  %5.0f GB/s\n", ms, (double)S * chunks * 16 * 4096 / ms / 1e6);
  ms = best_ms([&] { k_stream<16, 1><<<S, 256>>>(in, out, chunks, stride4); });
  printf("per-stream pattern, headline 64+4   %.3f ms  %5.0f GB/s\n", ms, (double)S * chunks * 17 * 4096 / ms / 1e6);
  ms = best_ms([&] { k_stream<12, 6><<<S, 256>>>(in, out, chunks, stride4); });
  printf("per-stream pattern, cfg2 48+24      %.3f ms  %5.0f GB/s\n", ms, (double)S * chunks * 18 * 4096 / ms / 1e6);
  ms = best_ms([&] { k_stream<16, 12><<<S, 256>>>(in, out, chunks, stride4); });
  printf("per-stream pattern, cfg3 64+48      %.3f ms  %5.0f GB/s\n", ms, (double)S * chunks * 28 * 4096 / ms / 1e6);
  return 0;
}
