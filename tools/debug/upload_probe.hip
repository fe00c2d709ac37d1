// Probe (not product): how fast a kernel reads pinned host memory over PCIe, by size, by loads per lane, and with the
// lines just written by CPU threads (as the group's staging rows are) or not.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int U>
__global__ __launch_bounds__(256) void up(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { size_t k = i + (size_t)u * gridDim.x * 256; if (k < n16) v[u] = __builtin_nontemporal_load(src + k); }
#pragma unroll
  for (int u = 0; u < U; ++u) { size_t k = i + (size_t)u * gridDim.x * 256; if (k < n16) dst[k] = v[u]; }
}
int main() {
  const size_t MB = 1 << 20;
  void *h, *d;
  hipHostMalloc(&h, 16 * MB, 0); hipMalloc(&d, 16 * MB);
  memset(h, 1, 16 * MB);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dirty = 0; dirty < 3; ++dirty)
    for (size_t bytes : {MB / 2, MB, 2 * MB, 8 * MB})
      for (int U : {1, 4}) {
        float acc = 0; const int R = 30;
        for (int r = 0; r < R; ++r) {
          if (dirty == 1) memset(h, r, bytes);                       // written by this thread just before
          if (dirty == 2) {                                          // written by 8 threads, as a pool would
            std::vector<std::thread> th;
            for (int t = 0; t < 8; ++t) th.emplace_back([=] { memset((char *)h + t * (bytes / 8), r + t, bytes / 8); });
            for (auto &x : th) x.join();
          }
          size_t n16 = bytes / 16; unsigned g = (unsigned)((n16 / U + 255) / 256);
          hipEventRecord(e0, st);
          if (U == 1) hipLaunchKernelGGL(up<1>, dim3(g), dim3(256), 0, st, (const u32x4 *)h, (u32x4 *)d, n16);
          else hipLaunchKernelGGL(up<4>, dim3(g), dim3(256), 0, st, (const u32x4 *)h, (u32x4 *)d, n16);
          hipEventRecord(e1, st); hipStreamSynchronize(st);
          float ms; hipEventElapsedTime(&ms, e0, e1); acc += ms;
        }
        printf("dirty %d  %4zu KB  loads/lane %d: %.1f us  %.1f GB/s\n", dirty, bytes / 1024, U, acc / R * 1e3, bytes / (acc / R * 1e-3) / 1e9);
      }
  return 0;
}
