// deep_prefetch_probe.hip — the headline kernel's traffic (16 planar rows x 4 KiB read per chunk, 4 KiB written, 512 workgroups) with the next
// chunk's loads issued ONE or TWO chunks ahead: what would a second set of prefetch registers buy at two workgroups per CU?
// `spin` = dependent VALU work per chunk (the kernel's ~2.2k cycles of compute).   hipcc --offload-arch=gfx950 -O3 ... -o tools/bin/deep_prefetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int ROWS = 16;
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void probe(const v4 *in, u4 *out, int chunks, int spin) {
  const int s = blockIdx.x, t = threadIdx.x;
  const v4 *src = in + (long)s * chunks * ROWS * 256;
  const long rstride = (long)chunks * 256;
  v4 x[DEPTH][ROWS];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int m = 0; m < ROWS; ++m) x[d][m] = __builtin_nontemporal_load(src + m * rstride + (long)(d < chunks ? d : 0) * 256 + t);
  for (int c0 = 0; c0 < chunks; c0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = c0 + d;
      if (c >= chunks) break;
      float a = 0.f;
#pragma unroll
      for (int m = 0; m < ROWS; ++m) {
        a += x[d][m].x + x[d][m].y + x[d][m].z + x[d][m].w;
        const int cn = c + DEPTH < chunks ? c + DEPTH : c;   // (the tail re-reads its own chunk: dropped)
        x[d][m] = __builtin_nontemporal_load(src + m * rstride + (long)cn * 256 + t);
      }
      for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
      out[((long)s * chunks + c) * 256 + t] = u4{__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
    }
  }
}
template <int DEPTH> float run(const v4 *in, u4 *out, int S, int chunks, int spin) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    probe<DEPTH><<<S, 256>>>(in, out, chunks, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  return best;
}
int main() {
  const int S = 512, chunks = 64, NI = 6, NO = 4;
  const size_t ib = (size_t)S * chunks * ROWS * 4096, ob = (size_t)S * chunks * 4096;
  std::vector<v4 *> in(NI); std::vector<u4 *> out(NO);
  for (auto &p : in) { hipMalloc(&p, ib); hipMemset(p, 0, ib); }
  for (auto &p : out) { hipMalloc(&p, ob); hipMemset(p, 0, ob); }
  for (int i = 0; i < NI; ++i)
    for (int o = 0; o < NO; ++o) {
      printf("in %d out %d:", i, o);
      for (int spin : {0, 300}) {
        const float a = run<1>(in[i], out[o], S, chunks, spin), b = run<2>(in[i], out[o], S, chunks, spin);
        printf("  spin %3d  depth1 %.3f ms %5.1f Gs/s  depth2 %.3f ms %5.1f Gs/s", spin, a, S * chunks * 1024.0 / a / 1e6, b, S * chunks * 1024.0 / b / 1e6);
      }
      printf("\n");
    }
  return 0;
}
