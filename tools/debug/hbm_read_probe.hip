// hbm_read_probe.hip — how fast can the render kernel's HBM access pattern go with no compute?
// Same geometry as render_fast_kernel<16,2>: one 256-thread workgroup per stream, per 1024-sample
// chunk every thread reads 16 x 16 B (16 channel rows, 4 KiB apart) and writes 16 B.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/hbm_read_probe.hip -o gpurun_out/hbm_probe && ./hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(256) void probe(const float4 *in, uint4 *out, int chunks, long stream_stride4) {
  const int s = blockIdx.x, t = threadIdx.x;
  const float4 *src = in + (long)s * stream_stride4;
  float4 x[DEPTH][16];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int m = 0; m < 16; ++m) x[d][m] = src[((long)d * 16 + m) * 256 + t];
  for (int c = 0; c < chunks; ++c) {
    const int slot = (c + DEPTH - 1) % DEPTH;
    if (c + DEPTH - 1 < chunks)
#pragma unroll
      for (int m = 0; m < 16; ++m) x[(DEPTH - 1)][m] = src[((long)(c + DEPTH - 1) * 16 + m) * 256 + t];
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) a += x[0][m].x + x[0][m].y + x[0][m].z + x[0][m].w;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int m = 0; m < 16; ++m) x[d][m] = x[d + 1][m];
    uint4 w = {__float_as_uint(a), 0u, 0u, (unsigned)slot};
    out[((long)s * chunks + c) * 256 + t] = w;
    __syncthreads();
  }
}

int main() {
  const int S = 512, chunks = 64;
  const size_t in_bytes = (size_t)S * chunks * 16 * 1024 * 4, out_bytes = (size_t)S * chunks * 256 * 16;
  float4 *in;
  uint4 *out;
  hipMalloc(&in, in_bytes);
  hipMalloc(&out, out_bytes);
  hipMemset(in, 1, in_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (variant == 0) probe<2><<<S, 256>>>(in, out, chunks, (long)chunks * 16 * 256);
      if (variant == 1) probe<3><<<S, 256>>>(in, out, chunks, (long)chunks * 16 * 256);
      if (variant == 2) probe<2><<<S * 4, 256>>>(in, out, chunks / 4, (long)(chunks / 4) * 16 * 256);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("variant %d: %.3f ms  -> %.0f GB/s (read+write %.2f GB)\n", variant, best,
           (in_bytes + out_bytes) / best / 1e6, (in_bytes + out_bytes) / 1e9);
  }
  return 0;
}
