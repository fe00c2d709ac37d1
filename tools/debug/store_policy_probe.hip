// store_policy_probe.hip — the render kernels' traffic shape (tools/debug/rw_mix_probe.hip) with every cache-policy
// combination gfx950 offers on the STORES (sc0 / sc1 / nt bits) and two on the loads: does any of them lift the
// read + write regime above what plain / non-temporal stores reach?
//   hipcc --offload-arch=gfx950 -O3 tools/debug/store_policy_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <int POL>
__device__ __forceinline__ void store_pol(u4 *to, u4 w) {
  if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(to), "v"(w) : "memory");
  if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(to), "v"(w) : "memory");
  if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(to), "v"(w) : "memory");
  if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(to), "v"(w) : "memory");
  if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(to), "v"(w) : "memory");
  if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(to), "v"(w) : "memory");
  if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(to), "v"(w) : "memory");
  if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(to), "v"(w) : "memory");
}

template <int ROWS, int PIECES, int POL, bool NTLOAD>
__global__ __launch_bounds__(256, 2) void probe(const v4 *in, u4 *out, int chunks) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (long)s * chunks * ROWS * 256;
  u4 *dst = out + (long)s * chunks * PIECES * 256;
  v4 x[ROWS];
  auto ld = [&](const v4 *p) { return NTLOAD ? __builtin_nontemporal_load(p) : *p; };
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = ld(src + m * 256 + t);
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    if (c + 1 < chunks) {
#pragma unroll
      for (int m = 0; m < ROWS; ++m) x[m] = ld(src + ((long)(c + 1) * ROWS + m) * 256 + t);
    }
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
#pragma unroll
    for (int k = 0; k < PIECES; ++k) store_pol<POL>(dst + ((long)c * PIECES * 4 + wave * PIECES + k) * 64 + lane, w);
  }
}

template <int ROWS, int PIECES, int POL, bool NTLOAD>
float time_one(const v4 *in, u4 *out, int S, int chunks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    probe<ROWS, PIECES, POL, NTLOAD><<<S, 256>>>(in, out, chunks);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

template <int ROWS, int PIECES>
void run(const char *name, const v4 *in, u4 *out, int S, int chunks) {
  const char *pol[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  const double bytes = (double)S * chunks * (ROWS + PIECES) * 4096;
  float ms[8][2];
  ms[0][0] = time_one<ROWS, PIECES, 0, false>(in, out, S, chunks); ms[0][1] = time_one<ROWS, PIECES, 0, true>(in, out, S, chunks);
  ms[1][0] = time_one<ROWS, PIECES, 1, false>(in, out, S, chunks); ms[1][1] = time_one<ROWS, PIECES, 1, true>(in, out, S, chunks);
  ms[2][0] = time_one<ROWS, PIECES, 2, false>(in, out, S, chunks); ms[2][1] = time_one<ROWS, PIECES, 2, true>(in, out, S, chunks);
  ms[3][0] = time_one<ROWS, PIECES, 3, false>(in, out, S, chunks); ms[3][1] = time_one<ROWS, PIECES, 3, true>(in, out, S, chunks);
  ms[4][0] = time_one<ROWS, PIECES, 4, false>(in, out, S, chunks); ms[4][1] = time_one<ROWS, PIECES, 4, true>(in, out, S, chunks);
  ms[5][0] = time_one<ROWS, PIECES, 5, false>(in, out, S, chunks); ms[5][1] = time_one<ROWS, PIECES, 5, true>(in, out, S, chunks);
  ms[6][0] = time_one<ROWS, PIECES, 6, false>(in, out, S, chunks); ms[6][1] = time_one<ROWS, PIECES, 6, true>(in, out, S, chunks);
  ms[7][0] = time_one<ROWS, PIECES, 7, false>(in, out, S, chunks); ms[7][1] = time_one<ROWS, PIECES, 7, true>(in, out, S, chunks);
  for (int p = 0; p < 8; ++p)
    printf("%-9s stores %-10s  plain loads %.3f ms %5.0f GB/s   nt loads %.3f ms %5.0f GB/s\n", name, pol[p], ms[p][0], bytes / ms[p][0] / 1e6,
           ms[p][1], bytes / ms[p][1] / 1e6);
}

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 512, chunks = 64;
  v4 *in;
  u4 *out;
  (void)hipMalloc(&in, (size_t)S * chunks * 16 * 4096);
  (void)hipMalloc(&out, (size_t)S * chunks * 12 * 4096);
  (void)hipMemset(in, 0, (size_t)S * chunks * 16 * 4096);
  run<16, 1>("headline", in, out, S, chunks);
  run<12, 6>("cfg2", in, out, S, chunks);
  run<16, 12>("cfg3", in, out, S, chunks);
  return 0;
}
