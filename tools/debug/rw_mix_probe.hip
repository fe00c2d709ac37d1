// rw_mix_probe.hip — the wide4 kernel's HBM traffic shape with no compute: one 256-thread workgroup per stream,
// per 1024-sample chunk every lane reads ROWS x 16 B (rows 4 KiB apart, prefetched one chunk ahead) and writes
// PIECES x 16 B into its stream's contiguous output (each store instruction = 1 KiB contiguous per wave).
//   cfg2: ROWS 12, PIECES 6 (72 B per sample-frame)    cfg3: ROWS 16, PIECES 12 (112 B)    headline: 16, 1
// Variants: stores off / plain / non-temporal; stores issued right after the loads or at the end of the iteration
// behind a dependent delay (as the kernel does: limiter phases between prefetch and stores).
//   hipcc --offload-arch=gfx950 -O3 tools/debug/rw_mix_probe.hip -o tools/bin/rw_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <int ROWS, int PIECES, int MODE>  // MODE 0: no stores, 1: plain, 2: nt
__global__ __launch_bounds__(256, 2) void probe(const v4 *in, u4 *out, int chunks, int spin) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (long)s * chunks * ROWS * 256;
  u4 *dst = out + (long)s * chunks * PIECES * 256;
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
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;  // the limiter phases: dependent VALU work, no memory
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
    if (MODE) {
#pragma unroll
      for (int k = 0; k < PIECES; ++k) {
        u4 *to = dst + ((long)c * PIECES * 4 + wave * PIECES + k) * 64 + lane;  // wave's PIECES KiB, 1 KiB per instruction
        if (MODE == 2) __builtin_nontemporal_store(w, to);
        else *to = w;
      }
    } else if (a == 123.456f) {
      dst[t] = w;
    }
  }
}

// MODE 3: the same traffic, but waves 0..3 only LOAD and a fifth wave issues every store of the workgroup:
// vmcnt retires in order per wave, so loads issued behind a store cannot be counted complete before the store
// is acknowledged; with the stores in another wave's queue the loading waves never wait for a store.
template <int ROWS, int PIECES>
__global__ __launch_bounds__(320, 2) void probe_split(const v4 *in, u4 *out, int chunks, int spin) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (long)s * chunks * ROWS * 256;
  u4 *dst = out + (long)s * chunks * PIECES * 256;
  __shared__ float sums[256];
  if (wave < 4) {
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
      for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
      sums[t] = a;
      __syncthreads();   // hand-over to the store wave
      __syncthreads();   // it has read the sums
    }
  } else {
    for (int c = 0; c < chunks; ++c) {
      __syncthreads();
      u4 w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = u4{__float_as_uint(sums[64 * q + lane]), (unsigned)c, (unsigned)lane, 0u};
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < PIECES; ++k) dst[((long)c * PIECES * 4 + q * PIECES + k) * 64 + lane] = w[q];
    }
  }
}

template <int ROWS, int PIECES>
void run(const char *name, const v4 *in, u4 *out, int S, int chunks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double rd = (double)S * chunks * ROWS * 4096, wr = (double)S * chunks * PIECES * 4096;
  for (int spin : {0, 400}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) probe<ROWS, PIECES, 0><<<S, 256>>>(in, out, chunks, spin);
        if (mode == 1) probe<ROWS, PIECES, 1><<<S, 256>>>(in, out, chunks, spin);
        if (mode == 2) probe<ROWS, PIECES, 2><<<S, 256>>>(in, out, chunks, spin);
        if (mode == 3) probe_split<ROWS, PIECES><<<S, 320>>>(in, out, chunks, spin);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double bytes = rd + (mode ? wr : 0.0);
      printf("%-9s spin %4d  %-6s %.3f ms  %.0f GB/s total  (%.1f Gchunk-samples/s)\n", name, spin,
             mode == 0 ? "none" : (mode == 1 ? "plain" : (mode == 2 ? "nt" : "split")), best, bytes / best / 1e6,
             (double)S * chunks * 1024 / best / 1e6);
    }
  }
}

int main() {
  const int S = 512, chunks = 64;
  v4 *in;
  u4 *out;
  hipMalloc(&in, (size_t)S * chunks * 16 * 4096);
  hipMalloc(&out, (size_t)S * chunks * 12 * 4096);
  hipMemset(in, 0, (size_t)S * chunks * 16 * 4096);
  run<16, 1>("headline", in, out, S, chunks);
  run<12, 6>("cfg2", in, out, S, chunks);
  run<16, 12>("cfg3", in, out, S, chunks);
  return 0;
}
