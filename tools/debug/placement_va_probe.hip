// placement_va_probe.hip — the two placement modes of NOTEBOOK.md 3 outside the bench: a dozen 2.1 GB allocations of the
// input (kept alive), the headline traffic shape timed on each, with the virtual address hipMalloc returned.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/placement_va_probe.hip -o /tmp/pvp && /tmp/pvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

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

int main(int argc, char **argv) {
  const bool contiguous = argc > 1;  // any argument: hipExtMallocWithFlags(hipDeviceMallocContiguous) instead of hipMalloc
  const int S = 512, chunks = 64, N = 48;
  const size_t in_bytes = (size_t)S * (chunks * 16 * 4096 + 4096);
  const long stride4 = (long)chunks * 16 * 256 + 256;
  u4 *out, *out2;
  (void)hipMalloc(&out, (size_t)S * chunks * 12 * 4096);
  (void)hipMemset(out, 0, (size_t)S * chunks * 12 * 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  v4 *in[N];
  for (int i = 0; i < N; ++i) {
    const hipError_t e = contiguous ? hipExtMallocWithFlags(reinterpret_cast<void **>(&in[i]), in_bytes, hipDeviceMallocContiguous)
                                    : hipMalloc(&in[i], in_bytes);
    if (e != hipSuccess) { printf("alloc %d failed: %s\n", i, hipGetErrorString(e)); return 1; }
    (void)hipMemset(in[i], 0, in_bytes);
  }
  (void)hipMalloc(&out2, (size_t)S * chunks * 4096);   // a second output buffer, allocated AFTER the inputs
  (void)hipMemset(out2, 0, (size_t)S * chunks * 4096);
  printf("out %p  out2 %p\n", (void *)out, (void *)out2);
  if (argc > 1 && argv[1][0] == 'w') {
    // "w": find one slow and one fast input buffer, then the same shape with fewer / more workgroups in flight and with
    // the streams laid out frame-major (all workgroups inside one 32 MB window at any time)
    float rate[N];
    auto time_it = [&](auto launch) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      return best;
    };
    int slow = -1, fast = -1;
    for (int i = 0; i < N; ++i) {
      const float ms = time_it([&] { k_stream<16, 1><<<S, 256>>>(in[i], out, chunks, stride4); });
      rate[i] = (float)((double)S * chunks * 17 * 4096 / ms / 1e6);
      if (rate[i] < 5700 && slow < 0 && i > 0) slow = i;
      if (rate[i] > 6100 && fast < 0) fast = i;
    }
    printf("slow buffer %d (%.0f GB/s), fast buffer %d (%.0f GB/s)\n", slow, slow >= 0 ? rate[slow] : 0.f, fast, fast >= 0 ? rate[fast] : 0.f);
    if (slow < 0 || fast < 0) return 0;
    for (int nwg : {64, 128, 256, 384, 512}) {
      const float a = time_it([&] { k_stream<16, 1><<<nwg, 256>>>(in[slow], out, chunks, stride4); });
      const float b = time_it([&] { k_stream<16, 1><<<nwg, 256>>>(in[fast], out, chunks, stride4); });
      printf("headline shape, %3d workgroups: slow %5.0f GB/s  fast %5.0f GB/s\n", nwg, (double)nwg * chunks * 17 * 4096 / a / 1e6,
             (double)nwg * chunks * 17 * 4096 / b / 1e6);
    }
    {
      const float a = time_it([&] { k_stream<16, 0><<<S, 256>>>(in[slow], out, chunks, stride4); });
      const float b = time_it([&] { k_stream<16, 0><<<S, 256>>>(in[fast], out, chunks, stride4); });
      printf("read only, 512 workgroups:      slow %5.0f GB/s  fast %5.0f GB/s\n", (double)S * chunks * 16 * 4096 / a / 1e6, (double)S * chunks * 16 * 4096 / b / 1e6);
    }
    {
      const float a = time_it([&] { k_stream<16, 12><<<S, 256>>>(in[slow], out, chunks, stride4); });
      const float b = time_it([&] { k_stream<16, 12><<<S, 256>>>(in[fast], out, chunks, stride4); });
      printf("cfg3 shape, 512 workgroups:     slow %5.0f GB/s  fast %5.0f GB/s\n", (double)S * chunks * 28 * 4096 / a / 1e6, (double)S * chunks * 28 * 4096 / b / 1e6);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'o') {
    // "out": the cfg3 shape (64 B read + 48 B written per sample-frame) with ONE input buffer (the first fast-looking one
    // is not known here: input 20, past the usual slow runs) against 40 candidate OUTPUT buffers of 1.6 GB
    const int NO = 40;
    u4 *outs[NO];
    for (int j = 0; j < NO; ++j) {
      if (hipMalloc(&outs[j], (size_t)S * chunks * 12 * 4096) != hipSuccess) { printf("out alloc %d failed\n", j); return 1; }
      (void)hipMemset(outs[j], 0, (size_t)S * chunks * 12 * 4096);
    }
    for (int which : {20, 2}) {
      for (int j = 0; j < NO; ++j) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          (void)hipEventRecord(e0);
          k_stream<16, 12><<<S, 256>>>(in[which], outs[j], chunks, stride4);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        printf("cfg3 input %2d output %2d  %.3f ms  %5.0f GB/s\n", which, j, best, (double)S * chunks * 28 * 4096 / best / 1e6);
      }
    }
    // the headline shape (4 B written per 64 B read) over the same output candidates, for every fourth input
    for (int which = 0; which < N; which += 4) {
      printf("headline input %2d outputs:", which);
      for (int j = 0; j < NO; j += 2) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          (void)hipEventRecord(e0);
          k_stream<16, 1><<<S, 256>>>(in[which], outs[j], chunks, stride4);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        printf(" %2.0f", (double)S * chunks * 17 * 4096 / best / 1e8);
      }
      printf("\n");
    }
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < N; ++i) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k_stream<16, 1><<<S, 256>>>(in[i], pass ? out2 : out, chunks, stride4);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("pass %d buffer %2d  va %p  (va >> 30) %% 64 = %2llu  %.3f ms  %5.0f GB/s\n", pass, i, (void *)in[i],
             (unsigned long long)(((uintptr_t)in[i]) >> 30) % 64, best, (double)S * chunks * 17 * 4096 / best / 1e6);
    }
  return 0;
}
