// Probe: ds_read_b128 from LDS addresses that are not 16-byte aligned (gfx950, unaligned access
// mode).  (1) data check at 2-byte granularity; (2) throughput: 8 independent reads per iteration,
// lane l at byte l*16 + mis, for mis = 0, 2, 4, 8.   Result on MI355X: see NOTEBOOK.md 4.2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void check(unsigned short *out, int stride) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr = (unsigned)(size_t)lds + 2u * (unsigned)((lane * stride) % 4000);
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  unsigned short *o = out + lane * 8;
  o[0] = v.x & 0xffff; o[1] = v.x >> 16; o[2] = v.y & 0xffff; o[3] = v.y >> 16;
  o[4] = v.z & 0xffff; o[5] = v.z >> 16; o[6] = v.w & 0xffff; o[7] = v.w >> 16;
}
__global__ void rate(long long *cyc, unsigned *sink, int mis, int iters) {
  __shared__ unsigned lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + mis + (threadIdx.x >> 6) * 8192;
  uint4 a = make_uint4(0, 0, 0, 0);
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) {
    uint4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
        "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
        "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
        : "v"(addr)
        : "memory");
    a.x ^= v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.y ^ v6.z ^ v7.w;
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (a.x == 0x12345678) sink[0] = a.x;
}
int main() {
  unsigned short *d; long long *c; unsigned *sk;
  hipMalloc(&d, 64 * 8 * 2); hipMalloc(&c, 8); hipMalloc(&sk, 4);
  for (int stride : {8, 9, 17}) {
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d, stride);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT stride %d\n", stride); return 1; }
    unsigned short h[512];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) if (h[l * 8 + j] != (unsigned short)((l * stride) % 4000 + j)) ++bad;
    printf("data check, lane stride %d halves: %d mismatches\n", stride, bad);
  }
  for (int waves : {1, 4})
    for (int mis : {0, 2, 4, 8}) {
      hipLaunchKernelGGL(rate, dim3(1), dim3(64 * waves), 0, 0, c, sk, mis, 2000);
      if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT mis %d\n", mis); return 1; }
      long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
      printf("%d wave(s), misalignment %d bytes: %.1f cycles per ds_read_b128 per wave\n", waves, mis, (double)cy / 2000 / 8);
    }
  return 0;
}
