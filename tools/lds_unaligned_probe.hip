// Probe: does ds_read_b128 accept a 2-byte-aligned LDS address on gfx950 (unaligned access mode),
// and what does it cost?  Each lane reads 8 halves starting at half index (lane * 9 + k) % 4000.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned short *out, long long *cyc, int stride, int iters) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned base = (unsigned)(size_t)lds;  // LDS byte address of the array
  uint4 acc = make_uint4(0, 0, 0, 0);
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) {
    unsigned addr = base + 2u * (unsigned)((lane * stride + k) % 4000);
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    if (k == 0) {
      unsigned short *o = out + lane * 8;
      o[0] = v.x & 0xffff; o[1] = v.x >> 16; o[2] = v.y & 0xffff; o[3] = v.y >> 16;
      o[4] = v.z & 0xffff; o[5] = v.z >> 16; o[6] = v.w & 0xffff; o[7] = v.w >> 16;
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[0] = t1 - t0;
  if (acc.x == 0x12345678) out[0] = 1;
}
int main() {
  unsigned short *d; long long *c;
  hipMalloc(&d, 64 * 8 * 2); hipMalloc(&c, 8);
  for (int stride : {8, 9, 17}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, stride, 1000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT stride %d\n", stride); return 1; }
    unsigned short h[512]; long long cy;
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) if (h[l * 8 + j] != (unsigned short)((l * stride) % 4000 + j)) ++bad;
    printf("stride %d halves: mismatches %d, %.1f cycles per read\n", stride, bad, (double)cy / 1000);
  }
  return 0;
}
