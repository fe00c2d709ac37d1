// Probe (not product): does hipExtStreamCreateWithCUMask confine a stream's kernels on this stack, and to which CUs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // spin a little so that workgroups spread over everything the stream may use
  long long t0 = clock64(); while (clock64() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void run(hipStream_t st, const char *name) {
  const int G = 4096;
  unsigned *d; hipMalloc(&d, G * 8);
  hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, st, d);
  std::vector<unsigned> h(2 * G);
  hipStreamSynchronize(st);
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  std::set<unsigned> cus; std::set<unsigned> xccs;
  for (int i = 0; i < G; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
  }
  printf("%s: %zu distinct CUs on %zu XCCs:", name, cus.size(), xccs.size());
  int k = 0; for (unsigned c : cus) { if (k++ < 20) printf(" %x", c); } printf("\n");
  hipFree(d);
}
int main() {
  hipStream_t s0; hipStreamCreate(&s0); run(s0, "unmasked");
  for (int bits : {16, 32}) {
    unsigned mask[8] = {0};
    for (int i = 0; i < bits; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t sm; hipError_t e = hipExtStreamCreateWithCUMask(&sm, 8, mask);
    printf("first %d bits: create -> %d (%s)\n", bits, (int)e, hipGetErrorString(e));
    if (e == hipSuccess) run(sm, "masked (low bits)");
    unsigned inv[8]; for (int i = 0; i < 8; ++i) inv[i] = ~mask[i];
    e = hipExtStreamCreateWithCUMask(&sm, 8, inv);
    printf("complement: create -> %d\n", (int)e);
    if (e == hipSuccess) run(sm, "masked (complement)");
  }
  return 0;
}
