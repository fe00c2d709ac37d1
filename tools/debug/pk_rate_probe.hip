// Probe (not product): issue rate of independent v_pk_fma_f32 / v_pk_add_f32 / v_fma_f32 / v_cndmask_b32_dpp, one and two
// waves per SIMD: cycles per wave-instruction from s_memtime around an unrolled loop of 32 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  f2 a[32];
  float s[32];
  for (int i = 0; i < 32; ++i) { a[i] = f2{(float)threadIdx.x * 1e-3f + i, 1.f}; s[i] = threadIdx.x * 1e-3f + i; }
  const f2 b = f2{1.0001f, 0.9999f}, c = f2{1e-6f, -1e-6f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 4) asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(s[(i + 7) & 31]) : "vcc");
      if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
      if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
      if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 32; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float *o; unsigned long long *c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 4096 * 8);
  const char *names[8] = {"v_pk_fma_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_cndmask_dpp", "v_add_f32", "v_pk_add opsel", "v_pk_fma opsel"};
  for (int threads : {64, 512, 1024}) {      // one WG on one CU: 1 wave (one SIMD), 4 waves (1 per SIMD), 8 waves (2 per SIMD)
    for (int kind = 0; kind < 8; ++kind) {
      const int iters = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        if (kind == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(threads), 0, 0, o, c, iters);
        hipDeviceSynchronize();
      }
      unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
      printf("%3d threads  %-14s: %.2f clock-counter ticks per wave-instruction (x waves per SIMD = SIMD ticks per instr)\n", threads, names[kind], (double)h / (iters * 32.0));
    }
  }
  return 0;
}
