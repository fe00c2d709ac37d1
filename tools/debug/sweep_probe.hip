// sweep_probe.hip — cycles per step of the limiter's trigger-run recurrence
//   G <- shr(G) - a1 * (shr(G) - ep)
// for different ways of moving G one lane up (tools only; not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/sweep_probe tools/debug/sweep_probe.hip && /tmp/sweep_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int V>
__global__ void probe(float *out, long long *cyc, float a1) {
  float G = 0.9f + threadIdx.x * 1e-4f, ep = 0.7f, t = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 16; ++it) {
    if (V == 0) {  // DPP folded into both subtractions, wave_shr:1
      asm volatile(REP64("s_nop 1\n v_sub_f32_dpp %1, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_e32 %1, %3, %1\n v_sub_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(G), "=&v"(t) : "v"(ep), "v"(a1));
    } else if (V == 1) {  // no lane movement at all: the floor of three dependent VALU operations
      asm volatile(REP64("v_sub_f32_e32 %1, %0, %2\n v_mul_f32_e32 %1, %3, %1\n v_sub_f32_e32 %0, %0, %1\n")
                   : "+v"(G), "=&v"(t) : "v"(ep), "v"(a1));
    } else if (V == 2) {  // row_shr:1 (row-local) folded
      asm volatile(REP64("s_nop 1\n v_sub_f32_dpp %1, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_e32 %1, %3, %1\n v_sub_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(G), "=&v"(t) : "v"(ep), "v"(a1));
    } else if (V == 3) {  // one mov_dpp, then plain VALU
      float t2;
      asm volatile(REP64("s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_sub_f32_e32 %2, %1, %3\n v_mul_f32_e32 %2, %4, %2\n v_sub_f32_e32 %0, %1, %2\n")
                   : "+v"(G), "+v"(t), "=&v"(t2) : "v"(ep), "v"(a1));
    } else if (V == 5) {  // FOUR samples per lane: one lane crossing per 12 dependent operations
      float g0 = G, g1 = G, g2 = G, e0 = 0.71f, e1 = 0.72f, e2 = 0.73f;
      asm volatile(REP64("s_nop 1\n v_sub_f32_dpp %4, %3, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_e32 %4, %9, %4\n v_sub_f32_dpp %0, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_sub_f32_e32 %4, %0, %6\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %1, %0, %4\n"
                         "v_sub_f32_e32 %4, %1, %7\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %2, %1, %4\n"
                         "v_sub_f32_e32 %4, %2, %8\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %3, %2, %4\n")
                   : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(G), "=&v"(t) : "v"(ep), "v"(e0), "v"(e1), "v"(e2), "v"(a1));
      G += g0 + g1 + g2;
    } else if (V == 6) {  // as 5 without the s_nop (timing only: is the hazard hardware-interlocked?)
      float g0 = G, g1 = G, g2 = G, e0 = 0.71f, e1 = 0.72f, e2 = 0.73f;
      asm volatile(REP64("v_sub_f32_dpp %4, %3, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_e32 %4, %9, %4\n v_sub_f32_dpp %0, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_sub_f32_e32 %4, %0, %6\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %1, %0, %4\n"
                         "v_sub_f32_e32 %4, %1, %7\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %2, %1, %4\n"
                         "v_sub_f32_e32 %4, %2, %8\n v_mul_f32_e32 %4, %9, %4\n v_sub_f32_e32 %3, %2, %4\n")
                   : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(G), "=&v"(t) : "v"(ep), "v"(e0), "v"(e1), "v"(e2), "v"(a1));
      G += g0 + g1 + g2;
    } else if (V == 7) {  // uniform chain: e from an SGPR, a v_cndmask capture per step (4 VALU, no lane crossing)
      float cap = 0.f;
      asm volatile(REP64("v_subrev_f32_e32 %1, %3, %0\n v_mul_f32_e32 %1, %4, %1\n v_sub_f32_e32 %0, %0, %1\n"
                         "v_cndmask_b32_e32 %2, %2, %0, vcc\n")
                   : "+v"(G), "=&v"(t), "+v"(cap) : "s"(ep), "v"(a1) : "vcc");
      G += cap;
    } else if (V == 4) {  // wave_shr:1 with only ONE nop wait state (is s_nop 0 enough?) - timing only
      asm volatile(REP64("s_nop 0\n v_sub_f32_dpp %1, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_e32 %1, %3, %1\n v_sub_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(G), "=&v"(t) : "v"(ep), "v"(a1));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = G;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float *d_out;
  long long *d_c, c;
  hipMalloc(&d_out, 64 * 4);
  hipMalloc(&d_c, 8);
  const char *names[] = {"wave_shr:1 folded (library)", "no shift (3-op floor)", "row_shr:1 folded",
                         "mov_dpp + 3 VALU", "wave_shr:1 folded, s_nop 0", "4 samples per lane (per 4 samples)",
                         "4 samples per lane, no s_nop (per 4)", "uniform chain + cndmask capture"};
#define RUN(V)                                                      \
  for (int k = 0; k < 2; ++k) {                                     \
    hipLaunchKernelGGL(probe<V>, dim3(1), dim3(64), 0, 0, d_out, d_c, 0.04f); \
    hipDeviceSynchronize();                                         \
  }                                                                 \
  hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);                     \
  printf("%-32s %.2f counter ticks per step\n", names[V], c / 1024.0);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  return 0;
}
