// fir16_loop_probe.hip — the K loop of render_fir16.hpp in isolation: 12 ds_read_b128 + 24 v_mfma_f32_16x16x32_f16 per
// step, two operand sets, one wave per SIMD and workgroup, WGS workgroups per CU.  Reports, per variant, shader
// cycles per MFMA per SIMD (s_memtime), the in-kernel clock (s_memtime / s_memrealtime) and wall time: what the
// loop can reach with nothing else in the kernel, and at which clock.
//   hipcc -O3 --offload-arch=gfx950 tools/debug/fir16_loop_probe.hip -o gpurun_out/fir16_loop_probe && gpurun_out/fir16_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTaps = 304, kSlice = 4096 + 256 + 32;
constexpr int kXBytes = 2 * kSlice * 2, kHBytes = 2 * 2 * 8 * kTaps * 2;

struct Ops {
  f16x8 a_hi[2], a_lo[2], b_hi[4], b_lo[4];
};

template <int MODE>  // 0 = reads + MFMAs (product order), 1 = MFMAs only (operands loaded once), 2 = reads only
__global__ __launch_bounds__(256, 2) void loop_kernel(const _Float16 *init, int iters, int ks, unsigned long long *stamps, float *sink) {
  extern __shared__ unsigned char lds[];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, col = lane & 15, g = lane >> 4;
  _Float16 *all = reinterpret_cast<_Float16 *>(lds);
  for (int i = t; i < (kXBytes + kHBytes) / 2; i += 256) all[i] = init[i];
  __syncthreads();
  const _Float16 *xh = all, *xl = xh + kSlice;
  const _Float16 *a0 = all + kXBytes / 2 + (col & 7) * kTaps + 8 * g + (col & 8);
  const int q0 = (4096 - 16) - 1024 * w - 16 * col + 8 * g;
  f32x4 acc_hh[2][4], acc_x[2][4];
  for (int e = 0; e < 2; ++e)
    for (int c = 0; c < 4; ++c) acc_hh[e][c] = acc_x[e][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto ld = [&](int s, Ops &o) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      o.a_hi[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 0) * 8 * kTaps + 32 * s);
      o.a_lo[e] = *reinterpret_cast<const f16x8 *>(a0 + (e * 2 + 1) * 8 * kTaps + 32 * s);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o.b_hi[c] = *reinterpret_cast<const f16x8 *>(xh + q0 - 256 * c + 32 * s);
      o.b_lo[c] = *reinterpret_cast<const f16x8 *>(xl + q0 - 256 * c + 32 * s);
    }
  };
  auto mm = [&](const Ops &o) {
    if (MODE == 2) {
      asm volatile("" ::"v"(o.a_hi[0]), "v"(o.a_lo[0]), "v"(o.a_hi[1]), "v"(o.a_lo[1]));
      asm volatile("" ::"v"(o.b_hi[0]), "v"(o.b_lo[0]), "v"(o.b_hi[1]), "v"(o.b_lo[1]), "v"(o.b_hi[2]), "v"(o.b_lo[2]), "v"(o.b_hi[3]), "v"(o.b_lo[3]));
      return;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc_hh[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[e], o.b_hi[c], acc_hh[e][c], 0, 0, 0);
        acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[e], o.b_lo[c], acc_x[e][c], 0, 0, 0);
        acc_x[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_lo[e], o.b_hi[c], acc_x[e][c], 0, 0, 0);
      }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    Ops o0, o1;
    ld(0, o0);
    if (MODE == 1) {
      ld(1, o1);
      for (int s = 0; s + 2 <= ks; s += 2) {
        asm volatile("" : "+v"(o0.a_hi[0]), "+v"(o1.a_hi[0]));
        mm(o0);
        mm(o1);
      }
    } else {
      int s = 0;
      for (; s + 2 <= ks; s += 2) {
        ld(s + 1, o1);
        mm(o0);
        ld(s + 2 < ks ? s + 2 : ks - 1, o0);
        mm(o1);
      }
      if (s < ks) mm(o0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float v = 0.f;
  for (int e = 0; e < 2; ++e)
    for (int c = 0; c < 4; ++c) v += acc_hh[e][c][0] + acc_x[e][c][1];
  if (v == 123.456f) sink[0] = v;
  if (lane == 0) {
    stamps[2 * (blockIdx.x * 4 + w)] = t1 - t0;
    stamps[2 * (blockIdx.x * 4 + w) + 1] = r1 - r0;
  }
}

template <int MODE>
void run(const char *name, int wgs_per_cu, const _Float16 *d_init, unsigned long long *d_st, float *d_sink) {
  const int iters = 2000, ks = 9, nwg = 256 * wgs_per_cu;
  const size_t lds = wgs_per_cu == 2 ? 70 * 1024 : 120 * 1024;  // LDS size sets the workgroups per CU
  hipFuncSetAttribute(reinterpret_cast<const void *>(&loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(nwg), dim3(256), lds, 0, d_init, iters, ks, d_st, d_sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st(2 * 4 * nwg);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int i = 0; i < 4 * nwg; ++i) {
    cyc.push_back((double)st[2 * i]);
    clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double n_mfma = (double)iters * ks * 24;
  const double med = cyc[cyc.size() / 2];
  printf("%-28s wgs/cu %d  wall %.3f ms  wave cycles/MFMA %.2f  -> SIMD cycles/MFMA %.2f  clock %.0f MHz  wall/MFMA/SIMD %.2f ns\n", name,
         wgs_per_cu, ms, med / n_mfma, med / n_mfma / wgs_per_cu, clk[clk.size() / 2], ms * 1e6 / (n_mfma * wgs_per_cu));
}

int main() {
  const size_t n = (kXBytes + kHBytes) / 2;
  std::vector<_Float16> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  _Float16 *d_init;
  unsigned long long *d_st;
  float *d_sink;
  hipMalloc(&d_init, n * 2);
  hipMalloc(&d_st, 2 * 4 * 512 * 8);
  hipMalloc(&d_sink, 4);
  hipMemcpy(d_init, h.data(), n * 2, hipMemcpyHostToDevice);
  for (int wgs = 1; wgs <= 2; ++wgs) {
    run<0>("reads + MFMAs (product)", wgs, d_init, d_st, d_sink);
    run<1>("MFMAs only", wgs, d_init, d_st, d_sink);
    run<2>("reads only", wgs, d_init, d_st, d_sink);
  }
  return 0;
}
