// fir16_stage_probe.hip — fir_stage16<16> of render_fir16.hpp (the product header itself) run as the HRTF kernel
// runs it (256 threads, 66 KB of LDS so that two workgroups share a CU, 16 passes of 4096 samples, 256 taps),
// with s_memtime stamps around its phases (hooks IAMF_F16_STAMP in the header).  Prints, per phase, the median over
// waves of the cycles spent there, the in-kernel clock and the wall time.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/debug/fir16_stage_probe.hip -o /tmp/p && /tmp/p [streams]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../include/iamf_hip.h"

constexpr int kNStamp = 8;
__device__ unsigned long long *g_stamp;  // [workgroup][wave][kNStamp]
#define IAMF_F16_STAMP_DECL                      \
  unsigned long long f16_acc[kNStamp] = {};      \
  unsigned long long f16_last = __builtin_amdgcn_s_memtime();
#define IAMF_F16_STAMP(k)                                        \
  {                                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    f16_acc[k] += now - f16_last;                                \
    f16_last = now;                                              \
  }
#define IAMF_F16_STAMP_END                                                                                      \
  if ((threadIdx.x & 63) == 0)                                                                                  \
    for (int k = 0; k < kNStamp; ++k) g_stamp[(blockIdx.x * 4 + (threadIdx.x >> 6)) * kNStamp + k] += f16_acc[k];

namespace {
#include "../iac_amd/csrc/render_common.hpp"
#include "../iac_amd/csrc/render_fir.hpp"
#include "../iac_amd/csrc/render_fir16.hpp"

__global__ __launch_bounds__(256, 2) void stage_kernel(RenderParams p, unsigned long long *clk, float *sink) {
  extern __shared__ float lds[];
  float *fir = lds;
  const int s = blockIdx.x;
  const float *in_s = p.in + (int64_t)s * p.in_stream_stride;
  const float *hist = p.fir_hist + (int64_t)s * 16 * kFirHist;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int c0 = 0; c0 < p.total; c0 += kF16Span) {
    fir_stage16<16>(p, in_s, hist, c0, fir, fir);
    acc += fir[threadIdx.x];
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 123.456f) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) {
    clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
    clk[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
  }
}
}  // namespace

#define CHK(x)                                                        \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main(int argc, char **argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 512, frames = 64, fs = 1024, M = 16;
  const bool zeros = argc > 2 && atoi(argv[2]) == 0;  // second argument 0: all-zero input and tables (the power test)
  const size_t n_in = (size_t)ns * frames * M * fs;
  std::vector<float> h_in(1 << 22);
  srand(3);
  for (auto &v : h_in) v = zeros ? 0.f : (rand() % 2001 - 1000) / 4000.0f;
  float *d_in, *d_hist, *d_sink;
  CHK(hipMalloc(&d_in, n_in * 4));
  for (size_t o = 0; o < n_in; o += h_in.size()) CHK(hipMemcpy(d_in + o, h_in.data(), std::min(h_in.size(), n_in - o) * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&d_hist, (size_t)ns * M * kFirHist * 4));
  CHK(hipMemset(d_hist, 0, (size_t)ns * M * kFirHist * 4));
  CHK(hipMalloc(&d_sink, 4));
  std::vector<_Float16> h_tab((size_t)M * kF16HBytes / 2);
  for (auto &v : h_tab) v = zeros ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  void *d_tab;
  CHK(hipMalloc(&d_tab, h_tab.size() * 2));
  CHK(hipMemcpy(d_tab, h_tab.data(), h_tab.size() * 2, hipMemcpyHostToDevice));
  unsigned long long *d_stamp, *d_clk;
  CHK(hipMalloc(&d_stamp, (size_t)ns * 4 * kNStamp * 8));
  CHK(hipMalloc(&d_clk, (size_t)ns * 4 * 2 * 8));
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &d_stamp, sizeof(d_stamp)));
  RenderParams p = {};
  p.in = d_in;
  p.in_stream_stride = (int64_t)frames * M * fs;
  p.in_frame_stride = (int64_t)M * fs;
  p.total = frames * fs;
  p.frame_size = fs;
  p.n_streams = ns;
  p.fir_taps = 256;
  p.fir_hist = d_hist;
  p.fir_h16 = d_tab;
  p.fir_inv_scale = 1.f;
  const size_t lds = 66176;  // what render_fast_kernel<16, 2, 2> asks for
  CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipMemset(d_stamp, 0, (size_t)ns * 4 * kNStamp * 8));
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(stage_kernel, dim3(ns), dim3(256), lds, 0, p, d_clk, d_sink);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> st((size_t)ns * 4 * kNStamp), ck((size_t)ns * 4 * 2);
  CHK(hipMemcpy(st.data(), d_stamp, st.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(ck.data(), d_clk, ck.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clock, total;
  for (int i = 0; i < ns * 4; ++i) {
    clock.push_back((double)ck[2 * i] / (double)ck[2 * i + 1] * 100.0);
    total.push_back((double)ck[2 * i]);
  }
  std::sort(clock.begin(), clock.end());
  std::sort(total.begin(), total.end());
  printf("%s data  streams %d  wall %.3f ms  in-kernel clock %.0f MHz (median)  wave lifetime %.0f cycles (median)\n", zeros ? "ZERO" : "random", ns, ms, clock[clock.size() / 2],
         total[total.size() / 2]);
  const char *names[kNStamp] = {"prologue (first channel)", "issue next channel's loads", "K loop", "barrier A", "LDS stores (+ wait for loads)",
                                "barrier B", "sums -> LDS", "-"};
  const double n_iter = 16.0 * 16.0;  // passes x channels
  for (int k = 0; k < 7; ++k) {
    std::vector<double> v;
    for (int i = 0; i < ns * 4; ++i) v.push_back((double)st[(size_t)i * kNStamp + k]);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    printf("  %-32s %10.0f cycles per wave  (%5.1f %%)  %8.1f per channel iteration\n", names[k], med, 100.0 * med / total[total.size() / 2], med / n_iter);
  }
  return 0;
}
