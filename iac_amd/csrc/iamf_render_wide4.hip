// iamf_render_wide4.hip — instantiations of render_wide4_kernel<M, C> (render_wide4.hpp), in a
// translation unit of their own so that the build compiles them next to iamf_render.hip.
// Compiled with -ffp-contract=off like every kernel of the library: the projection is bit-exact.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <atomic>
#include <string.h>

#include <type_traits>

#include "../../include/iamf_hip.h"

namespace {

#include "render_common.hpp"
#include "render_downmix.hpp"
#include "render_fir.hpp"
#include "render_fir16.hpp"
#include "render_fir_fft.hpp"
#include "render_fast.hpp"
#include "render_wide4.hpp"

template <int M, int C>
void launch_mc(const RenderParams &p, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)wide4_lds_floats(C, M, 0);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, false, false>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, true, false>), 80 * 1024);
    opted.end();
  }
  if (p.use_mfma)
    hipLaunchKernelGGL((render_wide4_kernel<M, C, true, false>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((render_wide4_kernel<M, C, false, false>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
}

// scalable channel audio: M decoded channels -> demixer -> the M channels of the target layout -> C
template <int M, int C>
void launch_mc_demixer(const RenderParams &p, hipStream_t st) {
  static_assert(wide4_lds_floats(C, M, kW4DmxFloats) <= 20480, "two workgroups per CU");
  const size_t lds = sizeof(float) * (size_t)wide4_lds_floats(C, M, kW4DmxFloats);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, false, true>), 80 * 1024);
    opted.end();
  }
  hipLaunchKernelGGL((render_wide4_kernel<M, C, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
}

// parametric down-mixer: M channels of the element's layout -> the C channels of a smaller IAMF layout
template <int M, int C>
void launch_mc_downmixer(const RenderParams &p, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)wide4_lds_floats(C, M, 0);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, false, false, true>), 80 * 1024);
    opted.end();
  }
  hipLaunchKernelGGL((render_wide4_kernel<M, C, false, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
}

template <int M>
bool launch_m(const RenderParams &p, hipStream_t st) {
  if (p.dmx_on) {
    if constexpr (M == 12 || M == 10 || M == 8) {
      if (p.out_ch == 10 && M == 12) { launch_mc_downmixer<M, (M > 10 ? 10 : 6)>(p, st); return true; }
      if (p.out_ch == 8 && M >= 10) { launch_mc_downmixer<M, (M > 8 ? 8 : 6)>(p, st); return true; }
      if (p.out_ch == 6) { launch_mc_downmixer<M, 6>(p, st); return true; }
    }
    return false;
  }
  if (p.demix_on) {
    if constexpr (M == 6 || M == 8 || M == 10 || M == 12) {
      switch (p.out_ch) {
        case 6: launch_mc_demixer<M, 6>(p, st); return true;
        case 8: launch_mc_demixer<M, 8>(p, st); return true;
        case 10: launch_mc_demixer<M, 10>(p, st); return true;
        case 12: launch_mc_demixer<M, 12>(p, st); return true;
        case 24: launch_mc_demixer<M, 24>(p, st); return true;
        default: return false;
      }
    }
    return false;
  }
  switch (p.out_ch) {
    case 6: launch_mc<M, 6>(p, st); return true;
    case 8: launch_mc<M, 8>(p, st); return true;
    case 10: launch_mc<M, 10>(p, st); return true;
    case 12: launch_mc<M, 12>(p, st); return true;
    case 14: launch_mc<M, 14>(p, st); return true;   // Sound System G (4+9+0)
    case 24: launch_mc<M, 24>(p, st); return true;
    default: return false;
  }
}

}  // namespace

// 1 if a render_wide4_kernel instance exists for m inputs and c output channels
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_has(int m, int c) {
  return (m == 4 || m == 6 || m == 8 || m == 9 || m == 10 || m == 12 || m == 16) &&
         (c == 6 || c == 8 || c == 10 || c == 12 || c == 14 || c == 24);
}

// 1 if the down-mixer variant exists: 7.1.4 -> {10, 8, 6}, 5.1.4 / 7.1.2 -> {8, 6}, 5.1.2 / 7.1 -> 6 channels
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_has_downmixer(int m, int c) {
  return (m == 12 && (c == 10 || c == 8 || c == 6)) || (m == 10 && (c == 8 || c == 6)) || (m == 8 && c == 6);
}

// 1 if the demixer variant exists: m = channels of the scalable element's target layout (5.1 .. 7.1.4)
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_has_demixer(int m, int c) {
  return (m == 6 || m == 8 || m == 10 || m == 12) && (c == 6 || c == 8 || c == 10 || c == 12 || c == 24);
}

// params: the caller's RenderParams (same definition, render_common.hpp); returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_launch(const void *params, int m, hipStream_t st) {
  RenderParams p;
  memcpy(&p, params, sizeof(p));
  switch (m) {
    case 4: return launch_m<4>(p, st) ? 1 : 0;
    case 6: return launch_m<6>(p, st) ? 1 : 0;
    case 8: return launch_m<8>(p, st) ? 1 : 0;
    case 9: return launch_m<9>(p, st) ? 1 : 0;
    case 10: return launch_m<10>(p, st) ? 1 : 0;
    case 12: return launch_m<12>(p, st) ? 1 : 0;
    case 16: return launch_m<16>(p, st) ? 1 : 0;
    default: return 0;
  }
}
