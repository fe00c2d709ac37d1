// iamf_render_wide4_mix.hip — instantiations of the mixing variant of render_wide4_kernel
// (render_wide4.hpp, MIX = true: second element and / or per-sample gain ramps), in a translation
// unit of their own so that the build compiles them next to the other kernels.
// Compiled with -ffp-contract=off like every kernel of the library.
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
  static_assert(wide4_lds_floats(C, M, kW4MixFloats) <= 20480, "two workgroups per CU");
  const size_t lds = sizeof(float) * (size_t)wide4_lds_floats(C, M, kW4MixFloats);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, false, false, false, true>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, true, false, false, true>), 80 * 1024);
    opted.end();
  }
  if (p.use_mfma)
    hipLaunchKernelGGL((render_wide4_kernel<M, C, true, false, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((render_wide4_kernel<M, C, false, false, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
}

template <int M>
bool launch_m(const RenderParams &p, hipStream_t st) {
  switch (p.out_ch) {
    case 6: launch_mc<M, 6>(p, st); return true;
    case 8: launch_mc<M, 8>(p, st); return true;
    case 10: launch_mc<M, 10>(p, st); return true;
    case 12: launch_mc<M, 12>(p, st); return true;
    default: return false;
  }
}

}  // namespace

// 1 if the mixing variant exists for m inputs of the first element and c output channels
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_has_mix(int m, int c) {
  return (m == 4 || m == 6 || m == 8 || m == 9 || m == 10 || m == 12 || m == 16) && (c == 6 || c == 8 || c == 10 || c == 12);
}

// params: the caller's RenderParams (same definition, render_common.hpp); returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_mix_launch(const void *params, int m, hipStream_t st) {
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
