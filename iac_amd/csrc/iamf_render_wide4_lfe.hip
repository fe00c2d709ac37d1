// iamf_render_wide4_lfe.hip — instantiations of the LFE variant of render_wide4_kernel (render_wide4.hpp,
// LFE = true: ambisonics elements of order 1..3 rendered to a layout with LFE channels while the HOA LFE
// generator is on), in a translation unit of their own so that the build compiles them next to the others.
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
  const size_t lds = sizeof(float) * (size_t)wide4_lds_floats(C, M, 0);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, false, false, false, false, true>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_wide4_kernel<M, C, true, false, false, false, true>), 80 * 1024);
    opted.end();
  }
  if (p.use_mfma)
    hipLaunchKernelGGL((render_wide4_kernel<M, C, true, false, false, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((render_wide4_kernel<M, C, false, false, false, false, true>), dim3((unsigned)p.n_launch), dim3(256), lds, st, p);
}

template <int M>
bool launch_m(const RenderParams &p, hipStream_t st) {
  switch (p.out_ch) {
    case 6: launch_mc<M, 6>(p, st); return true;
    case 8: launch_mc<M, 8>(p, st); return true;
    case 10: launch_mc<M, 10>(p, st); return true;
    case 12: launch_mc<M, 12>(p, st); return true;
    case 14: launch_mc<M, 14>(p, st); return true;
    case 24: launch_mc<M, 24>(p, st); return true;
    default: return false;
  }
}

}  // namespace

// 1 if the LFE variant exists for an ambisonics element of m channels and c output channels
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_has_lfe(int m, int c) {
  return (m == 4 || m == 9 || m == 16) && (c == 6 || c == 8 || c == 10 || c == 12 || c == 14 || c == 24);
}

// params: the caller's RenderParams (same definition, render_common.hpp); returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_wide4_lfe_launch(const void *params, int m, hipStream_t st) {
  RenderParams p;
  memcpy(&p, params, sizeof(p));
  if (!p.lfe || p.dmx_on || p.demix_on) return 0;
  switch (m) {
    case 4: return launch_m<4>(p, st) ? 1 : 0;
    case 9: return launch_m<9>(p, st) ? 1 : 0;
    case 16: return launch_m<16>(p, st) ? 1 : 0;
    default: return 0;
  }
}
