// iamf_render_fir_m2b.hip — the binaural FIR renderer for CHANNEL-BASED elements (the role of
// IAMF_element_renderer_render_M2B, reference m2b_rdr.c:103-121, call site IAMF_decoder.c:2562-2570):
// instantiations of render_fast_kernel<M, 2, FIR> for the channel counts of the IAMF loudspeaker
// layouts (stereo 2, 5.1 / 3.1.2 6, 5.1.2 / 7.1 8, 5.1.4 / 7.1.2 10, 7.1.4 12), in a translation unit of
// their own so that the build compiles them next to the ambisonics ones (iamf_render.hip: 1, 4, 9, 16).
//   y[ear][t] = sum_c sum_k h[ear][c][k] * x[c][t - k]        c = loudspeaker channel, playback order
// PARITY UNPINNED like the scene-based form: the reference hands this to BEAR (bear/iamf_bear_api.h),
// which is not in its tree; the formula above, with one HRIR pair per loudspeaker supplied by the
// caller, is this library's specification (render_fir.hpp).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "../../include/iamf_hip.h"

namespace {

#include "render_common.hpp"
#include "render_downmix.hpp"
#include "render_fir.hpp"
#include "render_fir16.hpp"
#include "render_fir_fft.hpp"
#include "render_fast.hpp"

template <int M>
void launch_fir_m(const RenderParams &p, hipStream_t st) {
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 1>), 120 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 2>), 120 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 3>), 120 * 1024);
    opted.end();
  }
  const dim3 grid((unsigned)p.n_launch);
  const int stage = fir_stage_choice(p);   // as launch_fir_m of iamf_render.hip (4 is handled by the caller)
  if (stage == 3 || stage == 4) {
    static_assert(fast_lds_floats(2, M, 3) * 4 <= 80 * 1024, "two workgroups per CU");
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 3>), grid, dim3(256), sizeof(float) * (size_t)fast_lds_floats(2, M, 3), st, p);
  } else if (stage == 2) {
    static_assert(fast_lds_floats(2, M, 2) * 4 <= 80 * 1024, "two workgroups per CU");
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 2>), grid, dim3(256), sizeof(float) * (size_t)fast_lds_floats(2, M, 2), st, p);
  } else {
    hipLaunchKernelGGL((render_fast_kernel<M, 2, 1>), grid, dim3(512), sizeof(float) * (size_t)fast_lds_floats(2, M, 1), st, p);
  }
}

template <int M>
void launch_fft_m(const RenderParams &p, hipStream_t st) {   // as in iamf_render.hip
  const dim3 g((unsigned)((p.total + kFftSpan - 1) / kFftSpan), (unsigned)p.n_launch);
  // (whole frames only: past a call that ends inside a frame the two-base fetch would read what the caller left in the rest
  //  of the frame — harmless to the samples that are kept unless it is a NaN, which a transform spreads over its block)
  if ((M & 1) == 0 && p.fir_pre && p.fir_pre_next && p.total % p.frame_size == 0 && !getenv("IAMF_HIP_FIR_GENERAL_FETCH"))
    hipLaunchKernelGGL((fir_fft_kernel<M, (M & 1) == 0>), g, dim3(256), sizeof(float) * (size_t)kFftLdsFloats, st, p, p.fir_y, 2 * (int64_t)p.total);
  else
    hipLaunchKernelGGL((fir_fft_kernel<M, false>), g, dim3(256), sizeof(float) * (size_t)kFftLdsFloats, st, p, p.fir_y, 2 * (int64_t)p.total);
}

}  // namespace

// the FFT stage alone (fir_fft_kernel); returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_fir_m2b_launch_fft(const void *params, int m, hipStream_t st) {
  RenderParams p;
  memcpy(&p, params, sizeof(p));
  switch (m) {
    case 2: launch_fft_m<2>(p, st); return 1;
    case 6: launch_fft_m<6>(p, st); return 1;
    case 8: launch_fft_m<8>(p, st); return 1;
    case 10: launch_fft_m<10>(p, st); return 1;
    case 12: launch_fft_m<12>(p, st); return 1;
    default: return 0;
  }
}

extern "C" __attribute__((visibility("hidden"))) int iamf_hip_fir_m2b_has(int m) {
  return m == 2 || m == 6 || m == 8 || m == 10 || m == 12;
}

// params: the caller's RenderParams (same definition, render_common.hpp); returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_fir_m2b_launch(const void *params, int m, hipStream_t st) {
  RenderParams p;
  memcpy(&p, params, sizeof(p));
  switch (m) {
    case 2: launch_fir_m<2>(p, st); return 1;
    case 6: launch_fir_m<6>(p, st); return 1;
    case 8: launch_fir_m<8>(p, st); return 1;
    case 10: launch_fir_m<10>(p, st); return 1;
    case 12: launch_fir_m<12>(p, st); return 1;
    default: return 0;
  }
}
