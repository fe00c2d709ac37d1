// iamf_render_lpcm.hip — the headline kernel fed with LPCM packets: render_fast_kernel<M, OC, 0, false, false, LP = true>
// (render_fast.hpp) for mono-coded ambisonics elements (M = 1, 4, 9, 16 sub-streams of one channel each) into one- and
// two-channel layouts, in a translation unit of its own (compiled beside iamf_render.hip).
//
// What it replaces: the reference decodes an LPCM sub-stream packet into its planar f32 decoder buffer
// (src/iamf_dec/pcm/IAMF_pcm_decoder.c:64-83, 133-149: sample / 2^15; channel order by IAMF_decoder.c:2230-2260) and the
// renderer reads that buffer (IAMF_decoder.c:2550-2640).  On the device the f32 copy is 64 of the path's 68 bytes per
// sample-frame; here the render kernel reads the 16-bit samples themselves (32 + 4 bytes per sample-frame) and converts
// them where it loads them, by the same expression — results are bit-identical to iamf_hip_lpcm_unpack followed by the
// f32 kernel (tests/test_gpu_lpcm.py).  Entry: iamf_hip_batch_render_lpcm (iamf_render.hip), which falls back to exactly
// that pair for every input this kernel does not take.
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
void launch_lp_m(const RenderParams &p, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)fast_lds_floats(p.out_ch, M);
  static OptIn opted;
  if (opted.begin()) {
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 1, 0, false, false, true, true>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 0, false, false, true, true>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 1, 0, false, false, true, false>), 80 * 1024);
    opted.set(reinterpret_cast<const void *>(&render_fast_kernel<M, 2, 0, false, false, true, false>), 80 * 1024);
    opted.end();
  }
  const dim3 grid((unsigned)p.n_launch);
  // up to four workgroups a CU: the early per-channel prefetch (latency bound); beyond: the plain one (issue bound).
  // Measured on MI355X, 512 / 4096 streams: 117 against 109 / 136 against 145 Gsamples/s (profiles/r04_ab_fast.txt)
  const bool early = p.n_launch <= 1024 && !getenv("IAMF_HIP_LP_LATE");
  if (p.out_ch == 1) {
    if (early) hipLaunchKernelGGL((render_fast_kernel<M, 1, 0, false, false, true, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((render_fast_kernel<M, 1, 0, false, false, true, false>), grid, dim3(256), lds, st, p);
  } else {
    if (early) hipLaunchKernelGGL((render_fast_kernel<M, 2, 0, false, false, true, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((render_fast_kernel<M, 2, 0, false, false, true, false>), grid, dim3(256), lds, st, p);
  }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int iamf_hip_fast_lpcm_has(int m, int oc) {
  return (m == 1 || m == 4 || m == 9 || m == 16) && (oc == 1 || oc == 2);
}

// returns 1 if launched
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_fast_lpcm_launch(const void *params, int m, hipStream_t st) {
  RenderParams p;
  memcpy(&p, params, sizeof(p));
  if (!p.lpcm || !iamf_hip_fast_lpcm_has(m, p.out_ch)) return 0;
  switch (m) {
    case 1: launch_lp_m<1>(p, st); return 1;
    case 4: launch_lp_m<4>(p, st); return 1;
    case 9: launch_lp_m<9>(p, st); return 1;
    case 16: launch_lp_m<16>(p, st); return 1;
    default: return 0;
  }
}
