// iamf_probe.hip — diagnostic entry point: the render kernels' HBM traffic SHAPE with no compute, so that a
// measured rate can be put next to what the memory system delivers for that shape (bench.py:
// roofline.same_traffic_no_compute).  One 256-thread workgroup per stream; per 1024-sample chunk every lane
// reads `rows` x 16 B (rows 4 KiB apart inside a frame of rows x 4 KiB, prefetched one chunk ahead, non-temporal)
// and writes `pieces` x 16 B, each store instruction covering 1 KiB contiguous per wave — the geometry of
// render_fast.hpp (rows 16, pieces 1) and render_wide4.hpp (cfg2: 12 / 6, cfg3: 16 / 12).
// Round-2 findings: read-only this pattern runs at 7.1 TB/s; with a share of writes at 6.2-6.5 TB/s (headline shape)
// when input and output lie in memory regions of different kinds and at 5.45 TB/s when they lie in regions of the
// same kind (tools/placement_va_probe.hip) — iamf_hip_pick_buffer_pair below finds a fast pair among
// candidates.  Nothing here is used by the render path.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/iamf_hip.h"

namespace {

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <int ROWS>
__global__ __launch_bounds__(256, 2) void traffic_probe_kernel(const v4 *in, int64_t in_stream_stride4, u4 *out,
                                                               int64_t out_stream_stride4, int chunks, int pieces) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (int64_t)s * in_stream_stride4;
  u4 *dst = out + (int64_t)s * out_stream_stride4;
  v4 x[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * 256 + t);
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    const int cn = c + 1 < chunks ? c + 1 : c;  // unconditional prefetch (the last one re-reads)
#pragma unroll
    for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + ((int64_t)cn * ROWS + m) * 256 + t);
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
    for (int k = 0; k < pieces; ++k)
      __builtin_nontemporal_store(w, dst + ((int64_t)c * pieces * 4 + wave * pieces + k) * 64 + lane);
  }
}

}  // namespace

extern "C" int iamf_hip_probe_traffic(int n_streams, int chunks, int rows, int pieces, const void *d_in,
                                      int64_t in_stream_stride_bytes, void *d_out, int64_t out_stream_stride_bytes,
                                      void *stream) {
  if (n_streams <= 0 || chunks <= 0 || pieces < 0 || pieces > 64 || !d_in || !d_out ||
      in_stream_stride_bytes < (int64_t)chunks * rows * 4096 || out_stream_stride_bytes < (int64_t)chunks * pieces * 4096 ||
      (in_stream_stride_bytes & 15) || (out_stream_stride_bytes & 15))
    return IAMF_HIP_ERR_BAD_ARG;
  const v4 *in = static_cast<const v4 *>(d_in);
  u4 *out = static_cast<u4 *>(d_out);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)n_streams), block(256);
  const int64_t is4 = in_stream_stride_bytes / 16, os4 = out_stream_stride_bytes / 16;
  switch (rows) {
#define CASE_R(R) \
  case R: hipLaunchKernelGGL(traffic_probe_kernel<R>, grid, block, 0, st, in, is4, out, os4, chunks, pieces); break;
    CASE_R(1) CASE_R(2) CASE_R(4) CASE_R(6) CASE_R(8) CASE_R(9) CASE_R(10) CASE_R(12) CASE_R(16)
#undef CASE_R
    default: return IAMF_HIP_ERR_UNIMPLEMENTED;
  }
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}

// Times the traffic kernel on every (input candidate, output candidate) pair — 1 + reps launches each, the median of
// the reps — and reports the fastest pair: what a host with long-lived stream buffers does once at start-up
// (INTEGRATION.md 5).  Synchronous; the candidates' contents are read / overwritten.
extern "C" int iamf_hip_pick_buffer_pair(int n_streams, int chunks, int rows, int pieces, const void *const *d_in_candidates,
                                         int n_in, int64_t in_stream_stride_bytes, void *const *d_out_candidates, int n_out,
                                         int64_t out_stream_stride_bytes, void *stream, int *best_in, int *best_out,
                                         float *ms_out) {
  if (!d_in_candidates || !d_out_candidates || n_in <= 0 || n_out <= 0 || n_in > 256 || n_out > 256 || !best_in || !best_out)
    return IAMF_HIP_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    return IAMF_HIP_ERR_DEVICE;
  }
  constexpr int kReps = 3;
  int rc = IAMF_HIP_OK;
  float best = 0.f;
  *best_in = *best_out = 0;
  for (int i = 0; i < n_in && rc == IAMF_HIP_OK; ++i)
    for (int j = 0; j < n_out && rc == IAMF_HIP_OK; ++j) {
      float t[kReps];
      for (int r = -1; r < kReps && rc == IAMF_HIP_OK; ++r) {  // r = -1: untimed first touch
        if (hipEventRecord(e0, st) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
        if (rc == IAMF_HIP_OK)
          rc = iamf_hip_probe_traffic(n_streams, chunks, rows, pieces, d_in_candidates[i], in_stream_stride_bytes,
                                      d_out_candidates[j], out_stream_stride_bytes, stream);
        if (rc == IAMF_HIP_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = IAMF_HIP_ERR_DEVICE;
        float ms = 0.f;
        if (rc == IAMF_HIP_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
        if (r >= 0) t[r] = ms;
      }
      if (rc != IAMF_HIP_OK) break;
      float a = t[0], b = t[1], c = t[2];   // median of three
      const float med = a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b));
      if (ms_out) ms_out[i * n_out + j] = med;
      if ((i == 0 && j == 0) || med < best) {
        best = med;
        *best_in = i;
        *best_out = j;
      }
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// iamf_hip_pair_alloc: stream buffers ASSEMBLED from device-memory chunks of chosen kinds (DESIGN.md 3,
// tools/vmm_probe.hip).  Physical 2 GiB chunks (hipMemCreate) are mapped one by one and sorted into kinds by the
// pair relation itself: the headline traffic shape reading chunk c and writing into the start of a reference chunk is
// ~14 % slower when both are of one kind.  The input buffer is then mapped (hipMemMap) from chunks of the most
// numerous kind, every output buffer from chunks of the other kinds, each as one contiguous virtual range.
// ------------------------------------------------------------------------------------------------------------------
struct iamf_hip_pair_alloc {
  size_t chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;   // every physical chunk still owned
  struct Range { void *va; size_t bytes; bool mapped; };
  std::vector<Range> ranges;                               // every reserved virtual range
  int device = 0;
};

namespace {
void pair_alloc_free(iamf_hip_pair_alloc *a) {
  if (!a) return;
  (void)hipDeviceSynchronize();
  for (auto &r : a->ranges) {
    if (r.mapped) (void)hipMemUnmap(r.va, r.bytes);
    (void)hipMemAddressFree(r.va, r.bytes);
  }
  for (auto h : a->handles) (void)hipMemRelease(h);
  delete a;
}
}  // namespace

extern "C" void iamf_hip_pair_alloc_destroy(iamf_hip_pair_alloc *a) { pair_alloc_free(a); }

extern "C" int iamf_hip_pair_alloc_create(int64_t in_bytes, int64_t out_bytes, int n_out, void *stream,
                                          iamf_hip_pair_alloc **out_alloc, void **d_in, void **d_out, int *kinds_found) {
  if (!out_alloc || !d_in || !d_out || in_bytes <= 0 || out_bytes <= 0 || n_out <= 0 || n_out > 16) return IAMF_HIP_ERR_BAD_ARG;
  *out_alloc = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  iamf_hip_pair_alloc *a = new iamf_hip_pair_alloc;
#define PA_CHK(x)                     \
  do {                                \
    if ((x) != hipSuccess) {          \
      pair_alloc_free(a);             \
      return IAMF_HIP_ERR_DEVICE;     \
    }                                 \
  } while (0)
  PA_CHK(hipGetDevice(&a->device));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = a->device;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t chunk = a->chunk = (size_t)2 << 30;
  const int n_in_ch = (int)(((size_t)in_bytes + chunk - 1) / chunk);
  const int n_out_ch = (int)(((size_t)out_bytes + chunk - 1) / chunk);
  int want = 2 * (n_in_ch + n_out * n_out_ch) + 8;   // enough to find n_in_ch of one kind and the rest of others
  if (want > 72) want = 72;
  if (want < n_in_ch + n_out * n_out_ch + 2) {
    pair_alloc_free(a);
    return IAMF_HIP_ERR_BAD_ARG;   // larger than this helper assembles (144 GB of chunks)
  }
  // 1. physical chunks, each mapped on its own
  std::vector<void *> va(want, nullptr);
  for (int i = 0; i < want; ++i) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {   // the card is fuller than expected: work with what there is
      want = i;
      va.resize(want);
      break;
    }
    a->handles.push_back(h);
    PA_CHK(hipMemAddressReserve(&va[i], chunk, 0, nullptr, 0));
    a->ranges.push_back({va[i], chunk, false});
    PA_CHK(hipMemMap(va[i], chunk, 0, h, 0));
    a->ranges.back().mapped = true;
    PA_CHK(hipMemSetAccess(va[i], chunk, &acc, 1));
  }
  if (want < n_in_ch + n_out * n_out_ch) {
    pair_alloc_free(a);
    return IAMF_HIP_ERR_DEVICE;
  }
  // 2. kinds.  Probe: 480 streams x 60 chunks of the headline shape (16 rows, 1 piece) reading chunk c, writing into
  //    the first 118 MB of the reference chunk r; kind[c] == kind[r] iff that pair is slow.
  const int S1 = 480, pc = 60;
  const int64_t is_b = (int64_t)pc * 16 * 4096 + 4096, os_b = (int64_t)pc * 4096;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PA_CHK(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    pair_alloc_free(a);
    return IAMF_HIP_ERR_DEVICE;
  }
  auto pair_ms = [&](int c, int r) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, st);
      (void)iamf_hip_probe_traffic(S1, pc, 16, 1, va[c], is_b, va[r], os_b, stream);
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    return best;
  };
  std::vector<int> kind(want, -1);
  int n_kinds = 0;
  // what "fast" is on this card: the smallest time in two columns of the pair matrix — against chunk 0 and against the
  // chunk that is slowest against chunk 0 (of chunk 0's kind if there is more than one kind: the other kinds are then
  // fast against it)
  std::vector<std::vector<float>> col(want);
  auto column = [&](int r) -> const std::vector<float> & {
    if (col[r].empty()) {
      col[r].assign(want, 0.f);
      for (int c = 0; c < want; ++c)
        if (c != r) col[r][c] = pair_ms(c, r);
    }
    return col[r];
  };
  float fast_ms = 1e30f;
  {
    const std::vector<float> &c0 = column(0);
    int slowest = 1;
    for (int c = 1; c < want; ++c) {
      if (c0[c] < fast_ms) fast_ms = c0[c];
      if (c0[c] > c0[slowest]) slowest = c;
    }
    const std::vector<float> &c1 = column(slowest);
    for (int c = 0; c < want; ++c)
      if (c != slowest && c1[c] < fast_ms) fast_ms = c1[c];
  }
  const float thr = 1.07f * fast_ms;
  for (int r = 0; r < want && n_kinds < 5; ++r) {
    if (kind[r] >= 0) continue;
    kind[r] = n_kinds;   // chunk r founds a kind; every unassigned chunk that is slow against it joins
    const std::vector<float> &t = column(r);
    for (int c = 0; c < want; ++c)
      if (kind[c] < 0 && t[c] > thr) kind[c] = n_kinds;
    ++n_kinds;
  }
  if (getenv("IAMF_HIP_PAIR_DEBUG")) {
    fprintf(stderr, "pair_alloc: fast pair %.3f ms; kinds of the %d chunks:", fast_ms, want);
    for (int c = 0; c < want; ++c) fprintf(stderr, " %d", kind[c]);
    fprintf(stderr, "\n");
  }
  for (int c = 0; c < want; ++c)
    if (kind[c] < 0) kind[c] = n_kinds - 1;
  if (kinds_found) *kinds_found = n_kinds;
  // 3. which kind for the input, which for the outputs: cross-kind pairs are not all alike (6.4 or 6.0 TB/s for the
  //    headline shape, against 5.45 for a same-kind pair), so every ordered pair of kinds that has the chunks is timed
  //    on its first members and the fastest taken.  (One kind only, or not enough chunks: whatever there is.)
  std::vector<int> count(n_kinds, 0), rep(n_kinds, -1);
  for (int c = 0; c < want; ++c) {
    if (rep[kind[c]] < 0) rep[kind[c]] = c;
    ++count[kind[c]];
  }
  const int need_out = n_out * n_out_ch;
  int kin = -1, kout = -1;
  float best_pair = 1e30f;
  for (int ki = 0; ki < n_kinds; ++ki)
    for (int ko = 0; ko < n_kinds; ++ko) {
      if (ki == ko || count[ki] < n_in_ch || count[ko] < need_out) continue;
      const float t = column(rep[ko])[rep[ki]];   // input = first chunk of kind ki, output = start of the first chunk of kind ko
      if (getenv("IAMF_HIP_PAIR_DEBUG")) fprintf(stderr, "pair_alloc: input kind %d, output kind %d: %.3f ms\n", ki, ko, t);
      if (t < best_pair) {
        best_pair = t;
        kin = ki;
        kout = ko;
      }
    }
  if (kin < 0) {   // no ordered pair of kinds has the chunks: the most numerous kind for the input, anything else for the outputs
    kin = 0;
    for (int k = 1; k < n_kinds; ++k)
      if (count[k] > count[kin]) kin = k;
  }
  if (count[kin] < n_in_ch) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    pair_alloc_free(a);
    return IAMF_HIP_ERR_DEVICE;
  }
  std::vector<int> in_ids, out_ids;
  for (int c = 0; c < want && (int)in_ids.size() < n_in_ch; ++c)
    if (kind[c] == kin) in_ids.push_back(c);
  std::vector<char> used(want, 0);
  for (int c : in_ids) used[c] = 1;
  for (int c = 0; c < want && (int)out_ids.size() < need_out; ++c)
    if (!used[c] && kind[c] == kout) { out_ids.push_back(c); used[c] = 1; }
  for (int c = 0; c < want && (int)out_ids.size() < need_out; ++c)
    if (!used[c] && kind[c] != kin) { out_ids.push_back(c); used[c] = 1; }
  for (int c = 0; c < want && (int)out_ids.size() < need_out; ++c)
    if (!used[c]) { out_ids.push_back(c); used[c] = 1; }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if ((int)out_ids.size() < need_out) {
    pair_alloc_free(a);
    return IAMF_HIP_ERR_DEVICE;
  }
  // 4. assemble: the chosen chunks leave their own ranges and are mapped side by side
  PA_CHK(hipStreamSynchronize(st));
  auto assemble = [&](const int *ids, int n, void **base) -> bool {
    if (hipMemAddressReserve(base, (size_t)n * chunk, 0, nullptr, 0) != hipSuccess) return false;
    a->ranges.push_back({*base, (size_t)n * chunk, false});
    for (int k = 0; k < n; ++k) {
      if (hipMemUnmap(va[ids[k]], chunk) != hipSuccess) return false;
      for (auto &r : a->ranges)
        if (r.va == va[ids[k]]) r.mapped = false;
      if (hipMemMap(static_cast<char *>(*base) + (size_t)k * chunk, chunk, 0, a->handles[ids[k]], 0) != hipSuccess) return false;
    }
    a->ranges.back().mapped = true;   // (a partially mapped range is unmapped as a whole: hipMemUnmap of the range)
    return hipMemSetAccess(*base, (size_t)n * chunk, &acc, 1) == hipSuccess;
  };
  if (!assemble(in_ids.data(), n_in_ch, d_in)) {
    pair_alloc_free(a);
    return IAMF_HIP_ERR_DEVICE;
  }
  for (int j = 0; j < n_out; ++j)
    if (!assemble(out_ids.data() + (size_t)j * n_out_ch, n_out_ch, &d_out[j])) {
      pair_alloc_free(a);
      return IAMF_HIP_ERR_DEVICE;
    }
  // 5. the chunks nobody needs go back
  for (int c = 0; c < want; ++c)
    if (!used[c]) {
      (void)hipMemUnmap(va[c], chunk);
      for (auto &r : a->ranges)
        if (r.va == va[c]) r.mapped = false;
      (void)hipMemRelease(a->handles[c]);
      a->handles[c] = nullptr;
    }
  {
    std::vector<hipMemGenericAllocationHandle_t> keep;
    for (auto h : a->handles)
      if (h) keep.push_back(h);
    a->handles.swap(keep);
  }
#undef PA_CHK
  *out_alloc = a;
  return IAMF_HIP_OK;
}
