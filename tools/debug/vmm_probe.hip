// vmm_probe.hip — can a large stream buffer be ASSEMBLED from device-memory chunks of one kind (NOTEBOOK.md 3)?
// Physical 2 GiB chunks from hipMemCreate, each mapped on its own and classified by the headline traffic shape
// against one small reference output (fast = another kind than the reference, slow = the reference's kind); then
// a 14 GiB input made of "fast" chunks and an 8 GiB output made of "slow" chunks (= a clean different-kind pair), the
// opposite assignment, and plain hipMalloc buffers, all timed with the cfg3 shape at 2048 streams.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/vmm_probe.hip -o /tmp/vmm && /tmp/vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

#define CHK(x)                                                                     \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);         \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

template <int ROWS, int PIECES>
__global__ __launch_bounds__(256, 2) void k_stream(const v4 *in, u4 *out, int chunks, long in_stride4) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const v4 *src = in + (long)s * in_stride4;
  u4 *dst = out + (long)s * chunks * PIECES * 256;
  v4 x[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * 256 + t);
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    if (c + 1 < chunks) {
#pragma unroll
      for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + ((long)(c + 1) * ROWS + m) * 256 + t);
    }
    __syncthreads();
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
#pragma unroll
    for (int k = 0; k < PIECES; ++k) __builtin_nontemporal_store(w, dst + ((long)c * PIECES * 4 + wave * PIECES + k) * 64 + lane);
  }
}

static hipEvent_t e0, e1;
template <typename F>
float best_ms(F launch) {
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  int dev = 0;
  CHK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t chunk = (size_t)2 << 30;   // 2 GiB physical chunks
  printf("allocation granularity %zu, chunk %zu\n", gran, chunk);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;

  const int NCH = 40;
  std::vector<hipMemGenericAllocationHandle_t> h(NCH);
  std::vector<void *> va(NCH);
  for (int i = 0; i < NCH; ++i) {
    CHK(hipMemCreate(&h[i], chunk, &prop, 0));
    CHK(hipMemAddressReserve(&va[i], chunk, 0, nullptr, 0));
    CHK(hipMemMap(va[i], chunk, 0, h[i], 0));
    CHK(hipMemSetAccess(va[i], chunk, &acc, 1));
    CHK(hipMemset(va[i], 0, chunk));
  }
  // classification: the headline shape, 480 streams (fits 2 GiB with the 4 KiB stagger), one small reference output
  const int S1 = 480, chunks = 64;
  const long stride4 = (long)chunks * 16 * 256 + 256;
  u4 *ref_out;
  CHK(hipMalloc(&ref_out, (size_t)S1 * chunks * 4096));
  std::vector<float> rate(NCH);
  // reference outputs: one small hipMalloc buffer, and the first 126 MB of every eighth chunk
  for (int r = 0; r < 6; ++r) {
    u4 *o = r == 0 ? ref_out : reinterpret_cast<u4 *>(va[8 * (r - 1)]);
    printf("inputs against reference output %d (%s): ", r, r == 0 ? "hipMalloc" : "start of a chunk");
    for (int i = 0; i < NCH; ++i) {
      if (r > 0 && i == 8 * (r - 1)) { printf(" --"); continue; }
      const float ms = best_ms([&] { k_stream<16, 1><<<S1, 256>>>(static_cast<const v4 *>(va[i]), o, chunks, stride4); });
      const float g = (float)((double)S1 * chunks * 17 * 4096 / ms / 1e6);
      if (r == 0) rate[i] = g;
      printf(" %2.0f", g / 100);
    }
    printf("   (last %.0f GB/s)\n", rate[NCH - 1]);
  }
  std::vector<int> fast, slow;
  for (int i = 0; i < NCH; ++i) (rate[i] > 5900 ? fast : slow).push_back(i);
  printf("%zu fast chunks, %zu slow chunks\n", fast.size(), slow.size());
  const int NIN = 7, NOUT = 4;   // cfg3 at 2048 streams: input 12.9 GB -> 7 chunks, output 6.4 GB -> 4 chunks
  if ((int)fast.size() < NIN + NOUT || (int)slow.size() < NOUT) {
    printf("not enough chunks of both kinds on this card for the assembled test\n");
    return 0;
  }
  auto assemble = [&](const std::vector<int> &ids, int n, void **out_va) -> int {
    CHK(hipMemAddressReserve(out_va, (size_t)n * chunk, 0, nullptr, 0));
    for (int k = 0; k < n; ++k) {
      CHK(hipMemUnmap(va[ids[k]], chunk));
      CHK(hipMemMap(static_cast<char *>(*out_va) + (size_t)k * chunk, chunk, 0, h[ids[k]], 0));
    }
    CHK(hipMemSetAccess(*out_va, (size_t)n * chunk, &acc, 1));
    return 0;
  };
  const int S = 2048;
  const long stride4_3 = (long)chunks * 16 * 256 + 256;
  auto run = [&](const char *name, const void *in, void *out) {
    const float ms = best_ms([&] { k_stream<16, 12><<<S, 256>>>(static_cast<const v4 *>(in), static_cast<u4 *>(out), chunks, stride4_3); });
    printf("cfg3 shape, 2048 streams, %-44s %.3f ms  %5.0f GB/s\n", name, ms, (double)S * chunks * 28 * 4096 / ms / 1e6);
  };
  // (a) plain hipMalloc buffers
  void *pin, *pout;
  CHK(hipMalloc(&pin, (size_t)S * stride4_3 * 16));
  CHK(hipMalloc(&pout, (size_t)S * chunks * 12 * 4096));
  CHK(hipMemset(pin, 0, (size_t)S * stride4_3 * 16));
  run("plain hipMalloc input and output", pin, pout);
  // (b) input of fast chunks, output of slow chunks (different kinds); (c) input and output both of fast chunks
  void *in_f, *out_s, *out_f;
  std::vector<int> f_in(fast.begin(), fast.begin() + NIN), f_out(fast.begin() + NIN, fast.begin() + NIN + NOUT),
      s_out(slow.begin(), slow.begin() + NOUT);
  if (assemble(f_in, NIN, &in_f) || assemble(s_out, NOUT, &out_s) || assemble(f_out, NOUT, &out_f)) return 1;
  run("input of F chunks, output of s chunks", in_f, out_s);
  run("input of F chunks, output of F chunks", in_f, out_f);
  run("input of F chunks, plain hipMalloc output", in_f, pout);
  run("plain hipMalloc input, output of s chunks", pin, out_s);
  return 0;
}
