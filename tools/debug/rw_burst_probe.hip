// rw_burst_probe.hip — VERDICT r3 #1: does the SHAPE of the headline kernel's PCM stores change what the 4 B written per
// 64 B read cost?  The headline kernel's traffic with no compute: one 256-thread workgroup per stream, per 1024-sample
// chunk every lane reads 16 rows x 16 B (planar input: channel rows `chunks` x 4 KiB apart, as render_fast_kernel<16, 2>
// reads a 64-frame call) one chunk ahead, and writes 16 B of "PCM" (4 KiB per workgroup and chunk).  Variants of the STORE
// side only:
//   K      chunks of PCM held in registers and stored back to back as one K x 4 KiB burst per workgroup (K = 1: the kernel today)
//   stag   the burst phase of a workgroup shifted by (workgroup / 8) % K: neighbours on an XCD do not burst together
//   fm     frame-major output: chunk c of stream s at (c * S + s) * 4 KiB — the 512 workgroups' stores of one time step
//          are one contiguous 2 MiB region instead of 512 regions 256 KiB apart
//   w0     the burst issued by wave 0 alone (through LDS): 4 x K KiB contiguous per wave-instruction sequence
// First the plain shape is timed on every (input, output) allocation pair to find a FAST and a SLOW pair on this card
// (NOTEBOOK §3: "kinds of region"); the variants run on both.
//   hipcc --offload-arch=gfx950 -O3 tools/debug/rw_burst_probe.hip -o tools/bin/rw_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using v4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int ROWS = 16;

// MODE 0: no stores; 1: register burst; 2: burst by wave 0 through LDS; 3: register burst released by the DEVICE-WIDE clock
// (s_memrealtime, 100 MHz): every workgroup stores what it holds when (now % period) < window, or when its K slots are
// full — all workgroups write in the same windows, the memory sees read phases and write phases
template <int K, int MODE, bool FM, bool STAG>
__global__ __launch_bounds__(256, 2) void probe(const v4 *in, u4 *out, int chunks, int spin, unsigned period = 0, unsigned window = 0) {
  const int s = blockIdx.x, t = threadIdx.x, S = gridDim.x;
  const v4 *src = in + (long)s * chunks * ROWS * 256;   // [stream][row][chunk][256 lanes] x 16 B
  const long rstride = (long)chunks * 256;
  __shared__ u4 stage[MODE == 2 ? K * 256 : 1];
  v4 x[ROWS];
#pragma unroll
  for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * rstride + t);
  u4 hold[K];
  const int phase = STAG ? (int)((blockIdx.x >> 3) % K) : 0;
  int held = 0, c_first = 0;
  bool first_burst = STAG && phase != 0;   // a staggered workgroup's first burst is `phase` chunks long, then every K
  for (int c = 0; c < chunks; ++c) {
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < ROWS; ++m) a += x[m].x + x[m].y + x[m].z + x[m].w;
    if (c + 1 < chunks) {
#pragma unroll
      for (int m = 0; m < ROWS; ++m) x[m] = __builtin_nontemporal_load(src + m * rstride + (long)(c + 1) * 256 + t);
    }
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    const u4 w = {__float_as_uint(a), (unsigned)c, (unsigned)t, 0u};
    if (MODE == 0) {
      if (a == 123.456f) out[t] = w;
      continue;
    }
    // hold[] is indexed statically: shift in (K <= 8 moves per chunk; the real kernel would unroll its chunk loop K-fold)
#pragma unroll
    for (int k = 0; k + 1 < K; ++k) hold[k] = hold[k + 1];
    hold[K - 1] = w;
    if (held == 0) c_first = c;
    ++held;
    bool fire = held == K || c + 1 == chunks || (first_burst && held == phase);
    if (MODE == 3) {
      const unsigned now = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_s_memrealtime() & 0xffffffffu));
      fire = fire || (now % period) < window;
    }
    if (!fire) continue;
    first_burst = false;
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int idx = k - (K - held);   // hold[K - held .. K - 1] are the `held` chunks c_first ..
        if (idx >= 0) {
          const long cc = c_first + idx;
          u4 *to = FM ? out + (cc * S + s) * 256 + t : out + ((long)s * chunks + cc) * 256 + t;
          *to = hold[k];
        }
      }
    } else {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; ++k) stage[k * 256 + t] = hold[k];
      __syncthreads();
      if (t < 64) {
        for (int k = K - held; k < K; ++k) {
          const long cc = c_first + (k - (K - held));
          u4 *to = FM ? out + (cc * S + s) * 256 : out + ((long)s * chunks + cc) * 256;
#pragma unroll
          for (int q = 0; q < 4; ++q) to[q * 64 + t] = stage[k * 256 + q * 64 + t];
        }
      }
    }
    held = 0;
  }
}

struct Timer {
  hipEvent_t e0, e1;
  Timer() { hipEventCreate(&e0); hipEventCreate(&e1); }
  template <class F> float best(F f, int reps = 5) {
    float b = 1e9f;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0);
      f();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < b) b = ms;
    }
    return b;
  }
};

template <int K, int MODE, bool FM, bool STAG>
void one(Timer &T, const char *pair, const v4 *in, u4 *out, int S, int chunks, int spin) {
  const float ms = T.best([&] { probe<K, MODE, FM, STAG><<<S, 256>>>(in, out, chunks, spin); });
  const double bytes = (double)S * chunks * (ROWS * 4096.0 + (MODE ? 4096.0 : 0.0));
  printf("%-5s spin %3d  K %d %-5s %-3s %-4s  %.3f ms  %5.0f GB/s  %6.1f Gsamples/s\n", pair, spin, K,
         MODE == 0 ? "none" : (MODE == 1 ? "regs" : "wave0"), FM ? "fm" : "sm", STAG ? "stag" : "-", ms, bytes / ms / 1e6,
         (double)S * chunks * 1024 / ms / 1e6);
  fflush(stdout);
}

template <int K>
void allk(Timer &T, const char *pair, const v4 *in, u4 *out, int S, int chunks, int spin) {
  one<K, 1, false, false>(T, pair, in, out, S, chunks, spin);
  if (K > 1) one<K, 1, false, true>(T, pair, in, out, S, chunks, spin);
  one<K, 1, true, false>(T, pair, in, out, S, chunks, spin);
  one<K, 2, false, false>(T, pair, in, out, S, chunks, spin);
}

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 512, chunks = 64, NI = 8, NO = 6;
  const size_t in_bytes = (size_t)S * chunks * ROWS * 4096, out_bytes = (size_t)S * chunks * 4096;
  std::vector<v4 *> in(NI);
  std::vector<u4 *> out(NO);
  for (auto &p : in) {
    if (hipMalloc(&p, in_bytes) != hipSuccess) { printf("input allocation failed\n"); return 1; }
    hipMemset(p, 0, in_bytes);
  }
  for (auto &p : out) {
    if (hipMalloc(&p, out_bytes) != hipSuccess) { printf("output allocation failed\n"); return 1; }
    hipMemset(p, 0, out_bytes);
  }
  Timer T;
  printf("# plain shape (K 1, stream-major) on every pair, GB/s; rows = inputs, columns = outputs; %d workgroups\n", S);
  int bi = 0, bo = 0, wi = 0, wo = 0;
  float bms = 1e9f, wms = 0.f;
  for (int i = 0; i < NI; ++i) {
    printf("input %d:", i);
    for (int o = 0; o < NO; ++o) {
      const float ms = T.best([&] { probe<1, 1, false, false><<<S, 256>>>(in[i], out[o], chunks, 0); }, 4);
      printf(" %5.0f", (double)S * chunks * 17 * 4096.0 / ms / 1e6);
      if (ms < bms) { bms = ms; bi = i; bo = o; }
      if (ms > wms) { wms = ms; wi = i; wo = o; }
    }
    printf("\n");
  }
  printf("# fast pair: input %d output %d (%.3f ms); slow pair: input %d output %d (%.3f ms)\n", bi, bo, bms, wi, wo, wms);
  for (int pr = 0; pr < 2; ++pr) {
    const char *name = pr ? "slow" : "fast";
    const v4 *ip = pr ? in[wi] : in[bi];
    u4 *op = pr ? out[wo] : out[bo];
    for (int spin : {0, 300})
      for (unsigned period : {1000u, 2000u, 4000u, 8000u})       // 10 / 20 / 40 / 80 us
        for (unsigned window : {period / 16, period / 8, period / 4}) {
          const float ms = T.best([&] { probe<8, 3, false, false><<<S, 256>>>(ip, op, chunks, spin, period, window); });
          printf("%-5s spin %3d  K 8 clock period %5.0f us window %5.1f us  %.3f ms  %5.0f GB/s  %6.1f Gsamples/s\n", name, spin,
                 period * 0.01, window * 0.01, ms, (double)S * chunks * 17 * 4096.0 / ms / 1e6, (double)S * chunks * 1024 / ms / 1e6);
          fflush(stdout);
        }
    for (int spin : {0, 300}) {
      one<1, 0, false, false>(T, name, ip, op, S, chunks, spin);
      allk<1>(T, name, ip, op, S, chunks, spin);
      allk<2>(T, name, ip, op, S, chunks, spin);
      allk<4>(T, name, ip, op, S, chunks, spin);
      allk<8>(T, name, ip, op, S, chunks, spin);
    }
  }
  return 0;
}
