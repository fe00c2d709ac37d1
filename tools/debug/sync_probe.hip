// sync_probe.hip — what does it cost a host to learn that a small kernel (one workgroup, ~5 us: a single decoder handle's frame) has finished?
//   (a) hipStreamSynchronize   (b) hipEventRecord + spinning on hipEventQuery   (c) the kernel's last lane writes a word of
//   pinned host memory behind a system-scope fence, the host spins on it   (d) as (c), the word written by a second tiny kernel
//   (e) NO launch per call: one resident workgroup polls a doorbell word in pinned memory (bounded: it leaves after max_polls reads
//   without a new request, and after n_calls requests), does the same work, writes the flag — what a persistent per-handle kernel would cost
// hipcc --offload-arch=gfx950 -O3 tools/debug/sync_probe.hip -o tools/bin/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(const float *in, float *out, volatile unsigned *flag, unsigned seq, int spin) {
  float a = in[threadIdx.x];
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  out[threadIdx.x] = a;   // pinned host memory, as the facade's PCM
  if (flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      *flag = seq;
    }
  }
}
__global__ void signal(volatile unsigned *flag, unsigned seq) {
  __threadfence_system();
  *flag = seq;
}
__global__ void persistent(const float *in, float *out, volatile unsigned *door, volatile unsigned *flag, int spin, unsigned n_calls,
                           unsigned max_polls) {
  __shared__ unsigned go;
  for (unsigned seq = 1; seq <= n_calls; ++seq) {
    if (threadIdx.x == 0) {
      unsigned polls = 0, v;
      while ((v = __atomic_load_n((const unsigned *)door, __ATOMIC_RELAXED)) != seq && ++polls < max_polls) {}
      go = v == seq;
    }
    __syncthreads();
    if (!go) return;   // nobody rang for max_polls reads: every wave leaves
    float a = in[threadIdx.x];
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    out[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      *flag = seq;
    }
    __syncthreads();
  }
}
int main() {
  float *in, *out;
  unsigned *flag;
  hipHostMalloc(&in, 4096, 0);
  hipHostMalloc(&out, 4096, 0);
  hipHostMalloc(&flag, 64, 0);
  *flag = 0;
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 3000, spin = 400;
  unsigned seq = 0;
  for (int mode = 0; mode < 4; ++mode) {
    for (int warm = 0; warm < 2; ++warm) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        ++seq;
        if (mode == 0) {
          work<<<1, 256, 0, st>>>(in, out, nullptr, seq, spin);
          hipStreamSynchronize(st);
        } else if (mode == 1) {
          work<<<1, 256, 0, st>>>(in, out, nullptr, seq, spin);
          hipEventRecord(ev, st);
          while (hipEventQuery(ev) == hipErrorNotReady) {}
        } else if (mode == 2) {
          work<<<1, 256, 0, st>>>(in, out, flag, seq, spin);
          while (*(volatile unsigned *)flag != seq) {}
        } else {
          work<<<1, 256, 0, st>>>(in, out, nullptr, seq, spin);
          signal<<<1, 1, 0, st>>>(flag, seq);
          while (*(volatile unsigned *)flag != seq) {}
        }
      }
      auto t1 = std::chrono::steady_clock::now();
      if (warm) printf("mode %d (%s): %.2f us per call\n", mode, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "event + hipEventQuery spin" : mode == 2 ? "kernel writes pinned flag, host spins" : "second kernel writes the flag",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    hipStreamSynchronize(st);
  }
  {  // (e)
    unsigned *door;
    hipHostMalloc(&door, 64, 0);
    for (int warm = 0; warm < 2; ++warm) {
      *door = 0;
      *flag = 0;
      persistent<<<1, 256, 0, st>>>(in, out, door, flag, spin, (unsigned)N, 4000000u);
      auto t0 = std::chrono::steady_clock::now();
      for (unsigned i = 1; i <= (unsigned)N; ++i) {
        in[0] = (float)i;   // the "frame"
        __atomic_store_n(door, i, __ATOMIC_RELEASE);
        while (*(volatile unsigned *)flag != i) {}
      }
      auto t1 = std::chrono::steady_clock::now();
      hipStreamSynchronize(st);
      if (warm) printf("mode 4 (resident workgroup polls a pinned doorbell, no launch): %.2f us per call\n",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
