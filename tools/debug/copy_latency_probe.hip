// Probe (not product): what a round of the group of handles costs on the device side — pinned H2D copies of various
// sizes and counts, a kernel behind them, a D2H behind that — and whether a copy issued early progresses while the
// host does something else.   hipcc --offload-arch=gfx950 -O2 -o copy_latency_probe copy_latency_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(const unsigned *in, unsigned *out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1;
}
static void spin(double us) { double t = now(); while ((now() - t) * 1e6 < us) {} }
int main() {
  const size_t MB = 1 << 20;
  void *h, *d, *d2, *h2;
  hipHostMalloc(&h, 16 * MB, 0); hipHostMalloc(&h2, 2 * MB, 0);
  hipMalloc(&d, 16 * MB); hipMalloc(&d2, 16 * MB);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (int warm = 0; warm < 3; ++warm) { hipMemcpyAsync(d, h, 16 * MB, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
  const int R = 50;
  for (size_t total : {2 * MB, 8 * MB}) {
    for (int parts : {1, 4, 8}) {
      for (int mode = 0; mode < 4; ++mode) {
        // 0: copies, sync.  1: copies, kernel, D2H 256 KB, sync.  2: as 1 with 150 us of host work between the copies and the kernel
        // 3: as 2 but an event recorded + queried after the copies (does a query flush the queue?)
        double t_issue = 0, t_wait = 0, t_all = 0;
        for (int r = 0; r < R; ++r) {
          double t0 = now();
          for (int p = 0; p < parts; ++p)
            hipMemcpyAsync((char *)d + p * (total / parts), (char *)h + p * (total / parts), total / parts, hipMemcpyHostToDevice, st);
          if (mode == 3) { hipEventRecord(ev, st); (void)hipEventQuery(ev); }
          if (mode >= 2) spin(150);
          if (mode >= 1) {
            hipLaunchKernelGGL(touch, dim3((unsigned)(total / 4 / 256)), dim3(256), 0, st, (const unsigned *)d, (unsigned *)d2, total / 4);
            hipMemcpyAsync(h2, d2, 256 * 1024, hipMemcpyDeviceToHost, st);
          }
          double t1 = now();
          hipStreamSynchronize(st);
          double t2 = now();
          t_issue += t1 - t0; t_wait += t2 - t1; t_all += t2 - t0;
        }
        printf("total %zu MB parts %d mode %d: issue %.1f us, wait %.1f us, all %.1f us\n", total / MB, parts, mode, t_issue / R * 1e6, t_wait / R * 1e6, t_all / R * 1e6);
      }
    }
  }
  // zero-copy: the kernel reads the pinned host buffer itself
  for (size_t total : {2 * MB, 8 * MB}) {
    double t_all = 0;
    for (int r = 0; r < R; ++r) {
      double t0 = now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(total / 4 / 256)), dim3(256), 0, st, (const unsigned *)h, (unsigned *)d2, total / 4);
      hipMemcpyAsync(h2, d2, 256 * 1024, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      t_all += now() - t0;
    }
    printf("zero-copy read of %zu MB by the kernel + D2H: all %.1f us\n", total / MB, t_all / R * 1e6);
  }
  return 0;
}
