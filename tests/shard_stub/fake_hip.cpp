/*
 * TEST INFRASTRUCTURE — not a CPU path of the product.
 *
 * A HIP runtime stand-in with N "devices" for the multi-device entry of the C ABI (iac_amd/csrc/iamf_shard.hip):
 * VERDICT r3 #4 — that file's N > 1 path (one host thread per device, ncclCommInitAll, per-peer send / recv offsets,
 * root != 0, the gather beside the next render) had only ever run with ONE device.  Here it runs with 2 .. 8 on a host
 * without GPUs, under ASan / UBSan and TSan.
 *
 * What is modelled, because the shard's correctness depends on it:
 *   - the current device is per host thread (hipSetDevice / hipGetDevice);
 *   - a stream is an in-order queue drained by its own thread: work is ASYNCHRONOUS to the host, and two streams run
 *     concurrently — a missing event wait shows as wrong bytes (every queued operation also sleeps a little so that
 *     the host runs ahead) and, under TSan, as a data race on the buffers;
 *   - hipEventRecord / hipStreamWaitEvent have HIP's semantics (a wait refers to the record calls made before it);
 *   - "device memory" is host memory tagged with the device that allocated it; a batch refuses calls while another
 *     device is current (as libiamf_hip.so's on_batch_device does);
 *   - the batch ABI stand-in "renders" stream j of a batch by copying the head of its input row, XOR a per-call key, into
 *     its PCM row on the given stream — what every byte of a gathered buffer must be is then known to the driver.
 * Exported beyond the HIP names: fake_stream_enqueue (used by the RCCL stand-in, fake_rccl.cpp) and fake_hip_* knobs.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "iamf_hip.h"

namespace {
int g_devices = 8;
std::atomic<int> g_op_delay_us{200}, g_render_delay_us{30};
std::atomic<long> g_live_streams{0}, g_live_events{0}, g_live_allocs{0};
thread_local int t_device = 0;

using Clock = std::chrono::steady_clock;
}  // namespace

struct ihipStream_t {
  int device = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false, busy = false;
  std::thread th;
  ihipStream_t() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty() && stop) return;
        std::function<void()> f = std::move(q.front());
        q.pop_front();
        busy = true;
        lk.unlock();
        f();
        lk.lock();
        busy = false;
        cv.notify_all();
      }
    });
  }
  void push(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(mu);
    q.push_back(std::move(f));
    cv.notify_all();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return q.empty() && !busy; });
  }
  ~ihipStream_t() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
      cv.notify_all();
    }
    th.join();
  }
};

struct ihipEvent_t {
  int device = 0;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, completed = 0;   // record calls made / record operations a stream has executed
  Clock::time_point when;
};

extern "C" {

/* knobs for the driver */
void fake_hip_set_devices(int n) { g_devices = n; }
void fake_hip_set_op_delay_us(int us) { g_op_delay_us = us; }
void fake_hip_set_render_delay_us(int us) { g_render_delay_us = us; }
long fake_hip_live_objects(void) { return g_live_streams + g_live_events + g_live_allocs; }
/* a queued operation on a stream (the RCCL stand-in's "kernels") */
void fake_stream_enqueue(hipStream_t s, void (*fn)(void *), void *arg) {
  s->push([fn, arg] { fn(arg); });
}
int fake_stream_device(hipStream_t s) { return s->device; }

hipError_t hipGetDeviceCount(int *n) { *n = g_devices; return hipSuccess; }
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= g_devices) return hipErrorInvalidDevice;
  t_device = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n) {
  *p = calloc(1, n ? n : 1);
  if (*p) ++g_live_allocs;
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  if (p) --g_live_allocs;
  free(p);
  return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) {
  (void)flags;
  *s = new ihipStream_t();
  (*s)->device = t_device;
  ++g_live_streams;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipErrorInvalidValue;
  s->drain();
  delete s;
  --g_live_streams;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  if (!s) return hipErrorInvalidValue;
  s->drain();
  return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) {
  (void)flags;
  *e = new ihipEvent_t();
  (*e)->device = t_device;
  ++g_live_events;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
  if (!e) return hipErrorInvalidValue;
  delete e;
  --g_live_events;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (!e || !s) return hipErrorInvalidValue;
  if (e->device != s->device) return hipErrorInvalidHandle;   /* HIP: event and stream must belong to one device */
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    seq = ++e->recorded;
  }
  s->push([e, seq] {
    std::lock_guard<std::mutex> lk(e->mu);
    if (seq > e->completed) e->completed = seq;
    e->when = Clock::now();
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags) {
  (void)flags;
  if (!e || !s) return hipErrorInvalidValue;
  uint64_t target;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    target = e->recorded;   /* the record calls made so far; an event never recorded is complete */
  }
  if (!target) return hipSuccess;
  s->push([e, target] {
    std::unique_lock<std::mutex> lk(e->mu);
    e->cv.wait(lk, [e, target] { return e->completed >= target; });
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  if (!e) return hipErrorInvalidValue;
  std::unique_lock<std::mutex> lk(e->mu);
  const uint64_t target = e->recorded;
  e->cv.wait(lk, [e, target] { return e->completed >= target; });
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
  if (!e) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> lk(e->mu);
  return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  if (!ms || !a || !b) return hipErrorInvalidValue;
  std::unique_lock<std::mutex> la(a->mu, std::defer_lock), lb(b->mu, std::defer_lock);
  std::lock(la, lb);
  if (!a->completed || !b->completed) return hipErrorInvalidHandle;
  if (a->completed < a->recorded || b->completed < b->recorded) return hipErrorNotReady;
  *ms = std::chrono::duration<float, std::milli>(b->when - a->when).count();
  return hipSuccess;
}

hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s) {
  (void)kind;
  if (!dst || !src || !s || width > dpitch || width > spitch) return hipErrorInvalidValue;
  s->push([=] {
    std::this_thread::sleep_for(std::chrono::microseconds(g_op_delay_us.load()));
    for (size_t r = 0; r < height; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  });
  return hipSuccess;
}

/* ---- the batch ABI stand-in ---- */
struct iamf_hip_batch {
  iamf_hip_batch_config cfg;
  int device;
  unsigned calls;
  int pad_left;
};

int iamf_hip_format_bytes(int f) { return f == 16 ? 2 : f == 24 ? 3 : (f == 32 || f == -32) ? 4 : 0; }

int iamf_hip_batch_create(const iamf_hip_batch_config *c, iamf_hip_batch **out) {
  if (!c || !out || c->frame_size <= 0 || c->out_channels <= 0 || c->n_streams <= 0) return IAMF_HIP_ERR_BAD_ARG;
  iamf_hip_batch *b = new iamf_hip_batch();
  b->cfg = *c;
  b->device = t_device;
  b->calls = 0;
  b->pad_left = c->limiter_enable ? 240 : 0;
  ++g_live_allocs;
  *out = b;
  return IAMF_HIP_OK;
}
void iamf_hip_batch_destroy(iamf_hip_batch *b) {
  if (b) --g_live_allocs;
  delete b;
}
/* the byte the "renderer" writes at position k of stream j's PCM row in call number `call` of its batch */
static inline uint8_t fake_pcm_byte(uint8_t in, unsigned call) { return (uint8_t)(in ^ (uint8_t)(0x5a + 17 * call)); }

static int fake_emit(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride, int n, void *d_pcm, int64_t pcm_stride,
                     hipStream_t st) {
  if (t_device != b->device) return IAMF_HIP_ERR_INVALID_STATE;   /* as the library: the batch's device must be current */
  if (!st || st->device != b->device) return IAMF_HIP_ERR_BAD_ARG;
  const int skip = n < b->pad_left ? n : b->pad_left;
  b->pad_left -= skip;
  n -= skip;
  const int64_t bytes = (int64_t)n * b->cfg.out_channels * iamf_hip_format_bytes(b->cfg.out_format);
  if (bytes > pcm_stride && b->cfg.n_streams > 1) return IAMF_HIP_ERR_BUFFER_TOO_SMALL;
  const unsigned call = b->calls++;
  const int ns = b->cfg.n_streams;
  st->push([=] {
    /* short by default: a render that does not wait for the previous gather overwrites what that gather still reads */
    std::this_thread::sleep_for(std::chrono::microseconds(g_render_delay_us.load()));
    for (int j = 0; j < ns; ++j) {
      uint8_t *row = (uint8_t *)d_pcm + (int64_t)j * pcm_stride;
      const uint8_t *src = d_in ? (const uint8_t *)(d_in + (int64_t)j * in_stream_stride) : nullptr;
      for (int64_t k = 0; k < bytes; ++k) row[k] = fake_pcm_byte(src ? src[k] : (uint8_t)(j + k), call);
    }
  });
  return n;
}
int iamf_hip_batch_render(iamf_hip_batch *b, const float *d_in, int64_t in_stream_stride, int64_t in_frame_stride, int32_t n_frames,
                          void *d_pcm, int64_t pcm_stride, void *stream) {
  (void)in_frame_stride;
  if (!b || !d_in || !d_pcm || n_frames <= 0) return IAMF_HIP_ERR_BAD_ARG;
  return fake_emit(b, d_in, in_stream_stride, n_frames * b->cfg.frame_size, d_pcm, pcm_stride, (hipStream_t)stream);
}
int iamf_hip_batch_flush(iamf_hip_batch *b, void *d_pcm, int64_t pcm_stride, void *stream) {
  if (!b || !d_pcm) return IAMF_HIP_ERR_BAD_ARG;
  if (!b->cfg.limiter_enable) return 0;
  const int keep = b->pad_left;
  b->pad_left = 0;
  return fake_emit(b, nullptr, 0, 240 - keep, d_pcm, pcm_stride, (hipStream_t)stream);
}

}  // extern "C"
