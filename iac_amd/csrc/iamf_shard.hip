// iamf_shard.hip — one node, several MI355X: the batch renderer sharded over devices from C, with the final gather of
// packed PCM over RCCL (include/iamf_hip.h: iamf_hip_shard_*).  Host code only.
//
// SURVEY 8(e) / BASELINE north_star: every piece of state of the rendering path is per decoder handle
// (IAMF_decoder_private.h:342-345, audio_effect_peak_limiter.h:48-73, speex_resampler.h:67-101), so independent streams
// shard over the GPUs of a node with NO exchange while rendering; the job's one exchange is the gather of the packed PCM
// to one device ("RCCL over xGMI only for the final batched gather").  Here: `n_streams` streams are split into
// contiguous blocks (iamf_hip_shard_split), each device gets one batch, one HIP stream for rendering, a second one for
// the gather and one host thread that issues its launches, so the devices' launch latencies overlap; the gather is
// ncclSend / ncclRecv inside one ncclGroupStart / ncclGroupEnd on the second streams, ordered behind the render by an
// event — step i's gather runs while step i + 1 renders.
//
// RCCL is loaded with dlopen on first use: libiamf_hip.so itself does not link librccl, a single-GPU host that never
// creates a shard never loads it, and a node without RCCL gets IAMF_HIP_ERR_UNIMPLEMENTED from iamf_hip_shard_gather
// (rendering still works).  One process drives all devices (ncclCommInitAll): the C-side counterpart of bench.py's
// one-process-per-GPU torch.distributed path, for hosts written in C like the reference's.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <rccl/rccl.h>   // types only: every entry point is resolved with dlsym

#include "../../include/iamf_hip.h"

namespace {

struct Rccl {
  void *so = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  Rccl() {
    // IAMF_HIP_RCCL_LIB: the library to load instead (a site's own RCCL build; the N-device rehearsal of this file on a
    // host without GPUs, tests/shard_stub/)
    const char *own = getenv("IAMF_HIP_RCCL_LIB");
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (own && *own) {
      so = dlopen(own, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char *n : names)
        if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!so) return;
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(so, name))
    SYM(GetVersion, "ncclGetVersion");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    ok = GetVersion && CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
  }
};
Rccl &rccl() {
  static Rccl r;
  return r;
}

// one host thread per device: runs the closures handed to it, in order, with its device current
struct Worker {
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, stop = false;
  void start() {
    th = std::thread([this] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || stop; });
        if (stop) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
  }
  void finish() {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return done; });
      stop = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct Dev {
  int device = 0, first = 0, count = 0;
  iamf_hip_batch *batch = nullptr;
  hipStream_t render = nullptr, gather = nullptr;
  hipEvent_t rendered = nullptr, gathered = nullptr;
  // (timing) around the device's share of a gather: a ring of pairs, so that closing the books of one gather never makes
  // the host wait for it while the next is being issued (that wait would throttle the run-ahead of step i + 1's render)
  static constexpr int kPairs = 4;
  hipEvent_t g_begin[kPairs] = {}, g_end[kPairs] = {};
  bool pending[kPairs] = {};   // pair k brackets a gather whose time is not yet in total_gather_ms
  int g_cur = 0;               // the pair of the last gather
  double last_gather_ms = 0.0;
  ncclComm_t comm = nullptr;
  Worker w;
  int result = 0;
  // rows of the device's streams packed back to back for the wire (and, on the root, received before they are spread
  // over a strided destination): grows with the largest gather seen
  void *pack = nullptr, *unpack = nullptr;
  size_t pack_bytes = 0, unpack_bytes = 0;
  // accounting (iamf_hip_shard_times)
  int64_t last_sent = 0, total_sent = 0, last_received = 0;
  double total_gather_ms = 0.0;
};

// room for `bytes` in a staging buffer of device d (current): reallocation waits for the gather stream, whose queued
// copies may still use the old one
int stage_room(Dev *d, void **buf, size_t *have, size_t bytes) {
  if (bytes <= *have) return 0;
  if (hipStreamSynchronize(d->gather) != hipSuccess) return 1;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *have = 0;
  if (hipMalloc(buf, bytes) != hipSuccess) return 1;
  *have = bytes;
  return 0;
}

// close the books of the last gather of device d (its events have completed, or are waited for): its time -> the total
// wait == false: only the gathers that have completed (hipEventQuery), no blocking; true: all of them
void settle_times(Dev *d, bool wait) {
  for (int k = 0; k < Dev::kPairs; ++k) {
    if (!d->pending[k]) continue;
    if (!wait && hipEventQuery(d->g_end[k]) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventSynchronize(d->g_end[k]) == hipSuccess && hipEventElapsedTime(&ms, d->g_begin[k], d->g_end[k]) == hipSuccess) {
      d->total_gather_ms += (double)ms;
      if (k == d->g_cur) d->last_gather_ms = (double)ms;
    }
    d->pending[k] = false;
  }
}
// the pair the next gather uses (if the ring has come round to a gather still in flight, that one is waited for)
int next_pair(Dev *d) {
  const int k = (d->g_cur + 1) % Dev::kPairs;
  if (d->pending[k]) {
    float ms = 0.f;
    if (hipEventSynchronize(d->g_end[k]) == hipSuccess && hipEventElapsedTime(&ms, d->g_begin[k], d->g_end[k]) == hipSuccess)
      d->total_gather_ms += (double)ms;
    d->pending[k] = false;
  }
  d->g_cur = k;
  return k;
}

}  // namespace

struct iamf_hip_shard {
  std::vector<Dev *> devs;
  int n_streams = 0;
  bool comms = false;
};

extern "C" {

int iamf_hip_shard_split(int n_streams, int n_devices, int index, int *first, int *count) {
  if (n_streams < 0 || n_devices <= 0 || index < 0 || index >= n_devices || !first || !count) return IAMF_HIP_ERR_BAD_ARG;
  // contiguous blocks whose sizes differ by at most one, the larger ones first (iac_amd/sharding.py: shard_streams)
  const int q = n_streams / n_devices, r = n_streams % n_devices;
  *first = index * q + (index < r ? index : r);
  *count = q + (index < r ? 1 : 0);
  return IAMF_HIP_OK;
}

const char *iamf_hip_shard_rccl_version(void) {
  static char buf[48];
  Rccl &R = rccl();
  int v = 0;
  if (!R.ok || R.GetVersion(&v) != ncclSuccess) return "";
  // NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch since 2.9
  snprintf(buf, sizeof(buf), "%d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
  return buf;
}

void iamf_hip_shard_destroy(iamf_hip_shard *s) {
  if (!s) return;
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (Dev *d : s->devs) {
    if (!d) continue;
    d->w.finish();
    if (hipSetDevice(d->device) == hipSuccess) {
      if (d->render) (void)hipStreamSynchronize(d->render);
      if (d->gather) (void)hipStreamSynchronize(d->gather);
      if (d->comm && rccl().ok) (void)rccl().CommDestroy(d->comm);
      if (d->batch) iamf_hip_batch_destroy(d->batch);
      if (d->rendered) (void)hipEventDestroy(d->rendered);
      if (d->gathered) (void)hipEventDestroy(d->gathered);
      for (int k = 0; k < Dev::kPairs; ++k) {
        if (d->g_begin[k]) (void)hipEventDestroy(d->g_begin[k]);
        if (d->g_end[k]) (void)hipEventDestroy(d->g_end[k]);
      }
      if (d->pack) (void)hipFree(d->pack);
      if (d->unpack) (void)hipFree(d->unpack);
      if (d->render) (void)hipStreamDestroy(d->render);
      if (d->gather) (void)hipStreamDestroy(d->gather);
    }
    delete d;
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  delete s;
}

int iamf_hip_shard_create(const iamf_hip_batch_config *cfg, const int *devices, int n_devices, iamf_hip_shard **out) {
  if (!cfg || !out || n_devices <= 0 || n_devices > 64 || cfg->n_streams < n_devices) return IAMF_HIP_ERR_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IAMF_HIP_ERR_DEVICE;
  for (int i = 0; i < n_devices; ++i) {
    const int dv = devices ? devices[i] : i;
    if (dv < 0 || dv >= ndev) return IAMF_HIP_ERR_BAD_ARG;
    for (int k = 0; k < i; ++k)
      if ((devices ? devices[k] : k) == dv) return IAMF_HIP_ERR_BAD_ARG;   // one shard per device
  }
  int cur = -1;
  (void)hipGetDevice(&cur);
  iamf_hip_shard *s = new (std::nothrow) iamf_hip_shard();
  if (!s) return IAMF_HIP_ERR_ALLOC_FAIL;
  s->n_streams = cfg->n_streams;
  int rc = IAMF_HIP_OK;
  for (int i = 0; i < n_devices && rc == IAMF_HIP_OK; ++i) {
    Dev *d = new (std::nothrow) Dev();
    if (!d) {
      rc = IAMF_HIP_ERR_ALLOC_FAIL;
      break;
    }
    s->devs.push_back(d);
    d->device = devices ? devices[i] : i;
    (void)iamf_hip_shard_split(cfg->n_streams, n_devices, i, &d->first, &d->count);
    if (hipSetDevice(d->device) != hipSuccess) {
      rc = IAMF_HIP_ERR_DEVICE;
      break;
    }
    iamf_hip_batch_config c = *cfg;
    c.n_streams = d->count;
    rc = iamf_hip_batch_create(&c, &d->batch);
    if (rc != IAMF_HIP_OK) break;
    if (hipStreamCreateWithFlags(&d->render, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&d->gather, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->rendered, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->gathered, hipEventDisableTiming) != hipSuccess) {
      rc = IAMF_HIP_ERR_DEVICE;
      break;
    }
    for (int k = 0; k < Dev::kPairs && rc == IAMF_HIP_OK; ++k)
      if (hipEventCreate(&d->g_begin[k]) != hipSuccess || hipEventCreate(&d->g_end[k]) != hipSuccess) rc = IAMF_HIP_ERR_DEVICE;
    if (rc != IAMF_HIP_OK) break;
    d->w.device = d->device;
    d->w.start();
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  if (rc != IAMF_HIP_OK) {
    iamf_hip_shard_destroy(s);
    return rc;
  }
  *out = s;
  return IAMF_HIP_OK;
}

int iamf_hip_shard_devices(const iamf_hip_shard *s) { return s ? (int)s->devs.size() : 0; }

int iamf_hip_shard_info(const iamf_hip_shard *s, int index, int *device, int *first, int *count) {
  if (!s || index < 0 || index >= (int)s->devs.size()) return IAMF_HIP_ERR_BAD_ARG;
  if (device) *device = s->devs[index]->device;
  if (first) *first = s->devs[index]->first;
  if (count) *count = s->devs[index]->count;
  return IAMF_HIP_OK;
}

iamf_hip_batch *iamf_hip_shard_batch(iamf_hip_shard *s, int index) {
  return (s && index >= 0 && index < (int)s->devs.size()) ? s->devs[index]->batch : nullptr;
}

void *iamf_hip_shard_render_stream(iamf_hip_shard *s, int index) {
  return (s && index >= 0 && index < (int)s->devs.size()) ? s->devs[index]->render : nullptr;
}

int iamf_hip_shard_render(iamf_hip_shard *s, const float *const *d_in, int64_t in_stream_stride, int64_t in_frame_stride,
                          int32_t n_frames, void *const *d_pcm, int64_t pcm_stream_stride_bytes) {
  if (!s || !d_in || !d_pcm) return IAMF_HIP_ERR_BAD_ARG;
  const int n = (int)s->devs.size();
  for (int i = 0; i < n; ++i)
    if (!d_in[i] || !d_pcm[i]) return IAMF_HIP_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {   // every device's thread issues its own launch: the launch latencies overlap
    Dev *d = s->devs[i];
    const float *in = d_in[i];
    void *pcm = d_pcm[i];
    d->w.submit([=] {
      // a PCM buffer that the previous step's gather is still reading must not be overwritten: the render waits for it
      (void)hipStreamWaitEvent(d->render, d->gathered, 0);
      d->result = iamf_hip_batch_render(d->batch, in, in_stream_stride, in_frame_stride, n_frames, pcm, pcm_stream_stride_bytes, d->render);
      if (d->result >= 0 && hipEventRecord(d->rendered, d->render) != hipSuccess) d->result = IAMF_HIP_ERR_DEVICE;
    });
  }
  int r = 0;
  for (int i = 0; i < n; ++i) {
    s->devs[i]->w.wait();
    if (s->devs[i]->result < 0) return s->devs[i]->result;
    r = s->devs[i]->result;   // the same on every device: the batches advance together
  }
  return r;
}

int iamf_hip_shard_flush(iamf_hip_shard *s, void *const *d_pcm, int64_t pcm_stream_stride_bytes) {
  if (!s || !d_pcm) return IAMF_HIP_ERR_BAD_ARG;
  const int n = (int)s->devs.size();
  for (int i = 0; i < n; ++i)   // checked before anything is submitted: no device is left flushed while another is not
    if (!d_pcm[i]) return IAMF_HIP_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    Dev *d = s->devs[i];
    void *pcm = d_pcm[i];
    d->w.submit([=] {
      (void)hipStreamWaitEvent(d->render, d->gathered, 0);
      d->result = iamf_hip_batch_flush(d->batch, pcm, pcm_stream_stride_bytes, d->render);
      if (d->result >= 0 && hipEventRecord(d->rendered, d->render) != hipSuccess) d->result = IAMF_HIP_ERR_DEVICE;
    });
  }
  int r = 0;
  for (int i = 0; i < n; ++i) {
    s->devs[i]->w.wait();
    if (s->devs[i]->result < 0) return s->devs[i]->result;
    r = s->devs[i]->result;
  }
  return r;
}

// Every shard's packed PCM -> d_dst on device `root_index`, stream s of the whole job at d_dst + s * dst_stream_stride_bytes.
// Only the ROWS travel: bytes_per_stream bytes of each stream's region (what the last render / flush emitted: samples x
// channels x sample bytes), not the regions' padding — a flush sends 240 sample-frames per stream, not a 64-frame region.
// A device whose rows are not back to back (stride != bytes_per_stream) packs them into a staging buffer on its gather
// stream first (hipMemcpy2DAsync, device to device), the root spreads what it received the same way if its destination is
// strided.  Asynchronous: ordered behind the devices' last render / flush on their gather streams; iamf_hip_shard_sync (or
// the next render's wait on `gathered`) completes it.
int iamf_hip_shard_gather_rows(iamf_hip_shard *s, int root_index, void *d_dst, int64_t dst_stream_stride_bytes,
                               void *const *d_pcm, int64_t pcm_stream_stride_bytes, int64_t bytes_per_stream) {
  if (!s || !d_dst || !d_pcm || root_index < 0 || root_index >= (int)s->devs.size()) return IAMF_HIP_ERR_BAD_ARG;
  if (bytes_per_stream <= 0 || pcm_stream_stride_bytes < bytes_per_stream || dst_stream_stride_bytes < bytes_per_stream)
    return IAMF_HIP_ERR_BAD_ARG;
  const int n = (int)s->devs.size();
  for (int i = 0; i < n; ++i)   // checked before any device gets work
    if (!d_pcm[i]) return IAMF_HIP_ERR_BAD_ARG;
  Rccl &R = rccl();
  if (!R.ok) return IAMF_HIP_ERR_UNIMPLEMENTED;   // no RCCL on this host
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (!s->comms) {   // communicators on first use: one process, all the shard's devices (ncclCommInitAll)
    std::vector<ncclComm_t> comms((size_t)n);
    std::vector<int> list((size_t)n);
    for (int i = 0; i < n; ++i) list[(size_t)i] = s->devs[i]->device;
    const ncclResult_t rc = R.CommInitAll(comms.data(), n, list.data());
    if (rc != ncclSuccess) {
      fprintf(stderr, "iamf_hip: ncclCommInitAll failed: %s\n", R.GetErrorString ? R.GetErrorString(rc) : "?");
      if (cur >= 0) (void)hipSetDevice(cur);
      return IAMF_HIP_ERR_DEVICE;
    }
    for (int i = 0; i < n; ++i) s->devs[i]->comm = comms[(size_t)i];
    s->comms = true;
  }
  Dev *root = s->devs[root_index];
  const bool pack_src = pcm_stream_stride_bytes != bytes_per_stream;
  const bool spread_dst = dst_stream_stride_bytes != bytes_per_stream;
  const size_t row = (size_t)bytes_per_stream;
  int err = 0;
  std::vector<const void *> wire((size_t)n);   // what device i sends: its rows back to back
  for (int i = 0; i < n && !err; ++i) {
    // the gather streams wait for the renders that produce what they send; the books of the previous gather are closed
    Dev *d = s->devs[i];
    if (hipSetDevice(d->device) != hipSuccess) {
      err = 1;
      break;
    }
    settle_times(d, false);
    const int pair = next_pair(d);
    if (hipStreamWaitEvent(d->gather, d->rendered, 0) != hipSuccess || hipEventRecord(d->g_begin[pair], d->gather) != hipSuccess) err = 1;
    wire[(size_t)i] = d_pcm[i];
    if (!err && pack_src && d->count > 1) {
      if (stage_room(d, &d->pack, &d->pack_bytes, row * (size_t)d->count) ||
          hipMemcpy2DAsync(d->pack, row, d_pcm[i], (size_t)pcm_stream_stride_bytes, row, (size_t)d->count, hipMemcpyDeviceToDevice,
                           d->gather) != hipSuccess)
        err = 1;
      wire[(size_t)i] = d->pack;
    }
  }
  char *landing = static_cast<char *>(d_dst);   // where the root receives: the destination itself if its rows are back to back
  if (!err && spread_dst) {
    if (hipSetDevice(root->device) != hipSuccess || stage_room(root, &root->unpack, &root->unpack_bytes, row * (size_t)s->n_streams))
      err = 1;
    landing = static_cast<char *>(root->unpack);
  }
  bool group = false;
  if (!err) {
    group = R.GroupStart() == ncclSuccess;
    if (!group) err = 1;
  }
  for (int i = 0; i < n && !err; ++i) {
    Dev *d = s->devs[i];
    const size_t bytes = (size_t)d->count * row;
    if (R.Send(wire[(size_t)i], bytes, ncclUint8, root_index, d->comm, d->gather) != ncclSuccess) err = 1;
    if (R.Recv(landing + (size_t)d->first * row, bytes, ncclUint8, i, root->comm, root->gather) != ncclSuccess) err = 1;
  }
  if (group && R.GroupEnd() != ncclSuccess) err = 1;   // a started group is always ended
  if (!err && spread_dst) {
    if (hipSetDevice(root->device) != hipSuccess ||
        hipMemcpy2DAsync(d_dst, (size_t)dst_stream_stride_bytes, landing, row, row, (size_t)s->n_streams, hipMemcpyDeviceToDevice,
                         root->gather) != hipSuccess)
      err = 1;
  }
  for (int i = 0; i < n; ++i) {   // always: a render that waits on `gathered` must find the latest state of the stream
    Dev *d = s->devs[i];
    if (hipSetDevice(d->device) != hipSuccess || hipEventRecord(d->g_end[d->g_cur], d->gather) != hipSuccess ||
        hipEventRecord(d->gathered, d->gather) != hipSuccess)
      err = 1;
    if (!err) {
      d->last_sent = (int64_t)d->count * bytes_per_stream;
      d->total_sent += d->last_sent;
      d->last_received = d == root ? (int64_t)s->n_streams * bytes_per_stream : 0;
      d->pending[d->g_cur] = true;
    }
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  return err ? IAMF_HIP_ERR_DEVICE : IAMF_HIP_OK;
}

// the whole regions (stride bytes per stream, padding included): source and destination strides must agree
int iamf_hip_shard_gather(iamf_hip_shard *s, int root_index, void *d_dst, int64_t dst_stream_stride_bytes,
                          void *const *d_pcm, int64_t pcm_stream_stride_bytes) {
  if (dst_stream_stride_bytes != pcm_stream_stride_bytes) return IAMF_HIP_ERR_BAD_ARG;
  return iamf_hip_shard_gather_rows(s, root_index, d_dst, dst_stream_stride_bytes, d_pcm, pcm_stream_stride_bytes, pcm_stream_stride_bytes);
}

// What the gather cost device `index`: bytes it sent in the last gather and in all, bytes it received in the last one (the
// root: every device's rows, its own included), and the time its gather stream spent between the last render's completion
// and the end of its share (pack, send / receive, spread) — last gather and all of them.  Waits for that device's last gather.
int iamf_hip_shard_times(iamf_hip_shard *s, int index, int64_t *last_sent_bytes, int64_t *total_sent_bytes,
                         int64_t *last_received_bytes, double *last_gather_ms, double *total_gather_ms) {
  if (!s || index < 0 || index >= (int)s->devs.size()) return IAMF_HIP_ERR_BAD_ARG;
  Dev *d = s->devs[index];
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (hipSetDevice(d->device) != hipSuccess) return IAMF_HIP_ERR_DEVICE;
  int rc = IAMF_HIP_OK;
  settle_times(d, true);
  const double ms = d->last_gather_ms;
  if (cur >= 0) (void)hipSetDevice(cur);
  if (last_sent_bytes) *last_sent_bytes = d->last_sent;
  if (total_sent_bytes) *total_sent_bytes = d->total_sent;
  if (last_received_bytes) *last_received_bytes = d->last_received;
  if (last_gather_ms) *last_gather_ms = ms;
  if (total_gather_ms) *total_gather_ms = d->total_gather_ms;
  return rc;
}

int iamf_hip_shard_sync(iamf_hip_shard *s) {
  if (!s) return IAMF_HIP_ERR_BAD_ARG;
  int cur = -1, err = 0;
  (void)hipGetDevice(&cur);
  for (Dev *d : s->devs) {
    d->w.wait();
    if (hipSetDevice(d->device) != hipSuccess || hipStreamSynchronize(d->render) != hipSuccess ||
        hipStreamSynchronize(d->gather) != hipSuccess)
      err = 1;
  }
  if (cur >= 0) (void)hipSetDevice(cur);
  return err ? IAMF_HIP_ERR_DEVICE : IAMF_HIP_OK;
}

}  // extern "C"
