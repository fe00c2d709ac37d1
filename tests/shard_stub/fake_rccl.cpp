/*
 * TEST INFRASTRUCTURE — an RCCL stand-in for the N-device rehearsal of iac_amd/csrc/iamf_shard.hip on a host without GPUs
 * (loaded through IAMF_HIP_RCCL_LIB; see fake_hip.cpp).  Only what the shard uses: ncclGetVersion, ncclCommInitAll,
 * ncclCommDestroy, ncclGroupStart / End, ncclSend / ncclRecv, ncclGetErrorString.
 *
 * Semantics kept from the real library because the shard relies on them:
 *   - send / recv are only legal between GroupStart and GroupEnd here (the shard always groups them), and nothing is
 *     issued before GroupEnd;
 *   - at GroupEnd every send must meet the recv its peer posted for it, with the same byte count (else ncclInvalidUsage);
 *   - all operations of one group that use one stream form ONE unit of work on that stream (RCCL fuses them into one
 *     kernel): a rank that both sends to and receives from itself, or the root that receives from everybody while it
 *     sends to itself, cannot deadlock on its own queue order;
 *   - the copy happens when BOTH streams have reached the unit, asynchronously to the host.
 * Failure injection (environment): FAKE_RCCL_FAIL_INIT=1; FAKE_RCCL_FAIL_SEND_RANK=r with FAKE_RCCL_FAIL_AT_GROUP=g
 * (ncclSend of rank r fails in the g-th group, counted from 0): the group is abandoned, as a failed RCCL group is.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

extern "C" void fake_stream_enqueue(hipStream_t s, void (*fn)(void *), void *arg);
extern "C" int fake_stream_device(hipStream_t s);

struct ncclComm {
  int rank, nranks, device;
  int world;   // id of the ncclCommInitAll call
};

namespace {

struct Mailbox {
  std::mutex mu;
  std::condition_variable cv;
  const void *src = nullptr;
  void *dst = nullptr;
  size_t bytes = 0;
  bool ready = false, done = false;
};

struct Op {
  bool send;
  const void *sbuf;
  void *rbuf;
  size_t bytes;
  int peer;
  ncclComm *comm;
  hipStream_t stream;
};

struct Unit {   // the work of one group on one stream
  std::vector<std::pair<std::shared_ptr<Mailbox>, bool>> parts;   // (mailbox, this side sends)
};

thread_local int t_depth = 0;
thread_local bool t_failed = false;
thread_local std::vector<Op> t_ops;
int g_worlds = 0, g_groups = 0;
std::mutex g_mu;

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

void run_unit(void *arg) {
  std::unique_ptr<Unit> u(static_cast<Unit *>(arg));
  for (auto &p : u->parts)
    if (p.second) {   // my send buffers are final from here on
      std::lock_guard<std::mutex> lk(p.first->mu);
      p.first->ready = true;
      p.first->cv.notify_all();
    }
  for (auto &p : u->parts)
    if (!p.second) {
      Mailbox &m = *p.first;
      std::unique_lock<std::mutex> lk(m.mu);
      m.cv.wait(lk, [&m] { return m.ready; });
      lk.unlock();
      std::this_thread::sleep_for(std::chrono::microseconds(150));   // the wire
      memcpy(m.dst, m.src, m.bytes);
      lk.lock();
      m.done = true;
      m.cv.notify_all();
    }
  for (auto &p : u->parts)
    if (p.second) {   // a send completes when its data has left
      Mailbox &m = *p.first;
      std::unique_lock<std::mutex> lk(m.mu);
      m.cv.wait(lk, [&m] { return m.done; });
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *v) {
  *v = 29999;   // "2.99.99": recognisably not a real RCCL
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
  if (!comms || n <= 0) return ncclInvalidArgument;
  if (env_int("FAKE_RCCL_FAIL_INIT", 0)) return ncclSystemError;
  int ndev = 0;
  (void)hipGetDeviceCount(&ndev);
  for (int i = 0; i < n; ++i) {
    const int d = devs ? devs[i] : i;
    if (d < 0 || d >= ndev) return ncclInvalidArgument;
    for (int k = 0; k < i; ++k)
      if ((devs ? devs[k] : k) == d) return ncclInvalidArgument;
  }
  int world;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    world = g_worlds++;
  }
  for (int i = 0; i < n; ++i) comms[i] = new ncclComm{i, n, devs ? devs[i] : i, world};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
  if (t_depth++ == 0) {
    t_ops.clear();
    t_failed = false;
  }
  return ncclSuccess;
}

static ncclResult_t post(bool send, const void *sbuf, void *rbuf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                         hipStream_t stream) {
  if (t_depth <= 0) return ncclInvalidUsage;   // the shard always groups its point-to-point calls
  if (!comm || !stream || peer < 0 || peer >= comm->nranks || (!sbuf && !rbuf) || type != ncclUint8) {
    t_failed = true;
    return ncclInvalidArgument;
  }
  if (fake_stream_device(stream) != comm->device) {   // the stream must belong to the communicator's device
    t_failed = true;
    return ncclInvalidArgument;
  }
  int group;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    group = g_groups;
  }
  if (send && comm->rank == env_int("FAKE_RCCL_FAIL_SEND_RANK", -1) && group == env_int("FAKE_RCCL_FAIL_AT_GROUP", 0)) {
    t_failed = true;
    return ncclSystemError;
  }
  t_ops.push_back(Op{send, sbuf, rbuf, count, peer, comm, stream});
  return ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(true, buf, nullptr, count, type, peer, comm, stream);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(false, nullptr, buf, count, type, peer, comm, stream);
}

ncclResult_t ncclGroupEnd(void) {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_groups;
  }
  std::vector<Op> ops;
  ops.swap(t_ops);
  if (t_failed) return ncclSystemError;   // nothing of an abandoned group is issued
  // match: the k-th send a -> b of this group meets the k-th recv that b posted for a (one world per group)
  std::vector<bool> used(ops.size(), false);
  std::vector<std::pair<hipStream_t, Unit *>> units;
  auto unit_of = [&units](hipStream_t s) {
    for (auto &u : units)
      if (u.first == s) return u.second;
    units.emplace_back(s, new Unit());
    return units.back().second;
  };
  bool bad = false;
  for (size_t i = 0; i < ops.size() && !bad; ++i) {
    if (!ops[i].send) continue;
    size_t j = 0;
    for (; j < ops.size(); ++j)
      if (!used[j] && !ops[j].send && ops[j].comm->world == ops[i].comm->world && ops[j].comm->rank == ops[i].peer &&
          ops[j].peer == ops[i].comm->rank)
        break;
    if (j == ops.size() || ops[j].bytes != ops[i].bytes) {
      bad = true;
      break;
    }
    used[j] = used[i] = true;
    auto m = std::make_shared<Mailbox>();
    m->src = ops[i].sbuf;
    m->dst = ops[j].rbuf;
    m->bytes = ops[i].bytes;
    unit_of(ops[i].stream)->parts.emplace_back(m, true);
    unit_of(ops[j].stream)->parts.emplace_back(m, false);
  }
  for (size_t j = 0; j < ops.size() && !bad; ++j)
    if (!used[j]) bad = true;   // a recv nobody sends to: the real library would hang
  if (bad) {
    for (auto &u : units) delete u.second;
    return ncclInvalidUsage;
  }
  for (auto &u : units) fake_stream_enqueue(u.first, run_unit, u.second);
  return ncclSuccess;
}

}  // extern "C"
