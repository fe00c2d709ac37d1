/*
 * TEST INFRASTRUCTURE: drives iamf_hip_shard_* (iac_amd/csrc/iamf_shard.hip, compiled as host C++) against the N-device HIP
 * stand-in (fake_hip.cpp) and the RCCL stand-in (fake_rccl.cpp, loaded by the shard through IAMF_HIP_RCCL_LIB).
 *   shard_driver <streams> <devices> <root> <steps> <mode> [device ordinals, comma separated]
 * modes: rows   render -> gather_rows per step WITHOUT waiting in between (step i's gather beside step i + 1's render, one
 *               PCM buffer per device re-used by every step), then flush -> gather, then check every byte of every step's
 *               destination; even steps gather into a dense destination, odd steps into a strided one
 *        whole  the same through iamf_hip_shard_gather (whole regions, equal strides)
 *        destroy  render, gather, destroy at once (a gather in flight)
 *        fail   a peer's ncclSend fails in the first gather (FAKE_RCCL_FAIL_SEND_RANK): the call reports it, the shard stays
 *               usable, the next step's gather is complete and correct
 * Prints "ok ..." lines; any mismatch prints "MISMATCH ..." and exits 1.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "iamf_hip.h"

extern "C" {
void fake_hip_set_devices(int n);
void fake_hip_set_op_delay_us(int us);
long fake_hip_live_objects(void);
}

static uint8_t in_byte(int step, int stream, int64_t k) { return (uint8_t)(step * 131 + stream * 29 + k * 7 + (k >> 8)); }
static uint8_t want_byte(uint8_t in, unsigned call) { return (uint8_t)(in ^ (uint8_t)(0x5a + 17 * call)); }

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  const int S = atoi(argv[1]), N = atoi(argv[2]), root = atoi(argv[3]), steps = atoi(argv[4]);
  const std::string mode = argv[5];
  std::vector<int> ordinals;
  if (argc > 6) {
    for (char *tok = strtok(argv[6], ","); tok; tok = strtok(nullptr, ",")) ordinals.push_back(atoi(tok));
    if ((int)ordinals.size() != N) return 2;
  }
  fake_hip_set_devices(8);
  fake_hip_set_op_delay_us(150);
  const int fs = 64, frames = 6, ch = 2;
  iamf_hip_batch_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n_streams = S;
  cfg.frame_size = fs;
  cfg.out_channels = ch;
  cfg.out_format = IAMF_HIP_FMT_S16;
  cfg.limiter_enable = 1;
  iamf_hip_shard *sh = nullptr;
  int rc = iamf_hip_shard_create(&cfg, ordinals.empty() ? nullptr : ordinals.data(), N, &sh);
  printf("create %d devices %d\n", rc, iamf_hip_shard_devices(sh));
  if (rc) return rc == IAMF_HIP_ERR_BAD_ARG ? 0 : 1;
  std::vector<int> first(N), count(N);
  int covered = 0;
  for (int i = 0; i < N; ++i) {
    int dev = -1;
    if (iamf_hip_shard_info(sh, i, &dev, &first[i], &count[i])) return 1;
    if (first[i] != covered || dev != (ordinals.empty() ? i : ordinals[i])) {
      printf("MISMATCH split: device %d ordinal %d first %d (want %d)\n", i, dev, first[i], covered);
      return 1;
    }
    covered += count[i];
    int f2, c2;
    iamf_hip_shard_split(S, N, i, &f2, &c2);
    if (f2 != first[i] || c2 != count[i]) return 1;
  }
  if (covered != S) {
    printf("MISMATCH split covers %d of %d\n", covered, S);
    return 1;
  }
  const int64_t full = (int64_t)frames * fs * ch * 2;      // bytes a full call emits per stream
  const int64_t pcm_stride = full + 256;                   // regions with padding
  const int64_t in_row_floats = (full + 3) / 4 + 8;
  // "device" buffers (the stand-in's memory is the host's): one PCM buffer per device for all steps, inputs per step
  std::vector<void *> pcm(N);
  for (int i = 0; i < N; ++i) pcm[i] = calloc((size_t)count[i], (size_t)pcm_stride);
  std::vector<std::vector<float *>> in(steps, std::vector<float *>(N));
  for (int t = 0; t < steps; ++t)
    for (int i = 0; i < N; ++i) {
      in[t][i] = (float *)calloc((size_t)count[i] * in_row_floats, 4);
      for (int j = 0; j < count[i]; ++j) {
        uint8_t *row = (uint8_t *)(in[t][i] + (int64_t)j * in_row_floats);
        for (int64_t k = 0; k < full; ++k) row[k] = in_byte(t, first[i] + j, k);
      }
    }
  struct Dst {
    uint8_t *p;
    int64_t stride, row;
    int step;   // -1: the flush
  };
  std::vector<Dst> dsts;
  int failures_seen = 0;
  const bool whole = mode == "whole";
  for (int t = 0; t < steps; ++t) {
    std::vector<const float *> ins(N);
    for (int i = 0; i < N; ++i) ins[i] = in[t][i];
    const int r = iamf_hip_shard_render(sh, ins.data(), in_row_floats, in_row_floats, frames, pcm.data(), pcm_stride);
    if (r != frames * fs - (t == 0 ? 240 : 0)) {
      printf("MISMATCH render step %d returned %d\n", t, r);
      return 1;
    }
    Dst d;
    d.row = (int64_t)r * ch * 2;
    d.stride = whole ? pcm_stride : ((t & 1) ? d.row + 192 : d.row);
    d.p = (uint8_t *)malloc((size_t)(d.stride * S));
    memset(d.p, 0xEE, (size_t)(d.stride * S));
    d.step = t;
    const int g = whole ? iamf_hip_shard_gather(sh, root, d.p, d.stride, pcm.data(), pcm_stride)
                        : iamf_hip_shard_gather_rows(sh, root, d.p, d.stride, pcm.data(), pcm_stride, d.row);
    if (mode == "fail" && t == 0) {
      printf("gather step 0 under a failing peer: %d\n", g);
      if (g != IAMF_HIP_ERR_DEVICE) return 1;
      ++failures_seen;
      free(d.p);
      continue;
    }
    if (g != IAMF_HIP_OK) {
      printf("MISMATCH gather step %d returned %d\n", t, g);
      return 1;
    }
    dsts.push_back(d);
    if (mode == "destroy" && t == steps - 1) {   // a gather in flight: destroy must wait for it and leave nothing behind
      iamf_hip_shard_destroy(sh);
      sh = nullptr;
      break;
    }
  }
  if (sh) {
    const int r = iamf_hip_shard_flush(sh, pcm.data(), pcm_stride);
    if (r != 240) {
      printf("MISMATCH flush returned %d\n", r);
      return 1;
    }
    Dst d;
    d.row = (int64_t)r * ch * 2;
    d.stride = whole ? pcm_stride : d.row;
    d.p = (uint8_t *)malloc((size_t)(d.stride * S));
    memset(d.p, 0xEE, (size_t)(d.stride * S));
    d.step = -1;
    const int g = whole ? iamf_hip_shard_gather(sh, root, d.p, d.stride, pcm.data(), pcm_stride)
                        : iamf_hip_shard_gather_rows(sh, root, d.p, d.stride, pcm.data(), pcm_stride, d.row);
    if (g != IAMF_HIP_OK) return 1;
    dsts.push_back(d);
    if (iamf_hip_shard_sync(sh)) return 1;
    int64_t sent_all = 0;
    for (int i = 0; i < N; ++i) {
      int64_t ls = 0, ts = 0, lr = 0;
      double lms = 0, tms = 0;
      if (iamf_hip_shard_times(sh, i, &ls, &ts, &lr, &lms, &tms)) return 1;
      printf("device %d: last sent %lld total sent %lld last received %lld gather ms %.3f / %.3f\n", i, (long long)ls, (long long)ts,
             (long long)lr, lms, tms);
      if (ls != (whole ? pcm_stride : d.row) * count[i] || lr != (i == root ? (whole ? pcm_stride : d.row) * S : 0) || tms < lms) {
        printf("MISMATCH accounting of device %d\n", i);
        return 1;
      }
      sent_all += ts;
    }
    int64_t want_all = 0;
    for (const Dst &q : dsts) want_all += (whole ? pcm_stride : q.row) * S;
    if (sent_all != want_all) {
      printf("MISMATCH bytes on the wire %lld, want %lld\n", (long long)sent_all, (long long)want_all);
      return 1;
    }
  }
  // every byte of every destination: rows from the right device, step and stream; nothing written between the rows
  for (const Dst &d : dsts) {
    // which call of its batch a destination belongs to: the flush follows `steps` renders
    const unsigned call = d.step >= 0 ? (unsigned)d.step : (unsigned)steps;
    for (int s = 0; s < S; ++s) {
      const uint8_t *row = d.p + (int64_t)s * d.stride;
      int local = 0;
      for (int i = 0; i < N; ++i)
        if (s >= first[i] && s < first[i] + count[i]) local = s - first[i];
      for (int64_t k = 0; k < d.row; ++k) {
        const uint8_t in = d.step >= 0 ? in_byte(d.step, s, k) : (uint8_t)(local + k);
        if (row[k] != want_byte(in, call)) {
          printf("MISMATCH step %d stream %d byte %lld: %02x, want %02x\n", d.step, s, (long long)k, row[k], want_byte(in, call));
          return 1;
        }
      }
      if (!whole)
        for (int64_t k = d.row; k < d.stride; ++k)
          if (row[k] != 0xEE) {
            printf("MISMATCH step %d stream %d: padding byte %lld written\n", d.step, s, (long long)k);
            return 1;
          }
    }
  }
  printf("ok %zu destinations, %d streams over %d devices, root %d, failures seen %d\n", dsts.size(), S, N, root, failures_seen);
  if (sh) iamf_hip_shard_destroy(sh);
  for (const Dst &d : dsts) free(d.p);
  for (int i = 0; i < N; ++i) free(pcm[i]);
  for (auto &v : in)
    for (float *p : v) free(p);
  if (fake_hip_live_objects() != 0) {
    printf("MISMATCH %ld streams / events / allocations left behind\n", fake_hip_live_objects());
    return 1;
  }
  printf("clean\n");
  return 0;
}
