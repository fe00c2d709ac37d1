#!/usr/bin/env python3
"""Throughput of the resampler kernel (iamf_resample.hip): S streams x ch channels x ns input samples
per call, interleaved f32 in HBM.   python tools/debug/resample_probe.py [in_rate out_rate]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import iac_amd as A


def main():
    in_rate = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
    out_rate = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
    S, ch, ns = 512, 2, 16384
    r = A.Resampler(S, ch, in_rate, out_rate)
    cap = r.out_capacity(ns)
    x = (torch.randn((S, ns * ch), device="cuda", dtype=torch.float32) * 0.2).contiguous()
    y = torch.zeros((S, cap * ch), device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        n = r.process(x.data_ptr(), ns * ch, ns, y.data_ptr(), cap * ch, st)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        n = r.process(x.data_ptr(), ns * ch, ns, y.data_ptr(), cap * ch, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out_sf = S * n
    byts = S * (ns + n) * ch * 4
    print("%d -> %d Hz: %d streams x %d ch x %d in -> %d out per stream: %.3f ms, %.2f G output sample-frames/s, "
          "%.0f GB/s of %d B per output sample-frame (read + write)" % (in_rate, out_rate, S, ch, ns, n, dt * 1e3, out_sf / dt / 1e9,
                                                                       byts / dt / 1e9, byts // out_sf))
    r.close()


if __name__ == "__main__":
    main()
