#!/usr/bin/env python3
"""Throughput of the resampler kernels (iamf_resample.hip): S streams x ch channels x ns input samples
per call, interleaved f32 in HBM; the register-blocked kernel against the tiled one (IAMF_HIP_RESAMPLE_TILE=1).
   python tools/debug/resample_probe.py [in_rate out_rate [channels [streams]]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import iac_amd as A


def rate(in_rate, out_rate, ch, S, ns=16384, reps=10):
    r = A.Resampler(S, ch, in_rate, out_rate)
    cap = r.out_capacity(ns)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn((S, ns * ch), device="cuda", dtype=torch.float32, generator=gen) * 0.2).contiguous()
    y = torch.zeros((S, cap * ch), device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        n = r.process(x.data_ptr(), ns * ch, ns, y.data_ptr(), cap * ch, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        n = r.process(x.data_ptr(), ns * ch, ns, y.data_ptr(), cap * ch, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    r.close()
    return S * n / dt / 1e9, S * (ns + n) * ch * 4 / dt / 1e9, dt * 1e3, y


def main():
    pairs = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(44100, 48000), (48000, 44100), (96000, 48000), (48000, 16000), (32000, 48000), (48000, 32000), (16000, 48000)]
    chans = [int(sys.argv[3])] if len(sys.argv) > 3 else [2, 6, 8, 12]
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    for a, b in pairs:
        for ch in chans:
            os.environ.pop("IAMF_HIP_RESAMPLE_TILE", None)
            g1, bw1, ms1, y1 = rate(a, b, ch, S)
            os.environ["IAMF_HIP_RESAMPLE_TILE"] = "1"
            g0, bw0, ms0, y0 = rate(a, b, ch, S)
            os.environ.pop("IAMF_HIP_RESAMPLE_TILE", None)
            print("%6d -> %6d Hz, %2d ch, %d streams: blocked %.3f ms = %6.2f G output sample-frames/s (%4.0f GB/s) | tiled %.3f ms = %6.2f | "
                  "x%.2f | equal bits: %s" % (a, b, ch, S, ms1, g1, bw1, ms0, g0, g1 / g0, bool(torch.equal(y1, y0))), flush=True)


if __name__ == "__main__":
    main()
