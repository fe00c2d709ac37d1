#!/usr/bin/env python3
"""The three launches of a rate-converting stream (IAMF_decoder.c:3459-3500: render -> iamf_resample -> limiter + pack) timed
one by one: S streams, F frames of 1024 samples at 44.1 kHz per call -> 48 kHz s16.
   python tools/debug/resample_pipeline_probe.py [streams [frames]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import iac_amd as A  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    fs = 1024
    st = torch.cuda.current_stream().cuda_stream
    for name, mx, m in (("stereo -> A", A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"]), 2),
                        ("TOA -> binaural", A.get_h2m_matrix(3, A.SS["BINAURAL"]) if "BINAURAL" in A.SS else A.get_h2m_matrix(3, A.SS["A"]), 16)):
        gen = torch.Generator(device="cuda").manual_seed(3)
        x = (torch.randn((S, F * m * fs), device="cuda", generator=gen) * 0.2).contiguous()
        s1 = A.Batch(S, mx, 2, frame_size=fs, sample_rate=44100, out_format=A.FMT_F32, limiter=False)
        rs = A.Resampler(S, 2, 44100, 48000)
        import ctypes as C
        eye = np.eye(2, dtype=np.float32)
        ident = A.Matrix()
        ident.kind, ident.in_id, ident.out_id, ident.channels, ident.lfe1, ident.lfe2, ident.m, ident.n = A.KIND_M2M, 0, 0, 2, -1, -1, 2, 2
        ident.mat = eye.ctypes.data_as(C.POINTER(C.c_float))
        s3 = A.Batch(S, ident, 2, frame_size=1, sample_rate=48000, out_format=A.FMT_S16, limiter=True)
        ns = F * fs
        mid = torch.zeros((S, ns * 2), dtype=torch.float32, device="cuda")
        cap = rs.out_capacity(ns)
        res = torch.zeros((S, cap * 2), dtype=torch.float32, device="cuda")
        pcm = torch.zeros((S, cap * 2), dtype=torch.int16, device="cuda")
        n2 = [0]
        t1 = timed(lambda: s1.render(x.data_ptr(), F * m * fs, m * fs, F, mid.data_ptr(), ns * 2 * 4, st))
        t2 = timed(lambda: n2.__setitem__(0, rs.process(mid.data_ptr(), ns * 2, ns, res.data_ptr(), cap * 2, st)))
        t3 = timed(lambda: s3.render(res.data_ptr(), cap * 2, 2, n2[0], pcm.data_ptr(), cap * 2 * 2, st))
        sf = S * ns
        print("%-16s %d streams x %d frames: render(f32, limiter off) %.3f ms = %.1f G in-sf/s | resample %.3f ms = %.1f G out-sf/s | "
              "limiter + pack (frame size 1, %d frames) %.3f ms = %.1f G sf/s | all three: %.1f G input sample-frames/s"
              % (name, S, F, t1, sf / t1 / 1e6, t2, S * n2[0] / t2 / 1e6, n2[0], t3, S * n2[0] / t3 / 1e6, sf / (t1 + t2 + t3) / 1e6), flush=True)
        s1.close(); rs.close(); s3.close()


if __name__ == "__main__":
    main()
