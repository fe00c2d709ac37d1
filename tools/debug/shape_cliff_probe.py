#!/usr/bin/env python3
"""Where the fast kernels stop: Gsamples/s of one batch over frame sizes, frames per call, limiter on / off and layouts
(s16).  Every row renders the same number of samples per stream per call unless it says otherwise.
   python tools/debug/shape_cliff_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import iac_amd as A  # noqa: E402


def rate(mx, m, oc, S, fs, F, limiter=True, sample_rate=48000):
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn((S, F * m * fs), device="cuda", generator=gen) * 0.2).contiguous()
    bt = A.Batch(S, mx, oc, frame_size=fs, sample_rate=sample_rate, out_format=A.FMT_S16, limiter=limiter)
    pcm = torch.zeros((S, F * fs * oc * 2 + 64), dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bt.render(x.data_ptr(), F * m * fs, m * fs, F, pcm.data_ptr(), F * fs * oc * 2 + 64, st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    bt.close()
    return S * F * fs / min(ts[1:]) / 1e6


def main():
    lay = (("TOA -> binaural", A.get_h2m_matrix(3, A.SS["A"]), 16, 2, 512),
           ("stereo -> A", A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"]), 2, 2, 1024),
           ("7.1.4 -> J", A.get_m2m_matrix(A.SS["L714"], A.SS["J"]), 12, 12, 1024),
           ("7.1.4 -> E (11 ch)", A.get_m2m_matrix(A.SS["L714"], A.SS["E"]), 12, 11, 1024),
           ("TOA -> G (14 ch)", A.get_h2m_matrix(3, A.SS["G"]), 16, 14, 1024))
    total = 61440   # samples per stream per call: a multiple of every frame size below
    for name, mx, m, oc, S in lay:
        row = []
        for fs in (120, 128, 240, 256, 480, 512, 960, 1000, 1024, 2048, 4096):
            F = total // fs
            row.append("%d: %.1f" % (fs, rate(mx, m, oc, S, fs, F)))
        print("%-20s frame size: %s" % (name, " | ".join(row)), flush=True)
        print("%-20s limiter off (1024): %.1f | one 1024-frame per call: %.1f | four per call: %.1f | 44.1 kHz tables (1024): %.1f"
              % (name, rate(mx, m, oc, S, 1024, 60, limiter=False), rate(mx, m, oc, S, 1024, 1), rate(mx, m, oc, S, 1024, 4),
                 rate(mx, m, oc, S, 1024, 60, sample_rate=44100)), flush=True)


if __name__ == "__main__":
    main()
