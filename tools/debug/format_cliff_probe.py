#!/usr/bin/env python3
"""What the PCM format costs per layout: S streams x 64 frames x 1024, hot-ish noise, s16 / s24 / s32 / f32 output.
   python tools/debug/format_cliff_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import iac_amd as A  # noqa: E402


def main():
    fs, F = 1024, 64
    st = torch.cuda.current_stream().cuda_stream
    cases = (("TOA -> binaural", A.get_h2m_matrix(3, A.SS["A"]), 16, 2, 512),
             ("7.1.4 -> J", A.get_m2m_matrix(A.SS["L714"], A.SS["J"]), 12, 12, 1024),
             ("TOA -> 22.2", A.get_h2m_matrix(3, A.SS["H"]), 16, 24, 1024),
             ("5.1 -> 5.1", A.get_m2m_matrix(A.SS["L51"], A.SS["B"]), 6, 6, 1024))
    for name, mx, m, oc, S in cases:
        gen = torch.Generator(device="cuda").manual_seed(5)
        x = (torch.randn((S, F * m * fs), device="cuda", generator=gen) * 0.2).contiguous()
        row = []
        for fmt, nm, b in ((A.FMT_S16, "s16", 2), (A.FMT_S24, "s24", 3), (A.FMT_S32, "s32", 4), (A.FMT_F32, "f32", 4)):
            bt = A.Batch(S, mx, oc, frame_size=fs, out_format=fmt, limiter=True)
            pcm = torch.zeros((S, F * fs * oc * b), dtype=torch.uint8, device="cuda")
            ts = []
            for i in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                bt.render(x.data_ptr(), F * m * fs, m * fs, F, pcm.data_ptr(), F * fs * oc * b, st)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            bt.close()
            row.append("%s %.1f" % (nm, S * F * fs / min(ts[1:]) / 1e6))
        print("%-16s %4d streams: Gsamples/s  %s" % (name, S, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
