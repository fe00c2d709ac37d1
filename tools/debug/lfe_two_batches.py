"""One-off: the LFE-on workload as ONE batch of 1024 streams against TWO batches of 512 on two HIP streams (does the
serial recurrence kernel of one half run beside the render kernel of the other?).  python tools/debug/lfe_two_batches.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iac_amd as A

dev = torch.device("cuda:0")
F, fs, ch, out_id = 64, 1024, 16, A.SS["B"]
mx = A.get_h2m_matrix(3, out_id)
oc = A.layout_channels(out_id)


def make(S, stream):
    b = A.Batch(S, mx, oc, frame_size=fs, out_format=A.FMT_S16, limiter=True, lfe_hoa=True)
    x = (torch.randn((S, F * ch * fs + 1024), device=dev) * 0.25).contiguous()
    pcm = [torch.zeros((S, F * fs * oc * 2), dtype=torch.uint8, device=dev) for _ in range(2)]
    return b, x, pcm, stream


def run(parts, steps):
    for i in range(steps):
        for b, x, pcm, st in parts:
            b.render(x.data_ptr(), F * ch * fs + 1024, ch * fs, F, pcm[i & 1].data_ptr(), F * fs * oc * 2, st.cuda_stream)


for label, sizes in (("one batch of 1024", [1024]), ("two batches of 512 on two streams", [512, 512]),
                     ("four batches of 256 on four streams", [256] * 4)):
    parts = [make(S, torch.cuda.Stream(device=dev)) for S in sizes]
    run(parts, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(parts, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-40s %.3f ms per step of %d streams  %.1f Gsamples/s" % (label, dt * 1e3, sum(sizes), sum(sizes) * F * fs / dt / 1e9))
    for b, *_ in parts:
        b.close()
