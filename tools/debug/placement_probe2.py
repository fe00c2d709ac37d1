"""Which buffer's placement makes the headline bimodal (tools/debug/placement_probe.py: ~79 or ~90 Gsamples/s quiet
for the same stride and the same low address bits)?  Re-allocate ONE of {input, PCM, batch state} at a time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import iac_amd as A  # noqa: E402

dev = torch.device("cuda", 0)
S, F, fs, M = 512, 64, 1024, 16
mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
st = torch.cuda.current_stream().cuda_stream
n = F * M * fs
stride = n + 1024


def rate(b, x, pcm, steps=10):
    for _ in range(2):
        b.render(x.data_ptr(), stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
    b.reset()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        ev[i][0].record()
        b.render(x.data_ptr(), stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
        ev[i][1].record()
    torch.cuda.synchronize()
    b.reset()
    ms = np.median([a.elapsed_time(c) for a, c in ev][2:])
    return S * F * fs / (ms * 1e-3) / 1e9


def new_x():
    x = torch.randn((S, stride), device=dev) * 0.05
    return x


def new_pcm():
    return torch.zeros((S, F * fs * 4), dtype=torch.uint8, device=dev)


x, pcm, b = new_x(), new_pcm(), A.Batch(S, mx, 2, frame_size=fs)
print("same buffers, 5 measurements:", ["%.1f" % rate(b, x, pcm) for _ in range(5)], flush=True)
out = []
for i in range(8):
    b.close()
    b = A.Batch(S, mx, 2, frame_size=fs)
    out.append("%.1f" % rate(b, x, pcm))
print("new batch state each time:   ", out, flush=True)
out = []
keep = []
for i in range(8):
    keep.append(pcm)            # keep the old one alive so that the new one lands elsewhere
    pcm = new_pcm()
    out.append("%.1f@%x" % (rate(b, x, pcm), pcm.data_ptr() >> 21))
print("new PCM buffer each time:    ", out, flush=True)
del keep
out = []
keep = []
for i in range(8):
    keep.append(x)
    x = new_x()
    out.append("%.1f@%x" % (rate(b, x, pcm), x.data_ptr() >> 21))
    if len(keep) > 3:
        keep.pop(0)
print("new input buffer each time:  ", out, flush=True)
# same input buffer, shifted start inside a larger allocation (2 MiB steps and 64 KiB steps)
big = torch.randn((S * stride + (64 << 20) // 4,), device=dev) * 0.05
for step_b, label in ((2 << 20, "2 MiB"), (64 << 10, "64 KiB"), (4 << 10, "4 KiB")):
    out = []
    for k in range(8):
        off = k * step_b // 4
        xv = big[off:off + S * stride].view(S, stride)
        out.append("%.1f" % rate(b, xv, pcm))
    print("input shifted by k x %-6s:  " % label, out, flush=True)
