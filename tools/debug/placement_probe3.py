"""Follow-up to placement_probe2: (1) does a plain copy of the same buffer see the slow / fast split?
(2) slices of ONE large early arena at 2.2 GB steps."""
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


def rate(b, x, pcm, steps=8):
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


def read_gbs(x, steps=8):
    """a streaming READ of the buffer (sum reduction, writes nothing to speak of)"""
    for _ in range(2):
        x.sum()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        ev[i][0].record()
        x.sum()
        ev[i][1].record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(c) for a, c in ev][2:])
    return x.numel() * 4 / (ms * 1e-3) / 1e9


arena = torch.empty((12 * S * stride,), dtype=torch.float32, device=dev)   # 26 GB, the process's first allocation
pcm = torch.zeros((S, F * fs * 4), dtype=torch.uint8, device=dev)
b = A.Batch(S, mx, 2, frame_size=fs)
out = []
for k in range(12):
    xv = arena[k * S * stride:(k + 1) * S * stride].view(S, stride)
    xv.normal_(0, 0.05)
    out.append("%.1f" % rate(b, xv, pcm))
print("slices of one early 26 GB arena:", out, flush=True)
keep, res = [], []
for i in range(14):
    x = torch.randn((S, stride), device=dev) * 0.05
    r = rate(b, x, pcm)
    res.append("%.1f render / %.0f GB/s read @%x" % (r, read_gbs(x), x.data_ptr() >> 21))
    keep.append(x)
    if len(keep) > 3:
        keep.pop(0)
print("fresh allocations:")
for r in res:
    print("   ", r, flush=True)
