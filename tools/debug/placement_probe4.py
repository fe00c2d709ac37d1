"""Inside ONE allocation that is in the slow mode (a large early arena, placement_probe3): is the mode a
property of the stream STRIDE?  Same base, many strides; then same stride, bases shifted by odd amounts."""
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


def rate(b, ptr, stride, pcm, steps=8):
    for _ in range(2):
        b.render(ptr, stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
    b.reset()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        ev[i][0].record()
        b.render(ptr, stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
        ev[i][1].record()
    torch.cuda.synchronize()
    b.reset()
    ms = np.median([a.elapsed_time(c) for a, c in ev][2:])
    return S * F * fs / (ms * 1e-3) / 1e9


arena = torch.empty((6 << 30) // 4 * 2, dtype=torch.float32, device=dev)   # 12 GB, first allocation
arena.normal_(0, 0.05)
pcm = torch.zeros((S, F * fs * 4), dtype=torch.uint8, device=dev)
b = A.Batch(S, mx, 2, frame_size=fs)
print("arena base %x" % arena.data_ptr(), flush=True)
for pad_b in (0, 64, 256, 1024, 4096, 4096 + 256, 3 * 4096, 5 * 4096, 7 * 4096, 16384, 17 * 4096, 65536, 65536 + 4096,
              33 * 4096, 1 << 20, (1 << 20) + 4096, (2 << 20) + 4096, (2 << 20) + 12288, 3 * (1 << 20) + 20480):
    stride = n + pad_b // 4
    print("stride 4 MiB + %8d B: %.1f" % (pad_b, rate(b, arena.data_ptr(), stride, pcm)), flush=True)
stride = n + 1024
for off_b in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 6 << 20, 32 << 20, (1 << 30) + (2 << 20)):
    print("base + %10d B, stride 4 MiB + 4 KiB: %.1f" % (off_b, rate(b, arena.data_ptr() + off_b, stride, pcm)), flush=True)
