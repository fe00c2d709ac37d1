"""Does the headline's rate depend on WHERE the input lies and on the stream stride?  One process, quiet
and hot programmes (512 streams x 64 frames x 1024), for several paddings of the stream stride and several
fresh allocations of the same size (a spacer allocation of varying size shifts the base address).
   python tools/debug/placement_probe.py > gpurun_out/placement.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import iac_amd as A  # noqa: E402

dev = torch.device("cuda", 0)
S, F, fs, M = 512, 64, 1024, 16
mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
st = torch.cuda.current_stream().cuda_stream


def run(x_flat, stride, steps=12):
    b = A.Batch(S, mx, 2, frame_size=fs)
    pcm = torch.zeros((S, F * fs * 4), dtype=torch.uint8, device=dev)
    for _ in range(3):
        b.render(x_flat.data_ptr(), stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
    b.close()
    b = A.Batch(S, mx, 2, frame_size=fs)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        ev[i][0].record()
        b.render(x_flat.data_ptr(), stride, M * fs, F, pcm.data_ptr(), F * fs * 4, st)
        ev[i][1].record()
    torch.cuda.synchronize()
    b.close()
    ms = np.array([a.elapsed_time(c) for a, c in ev][2:])
    return S * F * fs / (np.median(ms) * 1e-3) / 1e9, S * F * fs / (ms.max() * 1e-3) / 1e9


for sig in ("quiet", "hot"):
    base = bench.synth_hot_device(S, M, F, fs, 1000, dev) if sig == "hot" else torch.randn((S, F, M, fs), device=dev) * 0.05
    base = base.reshape(S, -1)
    n = base.shape[1]
    for pad_bytes in (0, 256, 1024, 4096, 4096 + 256, 8192, 16384, 65536, 65536 + 4096, 1 << 20):
        pad = pad_bytes // 4
        res = []
        for spacer_mb in (0, 3, 17):
            sp = torch.empty(spacer_mb * 262144 + 1, dtype=torch.float32, device=dev)   # shifts the next allocation
            xp = torch.zeros((S, n + pad), dtype=torch.float32, device=dev)
            xp[:, :n] = base
            g, gmin = run(xp, n + pad)
            res.append("%.1f (min %.1f) @%x" % (g, gmin, xp.data_ptr() & 0xffffff))
            del xp, sp
            torch.cuda.empty_cache()
        print("%-5s pad %7d B: %s" % (sig, pad_bytes, "   ".join(res)), flush=True)
