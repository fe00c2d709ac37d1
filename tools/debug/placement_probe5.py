"""Stream-major vs FRAME-major input layout on the headline kernel (quiet and hot), in a slow-mode arena and
in fresh allocations.  stream-major: [stream][frame][ch][fs] (+4 KiB per stream); frame-major:
[frame][stream][ch][fs]: at any moment the 512 workgroups read inside one contiguous 32 MiB window."""
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
n = F * M * fs


def rate(b, ptr, ss, fstr, pcm, steps=8):
    for _ in range(2):
        b.render(ptr, ss, fstr, F, pcm.data_ptr(), F * fs * 4, st)
    b.reset()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        ev[i][0].record()
        b.render(ptr, ss, fstr, F, pcm.data_ptr(), F * fs * 4, st)
        ev[i][1].record()
    torch.cuda.synchronize()
    b.reset()
    ms = np.median([a.elapsed_time(c) for a, c in ev][2:])
    return S * F * fs / (ms * 1e-3) / 1e9


arena = torch.empty((8 * S * (n + 1024),), dtype=torch.float32, device=dev)   # first allocation: slow mode
pcm = torch.zeros((S, F * fs * 4), dtype=torch.uint8, device=dev)
b = A.Batch(S, mx, 2, frame_size=fs)
for sig in ("quiet", "hot"):
    x = bench.synth_hot_device(S, M, F, fs, 1000, dev) if sig == "hot" else torch.randn((S, F, M, fs), device=dev) * 0.05
    # stream-major with the 4 KiB stagger, inside the arena
    sm = arena[:S * (n + 1024)].view(S, n + 1024)
    sm[:, :n] = x.reshape(S, n)
    r_sm = rate(b, sm.data_ptr(), n + 1024, M * fs, pcm)
    ref = pcm.clone()
    # frame-major inside the arena
    fm = arena[2 * S * (n + 1024):2 * S * (n + 1024) + S * n].view(F, S, M * fs)
    fm.copy_(x.reshape(S, F, M * fs).transpose(0, 1))
    r_fm = rate(b, fm.data_ptr(), M * fs, S * M * fs, pcm)
    same = bool(torch.equal(ref, pcm))
    print("%-5s arena (slow mode): stream-major %.1f   frame-major %.1f   (same PCM: %s)" % (sig, r_sm, r_fm, same), flush=True)
    keep = []
    for i in range(6):
        a = torch.empty((S, n + 1024), dtype=torch.float32, device=dev)
        a[:, :n] = x.reshape(S, n)
        r1 = rate(b, a.data_ptr(), n + 1024, M * fs, pcm)
        fmv = a.view(-1)[:S * n].view(F, S, M * fs)
        fmv.copy_(x.reshape(S, F, M * fs).transpose(0, 1))
        r2 = rate(b, fmv.data_ptr(), M * fs, S * M * fs, pcm)
        print("%-5s fresh @%x: stream-major %.1f   frame-major %.1f" % (sig, a.data_ptr() >> 21, r1, r2), flush=True)
        keep.append(a)
        if len(keep) > 3:
            keep.pop(0)
    del keep, x
    torch.cuda.empty_cache()
