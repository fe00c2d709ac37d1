#!/usr/bin/env python3
"""headline kernel, one input buffer, the PCM output at different OFFSETS inside arenas: which output addresses are fast with this
input?   python tools/debug/pcm_offset_probe.py"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
import iac_amd as A  # noqa: E402

args = b.parse_args(["--placement-tries", "1", "--pcm-placement-tries", "1", "--no-cpu-baseline"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
wl = b.Workload(A, args.workload, args, 0, dev)
if wl.x is None:
    wl.x = torch.zeros((wl.S, wl.stream_stride), dtype=torch.float32, device=dev)


def rate(ptr):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
    for a, bb in ev:
        a.record()
        wl.batch.render(wl.x.data_ptr(), wl.stream_stride, wl.frame_stride, wl.F, ptr, wl.stride_bytes, wl.stream)
        bb.record()
    torch.cuda.synchronize()
    wl.batch.reset()
    return round(wl.sf_per_step / (float(np.median([a.elapsed_time(bb) for a, bb in ev[1:]])) * 1e-3) / 1e9, 1)


need = wl.S * wl.stride_bytes
print("input at 0x%x (%.2f GiB), pcm[0] 0x%x pcm[1] 0x%x, one output %.0f MiB" % (wl.x.data_ptr(), wl.x.numel() * 4 / 2 ** 30, wl.pcm[0].data_ptr(), wl.pcm[1].data_ptr(), need / 2 ** 20))
print("pcm[0]", rate(wl.pcm[0].data_ptr()), "pcm[1]", rate(wl.pcm[1].data_ptr()))
arena = torch.zeros(need + (2 << 30), dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + 4095) & ~4095
print("arena at 0x%x" % base)
step = 32 << 20
print("offset MiB : Gsamples/s")
print(" ".join("%d:%.1f" % (k * step >> 20, rate(base + k * step)) for k in range(0, (2 << 30) // step)))
