#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch path: host-resident element PCM -> H2D -> render -> D2H of the
packed PCM, double-buffered on two HIP streams so that the copies of one step overlap the render
of the other.  NOT what bench.py reports (its inputs are resident in HBM); this number goes into
DESIGN.md as the rate a host-fed deployment sees.  Run on the GPU box:  python tools/debug/pcie_probe.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iac_amd as A  # noqa: E402


def main():
    S, F, fs, in_ch, out_ch, steps = 512, 8, 1024, 16, 2, 12
    dev = torch.device("cuda", 0)
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    batch = A.Batch(S, mx, out_ch, frame_size=fs, out_format=A.FMT_S16, limiter=True)
    h_in = [(torch.randn((S, F, in_ch, fs)) * 0.25).pin_memory() for _ in range(2)]
    d_in = [torch.empty((S, F, in_ch, fs), device=dev) for _ in range(2)]
    stride = F * fs * out_ch * 2
    d_pcm = [torch.zeros((S, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    h_pcm = [torch.zeros((S, stride), dtype=torch.uint8).pin_memory() for _ in range(2)]
    copy = [torch.cuda.Stream(), torch.cuda.Stream()]
    render = torch.cuda.Stream()   # one batch = one stream of launches: renders stay ordered
    up = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]

    def step(i):
        b = i & 1
        with torch.cuda.stream(copy[b]):
            d_in[b].copy_(h_in[b], non_blocking=True)
            up[b].record()
        render.wait_event(up[b])
        with torch.cuda.stream(render):
            batch.render(d_in[b].data_ptr(), F * in_ch * fs, in_ch * fs, F, d_pcm[b].data_ptr(), stride,
                         render.cuda_stream)
            done[b].record()
        copy[b].wait_event(done[b])
        with torch.cuda.stream(copy[b]):
            h_pcm[b].copy_(d_pcm[b], non_blocking=True)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sf = S * F * fs * steps
    in_bytes = S * F * in_ch * fs * 4 * steps
    print(json.dumps({"pcie_inclusive_Msamples_per_s": round(sf / el / 1e6, 1),
                      "h2d_GBps": round(in_bytes / el / 1e9, 1), "ms_per_step": round(el / steps * 1e3, 3),
                      "streams": S, "frames_per_step": F, "note": "host pinned -> H2D -> render -> D2H, double-buffered"}))
    batch.close()


if __name__ == "__main__":
    main()
