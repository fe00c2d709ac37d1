"""How often do the tolerance-mode projections differ from the exact path, and by how much?  (doc item 8)
   cfg3:  TOA -> Sound System H, PROJ_AUTO (f32 MFMA, k-ordered fma chain) vs PROJ_EXACT (the reference's separately
          rounded multiply and add), bench "hot" programme, s16
   N3:    projection-mode TOA -> binaural / 5.1, PROJ_AUTO (one composed matrix) vs PROJ_EXACT (two exact stages)
Prints the fraction of PCM words that differ and the largest difference.   python tools/debug/flip_rates.py"""
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
S, F, fs = 16, 16, 1024
st = torch.cuda.current_stream().cuda_stream


def run(mx, out_ch, x, proj, pmat=None):
    b = A.Batch(S, mx, out_ch, frame_size=fs, projection=proj)
    if pmat is not None:
        b.set_projection(pmat)
    cap = F * fs * out_ch * 2
    pcm = torch.zeros((S, cap), dtype=torch.uint8, device=dev)
    n = b.render(x.data_ptr(), x.shape[1], 16 * fs, F, pcm.data_ptr(), cap, st)
    torch.cuda.synchronize()
    b.close()
    return pcm.cpu().numpy().view(np.int16).reshape(S, -1)[:, :n * out_ch].astype(np.int32)


x = bench.synth_hot_device(S, 16, F, fs, 1000, dev).reshape(S, -1).contiguous()
for name, out in (("cfg3 TOA -> H (24 ch)", "H"), ("TOA -> J (12 ch)", "J"), ("TOA -> B (6 ch)", "B")):
    mx = A.get_h2m_matrix(3, A.SS[out])
    ch = A.layout_channels(A.SS[out])
    e, m = run(mx, ch, x, A.PROJ_EXACT), run(mx, ch, x, A.PROJ_AUTO)
    d = np.abs(e - m)
    print("%-28s MFMA vs exact: %.4f %% of %d PCM words differ, max |diff| %d LSB" % (name, 100.0 * (d > 0).mean(), d.size, d.max()))
rng = np.random.default_rng(5)
P = rng.integers(-6000, 6000, size=(16, 16)).astype(np.float32) * np.float32(2.0 ** -15)
P[np.arange(16), np.arange(16)] += np.float32(0.5)
for name, out in (("projection TOA -> binaural", "BINAURAL"), ("projection TOA -> B (6 ch)", "B")):
    mx = A.get_h2m_matrix(3, A.SS[out])
    ch = A.layout_channels(A.SS[out])
    e, m = run(mx, ch, x, A.PROJ_EXACT, P), run(mx, ch, x, A.PROJ_AUTO, P)
    d = np.abs(e - m)
    print("%-28s composed vs two-stage: %.4f %% of %d PCM words differ, max |diff| %d LSB" % (name, 100.0 * (d > 0).mean(), d.size, d.max()))
