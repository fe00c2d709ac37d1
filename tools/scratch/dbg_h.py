import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import iac_amd as A, gpu_util as G, oracle_lib as O, synth
fs, F = 1024, 3
x = synth.gaussian(13, 16, F * fs, 0.15)[None]
mx = A.get_h2m_matrix(3, A.SS["H"])
got = G.hip_render(mx, 24, x, frame_size=fs, limiter=True, flush=True, projection=A.PROJ_EXACT)[0]
want = O.stream_run(O.get_h2m(3, O.SS["H"]), 24, x[0], fs)
d = np.argwhere(got != want)
print("mismatches", len(d))
rows = np.unique(d[:, 0]); cols = np.unique(d[:, 1])
print("rows", rows[:40], "...", rows[-10:], len(rows))
print("cols", cols)
for r, c in d[:10]:
    print(r, c, got[r, c], want[r, c])
