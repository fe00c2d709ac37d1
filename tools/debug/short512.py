"""debug: order dependence of the lone 512-sample call (960 first, then 512), plain vs guarded input, with and without
releasing the previous guarded mapping"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import iac_amd as A, oracle_lib as O, synth, gpu_util as G
import test_gpu_guard as TG

L = C.CDLL(os.path.join(ROOT, "tests", "guard", "build", "libguard.so"))
L.guard_alloc.argtypes = [C.c_size_t, C.POINTER(TG.GuardBuf)]
L.guard_free.argtypes = [C.POINTER(TG.GuardBuf)]
L.guard_upload.argtypes = [C.POINTER(TG.GuardBuf), C.c_void_p, C.c_size_t]
torch.zeros(1, device="cuda")
mx, omx = A.get_m2m_matrix(A.SS["L714"], A.SS["J"]), O.get_m2m(O.SS["L714"], O.SS["J"])

def data(fs):
    x = np.stack([synth.hot(4100 + s, 12, fs, burst_phase=100, burst_period=400) for s in range(3)])
    return x, [O.stream_run(omx, 12, x[s], fs) for s in range(3)]

def report(tag, got, want):
    for s in range(3):
        d = got[s].astype(np.int32) - want[s].astype(np.int32)
        rows = np.nonzero(np.abs(d).max(axis=1))[0]
        zc = [c for c in range(12) if not got[s][:, c].any()]
        print(tag, "stream", s, "ok" if not len(rows) else "BAD rows %d..%d (%d) zero-ch %s" % (rows[0], rows[-1], len(rows), zc), flush=True)

mode = sys.argv[1]
if mode == "plain":
    for fs in (960, 512):
        x, want = data(fs)
        report("plain %d" % fs, G.hip_render(mx, 12, x, frame_size=fs, flush=True), want)
elif mode == "guard":
    for fs in (960, 512):
        x, want = data(fs)
        report("guard %d" % fs, TG._render_from_guarded(L, A, mx, 12, x, fs), want)
elif mode == "guard_keep":   # never free: the second mapping gets another address range
    orig = L.guard_free
    class NoFree:
        def __call__(self, g): return 0
    L.guard_free = NoFree()
    for fs in (960, 512):
        x, want = data(fs)
        report("guard_keep %d" % fs, TG._render_from_guarded(L, A, mx, 12, x, fs), want)
