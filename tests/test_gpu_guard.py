"""-m gpu: calls shorter than one 1024-sample chunk from an input buffer that ENDS where mapped memory ends.

ADVICE r2 (high): render_wide4_kernel's unconditional loads used `4 * tt` as the index of a lane past the end of the
call, which is only inside the call when it is at least one chunk long — a single 960-, 768- or 512-sample frame
(what the decoder facade renders per IAMF_decoder_decode, from a d_in of exactly one frame) was read up to a whole
frame past its end.  The values were discarded, so every parity test passed as long as the neighbouring memory
happened to be mapped.  Here it is not: tests/guard/guard_alloc.hip places the input at the end of a mapping that is
followed by reserved-but-unmapped address space, so an out-of-range load is a GPU memory fault, not a silent read.
The PCM is compared with the oracle as well (bit-exact on the VALU paths)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GuardBuf(C.Structure):
    _fields_ = [("base", C.c_void_p), ("reserved", C.c_size_t), ("mapped", C.c_size_t), ("handle", C.c_void_p),
                ("ptr", C.c_void_p)]


@pytest.fixture(scope="module")
def guard():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    torch.zeros(1, device="cuda")   # a context on device 0 before the helper's first HIP call
    so = os.path.join(ROOT, "tests", "guard", "build", "libguard.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "guard")], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.guard_alloc.argtypes = [C.c_size_t, C.POINTER(GuardBuf)]
    L.guard_free.argtypes = [C.POINTER(GuardBuf)]
    L.guard_upload.argtypes = [C.POINTER(GuardBuf), C.c_void_p, C.c_size_t]
    return L


_GUARD = {}   # ONE guarded mapping per process, reused by every case: unmapping a range and mapping new memory at the
              # same addresses later in the process returned stale data on this stack (the read-back below caught it), and
              # the driver's handling of that is not what these tests are about


def _guard_buf(guard, nbytes):
    if "g" not in _GUARD:
        g = GuardBuf()
        assert guard.guard_alloc(8 << 20, C.byref(g)) == 0
        _GUARD["g"] = g
    g = _GUARD["g"]
    assert nbytes <= g.mapped and nbytes % 16 == 0
    return g, g.base + g.mapped - nbytes     # the buffer ENDS where the mapping ends


def _render_from_guarded(guard, A, mx, out_ch, x, fs, **kw):
    """x [S][m][fs]: ONE frame per stream, tight strides, the whole input ending at the guard"""
    import torch
    S, m, _ = x.shape
    nbytes = S * m * fs * 4
    g, ptr = _guard_buf(guard, nbytes)
    stage = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    torch.cuda.synchronize()
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert rt.hipMemcpy(ptr, stage.data_ptr(), nbytes, 3) == 0      # device to device: through the L2, like the kernels' reads
    torch.cuda.synchronize()
    back = torch.empty_like(stage)   # what the kernel will read is what was uploaded
    assert rt.hipMemcpy(back.data_ptr(), ptr, nbytes, 3) == 0
    torch.cuda.synchronize()
    assert torch.equal(back, stage), "guarded buffer does not hold the uploaded input"
    b = A.Batch(S, mx, out_ch, frame_size=fs, out_format=A.FMT_S16, limiter=True, **kw)
    cap = max(fs, 240) * out_ch * 2
    pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    n1 = b.render(ptr, m * fs, m * fs, 1, pcm.data_ptr(), cap, st)
    torch.cuda.synchronize()
    first = pcm.cpu().numpy()
    pcm.zero_()
    n2 = b.flush(pcm.data_ptr(), cap, st)
    torch.cuda.synchronize()
    second = pcm.cpu().numpy()
    b.close()
    outs = []
    for s in range(S):
        outs.append(np.concatenate([first[s][:n1 * out_ch * 2].view(np.int16).reshape(n1, out_ch),
                                    second[s][:n2 * out_ch * 2].view(np.int16).reshape(n2, out_ch)]))
    return outs


@pytest.mark.parametrize("fs", [960, 512, 768, 256, 1024])
@pytest.mark.parametrize("src,out", [("L714", "J"), ("L51", "C"), ("L714", "H")])
def test_short_call_m2m_from_exactly_sized_input(guard, fs, src, out):
    import iac_amd as A
    mx, omx = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out])
    ch = A.layout_channels(A.SS[out])
    S = 3
    x = np.stack([synth.hot(4100 + s, mx.m, fs, burst_phase=100, burst_period=400) for s in range(S)])
    got = _render_from_guarded(guard, A, mx, ch, x, fs)
    for s in range(S):
        want = O.stream_run(omx, ch, x[s], fs)
        assert got[s].shape == want.shape and np.array_equal(got[s], want), (fs, src, out, s)


@pytest.mark.parametrize("fs", [960, 512, 1024])
@pytest.mark.parametrize("lfe", [False, True])
@pytest.mark.parametrize("proj", ["exact", "mfma"])
def test_short_call_hoa_from_exactly_sized_input(guard, fs, lfe, proj):
    """TOA -> 5.1 on the VALU (what the facade's PROJ_EXACT takes) and MFMA wide4 variants, with and without the HOA LFE
    generator (whose scratch row fetch had the same out-of-range fallback); 65 streams = a second 64-stream LFE block"""
    import iac_amd as A
    mx, omx = A.get_h2m_matrix(3, A.SS["B"]), O.get_h2m(3, O.SS["B"])
    S = 65 if lfe else 2
    x = np.stack([synth.hot(4200 + s, 16, fs, burst_phase=50, burst_period=300) for s in range(S)])
    got = _render_from_guarded(guard, A, mx, 6, x, fs, lfe_hoa=lfe,
                               projection=A.PROJ_EXACT if proj == "exact" else A.PROJ_MFMA)
    for s in (0, 1, S - 1):
        want = O.stream_run(omx, 6, x[s], fs, lfe_rate=48000 if lfe else 0)
        assert got[s].shape == want.shape
        d = np.abs(got[s].astype(np.int32) - want.astype(np.int32)).max()
        assert d <= (0 if proj == "exact" else 1), (fs, lfe, proj, s, int(d))


@pytest.mark.parametrize("fs", [960, 512])
def test_short_call_binaural_from_exactly_sized_input(guard, fs):
    import iac_amd as A
    mx, omx = A.get_h2m_matrix(3, A.SS["BINAURAL"]), O.get_h2m(3, O.SS["BINAURAL"])
    x = np.stack([synth.hot(4300 + s, 16, fs, burst_phase=10, burst_period=200) for s in range(2)])
    got = _render_from_guarded(guard, A, mx, 2, x, fs)
    for s in range(2):
        want = O.stream_run(omx, 2, x[s], fs)
        assert np.array_equal(got[s], want)
