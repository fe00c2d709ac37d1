"""-m gpu: the decoder facade switched to the reference's -DSAMSUNG_TV behaviour (upstream's default
build), against PCM the REAL reference built that way produced (tests/golden/tv.npz,
oracle/gen_golden_tv.py): TV layout->layout tables, 12-channel PCM stride incl. the > 12-channel
overlap, top layer of scalable elements.  Bit-exact.  Also the batch ABI's pcm_stride_channels."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G
import iac_amd as A
import tv_cases as T
from decoder_driver import decode_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available()
    L = C.CDLL(A.lib_path())
    L.iamf_hip_decoder_set_variant.argtypes = [C.c_void_p, C.c_int]
    return L


class _TV:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name != "IAMF_decoder_open":
            return f
        lib = self._lib

        def open_():
            lib.IAMF_decoder_open.restype = C.c_void_p
            d = lib.IAMF_decoder_open()
            assert lib.iamf_hip_decoder_set_variant(d, 1) == 0
            return d
        return open_


@pytest.mark.parametrize("name", sorted(T.CASES))
def test_facade_matches_the_samsung_tv_reference(lib, golden, name):
    c = T.case(name)
    pcm, rets = decode_stream(_TV(lib), T.build(name), c["layout"], **T.decode_kwargs(name))
    want = golden.npz("tv")[name]
    assert list(rets) == list(golden.npz("tv")[name + "_rets"]), name
    assert pcm.shape == want.shape
    assert np.array_equal(pcm, want), name


def test_tv_tables_differ_where_the_reference_differs():
    d = A.get_m2m_matrix(A.SS["L51"], A.SS["G"])
    t = A.get_m2m_matrix(A.SS["L51"], A.SS["G"], variant=1)
    assert (d.m, d.n) == (t.m, t.n) == (6, 14)
    a = np.ctypeslib.as_array(d.mat, shape=(84,)).copy()
    b = np.ctypeslib.as_array(t.mat, shape=(84,)).copy()
    assert not np.array_equal(a, b)
    s = A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"])
    st = A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"], variant=1)
    assert np.array_equal(np.ctypeslib.as_array(s.mat, shape=(4,)), np.ctypeslib.as_array(st.mat, shape=(4,)))


@pytest.mark.parametrize("out,fmt,bps", [("B", A.FMT_S16, 2), ("J", A.FMT_S24, 3), ("H", A.FMT_S32, 4), ("G", A.FMT_S16, 2)])
def test_batch_pcm_stride_12(out, fmt, bps):
    """iamf_hip_batch with pcm_stride_channels = 12 against the natural layout of the same call, re-laid on
    the host by the reference's loop (channel-major writes, last writer wins)"""
    import torch
    import synth
    S, fs, F = 3, 1024, 2
    ch = A.layout_channels(A.SS[out])
    mx = A.get_m2m_matrix(A.SS["L714"], A.SS[out], variant=1)
    x = np.stack([synth.hot(40 + s, 12, fs * F, burst_phase=300, burst_period=900) * np.float32(0.6) for s in range(S)])
    xin = torch.from_numpy(G.to_frames(x, fs)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for stride in (0, 12):
        b = A.Batch(S, mx, ch, frame_size=fs, out_format=fmt, pcm_stride_channels=stride)
        cap = (F * fs * max(ch, 12) + 16) * bps
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        pcm[:] = 0xEE
        n = b.render(xin.data_ptr(), F * 12 * fs, 12 * fs, F, pcm.data_ptr(), cap, st)
        torch.cuda.synchronize()
        outs[stride] = (n, pcm.cpu().numpy())
        b.close()
    n = outs[0][0]
    assert n == outs[12][0] == F * fs - 240
    for s in range(S):
        nat = outs[0][1][s][:n * ch * bps].reshape(n, ch, bps)
        want = np.full((n * 12 + max(0, ch - 12), bps), 0xEE, dtype=np.uint8)
        want[:n * 12] = 0                      # memset(dst, 0, bytes * frame_size * stride)
        for c in range(ch):                    # for c: for i: dst[i * stride + c] = ...
            want[np.arange(n) * 12 + c] = nat[:, c]
        got = outs[12][1][s][:want.size].reshape(-1, bps)
        assert np.array_equal(got, want), (out, s)
        assert (outs[12][1][s][want.size:want.size + 8] == 0xEE).all()   # nothing written past it


@pytest.mark.parametrize("name", sorted(T.SWITCH_CASES))
def test_run_time_layout_switch_matches_the_samsung_tv_reference(lib, golden, name):
    """IAMF_decoder_output_layout_set_* + IAMF_decoder_configure(h, NULL, 0, NULL) between two frames: the -DSAMSUNG_TV
    build re-opens the renderers in place and re-initialises the limiter (IAMF_decoder.c:3819-3881) while the stream time
    and the parameter timelines go on.  PCM and every return value against the reference built that way."""
    import e2e_cases as E
    from decoder_driver import decode_stream_switching
    c = T.SWITCH_CASES[name]
    stream = E.build(c["stream"])[0]
    chunks, rets = decode_stream_switching(_TV(lib), stream, c["layouts"], c["after"], bit_depth=E.CASES[c["stream"]].get("bit_depth", 16))
    g = golden.npz("tv")
    assert [r[1] if isinstance(r, tuple) else r for r in rets] == list(g[name + "_rets"]), name
    assert [len(x) for x in chunks] == list(g[name + "_lens"])
    assert np.array_equal(np.concatenate(chunks, axis=0), g[name]), name
