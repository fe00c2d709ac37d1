"""-m gpu: the HOA LFE generator (reference h2m_rdr.c:1151-1239 under the reference's build
switch -DDISABLE_LFE_HOA=0) on the HIP path, against goldens of the LFE-enabled reference
(tests/golden/lfe.npz, oracle/gen_golden_lfe.py) and against the oracle.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import gpu_util as G
import iac_amd as A
import lfe_cases as LC
import oracle_lib as O
from decoder_driver import decode_stream

pytestmark = pytest.mark.gpu


def same_floats(a, b):
    """bit-identical except for the sign of a zero (the mixer's 0 + y turns -0 into +0)"""
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


@pytest.mark.parametrize("name", sorted(LC.STAGE))
def test_stage_f32_bit_exact_vs_reference(golden, name):
    """render_H2M + generator over ragged consecutive calls (filter state carried), float output,
    limiter off: the reference's own floats.  frame_size 1 = sample-interleaved input."""
    order, oid, rate, sizes, _ = LC.STAGE[name]
    x = LC.stage_input(name)
    ch = O.OUT_CH[oid]
    got = G.hip_render(A.get_h2m_matrix(order, oid), ch, x[None], frame_size=1, fmt=A.FMT_F32, limiter=False,
                       flush=False, frames_per_call=sizes, sample_rate=rate, lfe_hoa=True)[0]
    want = np.ascontiguousarray(golden.npz("lfe")["stage_" + name].T)
    assert same_floats(got, want), name


def test_many_streams_lane_mapping_and_state_vs_oracle():
    """150 streams (three 64-stream blocks of the recurrence kernel, the last one partial) in calls of
    3 + 1 + 2 frames of 960 samples: every stream must get ITS filter state"""
    S, fs, F = 150, 960, 6
    x = np.stack([LC.programme(900 + s, 16, fs * F) * np.float32(0.5 + (s % 7) * 0.1) for s in range(S)])
    mx, omx = A.get_h2m_matrix(3, A.SS["B"]), O.get_h2m(3, O.SS["B"])
    got = G.hip_render(mx, 6, x, frame_size=fs, frames_per_call=[3, 1, 2], lfe_hoa=True, projection=A.PROJ_EXACT)
    for s in list(range(0, S, 13)) + [63, 64, 127, 128, 149]:
        want = O.stream_run(omx, 6, x[s], fs, lfe_rate=48000)
        assert np.array_equal(got[s], want), s
    assert np.abs(got[5][:, 3].astype(np.int32)).max() > 1000   # the LFE slot carries a signal


def test_denormal_decay_f32_vs_oracle():
    """after the programme stops the recurrence decays through the f32 denormal range (a 120 Hz pole pair:
    ~0.99 per sample); the CPU keeps gradual underflow, so must the kernel"""
    fs, F = 1024, 24
    x = LC.programme(77, 4, fs * F, silence_from=1500)
    mx, omx = A.get_h2m_matrix(1, A.SS["B"]), O.get_h2m(1, O.SS["B"])
    got = G.hip_render(mx, 6, x[None], frame_size=fs, fmt=A.FMT_F32, limiter=False, flush=False, lfe_hoa=True)[0]
    want = np.ascontiguousarray(O.render_h2m_lfe(omx, x, 6, 48000, [fs] * F).T)
    lfe = want[:, 3]
    tiny = np.abs(lfe[np.nonzero(lfe)[0]]).min()
    assert 0 < tiny < 1.2e-38, "the case must reach denormal outputs (got %g)" % tiny
    assert same_floats(got, want)


@pytest.mark.parametrize("fmt,bd", [(A.FMT_S16, 16), (A.FMT_S24, 24), (A.FMT_S32, 32)])
def test_pipeline_formats_vs_oracle(fmt, bd):
    fs, F = 1024, 4
    x = np.stack([LC.programme(300 + s, 16, fs * F) for s in range(3)])
    mx, omx = A.get_h2m_matrix(3, A.SS["J"]), O.get_h2m(3, O.SS["J"])
    got = G.hip_render(mx, 12, x, frame_size=fs, fmt=fmt, lfe_hoa=True, projection=A.PROJ_EXACT)
    for s in range(3):
        assert np.array_equal(got[s], O.stream_run(omx, 12, x[s], fs, bit_depth=bd, lfe_rate=48000)), s


W4 = [(1, "B", 6), (2, "C", 8), (3, "D", 10), (3, "J", 12), (3, "G", 14), (1, "H", 24), (3, "H", 24), (2, "L312", 6), (3, "L712", 10)]


@pytest.mark.parametrize("order,lay,ch", W4, ids=["o%d_%s" % (o, l) for o, l, _ in W4])
def test_wide4_lfe_variant_bit_exact_vs_oracle_and_generic(order, lay, ch, monkeypatch):
    """16-bit PCM, limiter on, whole 1024-sample chunks: render_wide4_kernel<.., LFE> fills the LFE slot.  A hot programme (the limiter works), three calls with the filter and limiter state carried, the last one
    ending in a short chunk.  Exact projection: bit-equal to the oracle, and to the generic kernel's output."""
    fs, calls = 1024, [3, 2, 1]
    m, F = (order + 1) ** 2, 6
    x = np.stack([LC.programme(500 + 7 * s + order, m, fs * F) * np.float32(2.5) for s in range(3)])
    mx, omx = A.get_h2m_matrix(order, A.SS[lay]), O.get_h2m(order, O.SS[lay])
    got = G.hip_render(mx, ch, x, frame_size=fs, frames_per_call=calls, lfe_hoa=True, projection=A.PROJ_EXACT)
    monkeypatch.setenv("IAMF_HIP_NO_WIDE4", "1")
    gen = G.hip_render(mx, ch, x, frame_size=fs, frames_per_call=calls, lfe_hoa=True, projection=A.PROJ_EXACT)
    monkeypatch.delenv("IAMF_HIP_NO_WIDE4")
    for s in range(3):
        want = O.stream_run(omx, ch, x[s], fs, lfe_rate=48000)
        assert np.array_equal(got[s], want), s
        assert np.array_equal(gen[s], want), s
    lfe_slots = [c for c in range(ch) if got[0][:, c].any() and not G.hip_render(
        mx, ch, x[:1], frame_size=fs, projection=A.PROJ_EXACT)[0][:, c].any()]
    assert lfe_slots == [mx.lfe1], lfe_slots   # silent without the generator, alive with it


@pytest.mark.parametrize("fs,calls", [(1024, [40]), (960, [37, 3])], ids=["1024x40", "960x37+3"])
def test_long_calls_vs_oracle(fs, calls):
    """calls of tens of frames (what a batch deployment issues): 70 streams = two blocks of the recurrence kernel"""
    S, F = 70, sum(calls)
    x = np.stack([LC.programme(1200 + s, 16, fs * F) * np.float32(1.0 + (s % 5) * 0.4) for s in range(S)])
    mx, omx = A.get_h2m_matrix(3, A.SS["J"]), O.get_h2m(3, O.SS["J"])
    got = G.hip_render(mx, 12, x, frame_size=fs, frames_per_call=calls, lfe_hoa=True, projection=A.PROJ_EXACT)
    for s in (0, 33, 63, 64, 69):
        assert np.array_equal(got[s], O.stream_run(omx, 12, x[s], fs, lfe_rate=48000)), s


def test_wide4_lfe_variant_on_the_mfma_projection_within_one_lsb():
    """the default projection of an ambisonics element (f32 MFMA, +-1 LSB)"""
    fs, F = 1024, 4
    x = np.stack([LC.programme(640 + s, 16, fs * F) for s in range(2)])
    mx, omx = A.get_h2m_matrix(3, A.SS["J"]), O.get_h2m(3, O.SS["J"])
    got = G.hip_render(mx, 12, x, frame_size=fs, lfe_hoa=True)
    for s in range(2):
        want = O.stream_run(omx, 12, x[s], fs, lfe_rate=48000)
        assert np.abs(got[s].astype(np.int32) - want.astype(np.int32)).max() <= 1
        assert np.abs(got[s][:, mx.lfe1].astype(np.int32)).max() > 1000


def test_switch_off_is_the_default_build():
    """lfe_hoa = 0 -> the LFE slot stays silent, exactly the default reference build"""
    fs, F = 1024, 2
    x = LC.programme(55, 16, fs * F)[None]
    mx, omx = A.get_h2m_matrix(3, A.SS["B"]), O.get_h2m(3, O.SS["B"])
    got = G.hip_render(mx, 6, x, frame_size=fs, projection=A.PROJ_EXACT)[0]   # (AUTO = the +-1 LSB MFMA projection)
    assert np.array_equal(got, O.stream_run(omx, 6, x[0], fs))
    assert not got[:, 3].any()


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available()
    L = C.CDLL(A.lib_path())
    L.iamf_hip_decoder_set_hoa_lfe.argtypes = [C.c_void_p, C.c_int]
    return L


class _LfeOn:
    """the decoder library with the generator switched on for every handle it opens"""

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
            assert lib.iamf_hip_decoder_set_hoa_lfe(d, 1) == 0
            return d
        return open_


@pytest.mark.parametrize("name", sorted(LC.E2E))
def test_facade_matches_lfe_enabled_reference_decoder(lib, golden, name):
    c = LC.E2E[name]
    stream, _ = LC.build(name)
    pcm, rets = decode_stream(_LfeOn(lib), stream, ("ss", LC.SS_ENUM[c["ss"]]), bit_depth=c["bit_depth"])
    want = golden.npz("lfe")["e2e_" + name]
    assert list(rets) == list(golden.npz("lfe")["e2e_" + name + "_rets"]), name
    assert pcm.shape == want.shape
    assert np.array_equal(pcm, want), name


def test_facade_switch_protocol(lib):
    lib.IAMF_decoder_open.restype = C.c_void_p
    lib.IAMF_decoder_close.argtypes = [C.c_void_p]
    d = lib.IAMF_decoder_open()
    assert lib.iamf_hip_decoder_set_hoa_lfe(d, 1) == 0
    assert lib.iamf_hip_decoder_set_hoa_lfe(None, 1) == -1
    lib.IAMF_decoder_close(d)
