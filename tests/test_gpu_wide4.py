"""-m gpu: render_wide4_kernel (iac_amd/csrc/render_wide4.hpp) — even 6..24-channel layouts, 16-bit
PCM, whole 1024-sample chunks.  The VALU variant must be bit-exact against the oracle; state must
carry over between calls that take different kernels (wide4 <-> wide <-> generic)."""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G


LAYOUTS = ["B", "C", "D", "J", "H"]  # 6, 8, 10, 12, 24 channels


SOURCES = dict(L714=12, L514=10, L71=8, L51=6, TOA=16, SOA=9, FOA=4)


@pytest.mark.parametrize("out", LAYOUTS)
@pytest.mark.parametrize("src", sorted(SOURCES))
def test_wide4_exact_multi_stream_multi_call(hip, src, out):
    A, G = hip
    S, fs, F = 3, 1024, 5
    oid = A.SS[out]
    ch = A.layout_channels(oid)
    m = SOURCES[src]
    if src in ("TOA", "SOA", "FOA"):
        order = {"TOA": 3, "SOA": 2, "FOA": 1}[src]
        mx, omx = A.get_h2m_matrix(order, oid), O.get_h2m(order, O.SS[out])
    else:
        try:
            mx, omx = A.get_m2m_matrix(A.SS[src], oid), O.get_m2m(O.SS[src], O.SS[out])
        except KeyError:
            pytest.skip("the reference has no %s -> %s matrix" % (src, out))
    x = np.stack([synth.hot(900 + 7 * s, m, F * fs, sigma=0.22, burst_phase=150 + 400 * s, burst_period=2300)
                  for s in range(S)])
    eg, og = [0.8, 1.0, 1.2], [1.0, 0.9, 1.0]
    got = G.hip_render(mx, ch, x, frame_size=fs, flush=True, frames_per_call=[1, 2, 1, 1],
                       gains=dict(element=eg, output=og), projection=A.PROJ_EXACT)
    for s in range(S):
        want = O.stream_run(omx, ch, x[s], fs, element_gain=eg[s], output_gain=og[s])
        assert got[s].shape == want.shape
        assert np.array_equal(got[s], want), (src, out, s)


def test_wide4_state_handoff_between_kernels(hip):
    """512-sample frames: calls of 1, 2, 3, 4 frames alternate between the 256-sample-chunk wide
    kernel (total % 1024 != 0) and wide4 (total % 1024 == 0); the flush takes the generic kernel"""
    A, G = hip
    fs, calls = 512, [1, 2, 3, 4, 2, 1]
    F = sum(calls)
    x = synth.hot(321, 12, F * fs, sigma=0.25, burst_phase=333, burst_period=1700)[None]
    oid = A.SS["J"]
    got = G.hip_render(A.get_m2m_matrix(A.SS["L714"], oid), 12, x, frame_size=fs, flush=True,
                       frames_per_call=calls)[0]
    want = O.stream_run(O.get_m2m(O.SS["L714"], O.SS["J"]), 12, x[0], fs)
    assert np.array_equal(got, want)


def test_wide4_mfma_state_handoff_within_1lsb(hip):
    A, G = hip
    fs, calls = 1024, [1, 3, 2]
    F = sum(calls)
    x = synth.hot(77, 16, F * fs, sigma=0.2, burst_phase=500, burst_period=2100)[None]
    for order in (3, 2, 1):
        m = (order + 1) ** 2
        for out in ("B", "J", "H"):
            oid = A.SS[out]
            ch = A.layout_channels(oid)
            got = G.hip_render(A.get_h2m_matrix(order, oid), ch, x[:, :m], frame_size=fs, flush=True,
                               frames_per_call=calls, projection=A.PROJ_MFMA)[0]
            want = O.stream_run(O.get_h2m(order, O.SS[out]), ch, x[0, :m], fs)
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            assert got.shape == want.shape and d.max() <= 1, (order, out, int(d.max()))


def test_wide4_quiet_signal_never_triggers(hip):
    A, G = hip
    fs, F = 1024, 3
    x = synth.quiet(5, 12, F * fs)[None]
    oid = A.SS["J"]
    got = G.hip_render(A.get_m2m_matrix(A.SS["L714"], oid), 12, x, frame_size=fs, flush=True)[0]
    want = O.stream_run(O.get_m2m(O.SS["L714"], O.SS["J"]), 12, x[0], fs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fs,calls", [(960, [1, 1, 2, 1, 3]), (1536, [1, 2, 1]), (320, [4, 1, 7, 2]), (1280, [1, 1, 1])])
def test_wide4_short_last_chunk(hip, fs, calls):
    """frame sizes that do not fill 1024-sample chunks (Opus-style 960, ...): every call ends with a
    short chunk (>= 256 samples) or falls back to another kernel; state carries across either way"""
    A, G = hip
    F = sum(calls)
    for src, m, out in (("L714", 12, "J"), ("TOA", 16, "H"), ("TOA", 16, "B")):
        x = synth.hot(404 + fs, m, F * fs, sigma=0.25, burst_phase=211, burst_period=1900)[None]
        oid = A.SS[out]
        ch = A.layout_channels(oid)
        if src == "TOA":
            mx, omx = A.get_h2m_matrix(3, oid), O.get_h2m(3, O.SS[out])
        else:
            mx, omx = A.get_m2m_matrix(A.SS[src], oid), O.get_m2m(O.SS[src], O.SS[out])
        got = G.hip_render(mx, ch, x, frame_size=fs, flush=True, frames_per_call=calls, projection=A.PROJ_EXACT)[0]
        want = O.stream_run(omx, ch, x[0], fs)
        assert got.shape == want.shape
        assert np.array_equal(got, want), (fs, src, out)
