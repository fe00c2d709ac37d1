"""-m gpu: the MFMA projection (v_mfma_f32_32x32x2_f32) of the wide kernel.  An f32 MFMA is an
exact-product, k-ordered fma chain, so it differs from the reference's separate multiply and add
roundings by at most ~1 ulp(f32) per term.  Tolerances (BASELINE north star): PCM within +-1 LSB
at the chosen bit depth; float tap within 2^-17 absolute (1/4 LSB at 16 bit)."""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

F32_TOL = 2.0 ** -17


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G


@pytest.mark.parametrize("order,out", [(3, "H"), (3, "J"), (2, "H"), (1, "J"), (3, "G"), (2, "L312"), (0, "B")])
def test_mfma_projection_pcm_within_1lsb(hip, order, out):
    A, G = hip
    m = (order + 1) ** 2
    fs, F = 1024, 4
    x = synth.hot(500 + order, m, F * fs, sigma=0.2, burst_phase=1000, burst_period=3000)[None]
    oid = A.SS[out]
    ch = A.layout_channels(oid)
    got = G.hip_render(A.get_h2m_matrix(order, oid), ch, x, frame_size=fs, projection=A.PROJ_MFMA)[0]
    want = O.stream_run(O.get_h2m(order, O.SS[out]), ch, x[0], fs)
    assert got.shape == want.shape
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, (order, out, int(d.max()))
    # measured on the bench programme (tools/debug/flip_rates.py): 0.013-0.018 % of PCM words, always 1 LSB
    assert (d != 0).mean() < 0.005  # only rounding ties may move


def test_mfma_projection_float_tap(hip):
    A, G = hip
    fs, F = 1024, 3
    x = synth.gaussian(13, 16, F * fs, 0.15)[None]
    got = G.hip_render(A.get_h2m_matrix(3, A.SS["H"]), 24, x, frame_size=fs, fmt=A.FMT_F32,
                       projection=A.PROJ_MFMA)[0]
    omx = O.get_h2m(3, O.SS["H"])
    y = O.render(omx, x[0], 24)
    z, _ = O.limiter_run(y, [fs] * F)
    assert got.shape == z.T.shape
    assert np.abs(got - z.T).max() <= F32_TOL
    assert np.all(got[:, 3] == 0) and np.all(got[:, 23] == 0)


def test_mfma_m2m_when_forced(hip):
    """channel-layout matrices default to the exact VALU path; MFMA can be forced and must agree
    within the same tolerance (12 -> 24, 12 -> 12)."""
    A, G = hip
    fs, F = 960, 4
    x = synth.hot(77, 12, F * fs, sigma=0.2, burst_phase=200, burst_period=2000)[None]
    for out in ("H", "J"):
        oid = A.SS[out]
        ch = A.layout_channels(oid)
        got = G.hip_render(A.get_m2m_matrix(A.SS["L714"], oid), ch, x, frame_size=fs,
                           projection=A.PROJ_MFMA)[0]
        want = O.stream_run(O.get_m2m(O.SS["L714"], O.SS[out]), ch, x[0], fs)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1


def test_auto_mode_is_exact_for_m2m_and_stereo(hip):
    A, G = hip
    fs, F = 1024, 3
    x = synth.hot(5, 12, F * fs, sigma=0.2, burst_phase=100, burst_period=2500)[None]
    got = G.hip_render(A.get_m2m_matrix(A.SS["L714"], A.SS["J"]), 12, x, frame_size=fs)[0]
    want = O.stream_run(O.get_m2m(O.SS["L714"], O.SS["J"]), 12, x[0], fs)
    assert np.array_equal(got, want)
