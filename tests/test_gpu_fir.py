"""-m gpu: the binaural HRTF FIR renderer (kind FIR, f32 MFMA Toeplitz GEMM).

PARITY UNPINNED: the reference's binauraliser (Resonance Audio / BEAR) is not in the reference
tree, so there is nothing of the reference to compare with.  The checker here is this repo's own
specification, y[e][t] = sum_c sum_k h[e][c][k] x[c][t-k], evaluated in float64 (numpy).
Tolerance: 2^-17 absolute on the f32 tap (the BASELINE float tolerance), +-1 LSB on PCM."""
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


def hrir_set(seed, channels, taps):
    rng = np.random.default_rng(seed)
    k = np.arange(taps)
    h = rng.standard_normal((2, channels, taps)) * np.exp(-k / (taps / 6.0)) * 0.08
    return h.astype(np.float32)


def fir64(h, x):
    """float64 direct-form reference: [2][n]"""
    n = x.shape[1]
    y = np.zeros((2, n))
    for e in range(2):
        for c in range(x.shape[0]):
            y[e] += np.convolve(x[c].astype(np.float64), h[e, c].astype(np.float64))[:n]
    return y


@pytest.mark.parametrize("m,taps,fs,calls", [(16, 256, 1024, [2, 1, 3]), (4, 100, 1024, [3]), (9, 33, 960, [1, 1, 2]),
                                             (1, 256, 1024, [2]), (16, 64, 2048, [1, 1])])
def test_fir_stage_matches_float64_convolution(hip, m, taps, fs, calls):
    A, G = hip
    F = sum(calls)
    S = 3
    x = np.stack([synth.gaussian(800 + s, m, F * fs, 0.1) for s in range(S)])
    h = hrir_set(5, m, taps)
    got = G.hip_render(A.fir_matrix(h), 2, x, frame_size=fs, fmt=A.FMT_F32, limiter=True, flush=True,
                       frames_per_call=calls, fir_taps=taps)
    for s in range(S):
        y = fir64(h, x[s])
        assert np.abs(y).max() < 0.85  # quiet enough that the limiter stays at unity gain
        # limiter at unity: output = input delayed by 240 (first 240 withheld, flush emits the tail)
        assert got[s].shape == (F * fs, 2)
        assert np.abs(got[s].T - y).max() <= F32_TOL, (s, float(np.abs(got[s].T - y).max()))


def test_fir_pipeline_with_limiter(hip):
    """hot programme: HRTF -> limiter -> int16.  Expected value: the oracle's limiter + pack run on
    the float64 convolution rounded to f32; the limiter is a feedback system, so allow the few
    +-1 LSB ties of the FIR rounding to move a handful of samples by more."""
    A, G = hip
    fs, F = 1024, 6
    x = synth.hot(900, 16, F * fs, sigma=0.2, burst_amp=1.2, burst_phase=700, burst_period=2500)[None]
    h = hrir_set(6, 16, 256)
    got = G.hip_render(A.fir_matrix(h), 2, x, frame_size=fs, fmt=A.FMT_S16, limiter=True, flush=True,
                       frames_per_call=[2, 4], fir_taps=256)[0]
    y = fir64(h, x[0]).astype(np.float32)
    assert np.abs(y).max() > 1.0  # the limiter has work to do
    z, _ = O.limiter_run(y, [fs] * F)
    want = O.pack(z, 16)
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == want.shape
    assert (d <= 1).mean() > 0.999 and d.max() <= 8
    assert np.abs(got.astype(np.int32)).max() <= 1.001 * 32768 * 10 ** (-1 / 20) + 1
