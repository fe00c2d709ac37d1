"""-m gpu parity tests: the HIP path (through the C ABI) against the golden vectors produced by
the real reference and against the oracle on seeded inputs.  Integer PCM and the f32 stage taps
of the VALU path must be BIT-EXACT; the tolerance-based MFMA path has its own tests."""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import iac_amd as A
    import gpu_util as G
    return A, G


def _same_f32(a, b):
    return a.shape == b.shape and np.array_equal(a, b)  # -0.0 == +0.0 is fine for a stage tap


def test_h2m_goldens_bit_exact(hip, golden):
    A, G = hip
    g = golden.npz("h2m")
    for key, meta in golden.manifest.items():
        if not key.startswith("h2m/") or "sentinel" in key:
            continue
        name = key.split("/")[1]
        m = (meta["order"] + 1) ** 2
        x = synth.gaussian(meta["seed"], m, meta["ns"], meta["sigma"])
        mx = A.get_h2m_matrix(meta["order"], meta["out_id"])
        ch = A.layout_channels(meta["out_id"])
        y = G.hip_render(mx, ch, x[None], frame_size=64, fmt=A.FMT_F32, limiter=False, flush=False)[0]
        assert _same_f32(y.T, g[name]), name


def test_m2m_goldens_bit_exact(hip, golden):
    A, G = hip
    g = golden.npz("m2m")
    for key, meta in golden.manifest.items():
        if not key.startswith("m2m/"):
            continue
        name = key.split("/")[1]
        x = synth.uniform(meta["seed"], meta["m"], meta["ns"], meta["amp"])
        mx = A.get_m2m_matrix(meta["in_id"], meta["out_id"])
        y = G.hip_render(mx, mx.n, x[None], frame_size=64, fmt=A.FMT_F32, limiter=False, flush=False)[0]
        assert _same_f32(y.T, g[name]), name


def _limiter_input(meta):
    total = sum(meta["sizes"])
    if meta["kind"] == "hot":
        return synth.hot(meta["seed"], meta["ch"], total, sigma=0.25, burst_phase=700, burst_period=6000)
    return synth.quiet(meta["seed"], meta["ch"], total)


@pytest.mark.parametrize("name", ["hot2", "quiet2", "hot24", "hot2_960", "hot2_ragged", "hot12"])
def test_limiter_goldens_bit_exact(hip, golden, name):
    A, G = hip
    meta = golden.manifest["limiter/" + name]
    x = _limiter_input(meta)
    sizes = meta["sizes"]
    if len(set(sizes)) == 1:
        fs, calls = sizes[0], [1] * len(sizes)
    else:  # ragged call sizes: frame_size 1, one call per reference block
        fs, calls = 1, sizes
    y = G.hip_render(G.identity_matrix(meta["ch"]), meta["ch"], x[None], frame_size=fs, fmt=A.FMT_F32,
                     limiter=True, flush=True, frames_per_call=calls)[0]
    ref = golden.npz("limiter")[name]
    assert y.shape == ref.T.shape
    assert np.array_equal(y.T.view(np.uint32), ref.view(np.uint32)), name


@pytest.mark.parametrize("fmt_bits", [16, 24, 32])
def test_pipeline_toa_binaural_vs_oracle(hip, fmt_bits):
    """cfg4 (reference-exact form): TOA -> binaural matrix + limiter + PCM, several streams,
    state carried over three calls, then flush."""
    A, G = hip
    S, fs, F = 5, 1024, 9
    x = np.stack([synth.hot(1000 + s, 16, F * fs, burst_phase=900 + 50 * s, burst_period=4000) for s in range(S)])
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[fmt_bits]
    got = G.hip_render(mx, 2, x, frame_size=fs, fmt=fmt, limiter=True, flush=True, frames_per_call=[4, 1, 4])
    omx = O.get_h2m(3, O.SS["BINAURAL"])
    for s in range(S):
        want = O.stream_run(omx, 2, x[s], fs, bit_depth=fmt_bits)
        assert got[s].shape == want.shape
        assert np.array_equal(got[s], want), (s, fmt_bits)


@pytest.mark.parametrize("fmt_bits", [24, 32])
@pytest.mark.parametrize("lay,in_id,m", [("B", "L51", 6), ("J", "L714", 12), ("E", "L714", 12), ("G", "L714", 12), ("H", None, 16), ("L312", "L51", 6)])
def test_wide_layouts_in_24_and_32_bit_vs_oracle(hip, fmt_bits, lay, in_id, m):
    """s24 / s32 PCM of the layouts with more than two channels (render_wide_kernel: a lane stores 12- / 16-byte pieces of
    the chunk's contiguous run; until round 4 its own sample-frame element by element): 6, 11, 12, 14, 24 channels, the
    first call (240 withheld samples), calls of 1 / 3 / 2 frames, frame sizes 1024 and 960, the flush."""
    A, G = hip
    oc = {"B": 6, "J": 12, "E": 11, "G": 14, "H": 24, "L312": 6}[lay]
    for fs in (1024, 960):
        F, S = 6, 3
        x = np.stack([synth.hot(4100 + 7 * s + oc, m, F * fs, sigma=0.3, burst_phase=500 + 90 * s, burst_period=3000) for s in range(S)])
        mx = A.get_h2m_matrix(3, A.SS[lay]) if in_id is None else A.get_m2m_matrix(A.SS[in_id], A.SS[lay])
        omx = O.get_h2m(3, O.SS[lay]) if in_id is None else O.get_m2m(O.SS[in_id], O.SS[lay])
        fmt = {24: A.FMT_S24, 32: A.FMT_S32}[fmt_bits]
        got = G.hip_render(mx, oc, x, frame_size=fs, fmt=fmt, limiter=True, flush=True, frames_per_call=[1, 3, 2],
                           projection=A.PROJ_EXACT)
        for s in range(S):
            want = O.stream_run(omx, oc, x[s], fs, bit_depth=fmt_bits)
            assert got[s].shape == want.shape
            assert np.array_equal(got[s], want), (lay, fs, s, fmt_bits)


@pytest.mark.parametrize("fmt_bits", [16, 24, 32])
@pytest.mark.parametrize("lay,in_id,m", [("A", "STEREO", 2), ("A", None, 16), ("B", "L51", 6), ("J", "L714", 12), ("E", "L714", 12), ("H", None, 16), ("MONO", "STEREO", 2)])
def test_limiter_off_layouts_and_formats_vs_oracle(hip, fmt_bits, lay, in_id, m):
    """the player's -disable_limiter (IAMF_decoder_peak_limiter_enable(h, 0)): render_nolim_kernel<M> where its shape fits
    (4 samples per lane, the tile packed in LDS, one contiguous run of 16-byte pieces), the general kernel elsewhere (11
    channels in 16 / 24 bit, 24 channels in 24 / 32 bit) — every sample is emitted at once, no delay; gains and loudness on,
    calls of 1 / 3 / 2 frames, frame sizes 1024, 960 and 100 (a last chunk that is not full)."""
    A, G = hip
    oc = {"A": 2, "B": 6, "J": 12, "E": 11, "H": 24, "MONO": 1}[lay]
    for fs in (1024, 960, 100):
        F, S = 6, 3
        x = np.stack([synth.hot(5200 + 11 * s + oc, m, F * fs, sigma=0.4, burst_phase=300 + 70 * s, burst_period=2000) for s in range(S)])
        mx = A.get_h2m_matrix(3, A.SS[lay]) if in_id is None else A.get_m2m_matrix(A.SS[in_id], A.SS[lay])
        omx = O.get_h2m(3, O.SS[lay]) if in_id is None else O.get_m2m(O.SS[in_id], O.SS[lay])
        fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[fmt_bits]
        eg, og, lg = [0.8, 1.0, 1.25], [1.0, 0.6, 1.0], [O.lib().orc_db2lin(-2.0), 1.0, O.lib().orc_db2lin(1.5)]
        got = G.hip_render(mx, oc, x, frame_size=fs, fmt=fmt, limiter=False, flush=True, frames_per_call=[1, 3, 2], loudness=True,
                           gains=dict(element=eg, output=og, loudness=lg), projection=A.PROJ_EXACT)
        for s in range(S):
            want = O.stream_run(omx, oc, x[s], fs, bit_depth=fmt_bits, limiter_on=0, element_gain=eg[s], output_gain=og[s],
                                loudness_on=1, loudness_gain=lg[s])
            assert got[s].shape == want.shape and want.shape[:2] == (F * fs, oc)
            assert np.array_equal(got[s], want), (lay, fs, s, fmt_bits)


def test_pipeline_gains_and_loudness_vs_oracle(hip):
    A, G = hip
    S, fs, F = 3, 960, 4
    x = np.stack([synth.hot(2000 + s, 12, F * fs, sigma=0.2, burst_phase=300, burst_period=2500) for s in range(S)])
    mx = A.get_m2m_matrix(A.SS["L714"], A.SS["J"])
    eg = [0.7, 1.0, 1.3]
    og = [1.0, 0.5, -1.0]  # the negative one must be ignored like the reference does
    lg = [O.lib().orc_db2lin(-3.0), 1.0, O.lib().orc_db2lin(2.5)]
    got = G.hip_render(mx, 12, x, frame_size=fs, limiter=True, flush=True, loudness=True,
                       gains=dict(element=eg, output=og, loudness=lg))
    omx = O.get_m2m(O.SS["L714"], O.SS["J"])
    for s in range(S):
        want = O.stream_run(omx, 12, x[s], fs, element_gain=eg[s], output_gain=og[s], loudness_on=1,
                            loudness_gain=lg[s])
        assert np.array_equal(got[s], want), s


def test_pipeline_toa_H_24ch_vs_oracle(hip):
    """cfg3: TOA -> Sound System H (22 feeds in 24 slots, LFE1 zero, slot 23 silent)."""
    A, G = hip
    fs, F = 1024, 3
    x = synth.gaussian(13, 16, F * fs, 0.15)[None]
    mx = A.get_h2m_matrix(3, A.SS["H"])
    got = G.hip_render(mx, 24, x, frame_size=fs, limiter=True, flush=True, projection=A.PROJ_EXACT)[0]
    want = O.stream_run(O.get_h2m(3, O.SS["H"]), 24, x[0], fs)
    assert np.array_equal(got, want)
    assert np.all(got[:, 3] == 0) and np.all(got[:, 23] == 0) and np.any(got[:, 22] != 0)


def test_limiter_off_emits_every_sample(hip):
    A, G = hip
    x = synth.uniform(5, 2, 2048, 0.9)[None]
    mx = A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"])
    got = G.hip_render(mx, 2, x, frame_size=1024, limiter=False, flush=True)[0]
    want = O.stream_run(O.get_m2m(O.SS["STEREO"], O.SS["A"]), 2, x[0], 1024, limiter_on=0)
    assert got.shape == (2048, 2) and np.array_equal(got, want)


def test_full_size_batch_properties(hip):
    """BASELINE cfg5 per-GPU shard shape: 512 streams x 1024-sample frames.  The oracle checks a
    spread of streams exactly; size-independent properties cover the rest: identical inputs give
    identical PCM, the limiter bounds the peak, nothing is emitted twice."""
    A, G = hip
    S, fs, F = 512, 1024, 8
    base = np.stack([synth.hot(1000 + s, 16, F * fs, burst_phase=700 + 13 * s, burst_period=5000)
                     for s in range(8)])
    x = base[np.arange(S) % 8]
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    got = G.hip_render(mx, 2, x, frame_size=fs, limiter=True, flush=True, frames_per_call=[3, 5])
    omx = O.get_h2m(3, O.SS["BINAURAL"])
    want = [O.stream_run(omx, 2, base[s], fs) for s in range(8)]
    # the limiter is a smoother, not a brick wall: the reference itself overshoots the threshold
    # by a few LSB while the attack converges, so the bound carries 0.1 % of slack
    thr_lsb = 1.001 * 32768 * 10 ** (-1 / 20) + 1
    for s in range(S):
        assert got[s].shape == (F * fs, 2)
        assert np.array_equal(got[s], want[s % 8]), s
        assert np.abs(got[s].astype(np.int32)).max() <= thr_lsb
