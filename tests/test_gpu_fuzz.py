"""-m gpu: seeded random configurations of the batch renderer against the oracle, bit for bit:
random source / output layouts, frame sizes, call partitions, bit depths, limiter on/off and
thresholds, sample rates, gains (incl. the values the reference ignores), loudness, stream counts.
Every case goes through whichever kernel the dispatcher picks (fast / wide4 / wide / generic) and
usually through several of them across its calls."""
import os

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

H2M_OUT = ["A", "B", "C", "D", "E", "F", "G", "H", "I", "J", "L712", "L312", "BINAURAL"]
M2M_IN = ["MONO", "STEREO", "L51", "L512", "L514", "L71", "L712", "L714", "L312"]
IN_CH = dict(MONO=1, STEREO=2, L51=6, L512=8, L514=10, L71=8, L712=10, L714=12, L312=6)


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G


def _partition(rng, total):
    parts = []
    while total:
        n = int(rng.integers(1, min(total, 3) + 1))
        parts.append(n)
        total -= n
    return parts


# IAMF_FUZZ_SEEDS=<n> widens both sweeps for a soak run (the defaults keep the suite short)
_N_SEEDS = int(os.environ.get("IAMF_FUZZ_SEEDS", "0"))


@pytest.mark.parametrize("seed", range(_N_SEEDS or 40))
def test_random_configuration_matches_oracle(hip, seed):
    A, G = hip
    rng = np.random.default_rng(9000 + seed)
    # renderer
    for _ in range(50):
        out = H2M_OUT[int(rng.integers(len(H2M_OUT)))]
        if rng.random() < 0.5:
            order = int(rng.integers(0, 4))
            m = (order + 1) ** 2
            try:
                mx, omx = A.get_h2m_matrix(order, A.SS[out]), O.get_h2m(order, O.SS[out])
            except KeyError:
                continue
        else:
            src = M2M_IN[int(rng.integers(len(M2M_IN)))]
            m = IN_CH[src]
            try:
                mx, omx = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out])
            except KeyError:
                continue
        break
    else:
        pytest.skip("no matrix drawn")
    ch = A.layout_channels(A.SS[out])
    fs = int(rng.choice([128, 256, 480, 960, 1024, 1024, 2048]))
    F = int(rng.integers(2, 7))
    S = int(rng.integers(1, 4))
    bits = int(rng.choice([16, 16, 24, 32]))
    fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[bits]
    limiter = bool(rng.random() < 0.8)
    thr = float(rng.choice([-1.0, -1.0, -3.0, -0.1]))
    rate = int(rng.choice([48000, 48000, 44100, 32000, 16000]))
    loud = bool(rng.random() < 0.3)
    pick = lambda: float(rng.choice([1.0, 1.0, 0.5, 1.7, -1.0, 0.0]))
    eg, og = [pick() for _ in range(S)], [pick() for _ in range(S)]
    lg = [float(rng.choice([1.0, 0.6, 1.4])) for _ in range(S)]
    sigma = float(rng.choice([0.05, 0.2, 0.3]))
    x = np.stack([synth.hot(seed * 10 + s, m, F * fs, sigma=sigma, burst_phase=int(rng.integers(0, fs)),
                            burst_period=int(rng.integers(900, 4000))) for s in range(S)])
    got = G.hip_render(mx, ch, x, frame_size=fs, fmt=fmt, limiter=limiter, flush=limiter,
                       frames_per_call=_partition(rng, F), gains=dict(element=eg, output=og, loudness=lg),
                       loudness=loud, threshold_db=thr, sample_rate=rate, projection=A.PROJ_EXACT)
    for s in range(S):
        want = O.stream_run(omx, ch, x[s], fs, flush=limiter, element_gain=eg[s], output_gain=og[s],
                            loudness_on=int(loud), loudness_gain=lg[s], limiter_on=int(limiter), thr_db=thr,
                            rate=rate, bit_depth=bits)
        assert got[s].shape == want.shape, (seed, s, got[s].shape, want.shape)
        assert np.array_equal(got[s], want), (seed, s)


@pytest.mark.parametrize("seed", range(_N_SEEDS or 30))
def test_random_mix_configuration_matches_oracle(hip, seed):
    """random mix presentations: a bed, optionally a second element (1..16 channels), optionally
    per-sample element / output gain ramps; mono / stereo / binaural outputs take the mixing variant
    of the fast kernel when the second element has <= 4 channels, everything else the generic kernel"""
    import torch
    from test_gpu_extras import _run_ex
    A, G = hip
    rng = np.random.default_rng(7000 + seed)
    outs = ["A", "BINAURAL", "MONO", "A", "BINAURAL", "B", "J"]
    for _ in range(50):
        out = outs[int(rng.integers(len(outs)))]
        try:
            if rng.random() < 0.5:
                order = int(rng.integers(1, 4))
                m = (order + 1) ** 2
                mx, omx = A.get_h2m_matrix(order, A.SS[out]), O.get_h2m(order, O.SS[out])
            else:
                src = M2M_IN[int(rng.integers(len(M2M_IN)))]
                m = IN_CH[src]
                mx, omx = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out])
            mx2 = omx2 = None
            m2 = 0
            if rng.random() < 0.7:
                src2 = ["MONO", "STEREO", "FOA", "L51", "TOA"][int(rng.integers(5))]
                if src2 in ("FOA", "TOA"):
                    o2 = 1 if src2 == "FOA" else 3
                    m2 = (o2 + 1) ** 2
                    mx2, omx2 = A.get_h2m_matrix(o2, A.SS[out]), O.get_h2m(o2, O.SS[out])
                else:
                    m2 = IN_CH[src2]
                    mx2, omx2 = A.get_m2m_matrix(A.SS[src2], A.SS[out]), O.get_m2m(O.SS[src2], O.SS[out])
        except (KeyError, AssertionError):
            continue
        break
    else:
        pytest.skip("no matrix drawn")
    ch = A.layout_channels(A.SS[out])
    fs = int(rng.choice([256, 960, 1024, 1024, 2048]))
    F = int(rng.integers(2, 6))
    S = int(rng.integers(1, 4))
    bits = int(rng.choice([16, 16, 24, 32]))
    fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[bits]
    n = F * fs
    pick = lambda: float(rng.choice([1.0, 0.5, 1.3, -1.0, 0.0]))
    eg, og, eg2 = [pick() for _ in range(S)], [pick() for _ in range(S)], [pick() for _ in range(S)]
    ramp = lambda: (0.4 + 0.8 * rng.random((S, n))).astype(np.float32)
    ramps = {}
    if rng.random() < 0.4:
        ramps["element"] = ramp()
    if m2 and rng.random() < 0.3:
        ramps["element2"] = ramp()
    if rng.random() < 0.4:
        ramps["output"] = ramp()
    x0 = np.stack([synth.hot(seed * 10 + s, m, n, sigma=0.2, burst_phase=int(rng.integers(0, fs)),
                             burst_period=int(rng.integers(900, 4000))) for s in range(S)])
    x1 = np.stack([synth.hot(seed * 10 + 5 + s, max(m2, 1), n, sigma=0.2, burst_phase=int(rng.integers(0, fs)),
                             burst_period=int(rng.integers(900, 4000))) for s in range(S)]) if m2 else None
    b = A.Batch(S, mx, ch, frame_size=fs, out_format=fmt, projection=A.PROJ_EXACT)
    b.set_gains(element=eg, output=og)
    if m2:
        b.set_second_element(mx2, eg2)
    got = _run_ex(A, G, torch, b, S, m, x0, fs, ch, fmt, x2=x1, m2=m2, ramps=ramps or None, calls=_partition(rng, F))
    b.close()
    f32 = np.float32
    for s in range(S):
        y = O.render(omx, x0[s], ch)[:ch]
        if "element" in ramps:
            y = (y * ramps["element"][s][None, :]).astype(f32)
        elif eg[s] != 1.0 and eg[s] > 0:
            y = (y * f32(eg[s])).astype(f32)
        z = (np.zeros_like(y) + y).astype(f32)
        if m2:
            y2 = O.render(omx2, x1[s], ch)[:ch]
            if "element2" in ramps:
                y2 = (y2 * ramps["element2"][s][None, :]).astype(f32)
            elif eg2[s] != 1.0 and eg2[s] > 0:
                y2 = (y2 * f32(eg2[s])).astype(f32)
            z = (z + y2).astype(f32)
        if "output" in ramps:
            z = (z * ramps["output"][s][None, :]).astype(f32)
        elif og[s] != 1.0 and og[s] > 0:
            z = (z * f32(og[s])).astype(f32)
        z, _ = O.limiter_run(np.ascontiguousarray(z), [fs] * F)
        want = O.pack(z, bits)
        assert got[s].shape == want.shape, (seed, s, got[s].shape, want.shape)
        assert np.array_equal(got[s], want), (seed, s)


_DEMIX_STACKS = [[1, 3, 7], [0, 1, 2, 5], [1, 8], [2, 4], [3, 4], [8, 3, 6], [7], [1, 2, 4], [0, 1, 2], [1, 2, 5], [2, 5],
                 [1, 3, 6], [3, 7], [2, 3, 7], [1, 2, 3, 4], [0, 1, 8, 3, 7], [5], [4], [1, 2, 3, 6]]


@pytest.mark.parametrize("seed", range(_N_SEEDS or 24))
def test_random_demixer_configuration_matches_oracle(hip, seed):
    """random scalable stacks, output gains, demixing-mode schedules, recon gains, frame sizes and call
    partitions: decoded layers -> demixer -> layout matrix -> limiter -> s16 (wide4 demixer variant
    where it applies, else the generic kernel) against the oracle demixer + renderer chain"""
    import demix_cases as D
    from test_gpu_wide4 import _demix_render, _LAYOUT_SS
    A, G = hip
    rng = np.random.default_rng(5000 + seed)
    layers = _DEMIX_STACKS[int(rng.integers(len(_DEMIX_STACKS)))]
    if layers[-1] < 2:
        layers = [1, 3, 7]
    gains = {}
    for li in range(len(layers)):
        if rng.random() < 0.4:
            gains[li] = (int(rng.integers(1, 64)), float(np.float32(10 ** (float(rng.integers(-6, 7)) / 20))))
    fs = int(rng.choice([256, 512, 960, 1024]))
    offset = int(rng.choice([0, 0, 4, 8, 37])) if fs >= 512 else 0
    try:
        c = D.make_case(layers, gains, default=(int(rng.choice([0, 1, 2, 4, 5, 6])), int(rng.integers(0, 11))),
                        offset=offset, fs=fs, seed=900 + seed)
    except Exception:
        pytest.skip("stack not accepted by the case builder")
    F = len(c["schedule"])
    ch = len(c["order"])
    S = int(rng.integers(1, 3))
    x = np.stack([np.stack([synth.uniform(c["seed"] + 100 * s + f, ch, fs, 0.8) for f in range(F)]) for s in range(S)])
    dem = [D.drive_demixer(O.lib(), "orc_demixer_", c, x[s]) for s in range(S)]
    src = _LAYOUT_SS[c["layout"]]
    out = str(rng.choice(["J", "B", "D", "A", "H", "id"]))
    if out == "id":
        mx, omx, och = G.identity_matrix(ch), None, ch
    else:
        try:
            mx, omx = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out])
        except (KeyError, AssertionError):
            mx, omx, och = G.identity_matrix(ch), None, ch
            out = "id"
        else:
            och = A.layout_channels(A.SS[out])
    calls = _partition(rng, F)
    got = _demix_render(A, c, mx, och, x, calls)
    for s in range(S):
        xd = np.ascontiguousarray(dem[s].transpose(1, 0, 2).reshape(ch, F * fs))
        if omx is None:
            z, _ = O.limiter_run(xd, [fs] * F)
            want = O.pack(z, 16)
        else:
            want = O.stream_run(omx, och, xd, fs)
        assert got[s].shape == want.shape, (seed, layers, out, s)
        assert np.array_equal(got[s], want), (seed, layers, out, s)


LFE_OUT = ["B", "C", "D", "F", "H", "I", "J", "L712", "L312"]


@pytest.mark.parametrize("seed", range(_N_SEEDS or 24))
def test_random_lfe_configuration_matches_oracle(hip, seed):
    """the same sweep with the HOA LFE generator on: ambisonics of order 1..3 to a layout with an LFE
    channel, the generator's filter designed for the drawn sample rate, its state carried over the call partition.
    The cases that are 16-bit, limited and chunk-aligned take render_wide4_kernel<.., LFE>, the others the generic
    kernel."""
    A, G = hip
    rng = np.random.default_rng(3000 + seed)
    out = LFE_OUT[int(rng.integers(len(LFE_OUT)))]
    order = int(rng.integers(1, 4))
    m = (order + 1) ** 2
    mx, omx = A.get_h2m_matrix(order, A.SS[out]), O.get_h2m(order, O.SS[out])
    assert mx.lfe1 >= 0
    ch = A.layout_channels(A.SS[out])
    fs = int(rng.choice([256, 480, 960, 1024, 1024, 1024, 2048]))
    F = int(rng.integers(2, 9))
    S = int(rng.integers(1, 4))
    bits = int(rng.choice([16, 16, 16, 24, 32]))
    fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[bits]
    limiter = bool(rng.random() < 0.85)
    thr = float(rng.choice([-1.0, -1.0, -3.0]))
    rate = int(rng.choice([48000, 48000, 44100, 96000, 16000]))
    pick = lambda: float(rng.choice([1.0, 1.0, 0.5, 1.7]))
    eg, og = [pick() for _ in range(S)], [pick() for _ in range(S)]
    x = np.stack([synth.hot(seed * 10 + s, m, F * fs, sigma=float(rng.choice([0.1, 0.3])), burst_phase=int(rng.integers(0, fs)),
                            burst_period=int(rng.integers(900, 4000))) for s in range(S)])
    t = np.arange(F * fs, dtype=np.float64) / rate
    x[:, 0] += (0.3 * np.sin(2 * np.pi * 60.0 * t)).astype(np.float32)    # something for the 120 Hz low-pass to pass
    got = G.hip_render(mx, ch, x, frame_size=fs, fmt=fmt, limiter=limiter, flush=limiter, frames_per_call=_partition(rng, F),
                       gains=dict(element=eg, output=og, loudness=[1.0] * S), threshold_db=thr, sample_rate=rate,
                       projection=A.PROJ_EXACT, lfe_hoa=True)
    for s in range(S):
        want = O.stream_run(omx, ch, x[s], fs, flush=limiter, element_gain=eg[s], output_gain=og[s], limiter_on=int(limiter),
                            thr_db=thr, rate=rate, bit_depth=bits, lfe_rate=rate)
        assert got[s].shape == want.shape, (seed, out, order, s, got[s].shape, want.shape)
        assert np.array_equal(got[s], want), (seed, out, order, fs, F, bits, limiter, rate, s)
    assert any(np.any(g[..., mx.lfe1, :] if g.ndim == 3 else g[:, mx.lfe1]) for g in got)
