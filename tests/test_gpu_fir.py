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
                                             (1, 256, 1024, [2]), (16, 64, 2048, [1, 1]),
                                             # calls longer than one 4096-sample pass of the stage (the last pass partial),
                                             # and the shortest filters (one and two K steps of 32 taps)
                                             (16, 256, 1024, [9, 5]), (9, 200, 960, [11]), (4, 1, 1024, [5]),
                                             (9, 17, 1024, [5, 1]), (4, 18, 1024, [6])])
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


@pytest.mark.parametrize("m,taps,fs,calls", [(2, 256, 1024, [2, 1]), (6, 128, 1024, [3]), (8, 256, 960, [1, 2]),
                                             (10, 200, 1024, [2]), (12, 256, 1024, [1, 1, 2]), (12, 256, 1024, [10, 3])])
def test_m2b_channel_based_element_matches_float64_convolution(hip, m, taps, fs, calls):
    """M2B (the role of m2b_rdr.c:103-121): a channel-based element of 2 / 6 / 8 / 10 / 12 loudspeaker
    channels through one HRIR pair per loudspeaker; the LFE's pair is zero where the layout has one"""
    A, G = hip
    F = sum(calls)
    S = 2
    x = np.stack([synth.gaussian(820 + s, m, F * fs, 0.1) for s in range(S)])
    h = hrir_set(7, m, taps)
    if m >= 6:
        h[:, 3, :] = 0.0   # playback order: L R C LFE ...
    got = G.hip_render(A.fir_matrix(h), 2, x, frame_size=fs, fmt=A.FMT_F32, limiter=True, flush=True,
                       frames_per_call=calls, fir_taps=taps)
    for s in range(S):
        y = fir64(h, x[s])
        assert np.abs(y).max() < 0.85
        assert got[s].shape == (F * fs, 2)
        assert np.abs(got[s].T - y).max() <= F32_TOL, (m, s, float(np.abs(got[s].T - y).max()))


def _fir_stage_output(A, G, h, x, fs, calls, taps):
    """what the FIR stage hands to the limiter: the same call with a threshold nothing reaches
    (+60 dB: the gain stays exactly 1.0f, y * 1.0f == y), float output"""
    return G.hip_render(A.fir_matrix(h), 2, x, frame_size=fs, fmt=A.FMT_F32, limiter=True, flush=True,
                        frames_per_call=calls, fir_taps=taps, threshold_db=60.0)


@pytest.mark.parametrize("m", [16, 12])
def test_fir_pipeline_with_limiter(hip, m):
    """hot programme: HRTF -> limiter -> int16, WITHOUT a tolerance on the pipeline.  The FIR stage is the
    only unpinned, inexact part (checked against float64 above and below); everything behind it is the
    reference's arithmetic, so the oracle's limiter + pack run on the kernel's OWN FIR output must
    reproduce the kernel's PCM bit for bit.  (The 8-LSB allowance this test used to have compared against
    a float64 convolution: a last-bit difference in one FIR sample can flip a trigger decision of the
    limiter, a feedback system, and move the following 200 ms of gains.)"""
    A, G = hip
    fs, F = 1024, 6
    x = synth.hot(900, m, F * fs, sigma=0.2, burst_amp=1.2, burst_phase=700, burst_period=2500)[None]
    h = hrir_set(6, m, 256)
    calls = [2, 4]
    got = G.hip_render(A.fir_matrix(h), 2, x, frame_size=fs, fmt=A.FMT_S16, limiter=True, flush=True,
                       frames_per_call=calls, fir_taps=256)[0]
    y_dev = _fir_stage_output(A, G, h, x, fs, calls, 256)[0]        # [n][2] f32
    y64 = fir64(h, x[0])
    assert np.abs(y64).max() > 1.0  # the limiter has work to do
    assert np.abs(y_dev.T - y64).max() <= F32_TOL * max(1.0, float(np.abs(y64).max()))
    z, _ = O.limiter_run(np.ascontiguousarray(y_dev.T), [fs] * F)
    want = O.pack(z, 16)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.abs(got.astype(np.int32)).max() <= 1.001 * 32768 * 10 ** (-1 / 20) + 1


def test_split_f16_stage_against_the_f32_mfma_stage(hip, monkeypatch):
    """the split-f16 stage (three f16 MFMAs per product block on split operands, render_fir16.hpp; the default until
    round 3, now IAMF_HIP_FIR_F16=1) against
    the f32-MFMA stage (render_fir.hpp, IAMF_HIP_FIR_F32=1) on the same input: two independent
    evaluations of the same sums.  Both within 2^-17 of float64; their mutual difference as a histogram in
    units of 2^-24 (the f32 ulp of a value in [0.5, 1))."""
    A, G = hip
    fs, F, m, taps = 1024, 4, 16, 256
    x = np.stack([synth.gaussian(860 + s, m, F * fs, 0.12) for s in range(2)])
    h = hrir_set(8, m, taps)
    monkeypatch.setenv("IAMF_HIP_FIR_F16", "1")
    y16 = _fir_stage_output(A, G, h, x, fs, [F], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_F16")
    monkeypatch.setenv("IAMF_HIP_FIR_F32", "1")
    y32 = _fir_stage_output(A, G, h, x, fs, [F], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_F32")
    for s in range(2):
        y64 = fir64(h, x[s]).T
        e16, e32 = np.abs(y16[s] - y64).max(), np.abs(y32[s] - y64).max()
        d = np.abs(y16[s].astype(np.float64) - y32[s].astype(np.float64)) * 2.0 ** 24
        hist = np.histogram(d, bins=[0, 0.5, 1.5, 2.5, 4.5, 8.5, 1e9])[0]
        print("stream %d: |f16-f64| max %.3g, |f32-f64| max %.3g, |f16-f32| in 2^-24 units: 0:%d 1:%d 2:%d 3-4:%d 5-8:%d >8:%d"
              % ((s, e16, e32) + tuple(int(v) for v in hist)))
        # measured on MI355X (16 ch x 256 taps, |y| up to 0.5): both stages 6.4e-7 / 7.9e-7 from float64 — the
        # f32 rounding of a 4096-term sum — and from each other 0:17% 1:35% 2:23% 3-4:19% 5-8:6% >8:0.4% (max < 16)
        assert e16 <= 2.0 ** -19 and e32 <= 2.0 ** -19   # 1/4 of the stated float tolerance 2^-17
        assert d.max() <= 16.0, float(d.max())           # 2^-20 absolute between the two stages
        assert (d <= 4.5).mean() > 0.90 and (d <= 8.5).mean() > 0.99


def test_fft_stage_against_the_two_mfma_stages(hip, monkeypatch):
    """the default stage since round 3 — overlap-save in the frequency domain on the VALU (render_fir_fft.hpp: 1024-point
    transforms, 768-sample hops, two channels per complex transform, both ears per inverse) — against the two direct-form
    stages on the matrix cores: three independent evaluations of one specification.  All within 2^-19 of float64 at
    |y| <= 0.5; the FFT stage's error is the f32 rounding of a 1024-point transform pair instead of a 4096-term sum."""
    A, G = hip
    fs, F, m, taps = 1024, 7, 16, 256      # 7 frames: two full passes of 3072 samples and a partial one
    x = np.stack([synth.gaussian(870 + s, m, F * fs, 0.12) for s in range(2)])
    h = hrir_set(9, m, taps)
    yf = _fir_stage_output(A, G, h, x, fs, [3, 4], taps)
    # the same stage fused into the limiter kernel (IAMF_HIP_FIR_FUSED=1) instead of running as a kernel of its own in
    # front of the two-channel matrix kernel: the same arithmetic, the same floats
    monkeypatch.setenv("IAMF_HIP_FIR_FUSED", "1")
    yfused = _fir_stage_output(A, G, h, x, fs, [3, 4], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_FUSED")
    for s in range(2):
        assert np.array_equal(yf[s], yfused[s]), s
    # the stage kernel's two-base sample fetch (frame sizes that are multiples of 256 from 1024 on, an even channel count;
    # the history kept at the input's channel stride) against its general per-run fetch: the same floats
    monkeypatch.setenv("IAMF_HIP_FIR_GENERAL_FETCH", "1")
    ygen = _fir_stage_output(A, G, h, x, fs, [3, 4], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_GENERAL_FETCH")
    for s in range(2):
        assert np.array_equal(yf[s], ygen[s]), s
    monkeypatch.setenv("IAMF_HIP_FIR_F16", "1")
    y16 = _fir_stage_output(A, G, h, x, fs, [3, 4], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_F16")
    monkeypatch.setenv("IAMF_HIP_FIR_F32", "1")
    y32 = _fir_stage_output(A, G, h, x, fs, [3, 4], taps)
    monkeypatch.delenv("IAMF_HIP_FIR_F32")
    for s in range(2):
        y64 = fir64(h, x[s]).T
        ef, e16, e32 = (float(np.abs(y - y64).max()) for y in (yf[s], y16[s], y32[s]))
        print("stream %d: |fft-f64| max %.3g, |f16-f64| max %.3g, |f32-f64| max %.3g, |fft-f32| max %.3g"
              % (s, ef, e16, e32, float(np.abs(yf[s] - y32[s]).max())))
        assert ef <= 2.0 ** -19 and e16 <= 2.0 ** -19 and e32 <= 2.0 ** -19
        assert not np.array_equal(yf[s], y16[s])   # it really is another stage that ran


@pytest.mark.parametrize("fs,m,S", [(2048, 12, 11), (1024, 2, 9), (1280, 4, 7), (1024, 16, 5)])
def test_two_base_fetch_slabs_and_frame_sizes(hip, monkeypatch, fs, m, S):
    """the history of G = frame size / 256 streams shares a slab (stream counts that are not multiples of G; G = 8, 4, 5),
    calls of one, two and five frames so that windows lie inside a frame, cross a frame boundary, start before the call and
    end past it: the two-base fetch must give the floats of the general fetch, and both the float64 convolution"""
    A, G = hip
    taps, calls = 256, [1, 2, 5, 1]
    F = sum(calls)
    x = np.stack([synth.gaussian(990 + s, m, F * fs, 0.1) for s in range(S)])
    h = hrir_set(13, m, taps)
    y2 = _fir_stage_output(A, G, h, x, fs, calls, taps)
    monkeypatch.setenv("IAMF_HIP_FIR_GENERAL_FETCH", "1")
    yg = _fir_stage_output(A, G, h, x, fs, calls, taps)
    monkeypatch.delenv("IAMF_HIP_FIR_GENERAL_FETCH")
    for s in range(S):
        assert np.array_equal(y2[s], yg[s]), s
        assert np.abs(y2[s].T - fir64(h, x[s])).max() <= 2.0 ** -19, s


@pytest.mark.parametrize("calls", [[24, 3, 37], [10, 10]])
def test_many_streams_long_calls_hot_programme(hip, calls):
    """64 streams, calls of 24 / 3 / 37 frames (eight passes of the stage kernel, one, and a partial last one), a programme
    the limiter works on: the oracle's limiter + pack run on the stage's own output must reproduce the PCM bit for bit —
    the stage kernel and the two-channel matrix kernel behind it see the same y."""
    A, G = hip
    fs, m, taps, S = 1024, 16, 256, 64
    F = sum(calls)
    x = np.stack([synth.hot(930 + s, m, F * fs, sigma=0.2, burst_amp=1.2, burst_phase=300 + 37 * s, burst_period=2500)
                  for s in range(S)])
    h = hrir_set(11, m, taps)
    kw = dict(frame_size=fs, limiter=True, flush=True, frames_per_call=calls, fir_taps=taps)
    got = G.hip_render(A.fir_matrix(h), 2, x, fmt=A.FMT_S16, **kw)
    y_dev = G.hip_render(A.fir_matrix(h), 2, x, fmt=A.FMT_F32, threshold_db=60.0, **kw)
    for s in (0, 31, 63):
        z, _ = O.limiter_run(np.ascontiguousarray(y_dev[s].T), [fs] * F)
        assert np.array_equal(got[s], O.pack(z, 16)), s
        assert np.abs(y_dev[s]).max() > 1.0


def test_a_call_that_ends_inside_a_frame_with_nans_behind_it(hip):
    """a trimmed last frame (n_samples = 512 of 1024): what the caller left in the rest of the frame buffer — here NaN — must
    not reach the output.  A transform spreads one NaN over its whole block, so the stage has to take zeros past the end
    of the call (its general fetch does; the two-base fetch is for whole frames only and must not be chosen here)."""
    import torch
    A, G = hip
    fs, m, taps, S = 1024, 4, 256, 3
    x = np.stack([synth.gaussian(1700 + s, m, 3 * fs, 0.1) for s in range(S)])
    h = hrir_set(17, m, taps)
    keep = 2 * fs + 512
    xin = G.to_frames(x, fs).copy()                       # [S][3][m][fs]
    xin.reshape(S, 3, m, fs)[:, 2, :, 512:] = np.nan
    d_in = torch.from_numpy(xin).cuda()
    b = A.Batch(S, A.fir_matrix(h), 2, frame_size=fs, out_format=A.FMT_F32, limiter=True, threshold_db=60.0, fir_taps=taps)
    st = torch.cuda.current_stream().cuda_stream
    outs = [[] for _ in range(S)]
    cap = 2 * fs * 2 * 4
    for f0, nf, ns in ((0, 2, 0), (2, 1, 512)):
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = d_in.data_ptr() + 4 * f0 * m * fs, 3 * m * fs, m * fs
        a.n_frames, a.n_samples, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = nf, ns, pcm.data_ptr(), cap, st
        n = b.render_ex(a)
        torch.cuda.synchronize()
        hp = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(hp[s][:n * 2 * 4].view(np.float32).reshape(n, 2).copy())
    pcm = torch.zeros((S, 240 * 2 * 4), dtype=torch.uint8, device="cuda")
    n = b.flush(pcm.data_ptr(), 240 * 2 * 4, st)
    torch.cuda.synchronize()
    hp = pcm.cpu().numpy()
    b.close()
    for s in range(S):
        y = np.concatenate(outs[s] + [hp[s][:n * 2 * 4].view(np.float32).reshape(n, 2)], axis=0)
        assert y.shape == (keep, 2)
        assert np.isfinite(y).all(), s
        assert np.abs(y.T - fir64(h, x[s][:, :keep])).max() <= 2.0 ** -19, s


def _fir_mixed_calls(A, G, h, x, fs, calls, taps):
    """calls = [(n_frames, n_samples)]: n_samples > 0 = a call that ends inside its (single) frame.  x is the stream's
    programme as the renderer consumes it (kept samples back to back); what lies behind a partial call's samples in its
    frame buffer is NaN.  Returns the stage's f32 output [S][kept][2] (limiter threshold +60 dB: gains exactly 1)."""
    import torch
    S, m = x.shape[0], x.shape[1]
    b = A.Batch(S, A.fir_matrix(h), 2, frame_size=fs, out_format=A.FMT_F32, limiter=True, threshold_db=60.0, fir_taps=taps)
    st = torch.cuda.current_stream().cuda_stream
    outs = [[] for _ in range(S)]
    pos = 0
    for nf, ns in calls:
        take = ns if ns else nf * fs
        buf = np.full((S, nf, m, fs), np.nan, np.float32)
        seg = x[:, :, pos:pos + take]
        if ns:
            buf[:, 0, :, :ns] = seg
        else:
            buf[:] = seg.reshape(S, m, nf, fs).transpose(0, 2, 1, 3)
        pos += take
        d_in = torch.from_numpy(buf).cuda()
        cap = nf * fs * 2 * 4
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = d_in.data_ptr(), nf * m * fs, m * fs
        a.n_frames, a.n_samples, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = nf, ns, pcm.data_ptr(), cap, st
        n = b.render_ex(a)
        assert n >= 0, n
        torch.cuda.synchronize()
        hp = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(hp[s][:n * 2 * 4].view(np.float32).reshape(n, 2).copy())
    pcm = torch.zeros((S, 240 * 2 * 4), dtype=torch.uint8, device="cuda")
    n = b.flush(pcm.data_ptr(), 240 * 2 * 4, st)
    torch.cuda.synchronize()
    hp = pcm.cpu().numpy()
    b.close()
    return [np.concatenate(outs[s] + [hp[s][:n * 2 * 4].view(np.float32).reshape(n, 2)], axis=0) for s in range(S)], pos


# m = 4 / 16: fir_fft_kernel<M> of iamf_render.hip (ambisonics); m = 6 / 12: the channel-based launcher (iamf_render_fir_m2b.hip)
@pytest.mark.parametrize("m", [4, 16, 6, 12])
@pytest.mark.parametrize("calls", [[(1, 512), (2, 0), (1, 960), (1, 0)], [(1, 960), (1, 0), (3, 0)], [(2, 0), (1, 64), (1, 0), (1, 512), (2, 0)]])
def test_partial_call_then_whole_frames_hands_the_history_over(hip, monkeypatch, m, calls):
    """a call that ends inside a frame runs the general fetch, the whole-frame call after it the two-base
    fetch — which takes its first hop's 256 history samples from the copy kept at the input's channel stride
    (RenderParams::fir_pre).  Every stage kernel must leave BOTH copies current: partial -> whole -> partial -> whole must
    give the floats of the general fetch throughout, and the float64 convolution of the programme."""
    A, G = hip
    fs, taps, S = 1024, 256, 5            # 5 streams: G = 4 streams per history slab, the last slab partly filled
    total = sum(ns if ns else nf * fs for nf, ns in calls)
    x = np.stack([synth.gaussian(2300 + s, m, total, 0.1) for s in range(S)])
    h = hrir_set(19, m, taps)
    y2, used = _fir_mixed_calls(A, G, h, x, fs, calls, taps)
    assert used == total
    monkeypatch.setenv("IAMF_HIP_FIR_GENERAL_FETCH", "1")
    yg, _ = _fir_mixed_calls(A, G, h, x, fs, calls, taps)
    monkeypatch.delenv("IAMF_HIP_FIR_GENERAL_FETCH")
    for s in range(S):
        assert y2[s].shape == (total, 2)
        assert np.isfinite(y2[s]).all(), s
        assert np.array_equal(y2[s], yg[s]), s
        assert np.abs(y2[s].T - fir64(h, x[s])).max() <= 2.0 ** -19, s
