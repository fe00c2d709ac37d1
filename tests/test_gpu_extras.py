"""-m gpu: the parametric down-mixer, two-element mixing and per-sample mix-gain ramps on the HIP
path (generic kernel), bit-exact against reference goldens / the oracle."""
import ctypes as C

import numpy as np
import pytest

import e2e_cases
import e2e_model
import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G, torch


def _run_ex(A, G, torch, batch, S, m, x, fs, out_ch, fmt, x2=None, m2=0, ramps=None, dmx_frames=None,
            calls=None, flush=True):
    """x: [S][m][total]; returns list of per-stream outputs"""
    total = x.shape[2]
    F = total // fs
    xin = torch.from_numpy(G.to_frames(x, fs)).cuda()
    xin2 = torch.from_numpy(G.to_frames(x2, fs)).cuda() if x2 is not None else None
    bps = {A.FMT_S16: 2, A.FMT_S24: 3, A.FMT_S32: 4, A.FMT_F32: 4}[fmt]
    d_ramps = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in (ramps or {}).items()}
    d_dmx = None
    if dmx_frames is not None:
        raw = np.frombuffer(bytes(dmx_frames), dtype=np.uint8).copy()
        d_dmx = torch.from_numpy(raw).cuda()
    outs = [[] for _ in range(S)]
    st = torch.cuda.current_stream().cuda_stream
    f0 = 0
    for nf in (calls or [F]):
        cap = max(nf * fs, 240) * out_ch * bps
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in = xin.data_ptr() + 4 * f0 * m * fs
        a.in_stream_stride, a.in_frame_stride = F * m * fs, m * fs
        if xin2 is not None:
            a.d_in2 = xin2.data_ptr() + 4 * f0 * m2 * fs
            a.in2_stream_stride, a.in2_frame_stride = F * m2 * fs, m2 * fs
        for key, field in (("element", "d_element_ramp"), ("element2", "d_element2_ramp"), ("output", "d_output_ramp")):
            if key in d_ramps:
                setattr(a, field, d_ramps[key].data_ptr() + 4 * f0 * fs)
        a.ramp_stream_stride = total
        if d_dmx is not None:
            assert calls is None  # one call: frames index from 0
            a.d_dmx_frames = d_dmx.data_ptr()
        a.n_frames = nf
        a.d_pcm = pcm.data_ptr()
        a.pcm_stream_stride_bytes = cap
        a.stream = st
        n = batch.render_ex(a)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(G._view(h[s], n, out_ch, fmt))
        f0 += nf
    if flush:
        cap = 240 * out_ch * bps
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        n = batch.flush(pcm.data_ptr(), cap, st)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(G._view(h[s], n, out_ch, fmt))
    return [np.concatenate(o, axis=0) for o in outs]


def test_downmixer_goldens_bit_exact(hip, golden):
    A, G, torch = hip
    g = golden.npz("dmx")
    L = A.lib()
    for key, meta in golden.manifest.items():
        if not key.startswith("dmx/") or key.endswith("_invalid"):
            continue
        name = key.split("/")[1]
        il, ol, ns = meta["in_layout"], meta["out_layout"], meta["ns"]
        sched = [tuple(s) for s in meta["schedule"]]
        F = len(sched)
        assert L.iamf_hip_dmx_valid(il, ol) == 1
        x = np.stack([synth.uniform(meta["seed0"] + f, O.LAYOUT_CH[il], ns, 0.5) for f in range(F)])  # [F][m][ns]
        # host control plane, as IAMF_decoder.c:2574-2583 drives DMRenderer_set_mode_weight
        st = A.DmxState()
        L.iamf_hip_dmx_state_init(C.byref(st))
        L.iamf_hip_dmx_set_mode_weight(C.byref(st), meta["default_mode"], meta["default_w"])
        frames = (A.DmxFrame * F)()
        for f, (mode, off) in enumerate(sched):
            frames[f].offset = off
            L.iamf_hip_dmx_coefficients(C.byref(st), frames[f].prev)
            if mode > -1:
                L.iamf_hip_dmx_set_mode_weight(C.byref(st), mode, -1)
            L.iamf_hip_dmx_coefficients(C.byref(st), frames[f].cur)
        m, oc = O.LAYOUT_CH[il], O.LAYOUT_CH[ol]
        xs = np.ascontiguousarray(x.transpose(1, 0, 2).reshape(1, m, F * ns))
        b = A.Batch(1, A.dmx_matrix(il, ol), oc, frame_size=ns, out_format=A.FMT_F32, limiter=False)
        y = _run_ex(A, G, torch, b, 1, m, xs, ns, oc, A.FMT_F32, dmx_frames=frames, flush=False)[0]
        b.close()
        want = g[name].transpose(0, 2, 1).reshape(F * ns, oc)
        assert y.shape == want.shape, name
        assert np.array_equal(y, want), name


def test_invalid_downmix_pairs_are_refused(hip, golden):
    A, _, _ = hip
    for a, b in golden.manifest["dmx/_invalid"]:
        assert A.lib().iamf_hip_dmx_valid(a, b) == 0


def test_two_elements_match_reference_decoder(hip, golden):
    """stereo bed + 3rd-order HOA scene mixed into Sound System A, 32-bit, limiter off"""
    A, G, torch = hip
    _, info = e2e_cases.build("two_elements_A_s32")
    c = info["case"]
    e0, e1 = info["elements"]
    out_id = e2e_model.out_id_of(c["layout"])
    b = A.Batch(1, A.get_m2m_matrix(e2e_model.LAYOUT_RID[e0["layout"]], out_id), 2, frame_size=c["fs"],
                out_format=A.FMT_S32, limiter=False)
    b.set_gains(element=[e2e_model.q78_to_lin(c["element_gain_q78"])])
    b.set_second_element(A.get_h2m_matrix(3, out_id), [1.0])
    got = _run_ex(A, G, torch, b, 1, 2, e0["x"][None], c["fs"], 2, A.FMT_S32, x2=e1["x"][None], m2=16,
                  calls=[2, 3])[0]
    b.close()
    want = golden.npz("e2e")["two_elements_A_s32"]
    assert got.shape == want.shape and np.array_equal(got, want)


def test_two_elements_with_limiter_vs_oracle(hip):
    A, G, torch = hip
    fs, F = 1024, 4
    x0 = synth.hot(61, 12, F * fs, sigma=0.2, burst_phase=400, burst_period=2600)
    x1 = synth.hot(62, 16, F * fs, sigma=0.2, burst_phase=1400, burst_period=2600)
    oid = A.SS["J"]
    b = A.Batch(1, A.get_m2m_matrix(A.SS["L714"], oid), 12, frame_size=fs)
    b.set_gains(element=[0.8], output=[1.1])
    b.set_second_element(A.get_h2m_matrix(3, oid), [0.6])
    got = _run_ex(A, G, torch, b, 1, 12, x0[None], fs, 12, A.FMT_S16, x2=x1[None], m2=16)[0]
    b.close()
    y0 = O.render(O.get_m2m(O.SS["L714"], O.SS["J"]), x0, 12)
    y1 = O.render(O.get_h2m(3, O.SS["J"]), x1, 12)
    O.lib().orc_frame_gain_const(O.fp(y0), 12, F * fs, 0.8)
    O.lib().orc_frame_gain_const(O.fp(y1), 12, F * fs, 0.6)
    z = ((np.zeros_like(y0) + y0) + y1).astype(np.float32)
    O.lib().orc_frame_gain_const(O.fp(z), 12, F * fs, 1.1)
    z, _ = O.limiter_run(z, [fs] * F)
    assert np.array_equal(got, O.pack(z, 16))


@pytest.mark.parametrize("second,m2", [("STEREO", 2), ("MONO", 1), ("FOA", 4)])
@pytest.mark.parametrize("out", ["BINAURAL", "A"])
def test_two_elements_fast_kernel_vs_oracle(hip, second, m2, out):
    """3rd-order HOA bed + a mono / stereo / first-order second element into binaural / stereo on
    render_fast_kernel<16, 2, .., IN2>: element gains, mixer, output gain in the reference's order"""
    A, G, torch = hip
    fs, F, S = 1024, 4, 2
    oid = A.SS[out]
    x0 = np.stack([synth.hot(161 + s, 16, F * fs, sigma=0.2, burst_phase=400 + 100 * s, burst_period=2600) for s in range(S)])
    x1 = np.stack([synth.hot(171 + s, m2, F * fs, sigma=0.3, burst_phase=1400, burst_period=2100) for s in range(S)])
    if second == "FOA":
        mx2, omx2 = A.get_h2m_matrix(1, oid), O.get_h2m(1, O.SS[out])
    else:
        mx2, omx2 = A.get_m2m_matrix(A.SS[second], oid), O.get_m2m(O.SS[second], O.SS[out])
    eg, eg2, og = [0.8, 1.0], [0.6, 1.0], [1.1, 1.0]
    b = A.Batch(S, A.get_h2m_matrix(3, oid), 2, frame_size=fs)
    b.set_gains(element=eg, output=og)
    b.set_second_element(mx2, eg2)
    got = _run_ex(A, G, torch, b, S, 16, x0, fs, 2, A.FMT_S16, x2=x1, m2=m2, calls=[1, 3])
    b.close()
    for s in range(S):
        y0 = O.render(O.get_h2m(3, O.SS[out]), x0[s], 2)
        y1 = O.render(omx2, x1[s], 2)
        if eg[s] != 1.0:
            O.lib().orc_frame_gain_const(O.fp(y0), 2, F * fs, eg[s])
        if eg2[s] != 1.0:
            O.lib().orc_frame_gain_const(O.fp(y1), 2, F * fs, eg2[s])
        z = ((np.zeros_like(y0) + y0) + y1).astype(np.float32)
        if og[s] != 1.0:
            O.lib().orc_frame_gain_const(O.fp(z), 2, F * fs, og[s])
        z, _ = O.limiter_run(z, [fs] * F)
        assert np.array_equal(got[s], O.pack(z, 16)), s


def test_projection_demapping_vs_oracle(hip):
    """projection-mode ambisonics: 12 decoded channels -> 9 ambisonics channels (f32, decoded-channel
    ascending, IAMF_core_decoder.c:116-130) -> H2M to 7.1.4 -> limiter, two streams"""
    A, G, torch = hip
    fs, F, S, L_in = 960, 3, 2, 12
    n = fs * F
    rng = np.random.default_rng(77)
    P = (rng.integers(-12000, 12000, size=(L_in, 9)).astype(np.float32) * np.float32(2.0 ** -15)).astype(np.float32)
    xd = np.stack([synth.hot(80 + s, L_in, n, sigma=0.15, burst_phase=200 + 300 * s, burst_period=1900) for s in range(S)])
    oid = A.SS["J"]
    b = A.Batch(S, A.get_h2m_matrix(2, oid), 12, frame_size=fs, projection=A.PROJ_EXACT)
    b.set_projection(P)
    got = _run_ex(A, G, torch, b, S, L_in, xd, fs, 12, A.FMT_S16, calls=[1, 2])
    b.close()
    for s in range(S):
        xa = np.zeros((9, n), np.float32)
        for l in range(L_in):
            xa = (xa + (xd[s][l][None, :] * P[l][:, None]).astype(np.float32)).astype(np.float32)
        y = O.render(O.get_h2m(2, O.SS["J"]), xa, 12)
        z, _ = O.limiter_run(y, [fs] * F)
        assert np.array_equal(got[s], O.pack(z, 16)), s


@pytest.mark.parametrize("order,l_in,out", [(3, 16, "BINAURAL"), (3, 16, "B"), (3, 16, "J"), (2, 9, "H"), (1, 4, "A"),
                                             (2, 12, "J")])
def test_projection_tolerance_mode_within_1lsb(hip, order, l_in, out):
    """IAMF_HIP_PROJ_AUTO / _MFMA: de-mapping and H2M matrix composed into one (W * P^T in double),
    so the call runs on the fast / wide4 kernels.  Tolerance: +-1 LSB of the 16-bit PCM against the
    reference's two f32 stages, and most samples identical."""
    A, G, torch = hip
    fs, F, S = 1024, 4, 2
    n = fs * F
    m = (order + 1) ** 2
    rng = np.random.default_rng(1000 * order + l_in)
    P = (rng.integers(-9000, 9000, size=(l_in, m)).astype(np.float32) * np.float32(2.0 ** -15)).astype(np.float32)
    P[np.arange(min(l_in, m)), np.arange(min(l_in, m))] += np.float32(0.5)
    xd = np.stack([synth.hot(90 + s, l_in, n, sigma=0.15, burst_phase=200 + 300 * s, burst_period=1900) for s in range(S)])
    oid = A.SS[out]
    ch = A.layout_channels(oid)
    b = A.Batch(S, A.get_h2m_matrix(order, oid), ch, frame_size=fs)   # projection=AUTO
    b.set_projection(P)
    got = _run_ex(A, G, torch, b, S, l_in, xd, fs, ch, A.FMT_S16, calls=[1, 3])
    b.close()
    for s in range(S):
        xa = np.zeros((m, n), np.float32)
        for l in range(l_in):
            xa = (xa + (xd[s][l][None, :] * P[l][:, None]).astype(np.float32)).astype(np.float32)
        y = O.render(O.get_h2m(order, O.SS[out]), xa, ch)
        z, _ = O.limiter_run(y, [fs] * F)
        want = O.pack(z, 16)
        d = np.abs(got[s].astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, (s, int(d.max()))
        assert (d != 0).mean() < 0.02, (s, float((d != 0).mean()))


def test_mix_gain_ramps_vs_oracle(hip):
    """per-sample element / output gains (linear and quadratic-Bezier ramps built the way
    IAMF_decoder.c:639-664 builds them) are applied unconditionally, sample by sample"""
    A, G, torch = hip
    fs, F = 960, 4
    n = fs * F
    x = synth.hot(71, 16, n, sigma=0.2, burst_phase=300, burst_period=2100)
    er = np.ones(n, dtype=np.float32)
    orr = np.ones(n, dtype=np.float32)
    L = O.lib()
    L.orc_mix_gain_linear(0.5, 1.25, fs, 0, fs, O.fp(er[0:fs]))
    L.orc_mix_gain_quad(1.25, 0.7, fs, 0.2, int(0.3 * (fs + 0.1)), 0, fs, O.fp(er[fs:2 * fs]))
    er[2 * fs:] = np.float32(0.7)
    L.orc_mix_gain_linear(1.0, 0.9, 2 * fs, 100, 2 * fs - 100, O.fp(orr[fs:3 * fs - 100]))
    oid = A.SS["BINAURAL"]
    b = A.Batch(1, A.get_h2m_matrix(3, oid), 2, frame_size=fs)
    got = _run_ex(A, G, torch, b, 1, 16, x[None], fs, 2, A.FMT_S16,
                  ramps=dict(element=er[None], output=orr[None]), calls=[1, 3])[0]
    b.close()
    y = O.render(O.get_h2m(3, O.SS["BINAURAL"]), x, 2)
    L.orc_frame_gain_ramp(O.fp(y), 2, n, O.fp(er))
    z = (np.zeros_like(y) + y).astype(np.float32)
    L.orc_frame_gain_ramp(O.fp(z), 2, n, O.fp(orr))
    z, _ = O.limiter_run(z, [fs] * F)
    assert np.array_equal(got, O.pack(z, 16))


def test_resampler_goldens_bit_exact(hip, golden):
    """every rate pair of the reference goldens (direct and interpolated filters, up and down,
    ragged call sizes, end-of-stream drain), two streams per batch"""
    A, G, torch = hip
    g = golden.npz("resample")
    st = torch.cuda.current_stream().cuda_stream
    for key, meta in golden.manifest.items():
        if not key.startswith("resample/"):
            continue
        name = key.split("/")[1]
        ch, sizes = meta["ch"], meta["sizes"]
        x = synth.hot(meta["seed"], ch, sum(sizes), sigma=0.3, burst_amp=1.2, burst_len=60, burst_phase=50,
                      burst_period=700)
        xs = np.stack([x, (x * np.float32(0.5)).astype(np.float32)])  # stream 1: half amplitude
        r = A.Resampler(2, ch, meta["in_rate"], meta["out_rate"])
        outs, rets, pos = [[], []], [], 0
        for ns in sizes:
            inter = torch.from_numpy(np.ascontiguousarray(xs[:, :, pos:pos + ns].transpose(0, 2, 1))).cuda()
            pos += ns
            cap = r.out_capacity(ns)
            o = torch.zeros((2, cap, ch), dtype=torch.float32, device="cuda")
            n = r.process(inter.data_ptr(), ns * ch, ns, o.data_ptr(), cap * ch, st)
            torch.cuda.synchronize()
            h = o.cpu().numpy()
            rets.append(n)
            for s in range(2):
                outs[s].append(h[s, :n].T.copy())
        cap = max(r.flush_capacity(), 1)
        o = torch.zeros((2, cap, ch), dtype=torch.float32, device="cuda")
        n = r.flush(o.data_ptr(), cap * ch, st)
        torch.cuda.synchronize()
        h = o.cpu().numpy()
        rets.append(n)
        for s in range(2):
            outs[s].append(h[s, :n].T.copy())
        r.close()
        assert rets == list(g[name + "_rets"]), name
        y0 = np.concatenate(outs[0], axis=1)
        assert np.array_equal(y0.view(np.uint32), g[name].view(np.uint32)), name
        y1, _ = O.resample_run(xs[1], meta["in_rate"], meta["out_rate"], sizes)
        assert np.array_equal(np.concatenate(outs[1], axis=1).view(np.uint32), y1.view(np.uint32)), name


@pytest.mark.parametrize("rates", [(44100, 48000), (16000, 48000), (48000, 44100)])
def test_resampler_streams_out_of_step_by_ranges(hip, rates):
    """round 4: one resampler, five streams that consume DIFFERENT sequences of call lengths through range calls (what a
    group of decoder handles does): each stream's phase and history are its own, so every stream's output must be the
    oracle's for ITS sequence; a range across streams in different states is refused, same_state says which are alike"""
    A, G, torch = hip
    ch, S = 2, 5
    plans = [[300, 1024, 77, 1024], [1024, 512, 300, 64], [300, 1024, 77, 1024], [512, 512, 512, 889], [1024, 512, 300, 64]]   # (a stream's state is a function of the samples it has consumed: the three plans differ in every running total)
    xs = [synth.hot(7700 + s, ch, sum(plans[s]), sigma=0.3, burst_amp=1.1, burst_len=50, burst_phase=40, burst_period=600) for s in range(S)]
    r = A.Resampler(S, ch, rates[0], rates[1])
    st = torch.cuda.current_stream().cuda_stream
    cap = r.out_capacity(1024)
    outs, pos = [[] for _ in range(S)], [0] * S
    for step in range(4):
        # streams with equal call length this step AND equal history of lengths form the ranges: {0, 2}, {1, 4}, {3}
        inter = torch.zeros((S, 1024, ch), dtype=torch.float32, device="cuda")
        for s in range(S):
            ns = plans[s][step]
            inter[s, :ns] = torch.from_numpy(np.ascontiguousarray(xs[s][:, pos[s]:pos[s] + ns].T)).cuda()
        o = torch.zeros((S, cap, ch), dtype=torch.float32, device="cuda")
        if step > 0 and rates[1] % rates[0]:   # (1 : 3 leaves every stream in the same phase after any whole call)
            assert r.same_state(0, 2) and r.same_state(1, 4) and not r.same_state(0, 1)
            assert r.process_range(inter.data_ptr(), 1024 * ch, 300, o.data_ptr(), cap * ch, 0, 2, st) == -5   # streams 0 and 1 differ
        got = {}
        for s0, cnt in ((0, 1), (2, 1), (1, 1), (4, 1), (3, 1)) if step == 0 else ((0, 1), (1, 1), (2, 1), (3, 1), (4, 1)):
            got[s0] = r.process_range(inter.data_ptr(), 1024 * ch, plans[s0][step], o.data_ptr(), cap * ch, s0, cnt, st)
            assert got[s0] >= 0, got[s0]
        torch.cuda.synchronize()
        h = o.cpu().numpy()
        for s in range(S):
            outs[s].append(h[s, :got[s]].T.copy())
            pos[s] += plans[s][step]
    o = torch.zeros((S, max(r.flush_capacity(), 1), ch), dtype=torch.float32, device="cuda")
    n02 = r.flush_range(o.data_ptr(), o.shape[1] * ch, 0, 1, st)
    tails = {0: n02}
    for s in range(1, S):
        tails[s] = r.flush_range(o.data_ptr(), o.shape[1] * ch, s, 1, st)
    torch.cuda.synchronize()
    h = o.cpu().numpy()
    r.close()
    for s in range(S):
        outs[s].append(h[s, :tails[s]].T.copy())
        want, _ = O.resample_run(xs[s], rates[0], rates[1], plans[s])
        assert np.array_equal(np.concatenate(outs[s], axis=1).view(np.uint32), want.view(np.uint32)), s


@pytest.mark.parametrize("rates", [(44100, 48000), (48000, 44100), (22050, 48000), (8000, 44100),
                                   (96000, 48000), (48000, 16000), (16000, 48000), (32000, 48000), (48000, 32000), (24000, 48000)])
@pytest.mark.parametrize("ch,streams", [(1, 3), (2, 70), (2, 260), (6, 70), (8, 5), (6, 260), (1, 300), (8, 260), (12, 70), (10, 3), (14, 5), (24, 3)])
def test_resampler_blocked_kernel_over_its_shapes(hip, rates, ch, streams, monkeypatch):
    """resample_block_kernel<C, R> (interpolated mode, 1 / 2 / 6 / 8 channels; R = 1 / 2 / 4 by the launch's stream count;
    den = 160, 147, 320, 441 threads' worth of phases) and resample_direct_kernel<C, N, NUMP> (direct mode: 2:1 and 3:1 with
    the de-interleaved window, 1:3 / 2:3 / 3:2 / 1:2 with the plain one; filter rows of 64 / 96 / 128 / 192 taps in
    registers): ragged calls incl. one that yields no output, then the drain.
    Streams 0 and the last one against the oracle (the reference's arithmetic), every stream against the tiled kernel."""
    A, G, torch = hip
    sizes = [1024, 3, 1, 700, 1024, 64]
    rng = np.random.default_rng(ch * 1000 + streams)
    x = (rng.standard_normal((streams, sum(sizes), ch)) * 0.3).astype(np.float32)
    x[:, 100:140] *= 4.0   # beyond +-1: the clamp
    st = torch.cuda.current_stream().cuda_stream

    def run():
        r = A.Resampler(streams, ch, rates[0], rates[1])
        outs, pos = [], 0
        for ns in sizes:
            inter = torch.from_numpy(np.ascontiguousarray(x[:, pos:pos + ns])).cuda()
            pos += ns
            cap = max(r.out_capacity(ns), 1)
            o = torch.full((streams, cap, ch), 9.0, dtype=torch.float32, device="cuda")
            n = r.process(inter.data_ptr(), ns * ch, ns, o.data_ptr(), cap * ch, st)
            torch.cuda.synchronize()
            assert n >= 0, n
            outs.append(o[:, :n].cpu().numpy())
            assert bool((o[:, n:] == 9.0).all()), "nothing is written past the call's outputs"
        cap = max(r.flush_capacity(), 1)
        o = torch.zeros((streams, cap, ch), dtype=torch.float32, device="cuda")
        n = r.flush(o.data_ptr(), cap * ch, st)
        torch.cuda.synchronize()
        outs.append(o[:, :n].cpu().numpy())
        r.close()
        return np.concatenate(outs, axis=1)

    monkeypatch.delenv("IAMF_HIP_RESAMPLE_TILE", raising=False)
    got = run()
    monkeypatch.setenv("IAMF_HIP_RESAMPLE_TILE", "1")
    tiled = run()
    monkeypatch.delenv("IAMF_HIP_RESAMPLE_TILE", raising=False)
    assert np.array_equal(got.view(np.uint32), tiled.view(np.uint32))
    for s_ in (0, streams - 1):
        want, _ = O.resample_run(np.ascontiguousarray(x[s_].T), rates[0], rates[1], sizes)
        assert np.array_equal(np.ascontiguousarray(got[s_].T).view(np.uint32), want.view(np.uint32)), s_


def test_stream_signal_writes_the_pinned_word_behind_the_queued_work(hip):
    """iamf_hip_stream_signal (what the single-handle facade waits on instead of hipStreamSynchronize)"""
    A, G, torch = hip
    flag = torch.zeros(16, dtype=torch.int32).pin_memory()
    big = torch.zeros(1 << 24, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for seq in (1, 2, 77):
        big.add_(1.0)    # queued in front of the signal
        assert A.lib().iamf_hip_stream_signal(st, flag.data_ptr(), seq) == 0
        for _ in range(2000000):
            if int(flag[0]) == seq:
                break
        assert int(flag[0]) == seq
    torch.cuda.synchronize()
    assert float(big[0]) == 3.0
    assert A.lib().iamf_hip_stream_signal(st, None, 1) == -1


@pytest.mark.parametrize("ch,n,S", [(2, 1024, 1), (6, 777, 3), (12, 1, 2), (24, 1024, 5), (8, 257, 4)])
def test_deinterleave_f32_is_a_transposition(hip, ch, n, S):
    """iamf_hip_deinterleave_f32: a batch's f32 sample-frames -> the planar form a batch reads (how the frame of an element
    rendered by a batch of its own reaches the mixing batch as a second element); strides, short rows, untouched padding"""
    A, G, torch = hip
    rng = np.random.default_rng(ch * 1000 + n)
    src_stride, ch_stride = n * ch + 24, n + 8
    dst_stride = ch * ch_stride + 16
    x = rng.standard_normal((S, src_stride)).astype(np.float32)
    src = torch.from_numpy(x).cuda()
    dst = torch.full((S, dst_stride), -7.0, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert A.lib().iamf_hip_deinterleave_f32(src.data_ptr(), src_stride, ch, S, n, dst.data_ptr(), dst_stride, ch_stride, st) == 0
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    for s in range(S):
        want = np.full(dst_stride, -7.0, np.float32)
        for c in range(ch):
            want[c * ch_stride:c * ch_stride + n] = x[s, :n * ch].reshape(n, ch)[:, c]
        assert np.array_equal(got[s], want), s
    assert A.lib().iamf_hip_deinterleave_f32(src.data_ptr(), src_stride, 25, S, n, dst.data_ptr(), dst_stride, ch_stride, st) == -1
    assert A.lib().iamf_hip_deinterleave_f32(src.data_ptr(), src_stride, ch, S, n, dst.data_ptr(), dst_stride, n - 1, st) == -1


def test_resampled_pipeline_vs_oracle(hip):
    """44.1 kHz stereo element -> Sound System A at 48 kHz: render (f32) -> resample -> limiter +
    pack as three launches, against the oracle's stages in the decoder's order
    (IAMF_decoder.c:3459-3500)."""
    A, G, torch = hip
    fs, F = 1024, 5
    x = synth.hot(91, 2, F * fs, sigma=0.25, burst_phase=200, burst_period=1800)
    st = torch.cuda.current_stream().cuda_stream
    mx = A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"])
    stage1 = A.Batch(1, mx, 2, frame_size=fs, sample_rate=44100, out_format=A.FMT_F32, limiter=False)
    rs = A.Resampler(1, 2, 44100, 48000)
    stage3 = A.Batch(1, G.identity_matrix(2), 2, frame_size=1, sample_rate=48000, out_format=A.FMT_S16, limiter=True)
    xin = torch.from_numpy(G.to_frames(x[None], fs)).cuda()
    got = []
    for f in range(F):
        mid = torch.zeros((fs, 2), dtype=torch.float32, device="cuda")
        n1 = stage1.render(xin.data_ptr() + 4 * f * 2 * fs, F * 2 * fs, 2 * fs, 1, mid.data_ptr(), fs * 2 * 4, st)
        assert n1 == fs
        cap = rs.out_capacity(fs)
        res = torch.zeros((cap, 2), dtype=torch.float32, device="cuda")
        n2 = rs.process(mid.data_ptr(), fs * 2, fs, res.data_ptr(), cap * 2, st)
        pcm = torch.zeros((max(n2, 240), 2), dtype=torch.int16, device="cuda")
        n3 = stage3.render(res.data_ptr(), cap * 2, 2, n2, pcm.data_ptr(), pcm.numel() * 2, st)
        torch.cuda.synchronize()
        got.append(pcm.cpu().numpy()[:n3])
    # end of stream (iamf_delay_buffer_handle, IAMF_decoder.c:3250-3301): resampler tail + 240 zeros
    cap = rs.flush_capacity()
    tail = torch.zeros((cap + 240, 2), dtype=torch.float32, device="cuda")
    n2 = rs.flush(tail.data_ptr(), (cap + 240) * 2, st)
    pcm = torch.zeros((n2 + 240, 2), dtype=torch.int16, device="cuda")
    n3 = stage3.render(tail.data_ptr(), (cap + 240) * 2, 2, n2 + 240, pcm.data_ptr(), pcm.numel() * 2, st)
    torch.cuda.synchronize()
    got.append(pcm.cpu().numpy()[:n3])
    got = np.concatenate(got, axis=0)

    y = O.render(O.get_m2m(O.SS["STEREO"], O.SS["A"]), x, 2)
    z = (np.zeros_like(y) + y).astype(np.float32)
    r, rets = O.resample_run(z, 44100, 48000, [fs] * F, flush=True)
    n_tail = rets[-1]
    body, tail_o = r[:, :r.shape[1] - n_tail], r[:, r.shape[1] - n_tail:]
    sizes = rets[:-1] + [n_tail + 240]
    lim_in = np.concatenate([body, tail_o, np.zeros((2, 240), dtype=np.float32)], axis=1)
    zl, _ = O.limiter_run(lim_in, sizes, flush=False)
    want = O.pack(zl, 16)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_demixer_stage_matches_reference(hip, golden):
    """scalable channel audio (N2): demixer in front of an identity renderer, f32 out, limiter off,
    against the REAL reference demixer_* outputs (tests/golden/demix.npz); per-frame records built
    with the ABI's own host control plane (iamf_hip_demix_*)"""
    import ctypes as C
    import demix_cases as D
    A, G, torch = hip
    gold = golden.npz("demix")
    for name, c in D.STAGE_CASES.items():
        x = D.case_input(c)                      # [frames][ch][fs]
        F, ch, fs = x.shape
        b = A.Batch(1, G.identity_matrix(ch), ch, frame_size=fs, out_format=A.FMT_F32, limiter=False)
        b.set_demixer(c["layout"], c["order"], c["gains"], c["offset"])
        st = A.DemixState()
        A.lib().iamf_hip_demix_state_init(C.byref(st))
        A.lib().iamf_hip_demix_set_info(C.byref(st), c["default"][0], c["default"][1])
        frames = (A.DemixFrame * F)()
        rec = (C.c_int32 * 12)(*c["recon"])
        cur = [1.0] * len(c["recon"])
        for f, (mode, rg) in enumerate(c["schedule"]):
            if rg is not None:
                cur = rg
            if mode > -1:
                A.lib().iamf_hip_demix_set_info(C.byref(st), mode, -1)
            A.lib().iamf_hip_demix_frame_fill(C.byref(st), len(cur), rec, (C.c_float * 12)(*cur), C.byref(frames[f]))
        d_fr = torch.from_numpy(np.frombuffer(bytes(frames), dtype=np.uint8).copy()).cuda()
        xin = torch.from_numpy(np.ascontiguousarray(x[None])).cuda()     # [1][F][ch][fs]
        pcm = torch.zeros((1, F * fs * ch * 4), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr(), F * ch * fs, ch * fs
        a.n_frames, a.d_pcm, a.pcm_stream_stride_bytes = F, pcm.data_ptr(), F * fs * ch * 4
        a.d_demix_frames = d_fr.data_ptr()
        a.stream = torch.cuda.current_stream().cuda_stream
        n = b.render_ex(a)
        torch.cuda.synchronize()
        b.close()
        assert n == F * fs
        got = pcm.cpu().numpy().view(np.float32).reshape(F, fs, ch).transpose(0, 2, 1)
        assert np.array_equal(got, gold[name]), name
