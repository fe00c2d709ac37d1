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


LAYOUTS = ["B", "C", "D", "J", "G", "H"]  # 6, 8, 10, 12, 14, 24 channels


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


# ---- scalable channel audio: the demixer in front of the projection (render_wide4_kernel<.., DMX>) ----
def _demix_frames(A, c, S):
    import ctypes as C
    F = len(c["schedule"])
    frames = (A.DemixFrame * (S * F))()
    st = A.DemixState()
    rec = (C.c_int32 * 12)(*c["recon"])
    for s in range(S):
        A.lib().iamf_hip_demix_state_init(C.byref(st))
        A.lib().iamf_hip_demix_set_info(C.byref(st), c["default"][0], c["default"][1])
        cur = [1.0] * len(c["recon"])
        for f, (mode, rg) in enumerate(c["schedule"]):
            if rg is not None:
                cur = rg
            if mode > -1:
                A.lib().iamf_hip_demix_set_info(C.byref(st), mode, -1)
            A.lib().iamf_hip_demix_frame_fill(C.byref(st), len(cur), rec, (C.c_float * 12)(*cur),
                                              C.byref(frames[s * F + f]))
    return frames


def _demix_render(A, c, mx, out_ch, x, calls):
    """x [S][F][ch][fs] decoded channels -> demixer -> mx -> limiter -> s16; one render_ex per entry of
    `calls` (frames), then the flush.  Returns [S][n][out_ch]."""
    import torch
    S, F, ch, fs = x.shape
    b = A.Batch(S, mx, out_ch, frame_size=fs, out_format=A.FMT_S16, limiter=True)
    b.set_demixer(c["layout"], c["order"], c["gains"], c["offset"])
    frames = _demix_frames(A, c, S)
    xin = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = [[] for _ in range(S)]
    f0 = 0
    for nf in calls + [0]:
        cap = max(nf * fs, 240) * out_ch * 2
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        if nf:
            # the records of this call's frames, [S][nf]
            raw = np.frombuffer(bytes(frames), dtype=np.uint8).reshape(S, F, -1)[:, f0:f0 + nf].copy()
            d_fr = torch.from_numpy(raw).cuda()
            a = A.RenderArgs()
            a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr() + 4 * f0 * ch * fs, F * ch * fs, ch * fs
            a.n_frames, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = nf, pcm.data_ptr(), cap, st
            a.d_demix_frames = d_fr.data_ptr()
            n = b.render_ex(a)
        else:
            n = b.flush(pcm.data_ptr(), cap, st)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(h[s][:n * out_ch * 2].view(np.int16).reshape(n, out_ch).copy())
        f0 += nf
    b.close()
    return [np.concatenate(o, axis=0) for o in outs]


def _demix_cases():
    import demix_cases as D
    cases = dict(D.STAGE_CASES)  # 256-sample frames: four frame records per 1024-sample chunk
    cases["714_fs1024_pre40"] = D.make_case([1, 3, 7], {0: (0b110000, 0.7079458), 1: (0b001111, 1.4125376)},
                                            offset=40, fs=1024, seed=770)
    cases["514_fs512"] = D.make_case([1, 2, 4], {1: (0b001100, 1.2)}, default=(5, 9), fs=512, seed=780)
    cases["712_fs960"] = D.make_case([8, 3, 6], {0: (0b110011, 1.1885022)}, default=(6, 2), offset=8, fs=960, seed=790)
    cases["510_fs320"] = D.make_case([0, 1, 2], default=(2, 4), fs=320, seed=800)
    return cases


_LAYOUT_SS = {2: "L51", 3: "L512", 4: "L514", 5: "L71", 6: "L712", 7: "L714", 8: "L312"}


@pytest.mark.parametrize("name", sorted(_demix_cases()))
def test_wide4_demixer_exact(hip, name, monkeypatch):
    """decoded layers -> demixer -> layout matrix -> limiter -> s16 on the wide4 kernel: bit-exact
    against (oracle demixer, pinned to the reference's demixer.c) -> (oracle renderer chain), and
    identical to the generic kernel's demixer"""
    import demix_cases as D
    A, G = hip
    c = _demix_cases()[name]
    S, fs, F = 2, c["fs"], len(c["schedule"])
    ch = len(c["order"])
    x = np.stack([np.stack([synth.uniform(c["seed"] + 100 * s + f, ch, fs, 0.9) for f in range(F)]) for s in range(S)])
    dem = [D.drive_demixer(O.lib(), "orc_demixer_", c, x[s]) for s in range(S)]   # [F][ch][fs]
    src = _LAYOUT_SS[c["layout"]]
    outs = ["J", "H"] if ch == 12 else ["J"]
    calls = [4, F - 4] if (4 * fs) % 1024 == 0 else [F]
    for out in outs + ["id"]:
        if out == "id":
            mx, omx, och = G.identity_matrix(ch), None, ch
        else:
            try:
                mx, omx = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out])
            except KeyError:
                continue
            och = A.layout_channels(A.SS[out])
        got = _demix_render(A, c, mx, och, x, calls)
        monkeypatch.setenv("IAMF_HIP_NO_WIDE4", "1")
        ref = _demix_render(A, c, mx, och, x, calls)
        monkeypatch.delenv("IAMF_HIP_NO_WIDE4")
        for s in range(S):
            xd = np.ascontiguousarray(dem[s].transpose(1, 0, 2).reshape(ch, F * fs))
            if omx is None:
                z, _ = O.limiter_run(xd, [fs] * F)
                want = O.pack(z, 16)
            else:
                want = O.stream_run(omx, och, xd, fs)
            assert got[s].shape == want.shape, (name, out, s)
            assert np.array_equal(ref[s], want), (name, out, s, "generic kernel")
            assert np.array_equal(got[s], want), (name, out, s)


# ---- parametric down-mixer on the 4-samples-per-lane kernels (render_downmix.hpp) ----
_DOWN_PAIRS = [(7, 6), (7, 4), (7, 3), (7, 8), (6, 3), (6, 8), (4, 3), (4, 8), (3, 8), (5, 2), (5, 1), (5, 0),
               (2, 1), (2, 0), (1, 0)]


@pytest.mark.parametrize("il,ol", _DOWN_PAIRS)
@pytest.mark.parametrize("fs", [1024, 256, 960])
def test_downmixer_fast_paths_exact(hip, il, ol, fs, monkeypatch):
    """element -> parametric down-mixer (mode per frame, previous mode for the first `offset` samples,
    offsets not multiples of 4 included) -> limiter -> s16 on render_fast_kernel<.., DOWN> (mono /
    stereo) and render_wide4_kernel<.., DOWN>: bit-exact against the oracle down-mixer (pinned to
    the reference's DMRenderer_*) + limiter + pack, and identical to the generic kernel"""
    import ctypes as C
    import torch
    A, G = hip
    L = A.lib()
    assert L.iamf_hip_dmx_valid(il, ol) == 1
    S, F = 2, 8 * 1024 // fs
    m, oc = O.LAYOUT_CH[il], O.LAYOUT_CH[ol]
    sched = [((-1, 1, 2, 4, 5, 6, 0, 2)[f % 8], (0, 0, 37, 128, 0, fs - 3, 4, 0)[f % 8]) for f in range(F)]
    x = np.stack([np.stack([synth.hot(500 + 31 * s + f, m, fs, sigma=0.3, burst_phase=100 + 50 * f, burst_period=700)
                            for f in range(F)]) for s in range(S)])               # [S][F][m][fs]
    frames = (A.DmxFrame * (S * F))()
    st = A.DmxState()
    for s in range(S):
        L.iamf_hip_dmx_state_init(C.byref(st))
        L.iamf_hip_dmx_set_mode_weight(C.byref(st), 1, 3)
        for f, (mode, off) in enumerate(sched):
            fr = frames[s * F + f]
            fr.offset = off
            L.iamf_hip_dmx_coefficients(C.byref(st), fr.prev)
            if mode > -1:
                L.iamf_hip_dmx_set_mode_weight(C.byref(st), mode, -1)
            L.iamf_hip_dmx_coefficients(C.byref(st), fr.cur)
    d_fr = torch.from_numpy(np.frombuffer(bytes(frames), dtype=np.uint8).copy()).cuda()
    xin = torch.from_numpy(np.ascontiguousarray(x)).cuda()

    def run():
        b = A.Batch(S, A.dmx_matrix(il, ol), oc, frame_size=fs, out_format=A.FMT_S16, limiter=True)
        outs = [[] for _ in range(S)]
        stt = torch.cuda.current_stream().cuda_stream
        for last in (False, True):
            cap = max(F * fs, 240) * oc * 2
            pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
            if not last:
                a = A.RenderArgs()
                a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr(), F * m * fs, m * fs
                a.n_frames, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = F, pcm.data_ptr(), cap, stt
                a.d_dmx_frames = d_fr.data_ptr()
                n = b.render_ex(a)
            else:
                n = b.flush(pcm.data_ptr(), cap, stt)
            torch.cuda.synchronize()
            h = pcm.cpu().numpy()
            for s in range(S):
                outs[s].append(h[s][:n * oc * 2].view(np.int16).reshape(n, oc).copy())
        b.close()
        return [np.concatenate(o, axis=0) for o in outs]

    got = run()
    monkeypatch.setenv("IAMF_HIP_FORCE_GENERIC", "1")
    ref = run()
    monkeypatch.delenv("IAMF_HIP_FORCE_GENERIC")
    for s in range(S):
        y = O.downmix_run(il, ol, x[s], sched, 1, 3)                     # [F][oc][fs]
        yd = np.ascontiguousarray(y.transpose(1, 0, 2).reshape(oc, F * fs))
        z, _ = O.limiter_run(yd, [fs] * F)
        want = O.pack(z, 16)
        assert got[s].shape == want.shape
        assert np.array_equal(ref[s], want), (il, ol, s, "generic kernel")
        assert np.array_equal(got[s], want), (il, ol, s)


# ---- mixing variant: second element and per-sample gain ramps (render_wide4_kernel<.., MIX>) ----
@pytest.mark.parametrize("bed,out,second,m2,proj", [
    ("L714", "J", "STEREO", 2, "exact"), ("L714", "B", "MONO", 1, "exact"), ("TOA", "J", "FOA", 4, "exact"),
    ("L51", "C", "STEREO", 2, "exact"), ("TOA", "B", "STEREO", 2, "mfma"), ("L514", "D", None, 0, "exact")])
def test_wide4_mixing_variant(hip, bed, out, second, m2, proj):
    """bed + optional second element (<= 4 channels) + element / output gain ramps into 6..12-channel
    layouts, 1024-sample frames, s16: bit-exact against the oracle chain with the exact projection,
    within 1 LSB with the MFMA projection"""
    import torch
    from test_gpu_extras import _run_ex
    A, G = hip
    fs, F, S = 1024, 5, 2
    n = fs * F
    oid = A.SS[out]
    ch = A.layout_channels(oid)
    if bed == "TOA":
        m, mx, omx = 16, A.get_h2m_matrix(3, oid), O.get_h2m(3, O.SS[out])
    else:
        m, mx, omx = SOURCES[bed], A.get_m2m_matrix(A.SS[bed], oid), O.get_m2m(O.SS[bed], O.SS[out])
    mx2 = omx2 = None
    if second == "FOA":
        mx2, omx2 = A.get_h2m_matrix(1, oid), O.get_h2m(1, O.SS[out])
    elif second:
        mx2, omx2 = A.get_m2m_matrix(A.SS[second], oid), O.get_m2m(O.SS[second], O.SS[out])
    rng = np.random.default_rng(hash((bed, out)) % 1000)
    x0 = np.stack([synth.hot(261 + s, m, n, sigma=0.2, burst_phase=300 + 100 * s, burst_period=2300) for s in range(S)])
    x1 = np.stack([synth.hot(271 + s, m2, n, sigma=0.3, burst_phase=900, burst_period=1700) for s in range(S)]) if m2 else None
    ramps = dict(element=(0.5 + 0.7 * rng.random((S, n))).astype(np.float32),
                 output=(0.6 + 0.5 * rng.random((S, n))).astype(np.float32))
    if m2 == 2:
        ramps["element2"] = (0.3 + 0.9 * rng.random((S, n))).astype(np.float32)
    eg2 = [0.6, 1.0]
    b = A.Batch(S, mx, ch, frame_size=fs, projection=A.PROJ_EXACT if proj == "exact" else A.PROJ_MFMA)
    if m2:
        b.set_second_element(mx2, eg2)
    got = _run_ex(A, G, torch, b, S, m, x0, fs, ch, A.FMT_S16, x2=x1, m2=m2, ramps=ramps, calls=[1, 2, 2])
    b.close()
    f32 = np.float32
    for s in range(S):
        y = (O.render(omx, x0[s], ch)[:ch] * ramps["element"][s][None, :]).astype(f32)
        z = (np.zeros_like(y) + y).astype(f32)
        if m2:
            y2 = O.render(omx2, x1[s], ch)[:ch]
            if "element2" in ramps:
                y2 = (y2 * ramps["element2"][s][None, :]).astype(f32)
            elif eg2[s] != 1.0:
                y2 = (y2 * f32(eg2[s])).astype(f32)
            z = (z + y2).astype(f32)
        z = (z * ramps["output"][s][None, :]).astype(f32)
        z, _ = O.limiter_run(np.ascontiguousarray(z), [fs] * F)
        want = O.pack(z, 16)
        assert got[s].shape == want.shape
        if proj == "exact":
            assert np.array_equal(got[s], want), (bed, out, s)
        else:
            assert np.abs(got[s].astype(np.int32) - want.astype(np.int32)).max() <= 1, (bed, out, s)


def test_shared_divisor_quotients_are_the_ieee_quotients_for_every_numerator(hip):
    """The wide4 demixer divides 8 numerators each by delta, beta and gamma of the frame (demixer.c:205-214,255-267,
    357-366) through the divisor's reciprocal: q = n r, e = fma(-d, q, n), q' = fma(e, r, q).  Swept on the device over ALL
    2^32 numerators for every divisor a demixing mode can produce (IAMF_utils.c:236-240: 1, 0.707f, 0.866f): inside the
    range the kernel uses it for (2^-100 <= |n| < 2^126) not one quotient differs from n / d in any bit; outside it some
    do (underflowing residual, overflow, -0), which is why the kernel divides those the IEEE way."""
    import ctypes as C
    A, _ = hip
    L = A.lib()
    L.iamf_hip_selftest_shared_divisor.argtypes = [C.c_float, C.POINTER(C.c_uint64)]
    outside_bad = 0
    for d in (1.0, 0.707, 0.866, 0.5, 0.70710678):
        c = (C.c_uint64 * 3)()
        assert L.iamf_hip_selftest_shared_divisor(C.c_float(d), c) == 0
        n_in, bad_in, bad_out = int(c[0]), int(c[1]), int(c[2])
        assert n_in == 2 * (226 << 23), (d, n_in)          # 226 binades x 2^23 significands x 2 signs
        assert bad_in == 0, (d, bad_in)
        outside_bad += bad_out
    assert outside_bad > 0      # the guard is not decoration
