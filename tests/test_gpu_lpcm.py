"""iamf_hip_batch_render_lpcm (element 0 handed over as LPCM packets) against the f32 path on the same samples.

The reference decodes an LPCM packet into its planar f32 decoder buffer (src/iamf_dec/pcm/IAMF_pcm_decoder.c:64-83:
sample / 2^(bits-1)) and renders from there (IAMF_decoder.c:2550-2640).  The f32 path of this library is pinned against
the oracle elsewhere (tests/test_gpu_pipeline.py, test_gpu_fuzz.py, test_gpu_facade.py); here: the entry that takes the
packets themselves gives THE SAME BYTES as `render` over numpy's int / 2^(bits-1) — in its fused form (the render kernel
converts the 16-bit samples where it loads them: mono-coded ambisonics into one / two channels) and in its general form
(device unpacker + f32 kernels: coupled sub-streams, 24 / 32 bit, big-endian, wide layouts) — over several calls,
permuted and padded packet rows, a trimmed frame, and with the fused form switched off by IAMF_HIP_LPCM_UNFUSED.
"""
import os

import numpy as np
import pytest
import torch

import iac_amd as A
from gpu_util import hip_render

pytestmark = pytest.mark.gpu


def _pack(v, bps, le):
    """int samples [..., n] -> bytes [..., n * bps] in the reference's byte orders (24-bit big-endian: bitstream.c:204-208:
    byte 1 is the top one, then byte 2, byte 0 is the low one)"""
    v = v.astype(np.int64)
    u = v & ((1 << (8 * bps)) - 1)
    b = [((u >> (8 * k)) & 0xff).astype(np.uint8) for k in range(bps)]   # b[0] = low byte
    if le:
        order = b
    elif bps == 3:
        order = [b[1], b[2], b[0]]   # reads24be: p[2] | p[0] << 8 | p[1] << 16
    else:
        order = b[::-1]
    return np.stack(order, axis=-1).reshape(v.shape[:-1] + (v.shape[-1] * bps,))


def _rows(ints, bps, le, widths, perm, head, pad, frame_size):
    """ints [S][F][ch][fs] -> packet rows [S][F][row bytes] + layout.  Sub-stream j carries widths[j] channels (1 = mono
    packet, 2 = coupled: samples interleaved); perm[c] = the decoded channel output channel c takes."""
    S, F, ch, fs = ints.shape
    assert sum(widths) == ch and fs == frame_size
    off = head
    ch_off, ch_step = [], []
    pieces = []
    c = 0
    for w in widths:
        blk = ints[:, :, c:c + w, :]                       # [S][F][w][fs]
        inter = np.ascontiguousarray(blk.transpose(0, 1, 3, 2)).reshape(S, F, fs * w)
        pieces.append((off, _pack(inter, bps, le)))
        for k in range(w):
            ch_off.append(off + k * bps)
            ch_step.append(w * bps)
        off += w * bps * fs + pad
        c += w
    row = (off + 15) & ~15
    raw = np.zeros((S, F, row), dtype=np.uint8)
    for o, data in pieces:
        raw[:, :, o:o + data.shape[-1]] = data
    L = A.LpcmLayout()
    L.sample_bytes, L.little_endian, L.channels, L.frame_size = bps, 1 if le else 0, ch, fs
    for p in range(ch):
        L.src_offset[p] = ch_off[perm[p]]
        L.src_step[p] = ch_step[perm[p]]
    return raw, L, row


def _render_lpcm(matrix, out_ch, raw, L, row, frame_size, calls, first=0, n_samples=0, fmt=A.FMT_S16):
    S, F, _ = raw.shape
    d_raw = torch.from_numpy(raw).cuda()
    bps_out = {A.FMT_S16: 2, A.FMT_S24: 3, A.FMT_S32: 4}[fmt]
    b = A.Batch(S, matrix, out_ch, frame_size=frame_size, out_format=fmt, limiter=True)
    st = torch.cuda.current_stream().cuda_stream
    outs = [[] for _ in range(S)]
    f0 = 0
    for nf in calls:
        cap = max(nf * frame_size, 240) * out_ch * bps_out
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        inp = A.LpcmInput()
        inp.d_raw = d_raw.data_ptr() + f0 * row
        inp.raw_stream_stride = F * row
        inp.raw_frame_stride = row
        inp.first_sample = first
        inp.layout = L
        a = A.RenderArgs()
        a.n_frames = nf
        a.n_samples = n_samples
        a.d_pcm = pcm.data_ptr()
        a.pcm_stream_stride_bytes = cap
        a.stream = st
        n = b.render_lpcm(inp, a)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(h[s][:n * out_ch * bps_out].copy())
        f0 += nf
    cap = 240 * out_ch * bps_out
    pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
    n = b.flush(pcm.data_ptr(), cap, st)
    torch.cuda.synchronize()
    h = pcm.cpu().numpy()
    for s in range(S):
        outs[s].append(h[s][:n * out_ch * bps_out].copy())
    b.close()
    return [np.concatenate(o) for o in outs]


def _reference_bytes(matrix, out_ch, ints, bps, perm, frame_size, calls, fmt=A.FMT_S16):
    """the f32 path over sample / 2^(bits-1) in the renderer's channel order"""
    S, F, ch, fs = ints.shape
    x = (ints[:, :, perm, :].astype(np.float64) / float(1 << (8 * bps - 1))).astype(np.float32)   # exact for 16 / 24 bit;
    if bps == 4:                                                                                  # 32 bit: int -> f32 rounds first
        x = ints[:, :, perm, :].astype(np.float32) * np.float32(1.0 / 2147483648.0)
    planar = np.ascontiguousarray(x.transpose(0, 2, 1, 3)).reshape(S, ch, F * fs)
    outs = hip_render(matrix, out_ch, planar, frame_size, fmt=fmt, frames_per_call=calls)
    return [np.ascontiguousarray(o).view(np.uint8).reshape(-1) for o in outs]


def _ints(rng, S, F, ch, fs, bps, level=0.35):
    full = float(1 << (8 * bps - 1))
    v = rng.standard_normal((S, F, ch, fs)) * level * full
    v[:, :, :, ::97] *= 3.0   # peaks: the limiter works
    return np.clip(np.rint(v), -full, full - 1).astype(np.int64)


@pytest.mark.parametrize("order,out", [(3, "binaural"), (2, "binaural"), (1, "stereo"), (0, "stereo")])
@pytest.mark.parametrize("unfused", [False, True])
def test_mono_coded_ambisonics_s16_equals_the_f32_path(order, out, unfused, monkeypatch):
    # the fused kernel's case: (order + 1)^2 mono sub-streams, 16-bit little-endian; rows with a 16-byte head (as the group
    # of decoder handles lays them out) and the channels in a permuted ambisonics channel mapping
    if unfused:
        monkeypatch.setenv("IAMF_HIP_LPCM_UNFUSED", "1")
    rng = np.random.default_rng(100 + order)
    ch, fs, S, F = (order + 1) ** 2, 1024, 5, 6
    out_id = A.SS["BINAURAL"] if out == "binaural" else A.SS["A"]
    mx = A.get_h2m_matrix(order, out_id)
    ints = _ints(rng, S, F, ch, fs, 2)
    perm = list(rng.permutation(ch))
    raw, L, row = _rows(ints, 2, True, [1] * ch, perm, head=16, pad=0, frame_size=fs)
    calls = [2, 1, 3]
    got = _render_lpcm(mx, 2, raw, L, row, fs, calls)
    want = _reference_bytes(mx, 2, ints, 2, perm, fs, calls)
    for s in range(S):
        assert np.array_equal(got[s], want[s]), "stream %d" % s


def test_weights_outside_the_range_of_the_folded_scale_take_the_unfused_form():
    """The fused kernel keeps the LPCM decoder's "/ 32768" in its weights ((w * 2^-15) * (float)s has the bits of
    w * (s * 2^-15) while everything stays normal); a caller's matrix with a weight below 2^-100 must not take that form.
    Same PCM as the f32 path either way — here with weights that would underflow if scaled first."""
    import ctypes as C
    rng = np.random.default_rng(321)
    ch, fs, S, F = 16, 1024, 3, 3
    base = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    w = np.ctypeslib.as_array(base.mat, shape=(base.m * base.n,)).copy()
    w[5] = np.float32(1.5e-38)          # normal, but w * 2^-15 is subnormal
    w[21] = np.float32(-3e-41)          # subnormal already
    w[7] = np.float32(2.0 ** -120)
    keep = w   # the batch copies the matrix at create; keep the array alive until then
    mx = A.Matrix()
    C.memmove(C.byref(mx), C.byref(base), C.sizeof(A.Matrix))
    mx.mat = keep.ctypes.data_as(C.POINTER(C.c_float))
    ints = _ints(rng, S, F, ch, fs, 2)
    perm = list(range(ch))
    raw, L, row = _rows(ints, 2, True, [1] * ch, perm, head=16, pad=0, frame_size=fs)
    got = _render_lpcm(mx, 2, raw, L, row, fs, [F])
    want = _reference_bytes(mx, 2, ints, 2, perm, fs, [F])
    for s in range(S):
        assert np.array_equal(got[s], want[s]), "stream %d" % s


@pytest.mark.parametrize("bps,le", [(2, False), (3, True), (3, False), (4, True), (4, False)])
def test_other_sample_formats_take_the_general_form(bps, le):
    rng = np.random.default_rng(200 + bps * 2 + le)
    ch, fs, S, F = 16, 1024, 3, 3
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    ints = _ints(rng, S, F, ch, fs, bps)
    perm = list(range(ch))
    raw, L, row = _rows(ints, bps, le, [1] * ch, perm, head=0, pad=0, frame_size=fs)
    got = _render_lpcm(mx, 2, raw, L, row, fs, [F])
    want = _reference_bytes(mx, 2, ints, bps, perm, fs, [F])
    for s in range(S):
        assert np.array_equal(got[s], want[s]), "stream %d" % s


def test_coupled_substreams_into_a_wide_layout():
    # 7.1.4 element: 5 coupled + 2 mono sub-streams in audio-layer order, rendered to Sound System J (a wide4 kernel)
    rng = np.random.default_rng(300)
    ch, fs, S, F = 12, 1024, 3, 4
    mx = A.get_m2m_matrix(A.SS["L714"], A.SS["J"])
    ints = _ints(rng, S, F, ch, fs, 2)
    perm = [0, 1, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9]   # some audio-layer -> playback permutation
    raw, L, row = _rows(ints, 2, True, [2, 2, 2, 2, 2, 1, 1], perm, head=16, pad=6, frame_size=fs)
    got = _render_lpcm(mx, 12, raw, L, row, fs, [1, 3])
    want = _reference_bytes(mx, 12, ints, 2, perm, fs, [1, 3])
    for s in range(S):
        assert np.array_equal(got[s], want[s]), "stream %d" % s


@pytest.mark.parametrize("first,count", [(0, 512), (256, 768), (4, 64), (6, 250)])
def test_a_trimmed_frame(first, count):
    # one call of one frame of which samples [first, first + count) are rendered: fused when first is a multiple of 4
    # and count one of 64, else the general form; both against the f32 path over the same samples
    rng = np.random.default_rng(400 + first)
    ch, fs, S = 16, 1024, 4
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    ints = _ints(rng, S, 1, ch, fs, 2)
    perm = list(range(ch))
    raw, L, row = _rows(ints, 2, True, [1] * ch, perm, head=16, pad=0, frame_size=fs)
    got = _render_lpcm(mx, 2, raw, L, row, fs, [1], first=first, n_samples=count)
    # the f32 path: a batch fed the kept samples as a frame shortened by n_samples
    x = (ints[:, 0, :, first:first + count].astype(np.float32) / np.float32(32768.0))
    xin = np.zeros((S, 1, ch, fs), dtype=np.float32)
    xin[:, 0, :, :count] = x
    d = torch.from_numpy(xin).cuda()
    b = A.Batch(S, mx, 2, frame_size=fs, out_format=A.FMT_S16, limiter=True)
    st = torch.cuda.current_stream().cuda_stream
    want = [[] for _ in range(S)]
    cap = fs * 2 * 2
    pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
    a = A.RenderArgs()
    a.d_in = d.data_ptr()
    a.in_stream_stride = ch * fs
    a.in_frame_stride = ch * fs
    a.n_frames = 1
    a.n_samples = count
    a.d_pcm = pcm.data_ptr()
    a.pcm_stream_stride_bytes = cap
    a.stream = st
    n = b.render_ex(a)
    torch.cuda.synchronize()
    h = pcm.cpu().numpy()
    for s in range(S):
        want[s].append(h[s][:n * 4].copy())
    pcm2 = torch.zeros((S, 240 * 4), dtype=torch.uint8, device="cuda")
    n = b.flush(pcm2.data_ptr(), 240 * 4, st)
    torch.cuda.synchronize()
    h = pcm2.cpu().numpy()
    for s in range(S):
        want[s].append(h[s][:n * 4].copy())
    b.close()
    for s in range(S):
        assert np.array_equal(got[s], np.concatenate(want[s])), "stream %d" % s


def test_bad_arguments_are_refused():
    mx = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    b = A.Batch(2, mx, 2, frame_size=1024, out_format=A.FMT_S16, limiter=True)
    raw = torch.zeros((2, 16 * 2048 + 16), dtype=torch.uint8, device="cuda")
    pcm = torch.zeros((2, 4096), dtype=torch.uint8, device="cuda")
    L = A.LpcmLayout()
    L.sample_bytes, L.little_endian, L.channels, L.frame_size = 2, 1, 16, 1024
    for c in range(16):
        L.src_offset[c], L.src_step[c] = 16 + 2048 * c, 2
    inp = A.LpcmInput()
    inp.d_raw, inp.raw_stream_stride, inp.raw_frame_stride, inp.layout = raw.data_ptr(), 16 * 2048 + 16, 16 * 2048 + 16, L
    a = A.RenderArgs()
    a.n_frames, a.d_pcm, a.pcm_stream_stride_bytes = 1, pcm.data_ptr(), 4096
    assert b.render_lpcm(inp, a) == 1024 - 240          # the well-formed call
    inp.raw_frame_stride = 16 * 2048                     # the last channel's packet would end outside the row
    with pytest.raises(A.IamfHipError):
        b.render_lpcm(inp, a)
    inp.raw_frame_stride = 16 * 2048 + 16
    L2 = A.LpcmLayout.from_buffer_copy(L)
    L2.channels = 9                                      # not the element's channel count
    inp.layout = L2
    with pytest.raises(A.IamfHipError):
        b.render_lpcm(inp, a)
    inp.layout = L
    inp.first_sample = 8                                 # a trimmed start needs n_samples
    with pytest.raises(A.IamfHipError):
        b.render_lpcm(inp, a)
    inp.first_sample = 0
    a.d_in = raw.data_ptr()                              # d_in must be NULL
    with pytest.raises(A.IamfHipError):
        b.render_lpcm(inp, a)
    torch.cuda.synchronize()
    b.close()
