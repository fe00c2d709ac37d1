"""Helpers for the -m gpu tests: drive libiamf_hip.so through its C ABI with torch tensors as
plain device memory."""
import numpy as np
import torch

import iac_amd as A

_NP_OUT = {A.FMT_S16: np.int16, A.FMT_S32: np.int32, A.FMT_F32: np.float32, A.FMT_S24: np.uint8}


def to_frames(x, frame_size):
    """[S][m][total] planar -> [S][F][m][frame_size] (the ABI's input layout)"""
    S, m, total = x.shape
    assert total % frame_size == 0
    F = total // frame_size
    return np.ascontiguousarray(x.reshape(S, m, F, frame_size).transpose(0, 2, 1, 3))


def hip_render(matrix, out_ch, x, frame_size, fmt=A.FMT_S16, limiter=True, flush=True,
               frames_per_call=None, gains=None, loudness=False, threshold_db=-1.0,
               sample_rate=48000, projection=0, fir_taps=0, lfe_hoa=False):
    """x: numpy [S][m][total].  Returns a list (per stream) of arrays [n_out][out_ch] (S24:
    [n_out][out_ch][3] bytes) — everything the calls emitted, concatenated."""
    S, m, total = x.shape
    xin = torch.from_numpy(to_frames(x, frame_size)).cuda()
    F = total // frame_size
    bps = {A.FMT_S16: 2, A.FMT_S24: 3, A.FMT_S32: 4, A.FMT_F32: 4}[fmt]
    b = A.Batch(S, matrix, out_ch, frame_size=frame_size, sample_rate=sample_rate, out_format=fmt,
                limiter=limiter, threshold_db=threshold_db, loudness=loudness, projection=projection,
                fir_taps=fir_taps, lfe_hoa=lfe_hoa)
    if gains:
        b.set_gains(**gains)
    calls = frames_per_call or [F]
    assert sum(calls) == F
    stream_stride = F * m * frame_size
    frame_stride = m * frame_size
    outs = [[] for _ in range(S)]
    f0 = 0
    st = torch.cuda.current_stream().cuda_stream
    for nf in calls:
        cap = max(nf * frame_size, 240) * out_ch * bps
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        n = b.render(xin.data_ptr() + 4 * f0 * frame_stride, stream_stride, frame_stride, nf,
                     pcm.data_ptr(), cap, st)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(_view(h[s], n, out_ch, fmt))
        f0 += nf
    if flush:
        cap = 240 * out_ch * bps
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        n = b.flush(pcm.data_ptr(), cap, st)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(_view(h[s], n, out_ch, fmt))
    b.close()
    return [np.concatenate(o, axis=0) for o in outs]


def _view(raw, n, ch, fmt):
    if fmt == A.FMT_S24:
        return raw[:n * ch * 3].reshape(n, ch, 3).copy()
    dt = _NP_OUT[fmt]
    return raw[:n * ch * np.dtype(dt).itemsize].view(dt).reshape(n, ch).copy()


def identity_matrix(ch):
    """an M2M matrix that passes ch channels straight through (for limiter / pack stage tests)"""
    import ctypes as C
    eye = np.eye(ch, dtype=np.float32)
    m = A.Matrix()
    m.kind, m.in_id, m.out_id, m.channels, m.lfe1, m.lfe2, m.m, m.n = A.KIND_M2M, 0, 0, ch, -1, -1, ch, ch
    m.mat = eye.ctypes.data_as(C.POINTER(C.c_float))
    m._keep = eye
    return m
