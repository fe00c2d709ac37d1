"""What the oracle says a bench.py workload's PCM must be.  TEST INFRASTRUCTURE ONLY (the checker of
bench.py's `verified` entry and of tests/test_gpu_bench_geometry.py): everything here runs on the host
through oracle/liboracle.so; nothing here is measured or shipped.

A bench launch renders, per stream, F frames of `fs` samples from a state that was just reset, so the
oracle's answer for stream s is: element renderer (h2m_rdr.c:1088-1150 / m2m_rdr.c:1820-1840 /
downmix_renderer.c / demixer.c in front) -> limiter (audio_effect_peak_limiter.c:94-204, which withholds the
first 240 sample-frames) -> FLOAT2INT16 (IAMF_decoder.c:100-167): F*fs - 240 sample-frames.
"""
import ctypes as C

import numpy as np

import oracle_lib as O

DEMIX_MODES = (0, 1, 2, 4, 5, 6)
DEMIX_LAYERS = [1, 3, 7]
DEMIX_LAYER_GAINS = {0: (0b110000, 0.7079458), 1: (0b001111, 1.4125376)}


def demix_recon_gains(s, f, n_rec):
    """the recon gains bench.py's scalable workload gives frame f of stream s (python floats -> c_float)"""
    return [0.75 + 0.25 * ((s * 7 + f * 3 + i) % 16) / 15.0 for i in range(n_rec)]


def planar(x_sfcn):
    """[F][ch][fs] -> [ch][F*fs]"""
    F, ch, fs = x_sfcn.shape
    return np.ascontiguousarray(x_sfcn.transpose(1, 0, 2).reshape(ch, F * fs))


def tolerance_lsb(kind, out_ch):
    """bit-exact everywhere except where the product's contract is +-1 LSB: the MFMA projection of an HOA element into a
    layout wider than stereo (include/iamf_hip.h IAMF_HIP_PROJ_AUTO) and the composed projection-mode matrix."""
    if kind == "h2m_proj":
        return 1
    if kind in ("h2m", "h2m_lfe") and out_ch > 2:
        return 1
    return 0


def oracle_pcm(kind, in_id, out_id, out_ch, x, fs, s=0, proj=None):
    """x: [F][in_ch][fs] float32 — the element PCM of stream `s` exactly as the kernel read it.
    Returns int16 [F*fs - 240][out_ch]."""
    F = x.shape[0]
    if kind in ("h2m", "h2m_lfe", "h2m_proj", "m2m", "h2m_lpcm"):   # h2m_lpcm: x is the decoded packets, sample / 32768
        mx = O.get_m2m(in_id, out_id) if kind == "m2m" else O.get_h2m(in_id, out_id)
        xp = planar(x)
        if kind == "h2m_proj":   # iamf_core_decoder_convert_projection (IAMF_core_decoder.c:116-130): f32, ascending l
            l_in, m = proj.shape
            acc = np.zeros((m, xp.shape[1]), dtype=np.float32)
            for l in range(l_in):
                acc = (acc + (xp[l][None, :] * proj[l][:, None]).astype(np.float32)).astype(np.float32)
            xp = acc
        kw = {"lfe_rate": 48000} if kind == "h2m_lfe" else {}
        return O.stream_run(mx, out_ch, xp, fs, flush=False, **kw)
    if kind == "dmx":
        sched = [(DEMIX_MODES[(s + f) % 6], 0) for f in range(F)]
        y = O.downmix_run(in_id, out_id, x, sched, 1, 3)          # [F][oc][fs]
        z, _ = O.limiter_run(planar(y), [fs] * F, flush=False)
        return O.pack(z, 16)
    if kind == "demix":
        import demix_cases as D
        c = D.make_case(DEMIX_LAYERS, DEMIX_LAYER_GAINS, default=(1, 3), offset=0, fs=fs)
        c["schedule"] = [(DEMIX_MODES[(s + f) % 6], [float(np.float32(g)) for g in demix_recon_gains(s, f, len(c["recon"]))])
                         for f in range(F)]
        dem = D.drive_demixer(O.lib(), "orc_demixer_", c, x)        # [F][12][fs], 7.1.4 playback order
        return O.stream_run(O.get_m2m(in_id, out_id), out_ch, planar(dem), fs, flush=False)
    raise KeyError(kind)


def fir64(h, xp):
    """this repo's HRTF specification in float64 (tests/test_gpu_fir.py): h [2][ch][taps], xp [ch][n] -> [2][n]"""
    n = xp.shape[1]
    y = np.zeros((2, n))
    for e in range(2):
        for c in range(xp.shape[0]):
            y[e] += np.convolve(xp[c].astype(np.float64), h[e, c].astype(np.float64))[:n]
    return y


def fir_pcm_from_stage(y_stage, fs, F):
    """limiter + pack of the oracle on the FIR stage's OWN f32 output ([n][2]): what the kernel's PCM must equal bit for
    bit (everything behind the FIR stage is the reference's arithmetic)"""
    z, _ = O.limiter_run(np.ascontiguousarray(y_stage.T, dtype=np.float32), [fs] * F, flush=False)
    return O.pack(z, 16)


def compare(got, want, tol):
    """-> (ok, max_lsb, differing fraction)"""
    if got.shape != want.shape:
        return False, None, None
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    return bool(d.max(initial=0) <= tol), int(d.max(initial=0)), float((d > 0).mean()) if d.size else 0.0
