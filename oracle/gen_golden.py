#!/usr/bin/env python3
"""Generate tests/golden/*.npz by calling the REAL reference (oracle/_ref/libiamf_ref.so).

TEST INFRASTRUCTURE.  Runs only in the authoring container (needs /root/reference to have been
compiled by `make -C oracle ref`).  The reference's exported stage symbols are called through
ctypes on seeded inputs (tests/synth.py); only the OUTPUTS and the case parameters are stored.
Stage symbols used (all exported by the reference .so):
  IAMF_element_renderer_get_H2M_matrix / _render_H2M      src/iamf_dec/h2m_rdr.c:1070,1088
  IAMF_element_renderer_get_M2M_matrix / _render_M2M      src/iamf_dec/m2m_rdr.c:1786,1820
  audio_effect_peak_limiter_create/_init/_process_block   src/iamf_dec/audio_effect_peak_limiter.c
  DMRenderer_open/_set_mode_weight/_downmix               src/iamf_dec/downmix_renderer.c
  speex_resampler_*                                       src/iamf_dec/resample.c
  IAMF_decoder_* (whole pipeline on synthetic LPCM .iamf) src/iamf_dec/IAMF_decoder.c
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

ref = C.CDLL(os.path.join(HERE, "_ref", "libiamf_ref.so"))

FP = C.POINTER(C.c_float)


class PredefSP(C.Structure):
    _fields_ = [("system", C.c_int), ("lfe1", C.c_int), ("lfe2", C.c_int)]


class LfeFilter(C.Structure):
    _fields_ = [("init", C.c_int), ("c", C.c_float), ("a1", C.c_float), ("a2", C.c_float),
                ("a3", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
                ("ih", C.c_float * 2), ("oh", C.c_float * 2)]


class SPLayout(C.Structure):
    _fields_ = [("sp_type", C.c_int), ("predefined_sp", C.POINTER(PredefSP)), ("lfe_f", LfeFilter)]


class HoaLayout(C.Structure):
    _fields_ = [("order", C.c_int), ("lfe_on", C.c_int)]


class H2M(C.Structure):
    _fields_ = [("in_", C.c_int), ("out", C.c_int), ("channels", C.c_int), ("lfe1", C.c_int),
                ("lfe2", C.c_int), ("mat", FP), ("m", C.c_int), ("n", C.c_int)]


class M2M(C.Structure):
    _fields_ = [("in_", C.c_int), ("out", C.c_int), ("mat", FP), ("m", C.c_int), ("n", C.c_int)]


SS = dict(A=0x020, B=0x050, C=0x250, D=0x450, E=0x451, F=0x370, G=0x490, H=0x9A3, I=0x070,
          J=0x470, STEREO=0x200, L51=0x510, L512=0x512, L514=0x514, L71=0x710, L714=0x714,
          MONO=0x100, L712=0x712, L312=0x312, BINAURAL=0x1020)
# channels of each output layout (IAMF_decoder.c:3998-4008 / sound-system table)
OUT_CH = {0x020: 2, 0x050: 6, 0x250: 8, 0x450: 10, 0x451: 11, 0x370: 12, 0x490: 14, 0x9A3: 24,
          0x070: 8, 0x470: 12, 0x712: 10, 0x312: 6, 0x1020: 2, 0x100: 1}


def rows(a):
    """array of float* to the rows of a C-contiguous 2-D float32 array"""
    arr = (FP * a.shape[0])()
    for i in range(a.shape[0]):
        arr[i] = a[i].ctypes.data_as(FP)
    return arr


def ref_h2m(order, out_id, x, prefill=0.0):
    hin = HoaLayout(order, 0)
    pout = PredefSP(out_id, 0, 0)
    h = H2M()
    assert ref.IAMF_element_renderer_get_H2M_matrix(C.byref(hin), C.byref(pout), C.byref(h)) == 0
    ns = x.shape[1]
    out = np.full((max(OUT_CH[out_id], h.n + 2), ns), prefill, dtype=np.float32)
    ref.IAMF_element_renderer_render_H2M(C.byref(h), rows(x), rows(out), ns, None)
    return out[:OUT_CH[out_id]].copy()


def ref_m2m(in_id, out_id, x):
    pin, pout = PredefSP(in_id, 0, 0), PredefSP(out_id, 0, 0)
    lin, lout = SPLayout(), SPLayout()
    lin.predefined_sp = C.pointer(pin)
    lout.predefined_sp = C.pointer(pout)
    m = M2M()
    assert ref.IAMF_element_renderer_get_M2M_matrix(C.byref(lin), C.byref(lout), C.byref(m)) == 0
    ns = x.shape[1]
    out = np.zeros((m.n, ns), dtype=np.float32)
    ref.IAMF_element_renderer_render_M2M(C.byref(m), rows(x), rows(out), ns)
    return out


ref.audio_effect_peak_limiter_create.restype = C.c_void_p
ref.audio_effect_peak_limiter_init.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float,
                                               C.c_float, C.c_int]
ref.audio_effect_peak_limiter_process_block.argtypes = [C.c_void_p, FP, FP, C.c_int]
ref.audio_effect_peak_limiter_destroy.argtypes = [C.c_void_p]


def ref_limiter(x, frame_sizes, thr_db=-1.0, rate=48000, flush=True):
    """x: [ch][total]; processed in blocks of the given sizes; returns concatenated planar output
    [ch][n_out] and the per-call return values (flush = one extra call with `delay` zeros)."""
    ch = x.shape[0]
    lim = ref.audio_effect_peak_limiter_create()
    ref.audio_effect_peak_limiter_init(lim, thr_db, rate, ch, 0.001, 0.200, 240)
    outs, rets, pos = [], [], 0
    blocks = [x[:, pos0:pos0 + n] for pos0, n in zip(np.cumsum([0] + frame_sizes[:-1]), frame_sizes)]
    if flush:
        blocks.append(np.zeros((ch, 240), dtype=np.float32))
    for b in blocks:
        b = np.ascontiguousarray(b, dtype=np.float32)
        n = b.shape[1]
        o = np.zeros((ch, n), dtype=np.float32)
        r = ref.audio_effect_peak_limiter_process_block(lim, b.ctypes.data_as(FP), o.ctypes.data_as(FP), n)
        rets.append(r)
        outs.append(o.reshape(-1)[:ch * r].reshape(ch, r).copy())
    ref.audio_effect_peak_limiter_destroy(lim)
    return np.concatenate(outs, axis=1), rets


ref.DMRenderer_open.restype = C.c_void_p
ref.DMRenderer_open.argtypes = [C.c_int, C.c_int]
ref.DMRenderer_close.argtypes = [C.c_void_p]
ref.DMRenderer_set_mode_weight.argtypes = [C.c_void_p, C.c_int, C.c_int]
ref.DMRenderer_downmix.argtypes = [C.c_void_p, FP, FP, C.c_uint32, C.c_uint32, C.c_uint32]
LAYOUT_CH = [1, 2, 6, 8, 10, 8, 10, 12, 6, 2]


def ref_downmix(in_l, out_l, x, schedule, default_mode, default_w):
    """schedule: list of (mode, offset) per frame, frame = x[f]; mirrors IAMF_decoder.c:2574-2583:
    samples [0, offset) use the previous mode/weight, the rest the new one."""
    d = ref.DMRenderer_open(in_l, out_l)
    if not d:
        return None
    ref.DMRenderer_set_mode_weight(d, default_mode, default_w)
    outs = []
    for f, (mode, off) in enumerate(schedule):
        xi = np.ascontiguousarray(x[f], dtype=np.float32)
        ns = xi.shape[1]
        o = np.zeros((LAYOUT_CH[out_l], ns), dtype=np.float32)
        if off:
            ref.DMRenderer_downmix(d, xi.ctypes.data_as(FP), o.ctypes.data_as(FP), 0, off, ns)
        if mode > -1:
            ref.DMRenderer_set_mode_weight(d, mode, -1)
        if ns > off:
            ref.DMRenderer_downmix(d, xi.ctypes.data_as(FP), o.ctypes.data_as(FP), off, ns - off, ns)
        outs.append(o)
    ref.DMRenderer_close(d)
    return np.stack(outs)


def main():
    manifest = {}

    # ---- K2: HOA -> layout ----
    h2m_cases = [("toa_H", 3, SS["H"]), ("toa_A", 3, SS["A"]), ("toa_BIN", 3, SS["BINAURAL"]),
                 ("toa_J", 3, SS["J"]), ("toa_F", 3, SS["F"]), ("toa_G", 3, SS["G"]),
                 ("soa_H", 2, SS["H"]), ("foa_J", 1, SS["J"]), ("zoa_MONO", 0, SS["MONO"]),
                 ("foa_B", 1, SS["B"]), ("soa_312", 2, SS["L312"])]
    out = {}
    for name, order, oid in h2m_cases:
        m = (order + 1) ** 2
        x = synth.gaussian(13 + order, m, 320, 0.15)
        out[name] = ref_h2m(order, oid, x)
        manifest["h2m/" + name] = dict(order=order, out_id=oid, seed=13 + order, ns=320, sigma=0.15)
    # sentinel case: which slots does the reference leave untouched?
    x = synth.gaussian(16, 16, 64, 0.15)
    out["toa_H_sentinel"] = ref_h2m(3, SS["H"], x, prefill=7.0)
    manifest["h2m/toa_H_sentinel"] = dict(order=3, out_id=SS["H"], seed=16, ns=64, sigma=0.15, prefill=7.0)
    np.savez_compressed(os.path.join(GOLD, "h2m.npz"), **out)

    # ---- K1: layout -> layout ----
    m2m_cases = [("714_J", SS["L714"], SS["J"], 12), ("714_H", SS["L714"], SS["H"], 12),
                 ("714_A", SS["L714"], SS["A"], 12), ("stereo_A", SS["STEREO"], SS["A"], 2),
                 ("51_B", SS["L51"], SS["B"], 6), ("mono_A", SS["MONO"], SS["A"], 1),
                 ("512_D", SS["L512"], SS["D"], 8), ("714_BIN", SS["L714"], SS["BINAURAL"], 12),
                 ("312_G", SS["L312"], SS["G"], 6)]
    out = {}
    for name, iid, oid, m in m2m_cases:
        x = synth.uniform(11, m, 320, 0.5)
        out[name] = ref_m2m(iid, oid, x)
        manifest["m2m/" + name] = dict(in_id=iid, out_id=oid, m=m, seed=11, ns=320, amp=0.5)
    np.savez_compressed(os.path.join(GOLD, "m2m.npz"), **out)

    # ---- K8: limiter ----
    out = {}
    lim_cases = [
        ("hot2", 2, [1024] * 30, "hot", 1000),
        ("quiet2", 2, [1024] * 4, "quiet", 1001),
        ("hot24", 24, [1024] * 3, "hot", 1002),
        ("hot2_960", 2, [960] * 6, "hot", 1003),
        ("hot2_ragged", 2, [100, 100, 100, 1, 239, 1024, 7, 2048], "hot", 1004),
        ("hot12", 12, [1024] * 3, "hot", 1005),
    ]
    for name, ch, sizes, kind, seed in lim_cases:
        total = sum(sizes)
        if kind == "hot":
            x = synth.hot(seed, ch, total, sigma=0.25, burst_phase=700, burst_period=6000)
        else:
            x = synth.quiet(seed, ch, total)
        y, rets = ref_limiter(x, sizes)
        out[name] = y
        out[name + "_rets"] = np.array(rets, dtype=np.int32)
        manifest["limiter/" + name] = dict(ch=ch, sizes=sizes, kind=kind, seed=seed)
    np.savez_compressed(os.path.join(GOLD, "limiter.npz"), **out)

    # ---- K3: parametric down-mix ----
    out = {}
    dmx_cases = [("714_512", 7, 3), ("714_312", 7, 8), ("714_514", 7, 4), ("714_712", 7, 6),
                 ("512_312", 3, 8), ("514_512", 4, 3), ("514_312", 4, 8), ("712_512", 6, 3),
                 ("712_312", 6, 8), ("710_510", 5, 2), ("510_stereo", 2, 1), ("710_stereo", 5, 1),
                 ("710_mono", 5, 0), ("stereo_mono", 1, 0), ("510_mono", 2, 0)]
    schedule = [(0, 0), (1, 0), (1, 50), (2, 0), (4, 37), (5, 0), (6, 0), (6, 0), (0, 12), (4, 0),
                (4, 0), (4, 0), (4, 0)]
    for name, il, ol in dmx_cases:
        x = np.stack([synth.uniform(300 + f, LAYOUT_CH[il], 96, 0.5) for f in range(len(schedule))])
        y = ref_downmix(il, ol, x, schedule, 1, 3)
        assert y is not None, name
        out[name] = y
        manifest["dmx/" + name] = dict(in_layout=il, out_layout=ol, seed0=300, ns=96,
                                       schedule=schedule, default_mode=1, default_w=3)
    # invalid pairs must refuse to open
    invalid = [(7, 1), (7, 2), (3, 2), (1, 7), (2, 3), (9, 1), (7, 9), (7, 7)]
    manifest["dmx/_invalid"] = [[a, b] for a, b in invalid if ref.DMRenderer_open(a, b) is None]
    assert len(manifest["dmx/_invalid"]) == len(invalid)
    np.savez_compressed(os.path.join(GOLD, "dmx.npz"), **out)

    # ---- K6: resampler (speex-derived), driven like iamf_resample (IAMF_decoder.c:3223-3248) ----
    ref.speex_resampler_init.restype = C.c_void_p
    ref.speex_resampler_init.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
    ref.speex_resampler_skip_zeros.argtypes = [C.c_void_p]
    ref.speex_resampler_process_interleaved_float.argtypes = [C.c_void_p, FP, C.POINTER(C.c_uint32), FP,
                                                              C.POINTER(C.c_uint32)]
    ref.speex_resampler_get_input_latency.argtypes = [C.c_void_p]
    ref.speex_resampler_get_output_latency.argtypes = [C.c_void_p]
    ref.speex_resampler_destroy.argtypes = [C.c_void_p]
    out = {}
    rs_cases = [("441_48_2ch", 44100, 48000, 2, [1024] * 6), ("16_48_2ch", 16000, 48000, 2, [960] * 4),
                ("48_16_6ch", 48000, 16000, 6, [1024] * 4), ("32_48_ragged", 32000, 48000, 1, [100, 333, 1024, 7, 640]),
                ("24_48_12ch", 24000, 48000, 12, [512] * 3), ("48_441_2ch", 48000, 44100, 2, [1024] * 4),
                ("8_48_1ch", 8000, 48000, 1, [160] * 5)]
    for name, ir, orate, ch, sizes in rs_cases:
        total = sum(sizes)
        x = synth.hot(600 + ch, ch, total, sigma=0.3, burst_amp=1.2, burst_len=60, burst_phase=50, burst_period=700)
        err = C.c_int(0)
        st = ref.speex_resampler_init(ch, ir, orate, 4, C.byref(err))
        assert st and err.value == 0
        ref.speex_resampler_skip_zeros(st)
        outs, rets, pos = [], [], 0
        for ns in sizes:
            inter = np.ascontiguousarray(x[:, pos:pos + ns].T)
            pos += ns
            cap = ns * (orate // ir + 1)
            o = np.zeros((cap, ch), dtype=np.float32)
            il, ol = C.c_uint32(ns), C.c_uint32(cap)
            ref.speex_resampler_process_interleaved_float(st, inter.ctypes.data_as(FP), C.byref(il),
                                                          o.ctypes.data_as(FP), C.byref(ol))
            assert il.value == ns
            outs.append(o[:ol.value].T.copy())
            rets.append(ol.value)
        il = C.c_uint32(ref.speex_resampler_get_input_latency(st))
        ol = C.c_uint32(ref.speex_resampler_get_output_latency(st))
        o = np.zeros((max(ol.value, 1), ch), dtype=np.float32)
        ref.speex_resampler_process_interleaved_float(st, None, C.byref(il), o.ctypes.data_as(FP), C.byref(ol))
        outs.append(o[:ol.value].T.copy())
        rets.append(ol.value)
        ref.speex_resampler_destroy(st)
        out[name] = np.concatenate(outs, axis=1)
        out[name + "_rets"] = np.array(rets, dtype=np.int32)
        manifest["resample/" + name] = dict(in_rate=ir, out_rate=orate, ch=ch, sizes=sizes, seed=600 + ch)
    np.savez_compressed(os.path.join(GOLD, "resample.npz"), **out)

    extra = os.path.join(HERE, "gen_golden_extra.py")
    if os.path.exists(extra):
        import importlib.util
        spec = importlib.util.spec_from_file_location("gen_golden_extra", extra)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.generate(ref, manifest, GOLD, synth)

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
