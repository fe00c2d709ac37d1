"""ctypes access to oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
TABLES = os.path.join(ROOT, "iac_amd", "data", "rdr_tables.bin")
FP = C.POINTER(C.c_float)

SS = dict(A=0x020, B=0x050, C=0x250, D=0x450, E=0x451, F=0x370, G=0x490, H=0x9A3, I=0x070,
          J=0x470, STEREO=0x200, L51=0x510, L512=0x512, L514=0x514, L71=0x710, L714=0x714,
          MONO=0x100, L712=0x712, L312=0x312, BINAURAL=0x1020)
OUT_CH = {0x020: 2, 0x050: 6, 0x250: 8, 0x450: 10, 0x451: 11, 0x370: 12, 0x490: 14, 0x9A3: 24,
          0x070: 8, 0x470: 12, 0x712: 10, 0x312: 6, 0x1020: 2, 0x100: 1}
LAYOUT_CH = [1, 2, 6, 8, 10, 8, 10, 12, 6, 2]


class Matrix(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_id", C.c_int), ("out_id", C.c_int), ("channels", C.c_int),
                ("lfe1", C.c_int), ("lfe2", C.c_int), ("m", C.c_int), ("n", C.c_int), ("mat", FP)]

    def array(self):
        return np.ctypeslib.as_array(self.mat, shape=(self.m * self.n,)).copy()


_lib = None


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.startswith("iamf_oracle")]
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_tables_load.argtypes = [C.c_char_p]
        L.orc_get_h2m.argtypes = [C.c_int, C.c_int, C.POINTER(Matrix)]
        L.orc_get_m2m.argtypes = [C.c_int, C.c_int, C.POINTER(Matrix)]
        L.orc_tables_entry.argtypes = [C.c_int, C.POINTER(Matrix)]
        L.orc_render_h2m.argtypes = [C.POINTER(Matrix), FP, FP, C.c_int]
        L.orc_render_m2m.argtypes = [C.POINTER(Matrix), FP, FP, C.c_int]
        L.orc_frame_gain_const.argtypes = [FP, C.c_int, C.c_int, C.c_float]
        L.orc_frame_gain_ramp.argtypes = [FP, C.c_int, C.c_int, FP]
        L.orc_mix_gain_linear.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, FP]
        L.orc_mix_gain_quad.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int,
                                        C.c_int, FP]
        L.orc_loudness.argtypes = [FP, C.c_int, C.c_int, C.c_float]
        L.orc_db2lin.argtypes = [C.c_float]
        L.orc_db2lin.restype = C.c_float
        L.orc_limiter_init.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float,
                                       C.c_int]
        L.orc_limiter_process.argtypes = [C.c_void_p, FP, FP, C.c_int]
        L.orc_pack.argtypes = [C.c_void_p, FP, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_dmx_open.restype = C.c_void_p
        L.orc_dmx_open.argtypes = [C.c_int, C.c_int]
        L.orc_dmx_close.argtypes = [C.c_void_p]
        L.orc_dmx_set_mode_weight.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_dmx_downmix.argtypes = [C.c_void_p, FP, FP, C.c_int, C.c_int, C.c_int]
        L.orc_stream_open.argtypes = [C.c_void_p, C.POINTER(Matrix), C.c_int, C.c_float, C.c_float,
                                      C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_stream_close.argtypes = [C.c_void_p]
        L.orc_stream_enable_lfe.argtypes = [C.c_void_p, C.c_int]
        L.orc_lfe_init.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_render_h2m_lfe.argtypes = [C.POINTER(Matrix), FP, FP, C.c_int, C.c_void_p]
        L.orc_stream_frame.argtypes = [C.c_void_p, FP, C.c_int, C.c_void_p]
        L.orc_stream_flush.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_stream_run_frames.restype = C.c_long
        L.orc_stream_run_frames.argtypes = [C.POINTER(Matrix), C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                            FP, C.c_int, C.c_int, C.c_void_p]
        if hasattr(L, "orc_resampler_open"):
            L.orc_resampler_open.restype = C.c_void_p
            L.orc_resampler_open.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
            L.orc_resampler_close.argtypes = [C.c_void_p]
            L.orc_resample.argtypes = [C.c_void_p, FP, FP, C.c_int]
            L.orc_resample_flush.argtypes = [C.c_void_p, FP]
            L.orc_resample_out_capacity.argtypes = [C.c_void_p, C.c_int]
            L.orc_resample_flush_capacity.argtypes = [C.c_void_p]
        assert L.orc_tables_load(TABLES.encode()) == 0
        _lib = L
    return _lib


def fp(a):
    return a.ctypes.data_as(FP)


def get_h2m(order, out_id):
    m = Matrix()
    assert lib().orc_get_h2m(order, out_id, C.byref(m)) == 0, (order, hex(out_id))
    return m


def get_m2m(in_id, out_id):
    m = Matrix()
    assert lib().orc_get_m2m(in_id, out_id, C.byref(m)) == 0, (hex(in_id), hex(out_id))
    return m


def render(mx, x, out_channels, prefill=0.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ns = x.shape[1]
    out = np.full((max(out_channels, mx.n + 2), ns), prefill, dtype=np.float32)
    if mx.kind == 0:
        lib().orc_render_h2m(C.byref(mx), fp(x), fp(out), ns)
    else:
        lib().orc_render_m2m(C.byref(mx), fp(x), fp(out), ns)
    return out[:out_channels].copy()




def render_h2m_lfe(mx, x, out_channels, rate, sizes):
    """orc_render_h2m_lfe over consecutive calls with one filter (state carried): [out_channels][total]"""
    L = lib()
    f = C.create_string_buffer(L.orc_sizeof_lfe())
    L.orc_lfe_init(f, 120.0, float(rate))
    parts, pos = [], 0
    for ns in sizes:
        xi = np.ascontiguousarray(x[:, pos:pos + ns], dtype=np.float32)
        o = np.zeros((max(out_channels, mx.n + 2), ns), dtype=np.float32)
        L.orc_render_h2m_lfe(C.byref(mx), fp(xi), fp(o), ns, f)
        parts.append(o[:out_channels].copy())
        pos += ns
    return np.concatenate(parts, axis=1)


class Limiter:
    def __init__(self, ch, thr_db=-1.0, rate=48000, atk=0.001, rel=0.200, delay=240):
        self.buf = C.create_string_buffer(lib().orc_sizeof_limiter())
        self.ch = ch
        lib().orc_limiter_init(self.buf, thr_db, rate, ch, atk, rel, delay)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[1]
        o = np.zeros((self.ch, max(n, 1)), dtype=np.float32)
        r = lib().orc_limiter_process(self.buf, fp(x), fp(o), n)
        return o.reshape(-1)[:self.ch * r].reshape(self.ch, r).copy(), r


def limiter_run(x, sizes, flush=True, **kw):
    lim = Limiter(x.shape[0], **kw)
    outs, rets, pos = [], [], 0
    for n in sizes:
        o, r = lim.process(x[:, pos:pos + n])
        pos += n
        outs.append(o)
        rets.append(r)
    if flush:
        o, r = lim.process(np.zeros((x.shape[0], 240), dtype=np.float32))
        outs.append(o)
        rets.append(r)
    return np.concatenate(outs, axis=1), rets


def pack(x, bit_depth, stride=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, ns = x.shape
    stride = stride or ch
    dst = np.zeros(ns * stride * bit_depth // 8, dtype=np.uint8)
    lib().orc_pack(dst.ctypes.data_as(C.c_void_p), fp(x), ns, ch, bit_depth, stride)
    if bit_depth == 16:
        return dst.view(np.int16).reshape(ns, stride)
    if bit_depth == 32:
        return dst.view(np.int32).reshape(ns, stride)
    return dst.reshape(ns, stride, 3)


class Downmixer:
    def __init__(self, in_l, out_l):
        self.h = lib().orc_dmx_open(in_l, out_l)
        self.out_ch = LAYOUT_CH[out_l] if 0 <= out_l < 10 else 0

    def ok(self):
        return bool(self.h)

    def set_mode_weight(self, mode, w):
        return lib().orc_dmx_set_mode_weight(self.h, mode, w)

    def downmix(self, x, out, s, dur):
        ns = x.shape[1]
        return lib().orc_dmx_downmix(self.h, fp(x), fp(out), s, dur, ns)

    def close(self):
        if self.h:
            lib().orc_dmx_close(self.h)
            self.h = None


def downmix_run(in_l, out_l, x, schedule, default_mode, default_w):
    d = Downmixer(in_l, out_l)
    if not d.ok():
        return None
    d.set_mode_weight(default_mode, default_w)
    outs = []
    for f, (mode, off) in enumerate(schedule):
        xi = np.ascontiguousarray(x[f], dtype=np.float32)
        ns = xi.shape[1]
        o = np.zeros((d.out_ch, ns), dtype=np.float32)
        if off:
            d.downmix(xi, o, 0, off)
        if mode > -1:
            d.set_mode_weight(mode, -1)
        if ns > off:
            d.downmix(xi, o, off, ns - off)
        outs.append(o)
    d.close()
    return np.stack(outs)




class Stream:
    """orc_stream: render -> gains -> mix -> loudness -> limiter -> pack for one stream."""

    def __init__(self, mx, out_channels, element_gain=1.0, output_gain=1.0, loudness_on=0,
                 loudness_gain=1.0, limiter_on=1, thr_db=-1.0, rate=48000, bit_depth=16, max_ns=6144,
                 lfe_rate=0):
        self.buf = C.create_string_buffer(lib().orc_sizeof_stream())
        self.ch = out_channels
        self.bd = bit_depth
        self.max_ns = max_ns
        r = lib().orc_stream_open(self.buf, C.byref(mx), out_channels, element_gain, output_gain,
                                  loudness_on, loudness_gain, limiter_on, thr_db, rate, bit_depth, max_ns)
        assert r == 0
        if lfe_rate:   # HOA LFE generator on (the reference built -DDISABLE_LFE_HOA=0), filter at the stream's rate
            lib().orc_stream_enable_lfe(self.buf, lfe_rate)

    def _view(self, raw, n):
        if self.bd == 16:
            return raw.view(np.int16)[:n * self.ch].reshape(n, self.ch).copy()
        if self.bd == 32:
            return raw.view(np.int32)[:n * self.ch].reshape(n, self.ch).copy()
        return raw[:n * self.ch * 3].reshape(n, self.ch, 3).copy()

    def frame(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        ns = x.shape[1]
        raw = np.zeros(ns * self.ch * 4 + 16, dtype=np.uint8)
        n = lib().orc_stream_frame(self.buf, fp(x), ns, raw.ctypes.data_as(C.c_void_p))
        return self._view(raw, n)

    def flush(self):
        raw = np.zeros(240 * self.ch * 4 + 16, dtype=np.uint8)
        n = lib().orc_stream_flush(self.buf, raw.ctypes.data_as(C.c_void_p))
        return self._view(raw, n)

    def close(self):
        lib().orc_stream_close(self.buf)


def stream_run(mx, out_channels, x, frame_size, flush=True, **kw):
    """x: [m][total]; returns interleaved PCM [n_out][out_channels]."""
    s = Stream(mx, out_channels, **kw)
    outs = []
    for p in range(0, x.shape[1], frame_size):
        outs.append(s.frame(x[:, p:p + frame_size]))
    if flush:
        outs.append(s.flush())
    s.close()
    return np.concatenate(outs, axis=0)


def resample_run(x, in_rate, out_rate, sizes, flush=True):
    """x [ch][total] -> (y [ch][n_out], per-call counts), driven like iamf_resample"""
    L = lib()
    ch = x.shape[0]
    r = L.orc_resampler_open(ch, in_rate, out_rate, 4)
    assert r
    outs, rets, pos = [], [], 0
    for ns in sizes:
        xi = np.ascontiguousarray(x[:, pos:pos + ns], dtype=np.float32)
        pos += ns
        cap = L.orc_resample_out_capacity(r, ns)
        o = np.zeros((ch, cap), dtype=np.float32)
        n = L.orc_resample(r, fp(xi), fp(o), ns)
        outs.append(o[:, :n].copy())
        rets.append(n)
    if flush:
        cap = max(L.orc_resample_flush_capacity(r), 1)
        o = np.zeros((ch, cap), dtype=np.float32)
        n = L.orc_resample_flush(r, fp(o))
        outs.append(o[:, :n].copy())
        rets.append(n)
    L.orc_resampler_close(r)
    return np.concatenate(outs, axis=1), rets
