"""ctypes binding of include/iamf_hip.h (libiamf_hip.so).  No arithmetic lives here.

If the shared library is missing or cannot be loaded this module raises: there is no CPU path.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iac_amd", "csrc")

FMT_S16, FMT_S24, FMT_S32, FMT_F32 = 16, 24, 32, -32
KIND_H2M, KIND_M2M, KIND_DMX, KIND_FIR = 0, 1, 2, 3
PROJ_AUTO, PROJ_EXACT, PROJ_MFMA = 0, 1, 2
SS = dict(A=0x020, B=0x050, C=0x250, D=0x450, E=0x451, F=0x370, G=0x490, H=0x9A3, I=0x070,
          J=0x470, STEREO=0x200, L51=0x510, L512=0x512, L514=0x514, L71=0x710, L714=0x714,
          MONO=0x100, L712=0x712, L312=0x312, BINAURAL=0x1020)

FP = C.POINTER(C.c_float)


class IamfHipError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with IAMF_HIP error %d" % (what, code))
        self.code = code


class Matrix(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_id", C.c_int32), ("out_id", C.c_int32),
                ("channels", C.c_int32), ("lfe1", C.c_int32), ("lfe2", C.c_int32),
                ("m", C.c_int32), ("n", C.c_int32), ("mat", FP)]


class BatchConfig(C.Structure):
    _fields_ = [("n_streams", C.c_int32), ("frame_size", C.c_int32), ("sample_rate", C.c_int32),
                ("out_channels", C.c_int32), ("out_format", C.c_int32), ("matrix", Matrix),
                ("limiter_enable", C.c_int32), ("limiter_threshold_db", C.c_float),
                ("loudness_enable", C.c_int32), ("projection", C.c_int32), ("fir_taps", C.c_int32), ("lfe_hoa", C.c_int32),
                ("pcm_stride_channels", C.c_int32), ("out_gain_channels", C.c_int32), ("reserved", C.c_int32 * 2)]


class DmxState(C.Structure):
    _fields_ = [("mode", C.c_int32), ("w_idx", C.c_int32), ("w_idx_offset", C.c_int32),
                ("reserved", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float),
                ("delta", C.c_float), ("gamma_w", C.c_float)]


class DmxFrame(C.Structure):
    _fields_ = [("offset", C.c_int32), ("prev", C.c_float * 5), ("cur", C.c_float * 5)]


class RenderArgs(C.Structure):
    _fields_ = [("d_in", C.c_void_p), ("in_stream_stride", C.c_int64), ("in_frame_stride", C.c_int64),
                ("d_in2", C.c_void_p), ("in2_stream_stride", C.c_int64), ("in2_frame_stride", C.c_int64),
                ("d_element_ramp", C.c_void_p), ("d_element2_ramp", C.c_void_p),
                ("d_output_ramp", C.c_void_p), ("ramp_stream_stride", C.c_int64),
                ("d_dmx_frames", C.c_void_p), ("n_frames", C.c_int32), ("n_samples", C.c_int32),
                ("d_pcm", C.c_void_p), ("pcm_stream_stride_bytes", C.c_int64), ("stream", C.c_void_p),
                ("d_demix_frames", C.c_void_p), ("demix_sample0", C.c_int32), ("lfe_pre_samples", C.c_int32),
                ("lfe_post_samples", C.c_int32), ("reserved0", C.c_int32)]


class LpcmLayout(C.Structure):   # iamf_hip_lpcm_layout
    _fields_ = [("sample_bytes", C.c_int32), ("little_endian", C.c_int32), ("channels", C.c_int32), ("frame_size", C.c_int32),
                ("src_offset", C.c_int32 * 32), ("src_step", C.c_int32 * 32)]


class LpcmInput(C.Structure):   # iamf_hip_lpcm_input
    _fields_ = [("d_raw", C.c_void_p), ("raw_stream_stride", C.c_int64), ("raw_frame_stride", C.c_int64),
                ("first_sample", C.c_int32), ("layout", LpcmLayout)]


class DemixConfig(C.Structure):
    _fields_ = [("layout", C.c_int32), ("n_in", C.c_int32), ("chs_in", C.c_int32 * 12), ("n_gain", C.c_int32),
                ("gain_ch", C.c_int32 * 12), ("gain", C.c_float * 12), ("frame_offset", C.c_uint32)]


class DemixState(C.Structure):
    _fields_ = [("mode", C.c_int32), ("last_mode", C.c_int32), ("w_idx", C.c_int32), ("last_w_idx", C.c_int32),
                ("last_sfavg", C.c_float * 24)]


class DemixFrame(C.Structure):
    _fields_ = [("prev", C.c_float * 5), ("cur", C.c_float * 5), ("n_recon", C.c_int32),
                ("recon_ch", C.c_int32 * 12), ("recon_prev", C.c_float * 12), ("recon_cur", C.c_float * 12)]


def lib_path():
    # IAMF_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one box)
    return os.environ.get("IAMF_HIP_LIB") or os.path.join(ROOT, "iac_amd", "lib", "libiamf_hip.so")


def build(force=False):
    """Compile every HIP/C source for gfx950 into iac_amd/lib/libiamf_hip.so (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-j4", "-C", CSRC]
    if force:
        subprocess.check_call(args + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return lib_path()


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("libiamf_hip.so is not built (run __graft_entry__.build()); "
                              "iac_amd has no CPU fallback")
        L = C.CDLL(path)
        L.iamf_hip_get_h2m_matrix.argtypes = [C.c_int, C.c_int, C.POINTER(Matrix)]
        L.iamf_hip_get_m2m_matrix.argtypes = [C.c_int, C.c_int, C.POINTER(Matrix)]
        L.iamf_hip_probe_traffic.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_void_p]
        L.iamf_hip_pick_buffer_pair.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int64,
                                                C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_void_p,
                                                C.POINTER(C.c_int), C.POINTER(C.c_int), FP]
        L.iamf_hip_get_m2m_matrix_variant.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Matrix)]
        L.iamf_hip_layout_channels.argtypes = [C.c_int]
        L.iamf_hip_batch_create.argtypes = [C.POINTER(BatchConfig), C.POINTER(C.c_void_p)]
        L.iamf_hip_batch_destroy.argtypes = [C.c_void_p]
        L.iamf_hip_batch_destroy.restype = None
        L.iamf_hip_batch_set_gains.argtypes = [C.c_void_p, FP, FP, FP]
        L.iamf_hip_batch_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                            C.c_void_p, C.c_int64, C.c_void_p]
        L.iamf_hip_batch_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.iamf_hip_batch_reset.argtypes = [C.c_void_p]
        L.iamf_hip_batch_render_range.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_int32, C.c_int32]
        L.iamf_hip_batch_flush_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32]
        L.iamf_hip_format_bytes.argtypes = [C.c_int]
        L.iamf_hip_version.restype = C.c_char_p
        L.iamf_hip_batch_render_ex.argtypes = [C.c_void_p, C.POINTER(RenderArgs)]
        L.iamf_hip_batch_render_lpcm.argtypes = [C.c_void_p, C.POINTER(LpcmInput), C.POINTER(RenderArgs)]
        L.iamf_hip_batch_set_second_element.argtypes = [C.c_void_p, C.POINTER(Matrix), FP]
        L.iamf_hip_resampler_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.iamf_hip_resampler_destroy.argtypes = [C.c_void_p]
        L.iamf_hip_resampler_destroy.restype = None
        L.iamf_hip_resampler_out_capacity.argtypes = [C.c_void_p, C.c_int]
        L.iamf_hip_resampler_flush_capacity.argtypes = [C.c_void_p]
        L.iamf_hip_resampler_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                 C.c_void_p]
        L.iamf_hip_resampler_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.iamf_hip_resampler_process_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                       C.c_void_p, C.c_int32, C.c_int32]
        L.iamf_hip_resampler_flush_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32]
        L.iamf_hip_resampler_same_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.iamf_hip_stream_signal.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.iamf_hip_deinterleave_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                                C.c_int64, C.c_void_p]
        L.iamf_hip_batch_set_projection.argtypes = [C.c_void_p, FP, C.c_int]
        L.iamf_hip_batch_share_lfe_state.argtypes = [C.c_void_p, C.c_void_p]
        L.iamf_hip_batch_lfe_advance.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
        L.iamf_hip_lpcm_unpack.argtypes = [C.POINTER(LpcmLayout), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int64, C.c_int32, C.c_void_p]
        L.iamf_hip_batch_set_demixer.argtypes = [C.c_void_p, C.POINTER(DemixConfig)]
        L.iamf_hip_demix_state_init.argtypes = [C.POINTER(DemixState)]
        L.iamf_hip_demix_state_init.restype = None
        L.iamf_hip_demix_set_info.argtypes = [C.POINTER(DemixState), C.c_int, C.c_int]
        L.iamf_hip_demix_frame_fill.argtypes = [C.POINTER(DemixState), C.c_int, C.POINTER(C.c_int32), FP,
                                                C.POINTER(DemixFrame)]
        L.iamf_hip_demix_frame_fill.restype = None
        L.iamf_hip_dmx_valid.argtypes = [C.c_int, C.c_int]
        L.iamf_hip_dmx_layout_channels.argtypes = [C.c_int]
        L.iamf_hip_dmx_state_init.argtypes = [C.POINTER(DmxState)]
        L.iamf_hip_dmx_state_init.restype = None
        L.iamf_hip_dmx_set_mode_weight.argtypes = [C.POINTER(DmxState), C.c_int, C.c_int]
        L.iamf_hip_dmx_coefficients.argtypes = [C.POINTER(DmxState), FP]
        L.iamf_hip_dmx_coefficients.restype = None
        L.iamf_hip_shard_split.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.iamf_hip_shard_create.argtypes = [C.POINTER(BatchConfig), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.iamf_hip_shard_destroy.argtypes = [C.c_void_p]
        L.iamf_hip_shard_destroy.restype = None
        L.iamf_hip_shard_devices.argtypes = [C.c_void_p]
        L.iamf_hip_shard_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.iamf_hip_shard_batch.argtypes = [C.c_void_p, C.c_int]
        L.iamf_hip_shard_batch.restype = C.c_void_p
        L.iamf_hip_shard_render_stream.argtypes = [C.c_void_p, C.c_int]
        L.iamf_hip_shard_render_stream.restype = C.c_void_p
        L.iamf_hip_shard_render.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int32,
                                            C.POINTER(C.c_void_p), C.c_int64]
        L.iamf_hip_shard_flush.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64]
        L.iamf_hip_shard_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_int64]
        L.iamf_hip_shard_gather_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_int64, C.c_int64]
        L.iamf_hip_shard_times.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.iamf_hip_shard_sync.argtypes = [C.c_void_p]
        L.iamf_hip_shard_rccl_version.restype = C.c_char_p
        _lib = L
    return _lib


def get_h2m_matrix(order, out_id):
    m = Matrix()
    if lib().iamf_hip_get_h2m_matrix(order, out_id, C.byref(m)) != 0:
        raise KeyError((order, hex(out_id)))
    return m


def get_m2m_matrix(in_id, out_id, variant=0):
    """variant 1 = the tables of the reference's -DSAMSUNG_TV build"""
    m = Matrix()
    if lib().iamf_hip_get_m2m_matrix_variant(variant, in_id, out_id, C.byref(m)) != 0:
        raise KeyError((hex(in_id), hex(out_id)))
    return m


def dmx_matrix(in_layout, out_layout):
    """config 'matrix' selecting the parametric down-mixer between two IAChannelLayoutType ids"""
    m = Matrix()
    m.kind, m.in_id, m.out_id = KIND_DMX, in_layout, out_layout
    m.channels = m.n = lib().iamf_hip_dmx_layout_channels(out_layout)
    m.m = lib().iamf_hip_dmx_layout_channels(in_layout)
    m.lfe1 = m.lfe2 = -1
    return m


def fir_matrix(hrir):
    """config 'matrix' for the binaural HRTF renderer; hrir: float32 ndarray [2][channels][taps]"""
    import numpy as np
    h = np.ascontiguousarray(hrir, dtype=np.float32)
    m = Matrix()
    m.kind, m.in_id, m.out_id, m.channels, m.lfe1, m.lfe2 = KIND_FIR, 0, SS["BINAURAL"], 2, -1, -1
    m.m, m.n = h.shape[1], 2
    m.mat = h.ctypes.data_as(FP)
    m._keep = h
    return m


def lpcm_unpack(layout, d_raw, raw_stream_stride, d_first_count, d_out, out_stream_stride, n_streams, stream=None,
                first_count_stride=2):
    """iamf_hip_lpcm_unpack on device pointers (ints); raises IamfHipError on a negative return"""
    r = lib().iamf_hip_lpcm_unpack(C.byref(layout), d_raw, raw_stream_stride, d_first_count, first_count_stride, d_out,
                                   out_stream_stride, n_streams, stream)
    if r < 0:
        raise IamfHipError(r, "iamf_hip_lpcm_unpack")
    return r


def layout_channels(out_id):
    return lib().iamf_hip_layout_channels(out_id)


def _fparr(a):
    if a is None:
        return None
    arr = (C.c_float * len(a))(*[float(v) for v in a])
    return arr


class Batch:
    """Thin handle on iamf_hip_batch_*; pointers are raw device addresses (ints)."""

    def __init__(self, n_streams, matrix, out_channels, frame_size=1024, sample_rate=48000,
                 out_format=FMT_S16, limiter=True, threshold_db=-1.0, loudness=False, projection=PROJ_AUTO, fir_taps=0,
                 lfe_hoa=False, pcm_stride_channels=0):
        cfg = BatchConfig()
        cfg.n_streams = n_streams
        cfg.frame_size = frame_size
        cfg.sample_rate = sample_rate
        cfg.out_channels = out_channels
        cfg.out_format = out_format
        cfg.matrix = matrix
        cfg.limiter_enable = 1 if limiter else 0
        cfg.limiter_threshold_db = threshold_db
        cfg.loudness_enable = 1 if loudness else 0
        cfg.projection = projection
        cfg.fir_taps = fir_taps
        cfg.lfe_hoa = 1 if lfe_hoa else 0
        cfg.pcm_stride_channels = pcm_stride_channels
        self.cfg = cfg
        self.bytes_per_sample = lib().iamf_hip_format_bytes(out_format)
        h = C.c_void_p()
        r = lib().iamf_hip_batch_create(C.byref(cfg), C.byref(h))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_create")
        self.h = h

    def set_gains(self, element=None, output=None, loudness=None):
        r = lib().iamf_hip_batch_set_gains(self.h, _fparr(element), _fparr(output), _fparr(loudness))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_set_gains")

    def render(self, d_in, in_stream_stride, in_frame_stride, n_frames, d_pcm, pcm_stream_stride_bytes,
               stream=None):
        r = lib().iamf_hip_batch_render(self.h, d_in, in_stream_stride, in_frame_stride, n_frames,
                                        d_pcm, pcm_stream_stride_bytes, stream)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_render")
        return r

    def render_ex(self, args):
        r = lib().iamf_hip_batch_render_ex(self.h, C.byref(args))
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_render_ex")
        return r

    def render_lpcm(self, lpcm_input, args):
        """iamf_hip_batch_render_lpcm: element 0 as LPCM packets (LpcmInput), `args` a RenderArgs with d_in = None"""
        r = lib().iamf_hip_batch_render_lpcm(self.h, C.byref(lpcm_input), C.byref(args))
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_render_lpcm")
        return r

    def render_range(self, args, stream0, n_streams):
        r = lib().iamf_hip_batch_render_range(self.h, C.byref(args), stream0, n_streams)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_render_range")
        return r

    def flush_range(self, d_pcm, pcm_stream_stride_bytes, stream, stream0, n_streams):
        r = lib().iamf_hip_batch_flush_range(self.h, d_pcm, pcm_stream_stride_bytes, stream, stream0, n_streams)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_flush_range")
        return r

    def set_second_element(self, matrix, gains=None):
        r = lib().iamf_hip_batch_set_second_element(self.h, C.byref(matrix), _fparr(gains))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_set_second_element")

    def set_projection(self, matrix):
        """matrix: float32 [decoded channels][ambisonics channels] (IAMF_core_decoder.c:116-130)"""
        rows = [list(r) for r in matrix]
        r = lib().iamf_hip_batch_set_projection(self.h, _fparr([v for row in rows for v in row]), len(rows))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_set_projection")

    def set_demixer(self, layout, chs_in, gains=(), frame_offset=0):
        """gains: [(IAChannel, linear gain)] (reference demixer.c)"""
        c = DemixConfig()
        c.layout, c.n_in, c.n_gain, c.frame_offset = layout, len(chs_in), len(gains), frame_offset
        for i, ch in enumerate(chs_in):
            c.chs_in[i] = ch
        for i, (ch, g) in enumerate(gains):
            c.gain_ch[i], c.gain[i] = ch, g
        r = lib().iamf_hip_batch_set_demixer(self.h, C.byref(c))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_set_demixer")

    def flush(self, d_pcm, pcm_stream_stride_bytes, stream=None):
        r = lib().iamf_hip_batch_flush(self.h, d_pcm, pcm_stream_stride_bytes, stream)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_batch_flush")
        return r

    def reset(self):
        r = lib().iamf_hip_batch_reset(self.h)
        if r != 0:
            raise IamfHipError(r, "iamf_hip_batch_reset")

    def close(self):
        if self.h:
            lib().iamf_hip_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Resampler:
    """Thin handle on iamf_hip_resampler_*; buffers are raw device addresses (interleaved f32)."""

    def __init__(self, n_streams, channels, in_rate, out_rate):
        h = C.c_void_p()
        r = lib().iamf_hip_resampler_create(n_streams, channels, in_rate, out_rate, C.byref(h))
        if r != 0:
            raise IamfHipError(r, "iamf_hip_resampler_create")
        self.h = h

    def out_capacity(self, ns):
        return lib().iamf_hip_resampler_out_capacity(self.h, ns)

    def flush_capacity(self):
        return lib().iamf_hip_resampler_flush_capacity(self.h)

    def process(self, d_in, in_stride, ns, d_out, out_stride, stream=None):
        r = lib().iamf_hip_resampler_process(self.h, d_in, in_stride, ns, d_out, out_stride, stream)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_resampler_process")
        return r

    def flush(self, d_out, out_stride, stream=None):
        r = lib().iamf_hip_resampler_flush(self.h, d_out, out_stride, stream)
        if r < 0:
            raise IamfHipError(r, "iamf_hip_resampler_flush")
        return r

    def process_range(self, d_in, in_stride, ns, d_out, out_stride, s0, cnt, stream=None):
        """streams [s0, s0 + cnt) only (they must be in one state); returns the library's value, errors included"""
        return lib().iamf_hip_resampler_process_range(self.h, d_in, in_stride, ns, d_out, out_stride, stream, s0, cnt)

    def flush_range(self, d_out, out_stride, s0, cnt, stream=None):
        return lib().iamf_hip_resampler_flush_range(self.h, d_out, out_stride, stream, s0, cnt)

    def same_state(self, a, b):
        return bool(lib().iamf_hip_resampler_same_state(self.h, a, b))

    def close(self):
        if self.h:
            lib().iamf_hip_resampler_destroy(self.h)
            self.h = None


def pick_buffer_pair(n_streams, chunks, rows, pieces, in_ptrs, in_stream_stride_bytes, out_ptrs, out_stream_stride_bytes,
                     stream=None):
    """iamf_hip_pick_buffer_pair: (best_in, best_out, ms[n_in][n_out]) over candidate device pointers (ints)"""
    import numpy as np
    n_in, n_out = len(in_ptrs), len(out_ptrs)
    ins = (C.c_void_p * n_in)(*in_ptrs)
    outs = (C.c_void_p * n_out)(*out_ptrs)
    bi, bo = C.c_int(-1), C.c_int(-1)
    ms = np.zeros((n_in, n_out), dtype=np.float32)
    r = lib().iamf_hip_pick_buffer_pair(n_streams, chunks, rows, pieces, ins, n_in, in_stream_stride_bytes, outs, n_out,
                                        out_stream_stride_bytes, C.c_void_p(stream or 0), C.byref(bi), C.byref(bo),
                                        ms.ctypes.data_as(FP))
    if r != 0:
        raise RuntimeError("iamf_hip_pick_buffer_pair: %d" % r)
    return bi.value, bo.value, ms

