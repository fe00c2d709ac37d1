"""Facade protocol decisions that need no GPU: what IAMF_decoder_configure refuses, it refuses before any device call.

VERDICT r2 missing #4: with the HOA LFE generator on (the reference built -DDISABLE_LFE_HOA=0) the reference keeps ONE
low-pass filter per output layout (src/iamf_dec/h2m_rdr.c:1151-1239, call site IAMF_decoder.c:2625-2636), so two
scene-based elements of one mix presentation would push their W channels through the same two-sample history in turn.
The batch keeps the generator's state per stream, not per (stream, element): the facade says IAMF_ERR_UNIMPLEMENTED for
that combination at configure time instead of rendering something else."""
import ctypes as C

import iamf_writer as W

IAMF_ERR_UNIMPLEMENTED = -6


def _two_scene_elements_stream(fs=256):
    pd = lambda pid: W.param_definition(pid, 48000, mode=1)
    s = W.sequence_header(1) + W.codec_config_lpcm(0, fs, 16, 48000)
    s += W.audio_element_ambisonics_mono(1, 0, 4, [0, 1, 2, 3])          # first-order, sub-streams 0..3
    s += W.audio_element_ambisonics_mono(2, 0, 4, [4, 5, 6, 7])          # a second scene-based element
    s += W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=0), dict(eid=2, pdef=pd(102), default_q78=0)],
                            dict(pdef=pd(101), default_q78=0), [("ss", 1)])   # Sound System B: a layout with an LFE
    s += W.temporal_delimiter()
    return s


def _lib():
    import iac_amd
    L = C.CDLL(iac_amd.lib_path())
    L.IAMF_decoder_open.restype = C.c_void_p
    L.IAMF_decoder_close.argtypes = [C.c_void_p]
    L.IAMF_decoder_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.IAMF_decoder_output_layout_set_sound_system.argtypes = [C.c_void_p, C.c_int]
    L.IAMF_decoder_set_bit_depth.argtypes = [C.c_void_p, C.c_uint32]
    L.iamf_hip_decoder_set_hoa_lfe.argtypes = [C.c_void_p, C.c_int]
    return L


def test_two_scene_based_elements_with_the_lfe_generator_are_refused_at_configure():
    L = _lib()
    s = _two_scene_elements_stream()
    d = L.IAMF_decoder_open()
    L.IAMF_decoder_set_bit_depth(d, 16)
    L.IAMF_decoder_output_layout_set_sound_system(d, 1)
    assert L.iamf_hip_decoder_set_hoa_lfe(d, 1) == 0
    rs = C.c_uint32(0)
    assert L.IAMF_decoder_configure(d, s, len(s), C.byref(rs)) == IAMF_ERR_UNIMPLEMENTED
    assert L.IAMF_decoder_close(d) == 0


def test_group_entry_points_validate_their_arguments_without_a_gpu():
    L = _lib()
    L.iamf_hip_decoder_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    g = C.c_void_p()
    assert L.iamf_hip_decoder_group_create(None, 4, 0, C.byref(g)) == -1
    d = L.IAMF_decoder_open()
    harr = (C.c_void_p * 1)(d)
    assert L.iamf_hip_decoder_group_create(harr, 0, 0, C.byref(g)) == -1
    assert L.iamf_hip_decoder_group_create(harr, 1, 0, C.byref(g)) == -5     # not configured: IAMF_ERR_INVALID_STATE
    assert L.IAMF_decoder_close(d) == 0
