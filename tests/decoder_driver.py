"""Drives an IAMF_decoder.h implementation (the real reference or libiamf_hip.so) through ctypes
the way test/tools/iamfplayer/player/iamfplayer.c:380-431,571-600 drives the reference."""
import ctypes as C

import numpy as np


def decode_stream(ref, stream_bytes, layout, bit_depth=16, out_rate=0, loudness=0.0, limiter=True,
               threshold=-1.0, pcm_channels=None):
    # pcm_channels: channel stride of the PCM the decoder writes (a -DSAMSUNG_TV build: always 12)
    """layout: ('ss', IAMF_SoundSystem enum value) or ('binaural',). Returns (pcm ndarray
    [n][ch] (24-bit: [n][ch][3] bytes), list of per-call return values)."""
    ref.IAMF_decoder_open.restype = C.c_void_p
    ref.IAMF_decoder_close.argtypes = [C.c_void_p]
    ref.IAMF_decoder_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    ref.IAMF_decoder_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_uint32), C.c_void_p]
    ref.IAMF_decoder_output_layout_set_sound_system.argtypes = [C.c_void_p, C.c_int]
    ref.IAMF_decoder_output_layout_set_binaural.argtypes = [C.c_void_p]
    ref.IAMF_decoder_set_normalization_loudness.argtypes = [C.c_void_p, C.c_float]
    ref.IAMF_decoder_set_bit_depth.argtypes = [C.c_void_p, C.c_uint32]
    ref.IAMF_decoder_peak_limiter_enable.argtypes = [C.c_void_p, C.c_uint32]
    ref.IAMF_decoder_peak_limiter_set_threshold.argtypes = [C.c_void_p, C.c_float]
    ref.IAMF_decoder_set_sampling_rate.argtypes = [C.c_void_p, C.c_uint32]
    ref.IAMF_decoder_set_pts.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
    ref.IAMF_layout_sound_system_channels_count.argtypes = [C.c_int]

    d = ref.IAMF_decoder_open()
    if not limiter:
        ref.IAMF_decoder_peak_limiter_enable(d, 0)
    else:
        ref.IAMF_decoder_peak_limiter_set_threshold(d, threshold)
    ref.IAMF_decoder_set_normalization_loudness(d, loudness)
    ref.IAMF_decoder_set_bit_depth(d, bit_depth)
    if out_rate:
        assert ref.IAMF_decoder_set_sampling_rate(d, out_rate) == 0
    if layout[0] == "ss":
        ref.IAMF_decoder_output_layout_set_sound_system(d, layout[1])
        ch = ref.IAMF_layout_sound_system_channels_count(layout[1])
    else:
        ref.IAMF_decoder_output_layout_set_binaural(d)
        ch = 2
    ch_layout = ch
    if pcm_channels:
        ch = pcm_channels
    ref.IAMF_decoder_set_pts(d, 0, 90000)
    rsize = C.c_uint32(0)
    r = ref.IAMF_decoder_configure(d, stream_bytes, len(stream_bytes), C.byref(rsize))
    assert r == 0, "configure failed: %d" % r
    used = rsize.value
    bps = bit_depth // 8
    pcm = C.create_string_buffer(bps * 6144 * 6 * max(ch, ch_layout))
    chunks, rets = [], []
    while used < len(stream_bytes):
        rsize.value = 0
        rest = stream_bytes[used:]
        n = ref.IAMF_decoder_decode(d, rest, len(rest), C.byref(rsize), pcm)
        if n == -5:   # IAMF_ERR_INVALID_STATE: a new IA sequence starts here -> configure again (iamfplayer.c:569-588,622-625)
            used += rsize.value
            rest = stream_bytes[used:]
            rsize.value = 0
            r = ref.IAMF_decoder_configure(d, rest, len(rest), C.byref(rsize))
            assert r == 0, "reconfigure failed: %d" % r
            used += rsize.value
            rets.append(-5)
            continue
        assert n >= 0, "decode failed: %d" % n
        if n > 0:
            chunks.append(pcm.raw[:n * ch * bps])
            rets.append(n)
        used += rsize.value
        if not rsize.value:
            break
    rsize.value = 0
    n = ref.IAMF_decoder_decode(d, None, 0, C.byref(rsize), pcm)
    if n > 0:
        chunks.append(pcm.raw[:n * ch * bps])
    rets.append(n)
    ref.IAMF_decoder_close(d)
    raw = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    if bit_depth == 16:
        out = raw.view(np.int16).reshape(-1, ch)
    elif bit_depth == 32:
        out = raw.view(np.int32).reshape(-1, ch)
    else:
        out = raw.reshape(-1, ch, 3)
    return out.copy(), rets


def decode_stream_switching(ref, stream_bytes, layouts, switch_after, bit_depth=16, pcm_channels=12):
    """As decode_stream, but the output layout is changed while decoding: after `switch_after[i]` decode calls that
    returned a frame, IAMF_decoder_output_layout_set_* (layouts[i + 1]) + IAMF_decoder_configure(h, NULL, 0, NULL) — the
    run-time switch of the reference's -DSAMSUNG_TV build (IAMF_decoder.c:3819-3881).  Returns (list of per-call PCM
    arrays [n][pcm_channels], rets incl. the configure results as ('cfg', rc))."""
    ref.IAMF_decoder_open.restype = C.c_void_p
    ref.IAMF_decoder_close.argtypes = [C.c_void_p]
    ref.IAMF_decoder_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    ref.IAMF_decoder_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_uint32), C.c_void_p]
    ref.IAMF_decoder_output_layout_set_sound_system.argtypes = [C.c_void_p, C.c_int]
    ref.IAMF_decoder_output_layout_set_binaural.argtypes = [C.c_void_p]
    ref.IAMF_decoder_set_bit_depth.argtypes = [C.c_void_p, C.c_uint32]
    ref.IAMF_decoder_set_pts.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]

    def set_layout(d, layout):
        if layout[0] == "ss":
            return ref.IAMF_decoder_output_layout_set_sound_system(d, layout[1])
        return ref.IAMF_decoder_output_layout_set_binaural(d)

    d = ref.IAMF_decoder_open()
    ref.IAMF_decoder_set_bit_depth(d, bit_depth)
    set_layout(d, layouts[0])
    ref.IAMF_decoder_set_pts(d, 0, 90000)
    rsize = C.c_uint32(0)
    assert ref.IAMF_decoder_configure(d, stream_bytes, len(stream_bytes), C.byref(rsize)) == 0
    used = rsize.value
    bps = bit_depth // 8
    ch = pcm_channels
    pcm = C.create_string_buffer(bps * 6144 * 6 * 24)
    chunks, rets, frames, li = [], [], 0, 0
    dt = {16: np.int16, 32: np.int32}[bit_depth]
    while used < len(stream_bytes):
        rsize.value = 0
        rest = stream_bytes[used:]
        n = ref.IAMF_decoder_decode(d, rest, len(rest), C.byref(rsize), pcm)
        assert n >= 0, n
        if n > 0:
            chunks.append(np.frombuffer(pcm.raw[:n * ch * bps], dtype=dt).reshape(n, ch).copy())
            rets.append(n)
            frames += 1
            if li < len(switch_after) and frames == switch_after[li]:
                li += 1
                set_layout(d, layouts[li])
                rets.append(("cfg", ref.IAMF_decoder_configure(d, None, 0, None)))
        used += rsize.value
        if not rsize.value:
            break
    n = ref.IAMF_decoder_decode(d, None, 0, C.byref(rsize), pcm)
    if n > 0:
        chunks.append(np.frombuffer(pcm.raw[:n * ch * bps], dtype=dt).reshape(n, ch).copy())
    rets.append(n)
    ref.IAMF_decoder_close(d)
    return chunks, rets
