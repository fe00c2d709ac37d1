"""Drives an IAMF_decoder.h implementation (the real reference or libiamf_hip.so) through ctypes
the way test/tools/iamfplayer/player/iamfplayer.c:380-431,571-600 drives the reference."""
import ctypes as C

import numpy as np


class _Anchor(C.Structure):
    _fields_ = [("anchor_element", C.c_uint8), ("anchored_loudness", C.c_int16)]


class _Loudness(C.Structure):
    _fields_ = [("info_type", C.c_uint8), ("integrated_loudness", C.c_int16), ("digital_peak", C.c_int16),
                ("true_peak", C.c_int16), ("num_anchor_loudness", C.c_uint8), ("anchor_loudness", C.POINTER(_Anchor))]


class _Param(C.Structure):
    _fields_ = [("parameter_length", C.c_int), ("parameter_definition_type", C.c_uint32), ("dmixp_mode", C.c_uint32)]


class _Extradata(C.Structure):   # IAMF_extradata, include/IAMF_decoder.h:222-235
    _fields_ = [("output_sound_system", C.c_int), ("number_of_samples", C.c_uint32), ("bitdepth", C.c_uint32),
                ("sampling_rate", C.c_uint32), ("output_sound_mode", C.c_int), ("num_loudness_layouts", C.c_int),
                ("loudness_layout", C.POINTER(C.c_uint8)), ("loudness", C.POINTER(_Loudness)),
                ("num_parameters", C.c_uint32), ("param", C.POINTER(_Param))]


META_COLUMNS = 64


def last_metadata(ref, d, owns_anchors):
    """IAMF_decoder_get_last_metadata -> one row of int64 (fixed width, -9999 padded): pts, sound system, samples, bit depth,
    rate, sound mode, layouts, then per layout (layout byte: type << 6 | sound system << 2, info type, integrated, peak,
    true peak, anchors, then (element, loudness) per anchor), then parameters, then (length, type, demixing mode) per
    parameter.  Frees what the call hands out the way the reference's contract says (IAMF_decoder.c:3668-3706; the
    reference itself lends its anchor arrays, this library copies them: owns_anchors)."""
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    ref.IAMF_decoder_get_last_metadata.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(_Extradata)]
    pts, m = C.c_int64(-1), _Extradata()
    rc = ref.IAMF_decoder_get_last_metadata(d, C.byref(pts), C.byref(m))
    assert rc == 0, rc
    row = [pts.value, m.output_sound_system, m.number_of_samples, m.bitdepth, m.sampling_rate, m.output_sound_mode,
           m.num_loudness_layouts]
    for i in range(m.num_loudness_layouts):
        lo = m.loudness[i]
        row += [m.loudness_layout[i], lo.info_type, lo.integrated_loudness, lo.digital_peak,
                lo.true_peak if lo.info_type & 1 else 0, lo.num_anchor_loudness]
        for k in range(lo.num_anchor_loudness):
            row += [lo.anchor_loudness[k].anchor_element, lo.anchor_loudness[k].anchored_loudness]
        if owns_anchors and lo.num_anchor_loudness:
            libc.free(C.cast(lo.anchor_loudness, C.c_void_p))
    row.append(m.num_parameters)
    for i in range(m.num_parameters):
        row += [m.param[i].parameter_length, m.param[i].parameter_definition_type, m.param[i].dmixp_mode]
    for ptr in (m.loudness_layout, m.loudness, m.param):
        if ptr:
            libc.free(C.cast(ptr, C.c_void_p))
    assert len(row) <= META_COLUMNS
    return row + [-9999] * (META_COLUMNS - len(row))


def decode_stream(ref, stream_bytes, layout, bit_depth=16, out_rate=0, loudness=0.0, limiter=True,
               threshold=-1.0, pcm_channels=None, metadata=None, pts=(0, 90000), mix_id=None):
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
    # metadata: dict(rows=[], owns_anchors=bool) — a row (last_metadata) after configure and after every decode call
    # that delivered a frame or flushed
    ref.IAMF_decoder_set_pts(d, pts[0], pts[1])
    if mix_id is not None:
        ref.IAMF_decoder_set_mix_presentation_id.argtypes = [C.c_void_p, C.c_uint64]
        assert ref.IAMF_decoder_set_mix_presentation_id(d, mix_id) == 0
    rsize = C.c_uint32(0)
    r = ref.IAMF_decoder_configure(d, stream_bytes, len(stream_bytes), C.byref(rsize))
    assert r == 0, "configure failed: %d" % r
    if metadata is not None:
        metadata["rows"].append(last_metadata(ref, d, metadata["owns_anchors"]))
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
            if metadata is not None:
                metadata["rows"].append(last_metadata(ref, d, metadata["owns_anchors"]))
                if metadata.get("set_pts_after") == len(rets):   # the caller re-bases its clock mid-stream
                    ref.IAMF_decoder_set_pts(d, *metadata["set_pts_to"])
        used += rsize.value
        if not rsize.value:
            break
    rsize.value = 0
    n = ref.IAMF_decoder_decode(d, None, 0, C.byref(rsize), pcm)
    if n > 0:
        chunks.append(pcm.raw[:n * ch * bps])
    rets.append(n)
    if metadata is not None:
        metadata["rows"].append(last_metadata(ref, d, metadata["owns_anchors"]))
        if not out_rate or not metadata.get("strict", True):
            # a second flush: another 240 sample-frames of zeros; behind a resampler (strict=False runs: the fuzz) another
            # output latency's worth of its filter tail in front of them — kept for the caller to compare
            rsize.value = 0
            n = ref.IAMF_decoder_decode(d, None, 0, C.byref(rsize), pcm)
            rets.append(n)
            if n > 0 and metadata.get("strict", True):
                assert not any(pcm.raw[:n * ch * bps]), "a second flush hands out zeros"
            metadata["flush2"] = pcm.raw[:max(n, 0) * ch * bps]
            metadata["rows"].append(last_metadata(ref, d, metadata["owns_anchors"]))
    ref.IAMF_decoder_close(d)
    raw = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    if bit_depth == 16:
        out = raw.view(np.int16).reshape(-1, ch)
    elif bit_depth == 32:
        out = raw.view(np.int32).reshape(-1, ch)
    else:
        out = raw.reshape(-1, ch, 3)
    return out.copy(), rets


def decode_stream_switching(ref, stream_bytes, layouts, switch_after, bit_depth=16, pcm_channels=12, out_rate=0, loudness=0.0,
                            limiter=True, threshold=-1.0):
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
    if out_rate or loudness != 0.0 or not limiter or threshold != -1.0:   # (the fuzz; the stored goldens use the defaults)
        ref.IAMF_decoder_set_normalization_loudness.argtypes = [C.c_void_p, C.c_float]
        ref.IAMF_decoder_peak_limiter_enable.argtypes = [C.c_void_p, C.c_uint32]
        ref.IAMF_decoder_peak_limiter_set_threshold.argtypes = [C.c_void_p, C.c_float]
        ref.IAMF_decoder_set_sampling_rate.argtypes = [C.c_void_p, C.c_uint32]
        if not limiter:
            ref.IAMF_decoder_peak_limiter_enable(d, 0)
        else:
            ref.IAMF_decoder_peak_limiter_set_threshold(d, threshold)
        ref.IAMF_decoder_set_normalization_loudness(d, loudness)
        if out_rate:
            assert ref.IAMF_decoder_set_sampling_rate(d, out_rate) == 0
    set_layout(d, layouts[0])
    ref.IAMF_decoder_set_pts(d, 0, 90000)
    rsize = C.c_uint32(0)
    assert ref.IAMF_decoder_configure(d, stream_bytes, len(stream_bytes), C.byref(rsize)) == 0
    used = rsize.value
    bps = bit_depth // 8
    ch = pcm_channels
    pcm = C.create_string_buffer(bps * 6144 * 6 * 24)
    chunks, rets, frames, li = [], [], 0, 0
    dt = {16: np.int16, 32: np.int32, 24: np.uint8}[bit_depth]
    if bit_depth == 24:
        ch, bps = 3 * pcm_channels, 1   # bytes: [n][channels * 3]
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


def decode_stream_blocks(ref, stream_bytes, layout, block, bit_depth=16, out_rate=0, loudness=0.0, limiter=True, threshold=-1.0,
                         pcm_channels=None, mix_id=None, max_calls=100000):
    """The reference player's own loop (test/tools/iamfplayer/player/iamfplayer.c:529-662, bs_input_wav_output) with a block
    buffer of `block` bytes instead of its 184 320: the file is read block by block, IAMF_decoder_configure is fed until it
    stops answering IAMF_ERR_BUFFER_TOO_SMALL, IAMF_decoder_decode is called while it consumes something, what is left of
    a block moves to the front of the next, the end of the file flushes.  Returns (pcm, events): events = every call's
    ('c' | 'd', return value, rsize) — the incremental-parsing protocol itself is compared, not only the samples."""
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
    ref.IAMF_decoder_set_mix_presentation_id.argtypes = [C.c_void_p, C.c_uint64]
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
    bps = bit_depth // 8
    pcm = C.create_string_buffer(bps * 6144 * 6 * max(ch, ch_layout))
    events, chunks = [], []
    left, pos, end, state, calls = b"", 0, False, 0, 0
    rsize = C.c_uint32(0)
    while calls < max_calls:
        if len(left) != block:
            got = stream_bytes[pos:pos + block - len(left)]
            pos += len(got)
            if not got:
                end = True
        else:
            got = b""
        data = left + got
        size, used, ret = len(data), 0, 0
        if state <= 0:
            if end:
                break
            rsize.value = 0
            if state == 0:
                ref.IAMF_decoder_set_pts(d, 0, 90000)
            if mix_id is not None:
                ref.IAMF_decoder_set_mix_presentation_id(d, mix_id)
            ret = ref.IAMF_decoder_configure(d, data, size, C.byref(rsize))
            calls += 1
            events.append(("c", ret, rsize.value))
            if ret == 0:
                state = 1
            elif ret != -2 or not rsize.value:
                break
            used += rsize.value
        if state > 0:
            while calls < max_calls:
                rsize.value = 0
                if not end:
                    rest = data[used:]
                    ret = ref.IAMF_decoder_decode(d, rest, len(rest), C.byref(rsize), pcm)
                else:
                    ret = ref.IAMF_decoder_decode(d, None, 0, C.byref(rsize), pcm)
                calls += 1
                events.append(("d", ret, rsize.value))
                if ret > 0:
                    chunks.append(pcm.raw[:ret * ch * bps])
                if end:
                    break
                used += rsize.value
                if ret == -5:
                    state = ret
                if ret < 0 or used >= size or not rsize.value:
                    break
        if end:
            break
        if len(data) - used == block and not got:   # a full block nothing is consumed from: the player would spin for ever
            events.append(("stuck", 0, 0))
            break
        left = data[used:]
    ref.IAMF_decoder_close(d)
    raw = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    if bit_depth == 16:
        out = raw.view(np.int16).reshape(-1, ch)
    elif bit_depth == 32:
        out = raw.view(np.int32).reshape(-1, ch)
    else:
        out = raw.reshape(-1, ch, 3)
    return out.copy(), events


def decode_stream_units(ref, descriptors, units, layout, bit_depth=16, out_rate=0, loudness=0.0, limiter=True, threshold=-1.0,
                        pcm_channels=None, mix_id=None):
    """The reference player's OTHER loop (iamfplayer.c:664-789, mp4_input_wav_output2: what a demuxer does): the descriptors
    in one IAMF_decoder_configure call with rsize == NULL, then ONE temporal unit per IAMF_decoder_decode call with
    rsize == NULL (include/IAMF_decoder.h:91-95), a flush at the end.  Returns (pcm, [configure's value, every decode's])."""
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
    ref.IAMF_decoder_set_mix_presentation_id.argtypes = [C.c_void_p, C.c_uint64]
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
    if mix_id is not None:
        ref.IAMF_decoder_set_mix_presentation_id(d, mix_id)
    bps = bit_depth // 8
    pcm = C.create_string_buffer(bps * 6144 * 6 * max(ch, ch_layout))
    rets, chunks = [ref.IAMF_decoder_configure(d, descriptors, len(descriptors), None)], []
    if rets[0] == 0:
        for u in units:
            n = ref.IAMF_decoder_decode(d, u, len(u), None, pcm)
            rets.append(n)
            if n > 0:
                chunks.append(pcm.raw[:n * ch * bps])
        n = ref.IAMF_decoder_decode(d, None, 0, None, pcm)
        rets.append(n)
        if n > 0:
            chunks.append(pcm.raw[:n * ch * bps])
    ref.IAMF_decoder_close(d)
    raw = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    if bit_depth == 16:
        out = raw.view(np.int16).reshape(-1, ch)
    elif bit_depth == 32:
        out = raw.view(np.int32).reshape(-1, ch)
    else:
        out = raw.reshape(-1, ch, 3)
    return out.copy(), rets
