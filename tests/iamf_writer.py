"""Minimal IAMF bitstream writer (LPCM only) for tests: builds the descriptor OBUs and temporal
units the reference decoder accepts (wire format per SURVEY.md Appendix B; written from the
IAMF OBU syntax, the parser being reference src/iamf_dec/IAMF_OBU.c:79-1248).

Only what the rendering-path tests need: ia sequence header, one `ipcm` codec config, channel-
based (single layer) and scene-based (ambisonics mono) audio elements, one mix presentation with
one sub-mix of 1..2 elements, mix-gain / demixing parameter blocks, audio frames.
"""
import struct

import numpy as np

OBU_CODEC_CONFIG, OBU_AUDIO_ELEMENT, OBU_MIX_PRESENTATION, OBU_PARAMETER_BLOCK = 0, 1, 2, 3
OBU_TEMPORAL_DELIMITER, OBU_AUDIO_FRAME, OBU_AUDIO_FRAME_ID0, OBU_SEQUENCE_HEADER = 4, 5, 6, 31

# loudspeaker_layout ids (IAChannelLayoutType) and their substream structure (coupled first)
LAYOUT_SUBSTREAMS = {0: (1, 0), 1: (1, 1), 2: (4, 2), 3: (5, 3), 4: (6, 4), 5: (5, 3), 6: (6, 4),
                     7: (7, 5), 8: (4, 2)}
LAYOUT_CHANNELS = {0: 1, 1: 2, 2: 6, 3: 8, 4: 10, 5: 8, 6: 10, 7: 12, 8: 6}

ANIM_STEP, ANIM_LINEAR, ANIM_BEZIER = 0, 1, 2


def leb128(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def obu(obu_type, payload, trim=None):
    hdr = (obu_type << 3) | (0x02 if trim else 0)
    body = b""
    if trim:
        body += leb128(trim[1]) + leb128(trim[0])  # num_samples_to_trim_at_end, then at_start
    body += payload
    return bytes([hdr]) + leb128(len(body)) + body


def sequence_header(profile=1):
    return obu(OBU_SEQUENCE_HEADER, b"iamf" + bytes([profile, profile]))


def codec_config_lpcm(cid, frame_size, sample_size=16, rate=48000, little_endian=True):
    p = leb128(cid) + b"ipcm" + leb128(frame_size) + struct.pack(">h", 0)
    p += bytes([1 if little_endian else 0, sample_size]) + struct.pack(">I", rate)
    return obu(OBU_CODEC_CONFIG, p)


def param_definition(pid, rate, mode=1, duration=0, constant_interval=0, intervals=None):
    p = leb128(pid) + leb128(rate) + bytes([0x80 if mode else 0x00])
    if not mode:
        p += leb128(duration) + leb128(constant_interval)
        if not constant_interval:
            p += leb128(len(intervals))
            for iv in intervals:
                p += leb128(iv)
    return p


def audio_element_channel(eid, cid, layout, substream_ids, demixing=None):
    """single-layer channel-based element; demixing = dict(pid, rate, frame, mode, w) or None"""
    ns, nc = LAYOUT_SUBSTREAMS[layout]
    assert len(substream_ids) == ns
    p = leb128(eid) + bytes([0 << 5]) + leb128(cid) + leb128(ns)
    for s in substream_ids:
        p += leb128(s)
    if demixing:
        p += leb128(1) + leb128(1)  # one parameter, type DEMIXING
        p += param_definition(demixing["pid"], demixing["rate"], mode=0, duration=demixing["frame"],
                              constant_interval=demixing["frame"])
        p += bytes([(demixing["mode"] & 7) << 5, (demixing["w"] & 15) << 4])
    else:
        p += leb128(0)
    p += bytes([1 << 5])  # num_layers = 1
    p += bytes([(layout << 4)]) + bytes([ns, nc])
    return obu(OBU_AUDIO_ELEMENT, p)


def audio_element_ambisonics_mono(eid, cid, channels, substream_ids):
    assert len(substream_ids) == channels
    p = leb128(eid) + bytes([1 << 5]) + leb128(cid) + leb128(channels)
    for s in substream_ids:
        p += leb128(s)
    p += leb128(0)                      # no parameters
    p += leb128(0)                      # ambisonics_mode = mono
    p += bytes([channels, channels]) + bytes(range(channels))
    return obu(OBU_AUDIO_ELEMENT, p)


def audio_element_ambisonics_projection(eid, cid, channels, substream_ids, coupled, matrix_q15):
    """matrix_q15: int16 [substreams + coupled][channels] (decoded channel major, IAMF_core_decoder.c:116-130)"""
    l_in = len(substream_ids) + coupled
    m = np.asarray(matrix_q15, dtype=np.int16)
    assert m.shape == (l_in, channels)
    p = leb128(eid) + bytes([1 << 5]) + leb128(cid) + leb128(len(substream_ids))
    for s in substream_ids:
        p += leb128(s)
    p += leb128(0)                      # no parameters
    p += leb128(1)                      # ambisonics_mode = projection
    p += bytes([channels, len(substream_ids), coupled])
    p += m.astype(">i2").tobytes()
    return obu(OBU_AUDIO_ELEMENT, p)


def mix_presentation(mid, elements, output_gain, layouts, loudness_q78=0, loudness_infos=None):
    """elements: list of dict(eid, gain_pdef(bytes), default_gain_q78, headphones_mode);
    output_gain: dict(pdef, default_q78); layouts: list of ('ss', n) / ('binaural',);
    loudness_infos: per layout dict(integrated, peak[, true_peak][, anchors=[(element, q78), ...]]) — loudness_info()
    with info_type bit 0 (true peak) / bit 1 (anchored loudness), IAMF_OBU.c:869-913"""
    p = leb128(mid) + leb128(0)         # no labels
    p += leb128(1)                      # one sub-mix
    p += leb128(len(elements))
    for e in elements:
        p += leb128(e["eid"])
        p += bytes([(e.get("headphones_mode", 0) & 3) << 6])
        p += leb128(0)                  # rendering_config_extension_size
        p += e["pdef"] + struct.pack(">h", e.get("default_q78", 0))
    p += output_gain["pdef"] + struct.pack(">h", output_gain.get("default_q78", 0))
    p += leb128(len(layouts))
    for lay in layouts:
        if lay[0] == "ss":
            p += bytes([(2 << 6) | (lay[1] << 2)])
        else:
            p += bytes([3 << 6])
        li = loudness_infos[layouts.index(lay)] if loudness_infos else None
        if li is None:
            p += bytes([0]) + struct.pack(">hh", loudness_q78, 0)  # info_type 0, loudness, digital peak
        else:
            it = (1 if "true_peak" in li else 0) | (2 if "anchors" in li else 0)
            p += bytes([it]) + struct.pack(">hh", li["integrated"], li["peak"])
            if it & 1:
                p += struct.pack(">h", li["true_peak"])
            if it & 2:
                p += bytes([len(li["anchors"])])
                for el, q in li["anchors"]:
                    p += bytes([el]) + struct.pack(">h", q)
    return obu(OBU_MIX_PRESENTATION, p)


def audio_element_scalable(eid, cid, layers, substream_ids, demixing=None, recon=None):
    """scalable channel audio (IAMF_OBU.c:491-530): layers = [dict(layout, nsub, ncoupled,
    out_gain=(6-bit flags, q7.8 dB) or None, recon=bool)]; demixing / recon = dict(pid, rate, frame, ...)"""
    p = leb128(eid) + bytes([0 << 5]) + leb128(cid) + leb128(len(substream_ids))
    for s in substream_ids:
        p += leb128(s)
    p += leb128((1 if demixing else 0) + (1 if recon else 0))
    if demixing:
        p += leb128(1)
        p += param_definition(demixing["pid"], demixing["rate"], mode=0, duration=demixing["frame"],
                              constant_interval=demixing["frame"])
        p += bytes([(demixing["mode"] & 7) << 5, (demixing["w"] & 15) << 4])
    if recon:
        p += leb128(2)
        p += param_definition(recon["pid"], recon["rate"], mode=0, duration=recon["frame"],
                              constant_interval=recon["frame"])
    p += bytes([len(layers) << 5])
    for l in layers:
        og = l.get("out_gain")
        p += bytes([(l["layout"] << 4) | ((1 if og else 0) << 3) | ((1 if l.get("recon") else 0) << 2)])
        p += bytes([l["nsub"], l["ncoupled"]])
        if og:
            p += bytes([(og[0] & 0x3f) << 2]) + struct.pack(">h", og[1])
    return obu(OBU_AUDIO_ELEMENT, p)


def recon_gain_block(pid, per_layer):
    """per_layer: for every layer WITH recon_gain_flag, in layer order, (flags, [gain bytes])"""
    p = leb128(pid)
    for flags, gains in per_layer:
        assert bin(flags).count("1") == len(gains)
        p += leb128(flags) + bytes(gains)
    return obu(OBU_PARAMETER_BLOCK, p)


def mix_gain_block(pid, segments, mode1=None):
    """segments: list of dict(anim, start, end=, control=, rel_time=) in q7.8 dB; when the
    parameter definition has mode 1 the block carries duration/interval itself (mode1 =
    dict(duration, constant_interval | intervals))"""
    p = leb128(pid)
    if mode1:
        p += leb128(mode1["duration"]) + leb128(mode1.get("constant_interval", 0))
        if not mode1.get("constant_interval", 0):
            p += leb128(len(segments))
    for i, s in enumerate(segments):
        if mode1 and not mode1.get("constant_interval", 0):
            p += leb128(mode1["intervals"][i])
        p += leb128(s["anim"]) + struct.pack(">h", s["start"])
        if s["anim"] != ANIM_STEP:
            p += struct.pack(">h", s["end"])
            if s["anim"] == ANIM_BEZIER:
                p += struct.pack(">h", s["control"]) + bytes([s["rel_time"]])
    return obu(OBU_PARAMETER_BLOCK, p)


def demixing_block(pid, mode):
    return obu(OBU_PARAMETER_BLOCK, leb128(pid) + bytes([(mode & 7) << 5]))


LE_DEFAULT = True   # the byte order lpcm_bytes writes when the caller does not say (tests/e2e_cases.py build() sets it per stream)


def lpcm_bytes(x, sample_size=16, little_endian=None):
    """x: [channels_in_substream(1 or 2)][n] float in [-1, 1) -> sample-interleaved bytes"""
    if little_endian is None:
        little_endian = LE_DEFAULT
    inter = np.ascontiguousarray(x.T)
    e = "<" if little_endian else ">"
    if sample_size == 16:
        return np.round(inter * 32768.0).clip(-32768, 32767).astype(e + "i2").tobytes()
    if sample_size == 32:
        return np.round(inter.astype(np.float64) * 2147483648.0).clip(-2 ** 31, 2 ** 31 - 1).astype(e + "i4").tobytes()
    v = np.round(inter.astype(np.float64) * 8388608.0).clip(-2 ** 23, 2 ** 23 - 1).astype("<i4")
    b = v.reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3]
    return (b if little_endian else b[:, ::-1]).tobytes()


def audio_frames(substreams, trim=None):
    """substreams: list of (substream_id, bytes)"""
    out = b""
    for sid, data in substreams:
        if sid <= 17:
            out += obu(OBU_AUDIO_FRAME_ID0 + sid, data, trim)
        else:
            out += obu(OBU_AUDIO_FRAME, leb128(sid) + data, trim)
    return out


def temporal_delimiter():
    return obu(OBU_TEMPORAL_DELIMITER, b"")


def quantize(x, sample_size=16):
    """the float values the decoder will reconstruct from lpcm_bytes(x)"""
    scale = float(1 << (sample_size - 1))
    q = np.round(x.astype(np.float64) * scale).clip(-scale, scale - 1)
    return (q.astype(np.float32) / np.float32(scale)).astype(np.float32)


def channel_element_substreams(layout, x_al, first_id, sample_size=16):
    """x_al: [channels][n] in AUDIO-LAYER order (coupled pairs first); returns [(id, bytes)]"""
    ns, nc = LAYOUT_SUBSTREAMS[layout]
    subs, c = [], 0
    for i in range(ns):
        w = 2 if i < nc else 1
        subs.append((first_id + i, lpcm_bytes(x_al[c:c + w], sample_size)))
        c += w
    return subs
