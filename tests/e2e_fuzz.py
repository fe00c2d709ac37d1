"""Seeded random IAMF streams for the decoder facade: one or two elements of random kinds, a random output layout, bit depth,
sample format, frame size, gains, ramps, trims, rates.  TEST INFRASTRUCTURE.

oracle/gen_golden_fuzz.py decodes every stream with the REAL reference (oracle/_ref/libiamf_ref.so) and stores, per seed, the
per-call return values and a SHA-256 of the PCM in tests/golden/fuzz.json — a few dozen bytes per stream, so the set can be
wide; tests/test_gpu_fuzz_facade.py decodes the same streams through libiamf_hip.so and compares both.  (The 80 + 20 + 25
streams whose PCM is stored sample by sample are in e2e.npz / lfe.npz / tv.npz: a hash says THAT something differs, they
say where.)"""
import hashlib

import numpy as np

import e2e_cases as E

KINDS = ["stereo", "l51", "l512", "l514", "l71", "l712", "l714", "l312", "mono", "l714dmx", "scalable", "toa_projection",
         "zoa", "foa", "soa", "toa"]
MODES = [0, 1, 2, 4, 5, 6]
RATES = [(44100, 48000), (48000, 44100), (32000, 48000), (16000, 48000), (96000, 48000)]
N_SEEDS = 240
# the reference's two other builds: -DDISABLE_LFE_HOA=0 (oracle/_ref_lfe; the facade after iamf_hip_decoder_set_hoa_lfe) and
# -DSAMSUNG_TV (oracle/_ref_tv; iamf_hip_decoder_set_variant: other layout -> layout tables, a 12-channel PCM stride)
# "wide": the default build again with the dimensions the first generator held fixed — scalable layer stacks, any layout
# with demixing info, the demixing defaults, big-endian samples, more stream / output rates (with and without conversion),
# frames trimmed away completely, frame sizes that are not multiples of 4
# "multi": three elements in the stream, two or three mix presentations of one or two of them each (own gains, one to three
# layouts with their own loudness), the caller naming one (IAMF_decoder_set_mix_presentation_id), a wrong one, or none — the
# reference then takes the FIRST presentation with the best layout score (IAMF_decoder.c:2997-3111) and the loudness of the
# best-scoring layout; sub-streams and parameter blocks of elements outside the chosen presentation are skipped
VARIANTS = dict(default=(0, N_SEEDS), lfe=(100000, 120), tv=(200000, 120), wide=(300000, 240), multi=(400000, 160),
                params=(500000, 200), concat=(600000, 120), syntax=(700000, 200), dparams=(800000, 160))
STACKS = [[1, 3, 7], [0, 1, 2, 5], [1, 8], [2, 4], [3, 4], [8, 3, 6], [1, 2], [1, 2, 3, 4], [2, 3], [1, 5], [2, 5, 6, 7], [0, 1],
          [1, 8, 3, 7], [2, 7], [0, 1, 8, 3, 4], [1, 2, 5, 6], [5, 7], [8, 6], [1, 7], [0, 2]]
WIDE_RATES = [(44100, 44100), (32000, 32000), (16000, 16000), (48000, 16000), (48000, 32000), (48000, 24000), (48000, 8000),
              (48000, 12000), (44100, 32000), (96000, 44100), (16000, 44100), (32000, 16000)] + RATES
SCENE = ["zoa", "foa", "soa", "toa", "toa_projection"]


def case(seed, variant="default"):
    """the e2e_cases-style description of stream `seed` of a variant's set"""
    rng = np.random.default_rng(900000 + VARIANTS[variant][0] + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    kinds = KINDS + 3 * SCENE if variant == "lfe" else KINDS   # the LFE generator works on scene-based elements
    two = rng.random() < 0.6
    pair = (pick(kinds), pick(kinds)) if two else (pick(kinds),)
    lay = int(rng.integers(0, 14))
    layout = ("binaural",) if lay == 13 else ("ss", lay)
    fs = pick([1024, 1024, 1024, 1024, 2048, 512, 960, 256, 240, 128])
    if "scalable" in pair or "toa_projection" in pair:
        fs = pick([1024, 1024, 2048, 512, 960])   # (the demixer's cross-fade windows are as long as a frame)
    frames = int(rng.integers(4, 9)) if fs >= 512 else int(rng.integers(12, 30))
    c = dict(layout=layout, bit_depth=pick([16, 16, 24, 32]), frames=frames, fs=fs, seed=910000 + 7 * seed,
             sample_size=pick([16, 16, 24, 32]), pair=pair,
             element_gain_q78=int(rng.integers(-1500, 300)), element2_gain_q78=int(rng.integers(-1500, 300)),
             output_gain_q78=int(rng.integers(-600, 300)),
             dmx_modes=[pick(MODES) for _ in range(32)], dmx_modes2=[pick(MODES) for _ in range(32)],
             scalable_modes1=[pick(MODES) for _ in range(32)], scalable_modes2=[pick(MODES) for _ in range(32)],
             recon_salt2=int(rng.integers(1, 50)))
    if rng.random() < 0.25:
        c["pair_ramps"] = True
    trims = {}
    if rng.random() < 0.3:
        trims[0] = (int(rng.integers(1, fs)), 0)
    if rng.random() < 0.3:
        trims[frames - 1] = (0, int(rng.integers(1, fs)))
    if trims:
        c["trims"] = trims
    if rng.random() < 0.2:
        c["rate"], c["out_rate"] = pick(RATES)
        # Sound System H from a scene-based element: render_H2M never writes slot 23 (h2m_rdr.c:1103-1150, the LFE2 slot is
        # not reserved for H), so the reference hands out what its frame buffer held before — silence without a resampler
        # (the buffers only ever rotate zeros into that slot), the previous frame's INTERLEAVED resampler output with one
        # (iamf_resample uses the frame buffer as scratch, IAMF_decoder.c:3235-3244).  Stale memory, not a result: this
        # library writes silence, and the combination is kept out of the comparison.
        if layout == ("ss", 7) and any(k in ("zoa", "foa", "soa", "toa", "toa_projection") for k in pair):
            del c["rate"], c["out_rate"]
    if rng.random() < 0.2:
        c["loudness"] = float(pick([-16.0, -24.0, -31.0]))
        c["mix_loudness_q78"] = int(rng.integers(-30, -10)) * 256
    if rng.random() < 0.15:
        c["limiter"] = False
    elif rng.random() < 0.3:
        c["threshold"] = float(pick([-3.0, -6.0, -0.5]))
    if variant == "dparams":   # scalable / demixing-info elements whose demixing and recon-gain blocks are missing for some frames
        kinds2 = ["scalable"] * 5 + ["dmx:%d" % l for l in (2, 3, 4, 5, 6, 7, 8)] + ["l714dmx"] * 2 + ["stereo", "toa"]
        c["pair"] = pair = tuple(pick(kinds2) for _ in pair)
        for k in ("1", "2"):
            st = pick(STACKS)
            c["scalable_layers" + k] = st
            c["scalable_gains" + k] = {int(li): (int(rng.integers(1, 64)), int(rng.integers(-1200, 600)))
                                       for li in range(len(st)) if rng.random() < 0.4}
            c["dmx_default" + k] = (pick(MODES), int(rng.integers(0, 11)))
        c["drop_blocks"] = (int(rng.integers(1, 1 << 30)), float(pick([0.15, 0.3, 0.6, 1.0])))
        if layout == ("ss", 7) and any(k in SCENE for k in pair):   # (slot 23 of H behind the resampler: see above)
            c.pop("rate", None), c.pop("out_rate", None)
        if c["fs"] < 512:
            c["fs"] = fs = 1024
            c["frames"] = frames = int(rng.integers(4, 9))
            c.pop("trims", None)
    if variant == "wide":
        kinds2 = KINDS + ["scalable"] * 4 + ["dmx:%d" % l for l in (2, 3, 4, 5, 6, 7, 8, 1)]
        pair = tuple(pick(kinds2) for _ in pair)
        c["pair"] = pair
        for k in ("1", "2"):
            st = pick(STACKS)
            c["scalable_layers" + k] = st
            c["scalable_gains" + k] = {int(li): (int(rng.integers(1, 64)), int(rng.integers(-1200, 600)))
                                       for li in range(len(st)) if rng.random() < 0.4}
            c["dmx_default" + k] = (pick(MODES), int(rng.integers(0, 11)))
        if rng.random() < 0.25:
            c["big_endian"] = True
        c.pop("rate", None), c.pop("out_rate", None)
        if rng.random() < 0.35:
            c["rate"], c["out_rate"] = pick(WIDE_RATES)
            if layout == ("ss", 7) and any(k in SCENE for k in pair):   # (slot 23 of H behind the resampler: see above)
                c["out_rate"] = c["rate"] if c["rate"] in (16000, 32000, 44100, 48000) else 48000
                c["rate"] = c["out_rate"]
        if not any(k in ("scalable", "toa_projection") for k in pair) and rng.random() < 0.25:
            c["fs"] = fs = pick([1000, 441, 250, 1023, 77, 2000])
            c["frames"] = frames = int(rng.integers(4, 9)) if fs >= 441 else int(rng.integers(12, 30))
            c.pop("trims", None)
            trims = {}
        if rng.random() < 0.2:   # a frame trimmed away completely, or all but one sample
            f0 = int(rng.integers(0, frames))
            a = int(rng.integers(0, fs + 1))
            tr = dict(c.get("trims", {}))
            tr[f0] = (a, fs - a) if rng.random() < 0.6 else (a, max(0, fs - a - 1))
            c["trims"] = tr
    return c


def case_multi(seed):
    c = case(seed, "wide")
    rng = np.random.default_rng(905000 + VARIANTS["multi"][0] + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    c.pop("pair_ramps", None)
    if any(k in ("scalable", "toa_projection") for k in c["pair"]) and c["fs"] < 512:
        c["fs"] = 1024
    c["elements"] = [pick(KINDS + ["scalable", "dmx:7", "dmx:4", "dmx:3"]) for _ in range(3)]
    if c["fs"] & 3 or c["fs"] < 240:
        c["fs"], c["frames"] = 1024, int(rng.integers(4, 8))
        c.pop("trims", None)
    pres = []
    for i in range(int(rng.integers(2, 4))):
        els = list(rng.permutation(3)[:int(rng.integers(1, 3))])
        lays = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(0, 14))
            lay = ("binaural",) if k == 13 else ("ss", k)
            if lay not in lays:
                lays.append(lay)
        pres.append(dict(id=int(pick([11, 22, 33, 44, 7])) if rng.random() < 0.2 else 10 * (i + 1) + i, elements=[int(e) for e in els],
                         gains=[int(rng.integers(-1200, 300)) for _ in els], out_gain=int(rng.integers(-500, 300)), layouts=lays,
                         loudness=[int(rng.integers(-30, -10)) * 256 for _ in lays]))
    c["presentations"] = pres
    r = rng.random()
    c["mix_id"] = -1 if r < 0.45 else (int(pick(pres)["id"]) if r < 0.9 else 999)
    if "loudness" not in c and rng.random() < 0.5:
        c["loudness"] = float(pick([-16.0, -24.0]))
    # (H from a scene-based element behind the resampler: see above — with several presentations any element may be chosen)
    if c["layout"] == ("ss", 7) and c.get("out_rate") and c.get("rate") != c.get("out_rate"):
        c["out_rate"] = c["rate"] = 48000
    return c


def build_multi(seed):
    import iamf_writer as W
    c = case_multi(seed)
    fs, F, ss, rate = c["fs"], c["frames"], c["sample_size"], c.get("rate", 48000)
    le = not c.get("big_endian", False)
    n = fs * F
    W.LE_DEFAULT = le
    try:
        stream = E._descriptor_prefix(fs, ss, rate, le)
        frames, sid = [], 0
        for k, kind in enumerate(c["elements"]):
            d, fr, _, ns = E._pair_element(kind, k + 1, sid, 200 + 10 * k, c["seed"] + k, n, fs, ss, rate, c)
            stream += d
            frames.append(fr)
            sid += ns
        for i, p in enumerate(c["presentations"]):
            els = [dict(eid=e + 1, pdef=E._pdef_static(100 + 10 * i + 2 * j, rate), default_q78=g)
                   for j, (e, g) in enumerate(zip(p["elements"], p["gains"]))]
            stream += W.mix_presentation(p["id"], els, dict(pdef=E._pdef_static(101 + 10 * i, rate), default_q78=p["out_gain"]), p["layouts"],
                                         loudness_infos=[dict(integrated=q, peak=0) for q in p["loudness"]])
        for f in range(F):
            stream += W.temporal_delimiter()
            subs = []
            for fr in frames:
                blocks, s_ = fr(f)
                stream += blocks
                subs += s_
            stream += W.audio_frames(subs, trim=c.get("trims", {}).get(f))
    finally:
        W.LE_DEFAULT = True
    return stream, c


def case_params(seed):
    """the default generator's stream with mix-gain parameter timelines of its own on the element gains and the output gain:
    definitions of mode 0 or 1, a parameter rate that is or is not the stream's, one to three sub-blocks per block with a
    constant or explicit intervals, every animation type, the odd block missing"""
    import iamf_writer as W
    c = case(seed, "default")
    c.pop("pair_ramps", None)
    rng = np.random.default_rng(907000 + VARIANTS["params"][0] + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    fs, F, rate = c["fs"], c["frames"], c.get("rate", 48000)
    sched = {}
    for pid in (100, 101, 102):
        if rng.random() < 0.25:
            continue
        prate = int(pick([rate, rate, rate, rate // 2, rate * 2, 48000, 44100, 90000, 16000]))
        D = max(1, int(round(fs * prate / rate)))
        mode = int(pick([0, 1]))
        nsub = int(pick([1, 1, 2, 3]))
        if nsub > D:
            nsub = 1
        ci, iv = 0, None
        if nsub == 1:
            ci = D
        elif D % nsub == 0 and rng.random() < 0.5:
            ci = D // nsub
        else:
            cuts = sorted(int(v) for v in rng.choice(np.arange(1, D), size=nsub - 1, replace=False))
            iv = [b - a for a, b in zip([0] + cuts, cuts + [D])]
        pdef = W.param_definition(pid, prate, mode=mode, duration=D, constant_interval=ci, intervals=iv)
        blocks = []
        for f in range(F):
            if rng.random() < 0.1:
                blocks.append(b"")
                continue
            segs = []
            for _ in range(nsub):
                anim = int(pick([W.ANIM_STEP, W.ANIM_LINEAR, W.ANIM_BEZIER]))
                a, b = int(rng.integers(-1500, 500)), int(rng.integers(-1500, 500))
                sg = dict(anim=anim, start=a)
                if anim != W.ANIM_STEP:
                    sg["end"] = b
                if anim == W.ANIM_BEZIER:
                    sg["control"] = int(rng.integers(-1500, 500))
                    sg["rel_time"] = int(rng.integers(0, 256))
                segs.append(sg)
            blocks.append(W.mix_gain_block(pid, segs, mode1=None if mode == 0 else dict(duration=D, constant_interval=ci, intervals=iv)))
        sched[pid] = dict(pdef=pdef, blocks=blocks)
    c["gain_sched"] = sched
    return c


def _leb(v, pad=0):
    """leb128 of v in len(minimal) + pad bytes (a non-minimal encoding is legal: bitstream.c:136-157 reads up to 8 bytes)"""
    out = []
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            break
    for _ in range(pad):
        out[-1] |= 0x80
        out.append(0)
    return bytes(out)


def _split_obus(stream):
    """[(type, redundant, trimming, extension, payload incl. trim / extension fields)]"""
    out, pos = [], 0
    while pos < len(stream):
        h = stream[pos]
        pos += 1
        size, shift = 0, 0
        while True:
            b = stream[pos]
            pos += 1
            size |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        out.append([h >> 3, (h >> 2) & 1, (h >> 1) & 1, h & 1, stream[pos:pos + size]])
        pos += size
    return out


def build_syntax(seed):
    """a stream of the default / wide sets rewritten OBU by OBU into other legal spellings of the same content: non-minimal
    leb128 sizes, OBU extension headers, reserved OBU types and parameter blocks of unknown ids in between, temporal
    delimiters dropped, the audio frames of a temporal unit in another order, redundant copies of descriptors in the
    middle of the data, a redundant copy of the sequence header"""
    rng = np.random.default_rng(911000 + seed)
    src = ("default", "wide")[seed & 1]
    stream, c = build(int(rng.integers(0, VARIANTS[src][1])), src)
    obus = _split_obus(stream)
    opts = dict(pad=rng.random() < 0.5, ext=rng.random() < 0.4, junk=rng.random() < 0.5, unknown_pb=rng.random() < 0.4,
                drop_td=rng.random() < 0.3, shuffle=rng.random() < 0.5, redundant=rng.random() < 0.4)
    descriptors = [o for o in obus if o[0] in (31, 0, 1, 2)]
    out = bytearray()

    def emit(t, red, trim, payload, allow_ext=True):
        ext = 1 if (allow_ext and opts["ext"] and rng.random() < 0.3) else 0
        body = bytes(payload)
        if ext:   # obu_extension_flag: extension_header_size + bytes, behind the trimming fields (IAMF_OBU.c:99-123)
            e = bytes(int(v) for v in rng.integers(0, 256, size=int(rng.integers(0, 6))))
            if trim:   # the two trim leb128s come first
                p = 0
                for _ in range(2):
                    while body[p] & 0x80:
                        p += 1
                    p += 1
                body = body[:p] + _leb(len(e)) + e + body[p:]
            else:
                body = _leb(len(e)) + e + body
        hdr = (t << 3) | (red << 2) | (trim << 1) | ext
        out.extend(bytes([hdr]) + _leb(len(body), int(rng.integers(0, 3)) if opts["pad"] else 0) + body)

    tu = []   # the audio frames of the temporal unit being collected

    def flush_tu():
        if opts["shuffle"] and len(tu) > 1:
            order = rng.permutation(len(tu))
            for i in order:
                emit(*tu[int(i)])
        else:
            for o in tu:
                emit(*o)
        tu.clear()

    in_data = False
    for t, red, trim, ext, payload in obus:
        is_frame = 5 <= t <= 23
        if not is_frame:
            flush_tu()
        if t == 4:
            in_data = True
            if opts["redundant"] and rng.random() < 0.3:   # a redundant copy of every descriptor, and of the sequence header
                for d in descriptors:
                    emit(d[0], 1, 0, d[4], allow_ext=False)
            if opts["drop_td"] and rng.random() < 0.5:
                continue
        if opts["junk"] and rng.random() < 0.2:
            emit(int(rng.integers(24, 31)), 0, 0, bytes(int(v) for v in rng.integers(0, 256, size=int(rng.integers(0, 40)))), allow_ext=False)
        if opts["unknown_pb"] and in_data and rng.random() < 0.15:
            emit(3, 0, 0, _leb(int(rng.integers(5000, 6000))) + bytes(int(v) for v in rng.integers(0, 256, size=int(rng.integers(0, 12)))), allow_ext=False)
        if is_frame:
            tu.append((t, red, trim, payload))
        else:
            emit(t, red, trim, payload, allow_ext=t not in (31,))
    flush_tu()
    c = dict(c, syntax=opts, source=src)
    return bytes(out), c


def build(seed, variant="default"):
    if variant == "syntax":
        return build_syntax(seed)
    if variant == "concat":
        # two or three IA sequences back to back: the decoder answers IAMF_ERR_INVALID_STATE at each new sequence header and is
        # configured again (iamfplayer.c:569-588,622-625; IAMF_decoder.c:2918-2921,3796-3806); the handle's settings are the
        # first stream's, the descriptors (elements, presentations, frame size, sample format, rate) change under it
        rng = np.random.default_rng(909000 + seed)
        parts = [build(int(rng.integers(0, VARIANTS["wide"][1])), "wide") for _ in range(int(rng.integers(2, 4)))]
        c = dict(parts[0][1])
        if c.get("out_rate") and any(p[1].get("rate", 48000) != c.get("rate", 48000) for p in parts):
            pass   # (a fixed output rate over sequences of different stream rates: the resampler is re-opened per sequence)
        # slot 23 of H behind a resampler is stale memory in the reference (see case()): no scene-based element into H when any
        # sequence is resampled
        outr = c.get("out_rate") or 48000
        if c["layout"] == ("ss", 7) and any(p[1].get("rate", 48000) != outr for p in parts) and \
                any(k in SCENE for p in parts for k in p[1]["pair"]):   # (the first sequence's resampler serves the later ones too)
            c["layout"] = ("ss", 9)
        return b"".join(p[0] for p in parts), c
    if variant == "multi":
        return build_multi(seed)
    if variant == "params":
        name = "fuzz_params_%d" % seed
        E.CASES[name] = case_params(seed)
        try:
            return E.build(name)[0], E.CASES[name]
        finally:
            del E.CASES[name]
    name = "fuzz_%s_%d" % (variant, seed)
    E.CASES[name] = case(seed, variant)
    try:
        return E.build(name)[0], E.CASES[name]
    finally:
        del E.CASES[name]


def reference_h_slot_23_is_stale(c, layouts):
    """slot 23 of Sound System H from a scene-based element behind a resampler is stale frame-buffer content in the reference
    (render_H2M never writes it for H; see case()): case() keeps that combination out of the sets, a run-time layout SWITCH to
    H (switch_case) can still reach it — found by hunting the switch set (8 of 3000 seeds, all of this kind: the one PCM
    column that carries slot 23).  Return values are compared, the PCM is not."""
    outr = c.get("out_rate") or 48000
    return ("ss", 7) in [tuple(l) for l in layouts] and c.get("rate", 48000) != outr and any(k in SCENE for k in c["pair"])


def reference_gain_list_overflows(c):
    """True if a scalable element of case `c` makes the reference write past its 12-entry output-gain arrays
    (iamf_stream_scale_demixer_configure, IAMF_decoder.c:2365-2380: chs[count] is stored before it is tested, so the 13th
    flagged bit — mapped or not — lands behind the array once 12 entries are collected).  Such streams are undefined behaviour
    there (seed 7214 of 'wide': its first gain turns 0) and are named, not compared, by the hunting tools."""
    surround = {0: 1, 1: 2, 2: 5, 3: 5, 4: 5, 5: 7, 6: 7, 7: 7, 8: 3}
    for k in ("1", "2"):
        layers, gains = c.get("scalable_layers" + k), c.get("scalable_gains" + k)
        if not layers or not gains:
            continue
        count = 0
        for li, lay in enumerate(layers):
            if li not in gains:
                continue
            flags = gains[li][0]
            for g in range(6):
                if not flags & (1 << g):
                    continue
                if count >= 12:
                    return True
                s_ = surround[lay]
                valid = {5: lay in (0, 1, 8), 4: lay in (1, 8), 3: s_ == 5, 2: s_ == 5, 1: True, 0: True}[g]
                count += 1 if valid else 0
    return False


def decode_kwargs(c, variant="default"):
    kw = dict(bit_depth=c["bit_depth"], out_rate=c.get("out_rate", 0), loudness=c.get("loudness", 0.0),
              limiter=c.get("limiter", True), threshold=c.get("threshold", -1.0))
    if variant == "tv":
        kw["pcm_channels"] = 12   # IAMF_decoder.c:3492-3495
    if "mix_id" in c and c["mix_id"] >= 0:
        kw["mix_id"] = c["mix_id"]
    return kw


def digest(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest()


# The streams above through the reference PLAYER's loop (decoder_driver.decode_stream_blocks: iamfplayer.c:529-662 with a block
# buffer of `block` bytes): configure until it stops asking for more, decode while it consumes something, the rest of a
# block in front of the next.  Compared: the PCM and every call's (return value, rsize).
N_BLOCKS = 200


def blocks_case(seed):
    """-> (variant, seed of that variant, block size)"""
    rng = np.random.default_rng(955000 + seed)
    variant = ["wide", "multi", "default"][seed % 3]
    block = int([184320, 65536, 30000, 20000, 12345, 9000, 6000, int(rng.integers(3000, 9000)), 2500, 777][int(rng.integers(0, 10))])
    return variant, int(rng.integers(0, VARIANTS[variant][1])), block


def events_digest(events):
    return hashlib.sha256(repr([(k, int(r), int(n)) for k, r, n in events]).encode()).hexdigest()


def meta_digest(md):
    """the rows decoder_driver.last_metadata collected (pts, sound system, samples, bit depth, rate, sound mode, loudness
    records, the DEMIXING record) and the PCM of the second flush call"""
    return hashlib.sha256(repr([[int(v) for v in r] for r in md["rows"]]).encode() + bytes(md.get("flush2", b""))).hexdigest()


# The -DSAMSUNG_TV build's run-time layout switch (IAMF_decoder.c:3819-3881) on the TV set's streams: one or two switches to
# random layouts after random numbers of delivered frames (decoder_driver.decode_stream_switching)
N_SWITCH = 120


def switch_case(seed):
    """-> (seed of the tv variant, layouts in turn, frames delivered before each switch)"""
    rng = np.random.default_rng(957000 + seed)
    vs = int(rng.integers(0, VARIANTS["tv"][1]))
    c = case(vs, "tv")
    lays = [c["layout"]]
    for _ in range(int(rng.integers(1, 3))):
        k = int(rng.integers(0, 14))
        lays.append(("binaural",) if k == 13 else ("ss", k))
    nfr = max(2, c["frames"] - 1)
    after = sorted(int(v) for v in rng.choice(np.arange(1, nfr + 1), size=min(len(lays) - 1, nfr), replace=False))
    return vs, lays[:len(after) + 1], after


# The streams through the reference player's demuxer loop (decoder_driver.decode_stream_units): descriptors in one configure
# call, one temporal unit per decode call, both with rsize == NULL.
N_UNITS = 160


def units_case(seed):
    """-> (variant, seed of that variant, descriptors, [temporal units])"""
    rng = np.random.default_rng(959000 + seed)
    variant = ["wide", "multi", "default", "params"][seed % 4]
    vs = int(rng.integers(0, VARIANTS[variant][1]))
    stream, c = build(vs, variant)
    obus = _split_obus(stream)
    first = next(i for i, o in enumerate(obus) if o[0] == 4)

    def pack(os_):
        out = bytearray()
        for t, red, trim, ext, payload in os_:
            out.extend(bytes([(t << 3) | (red << 2) | (trim << 1) | ext]) + _leb(len(payload)) + payload)
        return bytes(out)
    units, cur = [], []
    for o in obus[first:]:
        if o[0] == 4 and cur:
            units.append(pack(cur))
            cur = []
        cur.append(o)
    units.append(pack(cur))
    return variant, vs, pack(obus[:first]), units, c


# A group whose handles decode DIFFERENT streams of one topology: the same descriptors' shape (elements, layers, layouts,
# frame size, sample format, rates) but each stream its own audio, gains, ramps, demixing modes, recon gains and trims — a
# group keeps all of that per stream (ramp rows, stage records, launch runs by trim).  Expected per handle: the reference's
# decode of ITS stream.
N_GMIX = 60


def gmix_case(seed):
    """-> (variant, [seed-specific case dicts of one topology])"""
    rng = np.random.default_rng(961000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    variant = ["wide", "dparams", "default"][seed % 3]
    base = case(int(rng.integers(0, VARIANTS[variant][1])), variant)
    if base["fs"] & 3:
        base["fs"] = 1024
        base["frames"] = int(rng.integers(4, 9))
    base.pop("trims", None)
    fs, frames = base["fs"], base["frames"]
    out = []
    for k in range(int(rng.integers(3, 7))):
        c = dict(base)
        c["seed"] = base["seed"] + 1000 * (k + 1)
        c["element_gain_q78"] = int(rng.integers(-1500, 300))
        c["element2_gain_q78"] = int(rng.integers(-1500, 300))
        c["output_gain_q78"] = int(rng.integers(-600, 300))
        c["dmx_modes"] = [pick(MODES) for _ in range(32)]
        c["dmx_modes2"] = [pick(MODES) for _ in range(32)]
        c["scalable_modes1"] = [pick(MODES) for _ in range(32)]
        c["scalable_modes2"] = [pick(MODES) for _ in range(32)]
        c["recon_salt2"] = int(rng.integers(1, 50))
        if "drop_blocks" in base:
            c["drop_blocks"] = (int(rng.integers(1, 1 << 30)), base["drop_blocks"][1])
        if rng.random() < 0.5:
            c["pair_ramps"] = True
        else:
            c.pop("pair_ramps", None)
        trims = {}
        if rng.random() < 0.4:
            trims[0] = (int(rng.integers(1, fs)), 0)
        if rng.random() < 0.4:
            trims[frames - 1] = (0, int(rng.integers(1, fs)))
        if rng.random() < 0.2:
            a = int(rng.integers(1, fs))
            trims[int(rng.integers(1, max(2, frames - 1)))] = (a, fs - a)
        if trims:
            c["trims"] = trims
        out.append(c)
    return variant, out


def gmix_build(seed):
    variant, cases = gmix_case(seed)
    streams = []
    for k, c in enumerate(cases):
        name = "fuzz_gmix_%d_%d" % (seed, k)
        E.CASES[name] = c
        try:
            streams.append(E.build(name)[0])
        finally:
            del E.CASES[name]
    return variant, cases, streams
