"""Scalable channel audio: layer stacks, channel orders and the per-frame schedules of
the demixer cases.  Shared by the golden generator (which drives the REAL reference demixer,
src/iamf_dec/demixer.c, through its exported symbols), the oracle tests and the GPU tests.

The table logic below restates, for TEST purposes, the reference's helpers:
  channel ids                        IAMF_types.h:61-90
  playback / audio-layer orders      IAMF_utils.c:117-133,181-196
  surround / top counts              IAMF_utils.c:157-161
  new channels of a layer            IAMF_decoder.c:450-531   (iamf_channel_layout_get_new_channels)
  output-gain channel map            IAMF_decoder.c:533-600   (iamf_output_gain_channel_map)
  recon-gain flags -> channel order  IAMF_decoder.c:371-448
"""
import ctypes as C

import numpy as np

import synth

CH = dict(INVALID=0, L7=1, R7=2, C=3, LFE=4, SL7=5, SR7=6, BL7=7, BR7=8, HFL=9, HFR=10, HBL=11, HBR=12,
          MONO=13, L2=14, R2=15, TL=16, TR=17, L3=18, R3=19, SL5=20, SR5=21, HL=22, HR=23)
CH["L5"], CH["R5"] = CH["L7"], CH["R7"]
_n = lambda names: [CH[x] for x in names.split()]

# IAChannelLayoutType: 0 mono, 1 stereo, 2 5.1, 3 5.1.2, 4 5.1.4, 5 7.1, 6 7.1.2, 7 7.1.4, 8 3.1.2
PLAYBACK = [_n("MONO"), _n("L2 R2"), _n("L5 R5 C LFE SL5 SR5"), _n("L5 R5 C LFE SL5 SR5 HL HR"),
            _n("L5 R5 C LFE SL5 SR5 HFL HFR HBL HBR"), _n("L7 R7 C LFE SL7 SR7 BL7 BR7"),
            _n("L7 R7 C LFE SL7 SR7 BL7 BR7 HL HR"), _n("L7 R7 C LFE SL7 SR7 BL7 BR7 HFL HFR HBL HBR"),
            _n("L3 R3 C LFE TL TR")]
AUDIO_LAYER = [_n("MONO"), _n("L2 R2"), _n("L5 R5 SL5 SR5 C LFE"), _n("L5 R5 SL5 SR5 HL HR C LFE"),
               _n("L5 R5 SL5 SR5 HFL HFR HBL HBR C LFE"), _n("L7 R7 SL7 SR7 BL7 BR7 C LFE"),
               _n("L7 R7 SL7 SR7 BL7 BR7 HL HR C LFE"), _n("L7 R7 SL7 SR7 BL7 BR7 HFL HFR HBL HBR C LFE"),
               _n("L3 R3 TL TR C LFE")]
SURROUND = [1, 2, 5, 5, 5, 7, 7, 7, 3]
TOP = [0, 0, 0, 2, 4, 0, 2, 4, 2]
# (substreams, coupled) per layout when it is the first layer
LAYOUT_SUBSTREAMS = {0: (1, 0), 1: (1, 1), 2: (4, 2), 3: (5, 3), 4: (6, 4), 5: (5, 3), 6: (6, 4), 7: (7, 5), 8: (4, 2)}


def new_channels(last, cur):
    if last is None:
        return list(AUDIO_LAYER[cur])
    s1, s2, t1, t2 = SURROUND[last], SURROUND[cur], TOP[last], TOP[cur]
    out = []
    if s1 < 5 <= s2:
        out += _n("L5 R5")
    if s1 < 7 <= s2:
        out += _n("SL7 SR7")
    if t2 != t1 and t2 == 4:
        out += _n("HFL HFR")
    if t2 - t1 == 4:
        out += _n("HBL HBR")
    elif not t1 and t2 - t1 == 2:
        out += _n("TL TR") if s2 < 5 else _n("HL HR")
    if s1 < 3 <= s2:
        out += _n("C LFE")
    if s1 < 2 <= s2:
        out += _n("L2")
    return out


def channels_order(layers):
    """decoded channel order of a stack of layers (IAMF_decoder.c:1704-1709) and the per-layer
    (channels, substreams, coupled) counts a writer needs"""
    order, per_layer, last = [], [], None
    for lay in layers:
        nc = new_channels(last, lay)
        mono_like = {CH["C"], CH["LFE"], CH["MONO"], CH["L2"]} if last is not None else {CH["C"], CH["LFE"], CH["MONO"]}
        singles = [c for c in nc if c in mono_like]
        coupled = (len(nc) - len(singles)) // 2
        per_layer.append(dict(channels=len(nc), substreams=coupled + len(singles), coupled=coupled))
        order += nc
        last = lay
    return order, per_layer


GAIN_BITS = dict(RTF=0, LTF=1, RS=2, LS=3, R=4, L=5)  # IAOutputGainChannel, IAMF_decoder_private.h:62-70


def output_gain_channel(layout, g):
    s = SURROUND[layout]
    if g == "L":
        return {0: CH["MONO"], 1: CH["L2"], 8: CH["L3"]}.get(layout, 0)
    if g == "R":
        return {1: CH["R2"], 8: CH["R3"]}.get(layout, 0)
    if g == "LS":
        return CH["SL5"] if s == 5 else 0
    if g == "RS":
        return CH["SR5"] if s == 5 else 0
    if g == "LTF":
        return CH["TL"] if s < 5 else CH["HL"]
    if g == "RTF":
        return CH["TR"] if s < 5 else CH["HR"]
    return 0


def output_gain_list(layers, layer_gains):
    """layer_gains: {layer index: (flags bitmask over GAIN_BITS, linear gain)} -> [(channel, gain)]
    in the order iamf_stream_scale_demixer_configure builds it (IAMF_decoder.c:2365-2378)"""
    out = []
    names = sorted(GAIN_BITS, key=lambda k: GAIN_BITS[k])
    for li, lay in enumerate(layers):
        if li in layer_gains:
            flags, gain = layer_gains[li]
            for c, nm in enumerate(names):
                if flags & (1 << c):
                    ch = output_gain_channel(lay, nm)
                    if ch:
                        out.append((ch, gain))
    return out


RE = dict(L=0, C=1, R=2, LS=3, RS=4, LTF=5, RTF=6, LB=7, RB=8, LTB=9, RTB=10, LFE=11)
_RE_MAP = [
    _n("MONO") + [0] * 11,
    [CH["L2"], 0, CH["R2"]] + [0] * 9,
    [CH["L5"], CH["C"], CH["R5"], CH["SL5"], CH["SR5"], 0, 0, 0, 0, 0, 0, CH["LFE"]],
    [CH["L5"], CH["C"], CH["R5"], CH["SL5"], CH["SR5"], CH["HL"], CH["HR"], 0, 0, 0, 0, CH["LFE"]],
    [CH["L5"], CH["C"], CH["R5"], CH["SL5"], CH["SR5"], CH["HFL"], CH["HFR"], 0, 0, CH["HBL"], CH["HBR"], CH["LFE"]],
    [CH["L7"], CH["C"], CH["R7"], CH["SL7"], CH["SR7"], 0, 0, CH["BL7"], CH["BR7"], 0, 0, CH["LFE"]],
    [CH["L7"], CH["C"], CH["R7"], CH["SL7"], CH["SR7"], CH["HL"], CH["HR"], CH["BL7"], CH["BR7"], 0, 0, CH["LFE"]],
    [CH["L7"], CH["C"], CH["R7"], CH["SL7"], CH["SR7"], CH["HFL"], CH["HFR"], CH["BL7"], CH["BR7"], CH["HBL"],
     CH["HBR"], CH["LFE"]],
    [CH["L3"], CH["C"], CH["R3"], 0, 0, CH["TL"], CH["TR"], 0, 0, 0, 0, CH["LFE"]],
]
_RE_ORDER = [RE[k] for k in "L C R LS RS LTF RTF LB RB LTB RTB LFE".split()]


def recon_flags(l1, l2):
    if l1 == l2:
        return 0
    s1, s2, t1, t2 = SURROUND[l1], SURROUND[l2], TOP[l1], TOP[l2]
    f = 0
    if s1 != s2:
        if s2 <= 3:
            f |= (1 << RE["L"]) | (1 << RE["R"])
        elif s2 == 5:
            f |= (1 << RE["LS"]) | (1 << RE["RS"])
        elif s2 == 7:
            f |= (1 << RE["LB"]) | (1 << RE["RB"])
    if t2 != t1 and t2 == 4:
        f |= (1 << RE["LTB"]) | (1 << RE["RTB"])
    if s2 == 5 and t1 and t2 == t1:
        f |= (1 << RE["LTF"]) | (1 << RE["RTF"])
    return f


def recon_order(layout, flags):
    return [_RE_MAP[layout][c] for c in _RE_ORDER if flags & (1 << c)]


# ---- stage-level cases: what demixer_* of the reference is driven with ----
# schedule entries: (mode or -1 = keep, recon gains (one per recon channel, k/255) or None = keep)
def _sched(n_rec, seed):
    rng = np.random.default_rng(seed)
    modes = [-1, 1, 2, 2, 4, 5, 6, 0, 0, 1]
    out = []
    for f, m in enumerate(modes):
        rg = None
        if n_rec and f not in (0, 4):
            rg = [float(np.float32(int(v)) / np.float32(255.0)) for v in rng.integers(120, 256, size=n_rec)]
        out.append((m, rg))
    return out


def make_case(layers, layer_gains=None, default=(1, 3), offset=0, fs=256, seed=700):
    layout = layers[-1]
    order, per_layer = channels_order(layers)
    flags = recon_flags(layers[0], layout) if len(layers) > 1 else 0
    rec = recon_order(layout, flags)
    return dict(layers=layers, layout=layout, order=order, per_layer=per_layer, flags=flags, recon=rec,
                gains=output_gain_list(layers, layer_gains or {}), default=default, offset=offset, fs=fs,
                seed=seed, schedule=_sched(len(rec), seed))


STAGE_CASES = {
    "s2_512_714": make_case([1, 3, 7], {0: (0b110000, 0.7079458), 1: (0b001111, 1.4125376)}),
    "m_s_510_710": make_case([0, 1, 2, 5], {1: (0b110000, 0.5)}, default=(2, 5), seed=710),
    "s_312": make_case([1, 8], default=(4, 0), seed=720),
    "510_514": make_case([2, 4], seed=730),
    "512_514": make_case([3, 4], {0: (0b001111, 0.8)}, default=(5, 9), seed=740),
    "312_512_712": make_case([8, 3, 6], {0: (0b110011, 1.1885022)}, default=(6, 2), offset=37, seed=750),
    "714_single": make_case([7], seed=760),
}


def case_input(c):
    """decoded channels (in `order`) for every frame: [frames][channels][fs]"""
    n = len(c["schedule"])
    return np.stack([synth.uniform(c["seed"] + f, len(c["order"]), c["fs"], 0.5) for f in range(n)])


def drive_demixer(lib, prefix, c, x):
    """One demixer instance driven over the case's frames the way IAMF_decoder.c:2324-2386 drives it;
    `lib`/`prefix` select the reference (`demixer_`) or the oracle (`orc_demixer_`)."""
    IP, FP = C.POINTER(C.c_int), C.POINTER(C.c_float)
    f = lambda name: getattr(lib, prefix + name)
    f("open").restype = C.c_void_p
    d = C.c_void_p(f("open")(c["fs"]))
    arr_i = lambda v: (C.c_int * max(len(v), 1))(*v)
    arr_f = lambda v: (C.c_float * max(len(v), 1))(*v)
    f("set_channel_layout")(d, c["layout"])
    f("set_channels_order")(d, arr_i(c["order"]), len(c["order"]))
    f("set_output_gain")(d, arr_i([g[0] for g in c["gains"]]), arr_f([g[1] for g in c["gains"]]), len(c["gains"]))
    f("set_demixing_info")(d, c["default"][0], c["default"][1])
    rec = c["recon"]
    f("set_recon_gain")(d, len(rec), arr_i(rec), arr_f([1.0] * len(rec)), c["flags"])
    f("set_frame_offset")(d, c["offset"])
    outs = []
    demix = f("demixing") if prefix == "demixer_" else f("demix")
    for fr, (mode, rg) in enumerate(c["schedule"]):
        if rg is not None:
            f("set_recon_gain")(d, len(rec), arr_i(rec), arr_f(rg), c["flags"])
        if mode > -1:
            f("set_demixing_info")(d, mode, -1)
        src = np.ascontiguousarray(x[fr], dtype=np.float32).copy()
        dst = np.zeros_like(src)
        r = demix(d, dst.ctypes.data_as(FP), src.ctypes.data_as(FP), c["fs"])
        assert r == 0, (prefix, fr, r)
        outs.append(dst)
    f("close")(d)
    return np.stack(outs)


