"""Replays an e2e case (tests/e2e_cases.py) through the oracle's stages in the reference's stage
order (IAMF_decoder.c:3335-3500) -> packed PCM.  Used to check the oracle against the goldens
the real decoder produced, and as the expected value for the HIP path / facade on the same case."""
import ctypes as C

import numpy as np

import oracle_lib as O

LAYOUT_RID = {0: O.SS["MONO"], 1: O.SS["STEREO"], 2: O.SS["L51"], 3: O.SS["L512"], 4: O.SS["L514"],
              5: O.SS["L71"], 6: O.SS["L712"], 7: O.SS["L714"], 8: O.SS["L312"]}
SS_RID = [O.SS["A"], O.SS["B"], O.SS["C"], O.SS["D"], O.SS["E"], O.SS["F"], O.SS["G"], O.SS["H"],
          O.SS["I"], O.SS["J"], O.SS["L712"], O.SS["L312"], O.SS["MONO"]]


def q78_to_lin(q):
    """q_to_float(q, 8) then db2lin (fixedp11_5.c:45-47,72)"""
    db = np.float32(q) * np.float32(2.0 ** -8)
    return float(O.lib().orc_db2lin(float(db)))


def out_id_of(layout):
    return SS_RID[layout[1]] if layout[0] == "ss" else O.SS["BINAURAL"]


def element_matrix(el, out_id):
    if el["kind"] == "scene":
        return O.get_h2m(el["order"], out_id)
    return O.get_m2m(LAYOUT_RID[el["layout"]], out_id)


LAYOUT_SS = [12, 0, 1, 2, 3, 8, 10, 9, 11]   # layer layout -> IAMF_SoundSystem (IAMF_decoder.c:268-275)
LAYOUT_CH = [1, 2, 6, 8, 10, 8, 10, 12, 6]
SS_CH = [2, 6, 8, 10, 11, 12, 14, 24, 8, 12, 10, 6, 1]


def select_layer(layers, layout):
    """iamf_stream_set_output_layout, IAMF_decoder.c:1776-1822"""
    if len(layers) == 1:
        return 0
    if layout[0] != "ss":
        return len(layers) - 1
    for i, l in enumerate(layers):
        if LAYOUT_SS[l] == layout[1]:
            return i
    for i, l in enumerate(layers):
        if LAYOUT_CH[l] > SS_CH[layout[1]]:
            return i
    return len(layers) - 1


def demix_scalable(el, c):
    """decoded channels of the layers up to the selected one -> target layout in playback order, frame
    by frame through the oracle's demixer the way IAMF_decoder.c:2324-2386 drives the reference's"""
    import demix_cases as D
    import e2e_cases
    li = select_layer(el["layers"], c["layout"])
    layers = el["layers"][:li + 1]
    order, _ = D.channels_order(layers)
    layout = layers[-1]
    gains = D.output_gain_list(layers, {k: (f, q78_to_lin(q)) for k, (f, q) in el.get("gains", e2e_cases.SCALABLE_GAINS).items() if k <= li})
    flags = D.recon_flags(layers[0], layout) if li else 0
    rec = D.recon_order(layout, flags)
    sched = []
    for f in range(c["frames"]):
        rg = None
        if rec and el["wl"][li]["recon"]:   # qf_to_float(byte, 8): double division, then float (fixedp11_5.c:53)
            rg = [float(np.float32(np.float64(np.float32(v)) / 255.0)) for v in e2e_cases.scalable_recon_bytes(f, len(rec), el.get("salt", 0))]
        sched.append((el.get("modes", e2e_cases.SCALABLE_MODES)[f], rg))
    case = dict(layout=layout, order=order, gains=gains, default=(1, 3), recon=rec, flags=flags, offset=0,
                fs=c["fs"], schedule=sched)
    x = el["x"][:len(order)].reshape(len(order), c["frames"], c["fs"]).transpose(1, 0, 2)
    y = D.drive_demixer(O.lib(), "orc_demixer_", case, x)
    return layout, np.ascontiguousarray(y.transpose(1, 0, 2).reshape(len(order), -1))


def run_case(info):
    c = info["case"]
    out_id = out_id_of(c["layout"])
    ch = O.OUT_CH[out_id]
    fs = c["fs"]
    gains_q = [c.get("element_gain_q78", 0), c.get("element2_gain_q78", 0)][:len(info["elements"])]
    ys = []
    for el, gq in zip(info["elements"], gains_q):
        if el["kind"] == "scalable":
            layout, xd = demix_scalable(el, c)
            y = O.render(O.get_m2m(LAYOUT_RID[layout], out_id), xd, ch)
        else:
            y = O.render(element_matrix(el, out_id), el["x"], ch)
        g = q78_to_lin(gq)
        O.lib().orc_frame_gain_const(O.fp(y), ch, y.shape[1], g)
        ys.append(y)
    z = np.zeros_like(ys[0])
    for y in ys:
        z = (z + y).astype(np.float32)
    O.lib().orc_frame_gain_const(O.fp(z), ch, z.shape[1], q78_to_lin(c.get("output_gain_q78", 0)))
    if c.get("loudness", 0.0) != 0.0:
        mix_l = np.float32(c.get("mix_loudness_q78", 0)) * np.float32(2.0 ** -8)
        g = O.lib().orc_db2lin(float(np.float32(c["loudness"]) - mix_l))
        O.lib().orc_loudness(O.fp(z), z.shape[1], ch, g)
    if c.get("limiter", True):
        z, _ = O.limiter_run(z, [fs] * c["frames"], thr_db=c.get("threshold", -1.0))
    return O.pack(z, c.get("bit_depth", 16))
