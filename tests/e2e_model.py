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


def run_case(info):
    c = info["case"]
    out_id = out_id_of(c["layout"])
    ch = O.OUT_CH[out_id]
    fs = c["fs"]
    gains_q = [c.get("element_gain_q78", 0)] + [0] * (len(info["elements"]) - 1)
    ys = []
    for el, gq in zip(info["elements"], gains_q):
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
