"""-m gpu: the HIP render path fed with the element PCM of the end-to-end cases, against PCM the
REAL reference decoder produced from the corresponding .iamf streams (tests/golden/e2e.npz).
cfg1 of BASELINE.json (stereo element -> Sound System A, iamfplayer -o2 -s0) is `stereo_A_s16`."""
import numpy as np
import pytest

import e2e_cases
import e2e_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G


SINGLE = [n for n, c in sorted(e2e_cases.CASES.items()) if n != "two_elements_A_s32" and
          not (c.get("ramps") or c.get("concat") or c.get("dmx_modes") or c.get("out_rate") or c.get("trims") or c.get("scalable") or c.get("pair"))]


@pytest.mark.parametrize("name", SINGLE)
def test_hip_matches_reference_decoder_pcm(hip, golden, name):
    A, G = hip
    _, info = e2e_cases.build(name)
    c = info["case"]
    el = info["elements"][0]
    out_id = e2e_model.out_id_of(c["layout"])
    ch = A.layout_channels(out_id)
    if el["kind"] == "scene":
        mx = A.get_h2m_matrix(el["order"], out_id)
    else:
        mx = A.get_m2m_matrix(e2e_model.LAYOUT_RID[el["layout"]], out_id)
    fmt = {16: A.FMT_S16, 24: A.FMT_S24, 32: A.FMT_S32}[c.get("bit_depth", 16)]
    gains = dict(element=[e2e_model.q78_to_lin(c.get("element_gain_q78", 0))],
                 output=[e2e_model.q78_to_lin(c.get("output_gain_q78", 0))])
    loud = c.get("loudness", 0.0) != 0.0
    if loud:
        import oracle_lib as O
        mix_l = np.float32(c.get("mix_loudness_q78", 0)) * np.float32(2.0 ** -8)
        gains["loudness"] = [O.lib().orc_db2lin(float(np.float32(c["loudness"]) - mix_l))]
    got = G.hip_render(mx, ch, el["x"][None], frame_size=c["fs"], fmt=fmt, limiter=c.get("limiter", True),
                       flush=True, frames_per_call=[1] * c["frames"], gains=gains, loudness=loud,
                       threshold_db=c.get("threshold", -1.0), projection=A.PROJ_EXACT)[0]
    want = golden.npz("e2e")[name]
    assert got.shape == want.shape
    assert np.array_equal(got, want), name
