"""The checker behind bench.py's `verified` entry (tests/bench_verify.py) runs on the host alone: every workload kind
yields F*fs - 240 sample-frames of int16 from the oracle, and the comparison helper reports what it should."""
import numpy as np
import pytest

import bench_verify as V
import synth


@pytest.mark.parametrize("kind,in_id,out_id,in_ch,out_ch", [
    ("h2m", 3, 0x1020, 16, 2), ("h2m", 3, 0x9A3, 16, 24), ("m2m", 0x714, 0x470, 12, 12), ("h2m_lfe", 3, 0x050, 16, 6),
    ("h2m_proj", 3, 0x1020, 16, 2), ("dmx", 5, 1, 8, 2), ("dmx", 7, 3, 12, 8), ("demix", 0x714, 0x470, 12, 12)])
def test_oracle_pcm_shapes(kind, in_id, out_id, in_ch, out_ch):
    F, fs = 3, 1024
    x = synth.hot(77, in_ch, F * fs).reshape(in_ch, F, fs).transpose(1, 0, 2).copy()
    proj = None
    if kind == "h2m_proj":
        proj = (np.eye(in_ch, dtype=np.float32) * np.float32(0.5))
    want = V.oracle_pcm(kind, in_id, out_id, out_ch, x, fs, s=5, proj=proj)
    assert want.dtype == np.int16 and want.shape == (F * fs - 240, out_ch)
    assert np.abs(want.astype(np.int32)).max() > 1000   # the programme is there
    if kind == "h2m_proj":   # an exact half-scale identity de-mapping equals rendering x / 2
        direct = V.oracle_pcm("h2m", in_id, out_id, out_ch, (x * np.float32(0.5)).astype(np.float32), fs)
        assert np.array_equal(want, direct)


def test_compare_reports_lsb_and_fraction():
    a = np.zeros((10, 2), dtype=np.int16)
    b = a.copy()
    assert V.compare(a, b, 0) == (True, 0, 0.0)
    b[3, 1] = 1
    ok, w, fr = V.compare(a, b, 0)
    assert (ok, w) == (False, 1) and abs(fr - 0.05) < 1e-9
    assert V.compare(a, b, 1)[0] is True
    assert V.compare(a, b[:5], 1)[0] is False


def test_fir_pcm_from_stage_is_limiter_plus_pack():
    import oracle_lib as O
    F, fs = 2, 1024
    y = synth.hot(5, 2, F * fs)
    z, _ = O.limiter_run(y, [fs] * F, flush=False)
    assert np.array_equal(V.fir_pcm_from_stage(np.ascontiguousarray(y.T), fs, F), O.pack(z, 16))
