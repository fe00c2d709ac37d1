"""End-to-end: the stage order / gains / mixer / loudness / limiter / PCM pack of the oracle
against PCM produced by the REAL reference decoder (IAMF_decoder_*) on synthetic LPCM streams."""
import numpy as np
import pytest

import e2e_cases
import e2e_model


SIMPLE = [n for n, c in sorted(e2e_cases.CASES.items())
          if not (c.get("ramps") or c.get("pair_ramps") or c.get("concat") or c.get("dmx_modes") or c.get("out_rate") or c.get("trims") or "_dmx_" in n)]


@pytest.mark.parametrize("name", SIMPLE)
def test_oracle_pipeline_matches_reference_decoder(golden, name):
    _, info = e2e_cases.build(name)
    want = golden.npz("e2e")[name]
    got = e2e_model.run_case(info)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.array_equal(got, want), name


def test_int32_full_scale_wrap_is_what_the_reference_does(golden):
    """FLOAT2INT32 quirk (IAMF_decoder.c:114-119): clipped positive samples wrap to INT32_MIN"""
    pcm = golden.npz("e2e")["two_elements_A_s32"]
    assert (pcm == np.iinfo(np.int32).min).sum() > 10
    assert (pcm == np.iinfo(np.int32).max).sum() == 0
