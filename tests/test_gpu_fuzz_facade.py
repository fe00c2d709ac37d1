"""-m gpu: 240 seeded random IAMF streams (tests/e2e_fuzz.py: one or two elements of sixteen kinds, fourteen output layouts,
bit depths, sample formats, frame sizes from 128 to 2048, gains, ramps, trims, rate conversion, loudness, limiter settings)
through the IAMF_decoder.h facade of libiamf_hip.so against what the REAL reference returned for the same bytes
(oracle/gen_golden_fuzz.py -> tests/golden/fuzz.json: per-call return values and a SHA-256 of the PCM).  Bit-exact or fail."""
import json
import os

import numpy as np
import pytest

import e2e_fuzz as F
from decoder_driver import decode_stream

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz.json")))


from test_gpu_group import group_decode_all, lib  # noqa: E402,F401  (the fixture that declares the group entry points)


@pytest.mark.parametrize("seed", range(F.N_SEEDS))
def test_random_stream_matches_the_reference_decoder(lib, seed):
    want = GOLD[str(seed)]
    assert "sha256" in want, want   # (the reference decoded every stream of the committed set)
    stream, c = F.build(seed)
    pcm, rets = decode_stream(lib, stream, c["layout"], **F.decode_kwargs(c))
    desc = {k: v for k, v in c.items() if not k.endswith(("modes", "modes1", "modes2"))}
    assert [int(r) for r in rets] == want["rets"], (seed, desc)
    assert list(pcm.shape) == want["shape"], (seed, desc)
    assert F.digest(pcm) == want["sha256"], (seed, desc)


@pytest.mark.parametrize("seed", range(0, F.N_SEEDS, 5))
def test_random_stream_through_a_group_of_handles(lib, seed):
    """every fifth stream through five out-of-step handles of a group: each handle's PCM is the single handle's"""
    want = GOLD[str(seed)]
    stream, c = F.build(seed)
    case = dict(c)
    rc, outs = group_decode_all(lib, case, stream, 5, 2, starve=lambda r, i: (r + i) % 4 == 0 and i % 2 == 0)
    assert rc == 0, (seed, rc)
    for i, (pcm, rets) in enumerate(outs):
        assert [int(r) for r in rets] == want["rets"], (seed, i)
        assert F.digest(pcm) == want["sha256"], (seed, i)
