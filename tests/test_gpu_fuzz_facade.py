"""-m gpu: 240 seeded random IAMF streams (tests/e2e_fuzz.py: one or two elements of sixteen kinds, fourteen output layouts,
bit depths, sample formats, frame sizes from 128 to 2048, gains, ramps, trims, rate conversion, loudness, limiter settings)
through the IAMF_decoder.h facade of libiamf_hip.so against what the REAL reference returned for the same bytes
(oracle/gen_golden_fuzz.py -> tests/golden/fuzz.json: per-call return values and a SHA-256 of the PCM).  Bit-exact or fail."""
import json
import os

import numpy as np
import pytest

import e2e_fuzz as F
from decoder_driver import decode_stream, decode_stream_blocks, decode_stream_switching, decode_stream_units

pytestmark = pytest.mark.gpu

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(_G, "fuzz.json")))
GOLD_V = {v: json.load(open(os.path.join(_G, "fuzz_%s.json" % v))) for v in ("lfe", "tv", "wide", "multi", "params", "concat", "syntax", "dparams")}


from test_gpu_group import group_decode_all, lib  # noqa: E402,F401  (the fixture that declares the group entry points)


@pytest.mark.parametrize("seed", range(F.N_SEEDS))
def test_random_stream_matches_the_reference_decoder(lib, seed):
    want = GOLD[str(seed)]
    assert "sha256" in want, want   # (the reference decoded every stream of the committed set)
    stream, c = F.build(seed)
    md = dict(rows=[], owns_anchors=True, strict=False)
    pcm, rets = decode_stream(lib, stream, c["layout"], metadata=md, **F.decode_kwargs(c))
    desc = {k: v for k, v in c.items() if not k.endswith(("modes", "modes1", "modes2"))}
    assert [int(r) for r in rets][:len(want["rets"])] == want["rets"], (seed, desc)   # (+ the metadata run's second flush)
    assert list(pcm.shape) == want["shape"], (seed, desc)
    assert F.digest(pcm) == want["sha256"], (seed, desc)
    assert F.meta_digest(md) == want["meta"], (seed, desc, "IAMF_decoder_get_last_metadata rows")


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


class _Variant:
    """the decoder library with every handle it opens switched to one of the reference's other builds"""

    def __init__(self, lib, variant):
        self._lib, self._variant = lib, variant

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name != "IAMF_decoder_open":
            return f
        lib, variant = self._lib, self._variant

        def open_():
            import ctypes as C
            lib.IAMF_decoder_open.restype = C.c_void_p
            lib.iamf_hip_decoder_set_hoa_lfe.argtypes = [C.c_void_p, C.c_int]
            lib.iamf_hip_decoder_set_variant.argtypes = [C.c_void_p, C.c_int]
            d = lib.IAMF_decoder_open()
            assert (lib.iamf_hip_decoder_set_hoa_lfe(d, 1) if variant == "lfe" else lib.iamf_hip_decoder_set_variant(d, 1)) == 0
            return d
        return open_


@pytest.mark.parametrize("variant,seed", [(v, s) for v in ("lfe", "tv", "wide", "multi", "params", "concat", "syntax", "dparams") for s in range(F.VARIANTS[v][1])])
def test_random_stream_matches_the_other_builds_of_the_reference(lib, variant, seed):
    """the same generator against the reference built -DDISABLE_LFE_HOA=0 (scene-based elements three times as likely: one or
    two of them through the LFE generator, beside channel-based ones, behind the resampler) and -DSAMSUNG_TV (its own layout
    -> layout tables, 12-channel PCM stride); and "wide": the default build with what the first generator held fixed —
    twenty scalable layer stacks with random output gains, any layout with demixing info, the demixing defaults, big-endian
    samples, seventeen stream / output rate pairs with and without conversion, units trimmed away completely, frame sizes
    that are not multiples of 4; and "multi": three elements in the stream, two or three mix presentations (ids may repeat) of
    one or two of them, the caller naming one, a wrong one or none (IAMF_decoder_set_mix_presentation_id; the reference's
    matching score over the layouts, IAMF_decoder.c:2997-3111); and "params": mix-gain parameter timelines on the element and
    output gains — definitions of mode 0 or 1, a parameter rate that is or is not the stream's, one to three sub-blocks
    per block, every animation type, blocks missing.  On a few of those the REFERENCE dies of heap corruption (it writes
    past its gains[] behind a STEP sub-block, IAMF_decoder.c:921-960): this library must decode them without a fault
    (the same streams run under ASan in tests/test_facade_malformed.py), there is nothing to compare.  And "concat": two or
    three IA sequences of the wide set back to back on one handle — IAMF_ERR_INVALID_STATE at each new sequence header,
    configured again (IAMF_decoder.c:2918-2921,3796-3806); the reference hands the FIRST sequence's resampler on to the
    later ones (iamf_presentation_take_resampler, :3189-3199: no latency skipped again, the first ratio stays) and dies on
    one stream in nine (that resampler's buffer is sized for the first sequence's frames): those are decoded, not compared.
    And "syntax": streams of the default and wide sets rewritten OBU by OBU — non-minimal leb128 sizes, OBU extension headers,
    reserved OBU types and parameter blocks of unknown ids in between, temporal delimiters dropped, the audio frames of a
    unit in another order, redundant copies of the descriptors in the middle of the data.  Where the reference refuses a
    spelling (a reserved OBU or an extension inside the descriptors: IAMF_ERR_BUFFER_TOO_SMALL from configure), the facade
    must refuse it with the same code.  And "dparams": scalable and demixing-info elements whose demixing and recon-gain
    parameter blocks are missing for 15 % to all of the frames (the decoder goes on with the mode and gains it has)."""
    want = GOLD_V[variant][str(seed)]
    stream, c = F.build(seed, variant)
    dlib = lib if variant in ("wide", "multi", "params", "concat", "syntax", "dparams") else _Variant(lib, variant)
    md = dict(rows=[], owns_anchors=True, strict=False)
    if "error" in want and variant == "syntax":   # the reference refused this spelling: so must the facade, with the same code
        with pytest.raises(AssertionError) as ei:
            decode_stream(dlib, stream, c["layout"], metadata=md, **F.decode_kwargs(c, variant))
        assert str(ei.value) == want["error"], (variant, seed, c.get("syntax"))
        return
    if "crash" in want:   # the reference dies on this stream: nothing to compare; here it is decoded or refused, not fatal
        try:
            decode_stream(dlib, stream, c["layout"], metadata=md, **F.decode_kwargs(c, variant))
        except AssertionError:
            pass
        return
    pcm, rets = decode_stream(dlib, stream, c["layout"], metadata=md, **F.decode_kwargs(c, variant))
    if "error" in want:   # (the reference refused the stream with a decode error after a reconfiguration)
        return
    assert "sha256" in want, want
    desc = {k: v for k, v in c.items() if not k.endswith(("modes", "modes1", "modes2"))}
    assert [int(r) for r in rets][:len(want["rets"])] == want["rets"], (variant, seed, desc)
    assert list(pcm.shape) == want["shape"], (variant, seed, desc)
    assert F.digest(pcm) == want["sha256"], (variant, seed, desc)
    assert F.meta_digest(md) == want["meta"], (variant, seed, desc, "IAMF_decoder_get_last_metadata rows")


@pytest.mark.parametrize("seed", range(0, F.VARIANTS["lfe"][1], 3))
def test_random_lfe_stream_through_a_group_of_handles(lib, seed):
    """every third stream of the LFE set through five out-of-step handles: whole frames reach the generator of every stream
    (trimmed ones too: the filter runs over what is cut), runs of equal trims share a launch"""
    want = GOLD_V["lfe"][str(seed)]
    stream, c = F.build(seed, "lfe")
    case = dict(c, lfe_hoa=True)
    rc, outs = group_decode_all(lib, case, stream, 5, 2, starve=lambda r, i: (r + 2 * i) % 3 == 0 and i % 2 == 1)
    assert rc == 0, (seed, rc)
    for i, (pcm, rets) in enumerate(outs):
        assert [int(r) for r in rets] == want["rets"], (seed, i)
        assert F.digest(pcm) == want["sha256"], (seed, i)


@pytest.mark.parametrize("seed", range(0, F.VARIANTS["wide"][1], 4))
def test_random_wide_stream_through_a_group_of_handles(lib, seed):
    """every fourth stream of the wide set through a group: stacks of layers, units trimmed away (the demixers' and
    down-mixers' states move on, nothing is launched), new rate pairs.  Frame sizes that are not a multiple of 4 are a
    single-handle matter: the group says so at create (IAMF_ERR_INTERNAL from its unpack layout)."""
    want = GOLD_V["wide"][str(seed)]
    stream, c = F.build(seed, "wide")
    rc, outs = group_decode_all(lib, dict(c), stream, 4, 2, starve=lambda r, i: (r + i) % 3 == 0 and i % 2 == 1)
    if c["fs"] & 3:
        assert rc != 0
        return
    assert rc == 0, (seed, rc)
    for i, (pcm, rets) in enumerate(outs):
        assert [int(r) for r in rets] == want["rets"], (seed, i)
        assert F.digest(pcm) == want["sha256"], (seed, i)


# Hunting runs of the four sets below (tools/debug/fuzz_hunt_gen.py --sets ...): IAMF_FUZZ_HUNT=first:count takes the seeds
# first .. first + count - 1 instead of the committed ones, their goldens from tests/golden_tmp/hunt_<set>.json.
_HUNT = os.environ.get("IAMF_FUZZ_HUNT")


def _seeds(n):
    if not _HUNT:
        return range(n)
    first, count = (int(v) for v in _HUNT.split(":"))
    return range(first, first + count)


def _gold(name):
    if _HUNT:
        return json.load(open(os.path.join(os.path.dirname(_G), "golden_tmp", "hunt_%s.json" % name)))
    return json.load(open(os.path.join(_G, "fuzz_%s.json" % name)))


GOLD_B = _gold("blocks")


@pytest.mark.parametrize("seed", _seeds(F.N_BLOCKS))
def test_random_stream_through_the_players_block_loop(lib, seed):
    """the reference player's own loop (iamfplayer.c:529-662) with block buffers from 777 bytes to its 184 320: configure fed
    until it stops answering IAMF_ERR_BUFFER_TOO_SMALL, decode while it consumes something, the rest of a block in front of
    the next one, a block too small for an OBU leaves both stuck at the same call.  The PCM and EVERY call's return value and
    rsize must be the reference's (tests/golden/fuzz_blocks.json: hashes of both)."""
    want = GOLD_B[str(seed)]
    if "crash" in want:
        pytest.skip("the reference dies on this stream")
    variant, vs, block = F.blocks_case(seed)
    stream, c = F.build(vs, variant)
    pcm, events = decode_stream_blocks(lib, stream, c["layout"], block, **F.decode_kwargs(c, variant))
    assert len(events) == want["calls"] and [list(e) for e in events[-2:]] == want["last"], (seed, variant, vs, block, events[-3:], want)
    assert F.events_digest(events) == want["events"], (seed, variant, vs, block)
    assert list(pcm.shape) == want["shape"] and F.digest(pcm) == want["sha256"], (seed, variant, vs, block)


@pytest.mark.parametrize("variant,seed", [(v, s) for v in ("multi", "params") for s in range(0, F.VARIANTS[v][1], 5)])
def test_random_multi_and_params_streams_through_a_group_of_handles(lib, variant, seed):
    """every fifth stream of the sets with several mix presentations and with parameter timelines through four out-of-step
    handles of a group (the presentation is chosen per handle at configure; ramps of some streams of a round and constants
    of others share a launch)"""
    want = GOLD_V[variant][str(seed)]
    if "crash" in want:
        pytest.skip("the reference dies on this stream")
    stream, c = F.build(seed, variant)
    rc, outs = group_decode_all(lib, dict(c), stream, 4, 2, starve=lambda r, i: (r + 3 * i) % 4 == 0 and i % 2 == 0)
    if c["fs"] & 3:
        assert rc != 0
        return
    assert rc == 0, (variant, seed, rc)
    for i, (pcm, rets) in enumerate(outs):
        assert [int(r) for r in rets] == want["rets"], (variant, seed, i)
        assert F.digest(pcm) == want["sha256"], (variant, seed, i)


GOLD_S = _gold("switch")


@pytest.mark.parametrize("seed", _seeds(F.N_SWITCH))
def test_random_tv_stream_with_run_time_layout_switches(lib, seed):
    """the -DSAMSUNG_TV build's run-time layout switch (IAMF_decoder_output_layout_set_* + IAMF_decoder_configure(h, NULL, 0,
    NULL), IAMF_decoder.c:3819-3881) once or twice per stream, after random numbers of delivered frames, to random layouts:
    renderers, limiter and resampler re-opened, decoders, demixers and parameter clocks living on.  PCM and every return
    value (the configure calls' too) against the reference built that way; the streams it dies on are decoded only."""
    import numpy as np
    want = GOLD_S[str(seed)]
    vs, lays, after = F.switch_case(seed)
    stream, c = F.build(vs, "tv")
    try:
        chunks, rets = decode_stream_switching(_Variant(lib, "tv"), stream, lays, after, **F.decode_kwargs(c, "tv"))
    except AssertionError as e:
        assert want.get("error") == str(e), (seed, vs, lays, after, str(e), want)
        return
    if "sha256" not in want:
        return
    pcm = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 12), np.int16)
    got = [list(r) if isinstance(r, tuple) else int(r) for r in rets]
    assert got == want["rets"], (seed, vs, lays, after)
    assert list(pcm.shape) == want["shape"], (seed, vs, lays, after)
    # (a difference is accepted only where the reference's own PCM is undefined: one column, e2e_fuzz.reference_h_slot_23_is_stale;
    #  no committed seed needs it, 8 of 3000 hunted ones do)
    assert F.digest(pcm) == want["sha256"] or F.reference_h_slot_23_is_stale(c, lays), (seed, vs, lays, after)


GOLD_U = _gold("units")


@pytest.mark.parametrize("seed", _seeds(F.N_UNITS))
def test_random_stream_one_temporal_unit_per_call(lib, seed):
    """the reference player's demuxer loop (iamfplayer.c:664-789): the descriptors in one IAMF_decoder_configure call, ONE
    temporal unit per IAMF_decoder_decode call, rsize == NULL in both (include/IAMF_decoder.h:91-95), a flush at the end.
    Streams of four of the sets; every call's return value and the PCM against the reference."""
    want = GOLD_U[str(seed)]
    if "crash" in want:
        pytest.skip("the reference dies on this stream")
    variant, vs, desc, units, c = F.units_case(seed)
    pcm, rets = decode_stream_units(lib, desc, units, c["layout"], **F.decode_kwargs(c, variant))
    assert [int(r) for r in rets] == want["rets"], (seed, variant, vs)
    assert list(pcm.shape) == want["shape"] and F.digest(pcm) == want["sha256"], (seed, variant, vs)


GOLD_G = _gold("gmix")


@pytest.mark.parametrize("seed", _seeds(F.N_GMIX))
def test_group_whose_handles_decode_different_streams(lib, seed):
    """three to six handles of ONE topology, each with a stream of its own — other audio, gains, ramps, demixing modes, recon
    gains, missing blocks, trims (a unit trimmed away in one stream while the others render) — out of step through one group:
    every handle's PCM and return values are the reference's decode of ITS stream.  (The other group tests feed every
    handle the same bytes: a group that used stream 0's records for all would pass them.)"""
    want = GOLD_G[str(seed)]
    if "handles" not in want:
        pytest.skip("the reference dies on one of these streams")
    variant, cases, streams = F.gmix_build(seed)
    n = len(streams)
    rc, outs = group_decode_all(lib, dict(cases[0]), streams, n, 2, starve=lambda r, i: (r + 2 * i) % 5 == 0 and i % 2 == 1)
    if "toa_projection" in cases[0]["pair"]:   # every stream brings a de-mapping matrix of its own: not one topology
        assert rc == -1
        return
    assert rc == 0, (seed, rc)
    for i, (pcm, rets) in enumerate(outs):
        assert [int(r) for r in rets] == want["handles"][i]["rets"], (seed, variant, i)
        assert F.digest(pcm) == want["handles"][i]["sha256"], (seed, variant, i)
