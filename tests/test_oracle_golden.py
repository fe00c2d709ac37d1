"""The CPU restatement (oracle/) against golden vectors produced by the REAL reference
(oracle/gen_golden.py -> tests/golden).  Everything here must be bit-exact: the oracle is only
trusted as a checker for the HIP path because of these tests."""
import numpy as np
import pytest

import oracle_lib as O
import synth


def test_tables_blob_matches_manifest_dims():
    L = O.lib()
    assert L.orc_tables_count() == 196  # 4 orders x 14 layouts + 10 x 14 channel layouts
    h = O.get_h2m(3, O.SS["H"])
    assert (h.m, h.n, h.channels, h.lfe1, h.lfe2) == (16, 22, 24, 3, -1)
    b = O.get_h2m(3, O.SS["BINAURAL"])
    a = O.get_h2m(3, O.SS["A"])
    assert np.array_equal(a.array(), b.array())  # default build: binaural == stereo matrix
    m = O.get_m2m(O.SS["L714"], O.SS["J"])
    assert (m.m, m.n) == (12, 12)


def test_h2m_bit_exact(golden):
    g = golden.npz("h2m")
    for key, meta in golden.manifest.items():
        if not key.startswith("h2m/"):
            continue
        name = key.split("/")[1]
        m = (meta["order"] + 1) ** 2
        x = synth.gaussian(meta["seed"], m, meta["ns"], meta["sigma"])
        mx = O.get_h2m(meta["order"], meta["out_id"])
        y = O.render(mx, x, O.OUT_CH[meta["out_id"]], prefill=meta.get("prefill", 0.0))
        assert y.shape == g[name].shape, name
        assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32)), name


def test_h2m_sentinel_slots(golden):
    s = golden.npz("h2m")["toa_H_sentinel"]
    untouched = [c for c in range(24) if np.all(s[c] == 7.0)]
    zeroed = [c for c in range(24) if np.all(s[c] == 0.0)]
    assert untouched == [23] and zeroed == [3]  # LFE2 slot not reserved for H


def test_m2m_bit_exact(golden):
    g = golden.npz("m2m")
    for key, meta in golden.manifest.items():
        if not key.startswith("m2m/"):
            continue
        name = key.split("/")[1]
        x = synth.uniform(meta["seed"], meta["m"], meta["ns"], meta["amp"])
        mx = O.get_m2m(meta["in_id"], meta["out_id"])
        y = O.render(mx, x, mx.n)
        assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32)), name


def _limiter_input(meta):
    total = sum(meta["sizes"])
    if meta["kind"] == "hot":
        return synth.hot(meta["seed"], meta["ch"], total, sigma=0.25, burst_phase=700, burst_period=6000)
    return synth.quiet(meta["seed"], meta["ch"], total)


def test_limiter_bit_exact(golden):
    g = golden.npz("limiter")
    for key, meta in golden.manifest.items():
        if not key.startswith("limiter/"):
            continue
        name = key.split("/")[1]
        x = _limiter_input(meta)
        y, rets = O.limiter_run(x, meta["sizes"])
        assert rets == list(g[name + "_rets"]), name
        assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32)), name


def test_limiter_hot_case_really_limits(golden):
    meta = golden.manifest["limiter/hot2"]
    x = _limiter_input(meta)
    y = golden.npz("limiter")["hot2"]
    assert np.abs(x).max() > 1.2 and np.abs(y).max() <= 0.8914
    # the delayed input and the output differ on a large share of samples (gain != 1)
    n = y.shape[1]
    assert np.mean(y[:, :n - 240] != x[:, :n - 240]) > 0.3


def test_downmix_bit_exact(golden):
    g = golden.npz("dmx")
    for key, meta in golden.manifest.items():
        if not key.startswith("dmx/") or key.endswith("_invalid"):
            continue
        name = key.split("/")[1]
        sched = [tuple(s) for s in meta["schedule"]]
        x = np.stack([synth.uniform(meta["seed0"] + f, O.LAYOUT_CH[meta["in_layout"]], meta["ns"], 0.5)
                      for f in range(len(sched))])
        y = O.downmix_run(meta["in_layout"], meta["out_layout"], x, sched, meta["default_mode"],
                          meta["default_w"])
        assert y is not None, name
        assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32)), name


def test_demixer_bit_exact(golden):
    """scalable channel audio: oracle/iamf_oracle_demix.c against the real demixer_* (demixer.c)"""
    import demix_cases as D
    gold = golden.npz("demix")
    for name, c in D.STAGE_CASES.items():
        got = D.drive_demixer(O.lib(), "orc_demixer_", c, D.case_input(c))
        assert got.shape == gold[name].shape
        assert np.array_equal(got.view(np.uint32), gold[name].view(np.uint32)), name


def test_downmix_invalid_pairs_refused(golden):
    for a, b in golden.manifest["dmx/_invalid"]:
        assert not O.Downmixer(a, b).ok(), (a, b)


def test_pack_edge_values():
    # clamp, ties-to-even and sign handling of IAMF_decoder.c:100-119
    x = np.array([[0.0, 1.0, -1.0, 2.0, -2.0, 0.5 / 32768, 1.5 / 32768, 2.5 / 32768, -0.5 / 32768,
                   32766.5 / 32768, 0.999999, -0.999999, 1e-9, -1e-9]], dtype=np.float32)
    p = O.pack(x, 16)[:, 0]
    assert list(p[:9]) == [0, 32767, -32768, 32767, -32768, 0, 2, 2, 0]
    p32 = O.pack(x, 32)[:, 0]
    # reference quirk: 2147483647.f rounds to 2^31 in f32, so +full-scale is NOT clamped below
    # 2^31 and lrintf -> (int32_t) wraps it to INT32_MIN (IAMF_decoder.c:114-119)
    assert p32[1] == -2147483648 and p32[2] == -2147483648
    p24 = O.pack(x, 24)[:, 0]
    v = p24[:, 0].astype(np.int32) | (p24[:, 1].astype(np.int32) << 8) | (p24[:, 2].astype(np.int32) << 16)
    v = np.where(v & 0x800000, v - (1 << 24), v)
    assert v[1] == 8388607 and v[2] == -8388608 and v[0] == 0


def test_stream_pipeline_equals_stages():
    mx = O.get_h2m(3, O.SS["BINAURAL"])
    x = synth.hot(77, 16, 5 * 1024, burst_phase=500, burst_period=3000)
    pcm = O.stream_run(mx, 2, x, 1024)
    y = O.render(mx, x, 2)
    z, _ = O.limiter_run(y, [1024] * 5)
    assert np.array_equal(pcm, O.pack(z, 16))
    assert pcm.shape == (5 * 1024, 2)


def test_resampler_bit_exact(golden):
    g = golden.npz("resample")
    for key, meta in golden.manifest.items():
        if not key.startswith("resample/"):
            continue
        name = key.split("/")[1]
        total = sum(meta["sizes"])
        x = synth.hot(meta["seed"], meta["ch"], total, sigma=0.3, burst_amp=1.2, burst_len=60, burst_phase=50,
                      burst_period=700)
        y, rets = O.resample_run(x, meta["in_rate"], meta["out_rate"], meta["sizes"])
        assert rets == list(g[name + "_rets"]), (name, rets, list(g[name + "_rets"]))
        assert np.array_equal(y.view(np.uint32), g[name].view(np.uint32)), name
        assert np.abs(y).max() <= 1.0  # the reference clamps the float output


# ---- HOA LFE generator: goldens from the reference built -DDISABLE_LFE_HOA=0 (oracle/_ref_lfe) ----
def _lfe_manifest():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest_lfe.json")))


def test_lfe_stage_bit_exact(golden):
    import lfe_cases as LC
    g = golden.npz("lfe")
    man = _lfe_manifest()
    for name, (order, oid, rate, sizes, seed) in LC.STAGE.items():
        meta = man["stage/" + name]
        mx = O.get_h2m(order, oid)
        assert (mx.lfe1, mx.lfe2, mx.n) == (meta["lfe1"], meta["lfe2"], meta["n"]), name
        y = O.render_h2m_lfe(mx, LC.stage_input(name), O.OUT_CH[oid], rate, sizes)
        want = g["stage_" + name]
        assert y.shape == want.shape, name
        assert np.array_equal(y.view(np.uint32), want.view(np.uint32)), name
        if mx.lfe1 >= 0:   # the generator really produced a signal there (the default build writes zeros)
            assert np.abs(want[mx.lfe1]).max() > 0.02, name
            if mx.lfe2 >= 0:
                assert np.array_equal(want[mx.lfe2], want[mx.lfe1]), name


def test_lfe_e2e_through_the_oracle_stream(golden):
    """decoder-level goldens (IAMF_decoder_* of the LFE-enabled reference) against the oracle's stream
    pipeline with the generator on"""
    import lfe_cases as LC
    g = golden.npz("lfe")
    for name, c in LC.E2E.items():
        if c.get("second"):   # two elements (one shared filter, a stage per element): the facade against this golden (-m gpu)
            continue
        _, xq = LC.build(name)
        oid = LC.SS[c["ss"]]
        mx = O.get_h2m(c["order"], oid)
        rate = c.get("rate", 48000)
        want = g["e2e_" + name]
        if c.get("projection"):   # the de-mapping stage sits in front: checked on the facade (-m gpu)
            continue
        if rate != 48000:   # resampled to the 48 kHz default: checked on the facade against this golden (-m gpu)
            continue
        if c.get("trims"):
            # the reference renders every frame whole — the generator's filter runs over what is cut — and trims the result
            # (IAMF_decoder.c:3424-3430); then limiter and pack over the kept samples
            fs, F = c["fs"], c["frames"]
            z = O.render_h2m_lfe(mx, xq, O.OUT_CH[oid], rate, [fs] * F)
            keep = [(f * fs + c["trims"].get(f, (0, 0))[0], (f + 1) * fs - c["trims"].get(f, (0, 0))[1]) for f in range(F)]
            z = np.ascontiguousarray(np.concatenate([z[:, a:b] for a, b in keep], axis=1))
            z, _ = O.limiter_run(z, [b - a for a, b in keep])
            y = O.pack(z, c["bit_depth"])
        else:
            y = O.stream_run(mx, O.OUT_CH[oid], xq, c["fs"], bit_depth=c["bit_depth"], lfe_rate=rate)
        assert y.shape == want.shape, (name, y.shape, want.shape)
        assert np.array_equal(y, want), name


def test_tv_table_blob_is_what_the_tv_reference_returns():
    """iac_amd/data/rdr_tables_tv.bin = oracle/dump_tables.c linked against the reference built -DSAMSUNG_TV
    (checked where that build exists: the authoring container); everywhere: same index as the default blob,
    HOA tables identical, 47 layout->layout matrices different"""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(p):
        b = open(p, "rb").read()
        assert b[:8] == b"IARDRTB1"
        n = struct.unpack("<I", b[8:12])[0]
        ents = [struct.unpack("<3I5iI", b[12 + 36 * i:12 + 36 * (i + 1)]) for i in range(n)]
        data = np.frombuffer(b[12 + 36 * n:], dtype=np.float32)
        return {(e[0], e[1], e[2]): (e[3:8], data[e[8]:e[8] + e[6] * e[7]]) for e in ents}

    dflt = load(os.path.join(root, "iac_amd", "data", "rdr_tables.bin"))
    tv = load(os.path.join(root, "iac_amd", "data", "rdr_tables_tv.bin"))
    assert set(dflt) == set(tv) and len(tv) == 196
    diff = [k for k in dflt if dflt[k][0] != tv[k][0] or not np.array_equal(dflt[k][1], tv[k][1])]
    assert all(k[0] == 1 for k in diff) and len(diff) == 47
    tool = os.path.join(root, "oracle", "_ref_tv", "dump_tables_tv")
    if os.path.exists(tool) and os.path.isdir("/root/reference"):
        out = os.path.join("/tmp", "rdr_tables_tv_check.bin")
        subprocess.check_call([tool, out], stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == open(os.path.join(root, "iac_amd", "data", "rdr_tables_tv.bin"), "rb").read()
