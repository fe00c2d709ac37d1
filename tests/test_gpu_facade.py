"""-m gpu: the IAMF_decoder.h facade of libiamf_hip.so (OBU parsing + LPCM on the host, rendering
on the GPU) on the end-to-end streams, against PCM the REAL reference decoder produced from the same
bytes.  `stereo_A_s16` is BASELINE.json configs[0] (iamfplayer -o2 -s0)."""
import ctypes as C

import numpy as np
import pytest

import e2e_cases
from decoder_driver import decode_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available()
    import iac_amd
    return C.CDLL(iac_amd.lib_path())


@pytest.mark.parametrize("name", sorted(e2e_cases.CASES))
def test_facade_matches_reference_decoder(lib, golden, name):
    case = e2e_cases.CASES[name]
    stream, _ = e2e_cases.build(name)
    pcm, rets = decode_stream(lib, stream, case["layout"], bit_depth=case.get("bit_depth", 16),
                              out_rate=case.get("out_rate", 0), loudness=case.get("loudness", 0.0),
                              limiter=case.get("limiter", True), threshold=case.get("threshold", -1.0))
    want = golden.npz("e2e")[name]
    assert list(rets) == list(golden.npz("e2e")[name + "_rets"]), name
    assert pcm.shape == want.shape
    assert np.array_equal(pcm, want), name


@pytest.mark.parametrize("name", [n for n in sorted(e2e_cases.CASES) if "binaural" in n or n.startswith(("foa", "soa", "toa"))])
def test_facade_host_unpack_form_of_the_fused_streams(lib, golden, name, monkeypatch):
    """Streams of one mono-coded ambisonics element (16-bit, <= 2 output channels) hand their packets straight to the render
    kernel (iamf_hip_batch_render_lpcm, the fused LPCM form: what test_facade_matches_reference_decoder runs for them);
    IAMF_HIP_FACADE_UNPACK=1 keeps the host unpacker + f32 kernels for them as for every other stream.  Same goldens."""
    monkeypatch.setenv("IAMF_HIP_FACADE_UNPACK", "1")
    case = e2e_cases.CASES[name]
    stream, _ = e2e_cases.build(name)
    pcm, rets = decode_stream(lib, stream, case["layout"], bit_depth=case.get("bit_depth", 16),
                              out_rate=case.get("out_rate", 0), loudness=case.get("loudness", 0.0),
                              limiter=case.get("limiter", True), threshold=case.get("threshold", -1.0))
    want = golden.npz("e2e")[name]
    assert list(rets) == list(golden.npz("e2e")[name + "_rets"]), name
    assert pcm.shape == want.shape and np.array_equal(pcm, want), name


@pytest.mark.parametrize("name", sorted(e2e_cases.META_CASES))
def test_get_last_metadata_matches_the_reference(lib, golden, name):
    """IAMF_decoder_get_last_metadata, one of the 19 boundary functions, had no test.  Rows recorded from
    the REAL reference (oracle/gen_golden_extra.py: generate_meta) after configure, after every delivered frame and after
    each of two flush calls: pts (another time base, a start offset, a clock re-based mid-stream, trims, the resampler's
    rates: IAMF_decoder.c:3410-3415,3521-3522,4150-4160), sound system and sound mode (:241-254,1555-1578), the
    presentation's loudness records incl. true peak and anchored loudness (:3632-3645), the DEMIXING record and the mode
    each frame used (:3436-3442,3647-3662).  The PCM of the same run is checked once more on the way."""
    case, mc = e2e_cases.CASES[name], e2e_cases.META_CASES[name]
    stream, _ = e2e_cases.build(name)
    md = dict(rows=[], owns_anchors=True, **{k: v for k, v in mc.items() if k != "pts"})
    pcm, rets = decode_stream(lib, stream, case["layout"], bit_depth=case.get("bit_depth", 16), out_rate=case.get("out_rate", 0),
                              loudness=case.get("loudness", 0.0), limiter=case.get("limiter", True),
                              threshold=case.get("threshold", -1.0), metadata=md, pts=mc["pts"])
    want = golden.npz("meta")[name]
    got = np.array(md["rows"], dtype=np.int64)
    assert got.shape == want.shape, (got.shape, want.shape)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (name, "row %d" % i, [int(v) for v in g if v != -9999], [int(v) for v in w if v != -9999])
    assert np.array_equal(pcm, golden.npz("e2e")[name])


def test_metadata_before_configure_and_bad_arguments(lib):
    from decoder_driver import _Extradata
    lib.IAMF_decoder_open.restype = C.c_void_p
    lib.IAMF_decoder_close.argtypes = [C.c_void_p]
    lib.IAMF_decoder_get_last_metadata.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(_Extradata)]
    d = lib.IAMF_decoder_open()
    pts, m = C.c_int64(-1), _Extradata()
    assert lib.IAMF_decoder_get_last_metadata(d, None, C.byref(m)) == -1
    assert lib.IAMF_decoder_get_last_metadata(d, C.byref(pts), None) == -1
    assert lib.IAMF_decoder_get_last_metadata(None, C.byref(pts), C.byref(m)) == -1
    # an open handle that has not been configured: the zeroed context, nothing to free (what the reference returns:
    # [0, 0, 0, 0, 0, 0, 0, 0] through decoder_driver.last_metadata, and the pts of IAMF_decoder_set_pts alone)
    assert lib.IAMF_decoder_get_last_metadata(d, C.byref(pts), C.byref(m)) == 0
    assert (pts.value, m.output_sound_system, m.output_sound_mode, m.num_loudness_layouts, m.num_parameters) == (0, 0, 0, 0, 0)
    lib.IAMF_decoder_set_pts.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
    lib.IAMF_decoder_set_pts(d, 55, 1000)
    assert lib.IAMF_decoder_get_last_metadata(d, C.byref(pts), C.byref(m)) == 0 and pts.value == 55
    lib.IAMF_decoder_close(d)


def test_facade_api_surface(lib):
    lib.IAMF_decoder_open.restype = C.c_void_p
    lib.IAMF_decoder_close.argtypes = [C.c_void_p]
    lib.IAMF_decoder_get_codec_capability.restype = C.c_void_p
    lib.IAMF_decoder_peak_limiter_get_threshold.restype = C.c_float
    lib.IAMF_decoder_peak_limiter_get_threshold.argtypes = [C.c_void_p]
    lib.IAMF_decoder_set_sampling_rate.argtypes = [C.c_void_p, C.c_uint32]
    lib.IAMF_decoder_set_bit_depth.argtypes = [C.c_void_p, C.c_uint32]
    assert [lib.IAMF_layout_sound_system_channels_count(i) for i in range(13)] == [2, 6, 8, 10, 11, 12, 14, 24, 8, 12, 10, 6, 1]
    assert lib.IAMF_layout_binaural_channels_count() == 2
    d = lib.IAMF_decoder_open()
    assert abs(lib.IAMF_decoder_peak_limiter_get_threshold(d) + 1.0) < 1e-7  # default -1 dBFS
    assert lib.IAMF_decoder_set_sampling_rate(d, 44100) == 0
    assert lib.IAMF_decoder_set_sampling_rate(d, 22050) == -1  # not in the reference's whitelist
    assert lib.IAMF_decoder_set_bit_depth(d, 20) == -1
    cap = C.cast(lib.IAMF_decoder_get_codec_capability(), C.c_char_p).value
    assert b"ipcm" in cap
    lib.IAMF_decoder_close(d)


def test_player_cli_writes_the_reference_wav_payload(golden, tmp_path):
    """BASELINE configs[0] end to end: `iamfplayer -o2 -s0 stereo.iamf` -> WAV, here with the player
    built on libiamf_hip.so; the PCM payload must equal what the reference decoder produced."""
    import os
    import subprocess

    import iac_amd
    exe = os.path.join(os.path.dirname(iac_amd.lib_path()), "iamfplayer_hip")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    stream, _ = e2e_cases.build("stereo_A_s16")
    src = tmp_path / "stereo.iamf"
    src.write_bytes(stream)
    out = tmp_path / "ss0_stereo.wav"
    r = subprocess.run([exe, "-o2", "-s0", "-out", str(out), str(src)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Get 21 frames" in r.stdout  # 20 frames + the flushed tail, like the reference player
    raw = out.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt "
    pcm = np.frombuffer(raw[44:], dtype=np.int16).reshape(-1, 2)
    assert np.array_equal(pcm, golden.npz("e2e")["stereo_A_s16"])
