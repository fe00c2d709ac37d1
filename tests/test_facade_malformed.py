"""Malformed and hostile bitstreams against the decoder facade's host side, on the CPU build under
AddressSanitizer + UBSan: tests/facade_stub/ links iamf_decoder_facade.c against a
malloc-backed stand-in of the device layer, so every host-side length derived from an untrusted field
is checked by ASan.  Also the reconfiguration protocol (two IA sequences in one file)."""
import os
import subprocess

import numpy as np
import pytest

import iamf_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "facade_stub")
BIN = os.path.join(STUB, "build", "facade_driver_asan")


@pytest.fixture(scope="module")
def driver():
    os.makedirs(os.path.join(STUB, "build"), exist_ok=True)
    srcs = [os.path.join(ROOT, "iac_amd", "csrc", "iamf_decoder_facade.c"),
            os.path.join(STUB, "device_stub.c"), os.path.join(STUB, "facade_driver.c")]
    if not os.path.exists(BIN) or any(os.path.getmtime(s) > os.path.getmtime(BIN) for s in srcs):
        subprocess.check_call(["gcc", "-g", "-O1", "-std=gnu11", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                               "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include"] + srcs +
                              ["-lm", "-o", BIN])
    return BIN


def run(driver, tmp_path, stream, layout="0", bits=16):
    p = os.path.join(str(tmp_path), "s.iamf")
    open(p, "wb").write(stream)
    r = subprocess.run([driver, p, layout, str(bits)], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    return r.stdout.strip().splitlines()


FS = 64


def descriptors(element=None, fs=FS, rate=48000):
    pd = lambda pid: W.param_definition(pid, 48000, mode=1)
    s = W.sequence_header(1) + W.codec_config_lpcm(0, fs, 16, rate)
    s += element if element is not None else W.audio_element_channel(1, 0, 1, [0])
    s += W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=0)], dict(pdef=pd(101), default_q78=0), [("ss", 0)])
    return s


def stereo_frame(n=FS, trim=None, seed=0):
    x = (np.random.default_rng(seed).uniform(-0.5, 0.5, (2, n))).astype(np.float32)
    payload = W.lpcm_bytes(W.quantize(x, 16), 16)
    return W.temporal_delimiter() + W.obu(W.OBU_AUDIO_FRAME_ID0, payload, trim=trim)


def test_well_formed_stream_through_the_stub(driver, tmp_path):
    out = run(driver, tmp_path, descriptors() + b"".join(stereo_frame(seed=i) for i in range(8)))
    assert out[0].startswith("configure 0")
    assert out[-1] == "total %d configs 1" % (8 * FS)


@pytest.mark.parametrize("trim", [(0xFFFFFFFF, 0), (0, 0xFFFFFFFF), (0x7FFFFFFF, 0x7FFFFFFF), (FS - 1, 2),
                                  (1 << 40, 0), (0x80000000, 0x80000000), (FS + 1, 0)])
def test_hostile_trims_are_refused_not_applied(driver, tmp_path, trim):
    s = descriptors() + stereo_frame(seed=1) + stereo_frame(trim=trim, seed=2) + stereo_frame(seed=3)
    out = run(driver, tmp_path, s)
    decodes = [int(l.split()[1]) for l in out if l.startswith("decode")]
    assert decodes[1] == -1                      # IAMF_ERR_BAD_ARG, as iamf_frame_trim (IAMF_decoder.c:1364-1370)
    assert decodes[0] >= 0 and decodes[2] >= 0   # and the stream goes on:
    assert out[-1] == "total %d configs 1" % (2 * FS)   # the two good frames come out (limiter delay via flush)


@pytest.mark.parametrize("trim,kept", [((0, 0), FS), ((3, 0), FS - 3), ((0, 5), FS - 5), ((7, 9), FS - 16),
                                       ((FS, 0), 0), ((0, FS), 0), ((FS // 2, FS // 2), 0)])
def test_valid_trims_still_work(driver, tmp_path, trim, kept):
    s = descriptors() + stereo_frame(seed=1) + stereo_frame(trim=trim if any(trim) else None, seed=2)
    out = run(driver, tmp_path, s)
    assert out[-1] == "total %d configs 1" % (FS + kept)


def _element_raw(nsub, body):
    """an audio element OBU with an arbitrary sub-stream count and type-specific tail"""
    p = W.leb128(1) + bytes([0 << 5]) + W.leb128(0) + W.leb128(nsub)
    p += b"".join(W.leb128(i) for i in range(min(nsub, 64))) + W.leb128(0) + body
    return W.obu(W.OBU_AUDIO_ELEMENT, p)


def test_elements_without_substreams_are_rejected(driver, tmp_path):
    # channel-based, one mono layer claiming 0 sub-streams / 1 coupled: tu_complete() would be trivially true
    el = _element_raw(0, bytes([1 << 5, (0 << 4), 0, 1]))
    out = run(driver, tmp_path, descriptors(el) + stereo_frame())
    assert out[0].startswith("configure -") and "configs 0" in out[-1]
    # scene-based mono mapping with substream_count 0
    p = W.leb128(1) + bytes([1 << 5]) + W.leb128(0) + W.leb128(0) + W.leb128(0) + W.leb128(0) + bytes([4, 0, 0, 0, 0, 0])
    out = run(driver, tmp_path, descriptors(W.obu(W.OBU_AUDIO_ELEMENT, p)) + stereo_frame(), layout="b")
    assert out[0].startswith("configure -")
    # a sub-stream count that is negative once narrowed to int
    el = _element_raw(0xFFFFFFFF, bytes([1 << 5, (1 << 4), 1, 1]))
    out = run(driver, tmp_path, descriptors(el) + stereo_frame())
    assert out[0].startswith("configure -")
    # a layer with more coupled sub-streams than sub-streams
    el = _element_raw(1, bytes([1 << 5, (1 << 4), 1, 3]))
    out = run(driver, tmp_path, descriptors(el) + stereo_frame())
    assert out[0].startswith("configure -")


def test_absurd_frame_size_is_rejected(driver, tmp_path):
    out = run(driver, tmp_path, descriptors(fs=(1 << 31) + 5) + stereo_frame())
    assert out[0].startswith("configure -")


def test_truncated_and_bit_flipped_streams_never_fault(driver, tmp_path):
    good = descriptors() + b"".join(stereo_frame(seed=i, trim=(i, 0) if i % 3 == 0 and i else None) for i in range(6))
    rng = np.random.default_rng(11)
    for cut in list(range(1, len(good), 17)):
        run(driver, tmp_path, good[:cut])
    for _ in range(150):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        run(driver, tmp_path, bytes(b))


def test_scene_and_scalable_descriptors_fuzz(driver, tmp_path):
    """the e2e streams with the richest descriptors (projection matrix, three-layer scalable element,
    parameter blocks with ramps / demixing / recon gain), intact and with bytes flipped in descriptors,
    parameter blocks and frame headers"""
    import e2e_cases as E
    rng = np.random.default_rng(5)
    for n in ["toa_projection_B_s16", "scalable_J_s16", "scalable_312_dmx_s16", "l714_J_ramps", "l714_C_dmx",
              "two_elements_A_s32", "stereo_441_to_48k", "scalable_plus_scalable_J", "scalable_plus_l714dmx_312",
              "projection_plus_scalable_C"]:
        case = E.CASES[n]
        stream, _ = E.build(n)
        lay = "b" if case["layout"][0] == "binaural" else str(case["layout"][1])
        out = run(driver, tmp_path, stream, layout=lay, bits=case["bit_depth"])
        assert out[0].startswith("configure 0"), (n, out[:3])
        head = len(stream) - case["frames"] * (len(stream) // (case["frames"] + 1))   # roughly the descriptors
        for _ in range(30):
            b = bytearray(stream)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, max(64, min(len(b), head + 4096))))] = int(rng.integers(0, 256))
            run(driver, tmp_path, bytes(b), layout=lay, bits=case["bit_depth"])


def test_two_concatenated_sequences_reconfigure_cleanly(driver, tmp_path):
    """decode meets a new IA sequence header -> IAMF_ERR_INVALID_STATE until the caller configures
    again; the second configuration starts from a clean database (IAMF_decoder.c:2918-2921,3796-3806)"""
    seq1 = descriptors() + b"".join(stereo_frame(seed=i) for i in range(3))
    seq2 = descriptors(fs=2 * FS) + b"".join(stereo_frame(n=2 * FS, seed=i) for i in range(2))
    out = run(driver, tmp_path, seq1 + seq2)
    assert [l.split()[1] for l in out if l.startswith("configure")] == ["0", "0"]
    assert sum(l.startswith("decode -5 ") for l in out) == 1          # IAMF_ERR_INVALID_STATE, once
    # sequence 1: 3 frames minus the limiter's 240-sample delay (its tail is lost at the reconfiguration,
    # as in the reference); sequence 2: both frames of the new size, delay re-emitted by the final flush
    assert out[-1] == "total %d configs 2" % (3 * FS - min(240, 3 * FS) + 2 * 2 * FS)


# ---- the group of handles (iamf_decoder_group.inc) under the same sanitizers ----
GBIN = os.path.join(STUB, "build", "group_driver")


@pytest.fixture(scope="module")
def group_driver():
    os.makedirs(os.path.join(STUB, "build"), exist_ok=True)
    srcs = [os.path.join(ROOT, "iac_amd", "csrc", "iamf_decoder_facade.c"), os.path.join(ROOT, "iac_amd", "csrc", "iamf_decoder_group.inc"),
            os.path.join(STUB, "device_stub.c"), os.path.join(STUB, "group_driver.c")]
    if not os.path.exists(GBIN) or any(os.path.getmtime(x) > os.path.getmtime(GBIN) for x in srcs):
        subprocess.check_call(["gcc", "-g", "-O1", "-std=gnu11", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                               "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include"] +
                              [x for x in srcs if x.endswith(".c")] + ["-lm", "-lpthread", "-o", GBIN])
    return GBIN


def run_group(gdriver, tmp_path, stream, n, threads, layout="0", bits=16):
    p = os.path.join(str(tmp_path), "g.iamf")
    open(p, "wb").write(stream)
    r = subprocess.run([gdriver, p, layout, str(bits), str(n), str(threads)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    return r.stdout.strip().splitlines()


# (8 / 9 / 17 / 70 handles: whole upload chunks, a chunk of one handle, chunks of nine — iamf_decoder_group.inc cuts a round
#  into at most eight chunks of at least eight handles; 1 thread = no pool, the calling thread parses and uploads)
@pytest.mark.parametrize("n,threads", [(1, 1), (5, 3), (13, 0), (8, 2), (9, 1), (17, 5), (70, 4)])
def test_group_of_handles_out_of_step_matches_the_single_handle_totals(driver, group_driver, tmp_path, n, threads):
    """N handles on one stream, starved in different rounds and finishing at different times: every handle must emit
    what a handle alone emits (here: the sample count; the PCM itself is checked on the GPU, tests/test_gpu_group.py)"""
    frames = [stereo_frame(seed=1), stereo_frame(trim=(3, 0), seed=2), stereo_frame(seed=3), stereo_frame(trim=(0, 5), seed=4),
              stereo_frame(seed=5), stereo_frame(trim=(FS, 0), seed=6), stereo_frame(seed=7)]
    s = descriptors() + b"".join(frames)
    single = run(driver, tmp_path, s)
    want = int(single[-1].split()[1])
    out = run_group(group_driver, tmp_path, s, n, threads)
    assert "group_create 0" in out
    # a grouped handle refuses IAMF_decoder_decode / _close (IAMF_ERR_INVALID_STATE = -5)
    assert "single_decode_while_grouped -5 close -5" in out
    totals = [int(l.split()[2]) for l in out if l.startswith("total h")]
    assert totals == [want] * n, (totals, want)


def test_group_with_a_hostile_trim_goes_on_like_the_single_handle(driver, group_driver, tmp_path):
    s = descriptors() + stereo_frame(seed=1) + stereo_frame(trim=(1 << 40, 0), seed=2) + stereo_frame(seed=3)
    want = int(run(driver, tmp_path, s)[-1].split()[1])
    out = run_group(group_driver, tmp_path, s, 4, 2)
    assert [int(l.split()[2]) for l in out if l.startswith("total h")] == [want] * 4
    assert "decode -1 rsize" in "\n".join(out)   # handle 0 reported IAMF_ERR_BAD_ARG for that unit


def test_group_refuses_handles_that_differ_in_lpcm_byte_order(group_driver, tmp_path):
    """the batch's one unpack layout is built from handle 0, so handles that differ in the codec
    config's sample_format_flags (pcm/IAMF_pcm_decoder.c:60-62) must not share a group: IAMF_ERR_BAD_ARG, as the
    header promises for differing codec configuration."""
    pd = lambda pid: W.param_definition(pid, 48000, mode=1)

    def desc(le):
        s = W.sequence_header(1) + W.codec_config_lpcm(0, FS, 16, 48000, little_endian=le)
        s += W.audio_element_channel(1, 0, 1, [0])
        return s + W.mix_presentation(1, [dict(eid=1, pdef=pd(100), default_q78=0)], dict(pdef=pd(101), default_q78=0), [("ss", 0)])

    le = desc(True) + stereo_frame(seed=1)
    be = desc(False) + stereo_frame(seed=1)
    p1, p2 = os.path.join(str(tmp_path), "le.iamf"), os.path.join(str(tmp_path), "be.iamf")
    open(p1, "wb").write(le)
    open(p2, "wb").write(be)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([group_driver, p1, "0", "16", "4", "2", p2], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    assert "group_create -1" in r.stdout.splitlines(), r.stdout
    # the same four handles from ONE stream form a group
    r = subprocess.run([group_driver, p1, "0", "16", "4", "2", p1], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "group_create 0" in r.stdout.splitlines(), (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("n,threads", [(1, 1), (5, 3), (9, 2)])
def test_group_of_resampling_handles_matches_the_single_handle_totals(driver, group_driver, tmp_path, n, threads):
    """round 4: handles whose stream rate differs from the output rate (IAMF_decoder.c:3193-3199) form a group — three stages
    per launch run (render -> resample -> limiter / pack).  Here: the host side under ASan / UBSan against the stub, handles
    out of step; the PCM itself is checked on the GPU (tests/test_gpu_group.py)."""
    frames = [stereo_frame(seed=1), stereo_frame(trim=(3, 0), seed=2), stereo_frame(seed=3), stereo_frame(trim=(0, 5), seed=4),
              stereo_frame(seed=5)]
    s = descriptors(rate=44100) + b"".join(frames)
    want = int(run(driver, tmp_path, s)[-1].split()[1])
    out = run_group(group_driver, tmp_path, s, n, threads)
    assert "group_create 0" in out
    totals = [int(l.split()[2]) for l in out if l.startswith("total h")]
    assert totals == [want] * n, (totals, want)


@pytest.mark.parametrize("name,n,threads", [("scalable_plus_scalable_J", 5, 2), ("l714dmx_plus_l714dmx_C", 9, 3),
                                            ("projection_plus_projection_binaural", 3, 1)])
def test_group_of_handles_with_a_batch_per_element(driver, group_driver, tmp_path, name, n, threads):
    """round 4: presentations BOTH of whose elements need a per-stream stage (demixer / down-mixer / projection): element 1
    goes through a batch of its own, its frame is transposed on the device and mixed in as a second element — single
    handles and groups.  Here: the host side (two sets of stage records, the second batch's rows, the planar hand-over
    buffer) under ASan / UBSan against the stub, handles out of step; PCM on the GPU (tests/test_gpu_group.py)."""
    import e2e_cases as E
    case = E.CASES[name]
    stream, _ = E.build(name)
    lay = "b" if case["layout"][0] == "binaural" else str(case["layout"][1])
    want = int(run(driver, tmp_path, stream, layout=lay, bits=case["bit_depth"])[-1].split()[1])
    assert want == case["frames"] * case["fs"]
    out = run_group(group_driver, tmp_path, stream, n, threads, layout=lay, bits=case["bit_depth"])
    assert "group_create 0" in out
    assert [int(l.split()[2]) for l in out if l.startswith("total h")] == [want] * n


def test_parameter_timelines_the_reference_dies_on(driver, tmp_path):
    """tests/e2e_fuzz.py "params": random mix-gain parameter timelines.  Behind a STEP sub-block that fills part of a frame
    the reference still takes the whole frame for the animated sub-block that follows and writes past its gains[duration]
    (IAMF_decoder.c:921-960) — heap corruption, on a few of these streams fatal.  The facade's mirror of that function once
    did the same to its pinned ramp buffer; here the first 120 streams of the set run under ASan / UBSan."""
    import e2e_fuzz as F
    for seed in range(120):
        stream, c = F.build(seed, "params")
        lay = "b" if c["layout"][0] == "binaural" else str(c["layout"][1])
        out = run(driver, tmp_path, stream, layout=lay, bits=c["bit_depth"])
        assert out[0].startswith("configure 0"), (seed, out[:2])


@pytest.mark.parametrize("variant", ["wide", "multi", "concat", "syntax", "dparams"])
def test_fuzz_streams_under_the_sanitizers(driver, tmp_path, variant):
    """the host side of the facade (parser, parameter timelines, selection, reconfiguration, the two-batch pipelines) on 50
    streams of each of the richer fuzz sets under ASan / UBSan against the device stand-ins: whatever the stream, no fault
    (PCM and return values are compared on the GPU, tests/test_gpu_fuzz_facade.py)"""
    import e2e_fuzz as F
    for seed in range(50):
        stream, c = F.build(seed, variant)
        lay = "b" if c["layout"][0] == "binaural" else str(c["layout"][1])
        p = os.path.join(str(tmp_path), "s.iamf")
        open(p, "wb").write(stream)
        r = subprocess.run([driver, p, lay, str(c["bit_depth"])], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"))
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (variant, seed, r.stderr[-2500:])


def test_layer_output_gain_lists_are_bounded(driver, tmp_path):
    """tests/e2e_fuzz.py 'wide' seed 7214: a scalable element of four layers (stereo, 3.1.2, 5.1.2, 7.1.4) whose output-gain
    flags (43, 63, 27, 27) map to exactly 12 channels with two more flag bits behind them — the reference stores the 13th
    mapped value before testing it (IAMF_decoder.c:2371: chs[12] on a 12-entry stack array) and zeroes its first gain.  The
    facade's own list had no bound either until this was looked into; now entries naming a channel no layer carries are
    dropped, more than 12 of the rest refused.  This stream must configure and decode under ASan / UBSan; so must six
    layers with every flag set (24 mapped entries: 10 name decoded channels)."""
    import e2e_cases as E
    import e2e_fuzz as F
    stream, c = F.build(7214, "wide")
    assert F.reference_gain_list_overflows(c)
    out = run(driver, tmp_path, stream, layout=str(c["layout"][1]), bits=c["bit_depth"])
    assert out[0].startswith("configure 0"), out[:3]
    c6 = dict(c, pair=("scalable",), scalable_layers1=[0, 1, 8, 3, 4, 7], scalable_gains1={i: (63, -100 * i) for i in range(6)})
    c6.pop("trims", None)
    assert F.reference_gain_list_overflows(c6)
    E.CASES["six_layers_tmp"] = c6
    try:
        stream6 = E.build("six_layers_tmp")[0]
    finally:
        del E.CASES["six_layers_tmp"]
    out = run(driver, tmp_path, stream6, layout=str(c["layout"][1]), bits=c["bit_depth"])
    assert out[0].startswith("configure 0"), out[:3]
