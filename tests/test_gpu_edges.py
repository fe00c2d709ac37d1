"""-m gpu: edge cases and error behaviour of the C ABI: empty / ragged / maximum sizes, other
limiter settings and sample rates, state reset, and the error codes (same numeric values as the
reference's IAMF_ERR_*)."""
import ctypes as C

import numpy as np
import pytest

import e2e_cases
import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available()
    import iac_amd as A
    import gpu_util as G
    return A, G, torch


def test_empty_call_and_flush_only(hip):
    A, G, torch = hip
    b = A.Batch(2, A.get_h2m_matrix(3, A.SS["A"]), 2)
    pcm = torch.zeros((2, 4096), dtype=torch.uint8, device="cuda")
    x = torch.zeros((2, 16 * 1024), dtype=torch.float32, device="cuda")
    assert b.render(x.data_ptr(), 16 * 1024, 16 * 1024, 0, pcm.data_ptr(), 4096) == 0  # n_frames = 0
    assert b.flush(pcm.data_ptr(), 4096) == 0  # 240 zeros in, the 240-sample pad swallows them all
    with pytest.raises(A.IamfHipError) as e:
        b.render(x.data_ptr(), 16 * 1024, 16 * 1024, 1, pcm.data_ptr(), 4096)
    assert e.value.code == -5  # IAMF_ERR_INVALID_STATE after the flush
    b.reset()
    assert b.render(x.data_ptr(), 16 * 1024, 16 * 1024, 1, pcm.data_ptr(), 4096) == 1024 - 240
    b.close()


def test_error_codes(hip):
    A, G, torch = hip
    mx = A.get_h2m_matrix(3, A.SS["A"])
    with pytest.raises(A.IamfHipError) as e:
        A.Batch(0, mx, 2)
    assert e.value.code == -1
    with pytest.raises(A.IamfHipError) as e:
        A.Batch(1, mx, 2, out_format=20)
    assert e.value.code == -1
    with pytest.raises(A.IamfHipError) as e:
        A.Batch(1, mx, 25)
    assert e.value.code == -1
    odd = G.identity_matrix(3)  # 3 input channels: no kernel instantiation
    b = A.Batch(1, odd, 3)
    x = torch.zeros((3 * 1024,), dtype=torch.float32, device="cuda")
    pcm = torch.zeros((1024 * 3 * 2,), dtype=torch.uint8, device="cuda")
    with pytest.raises(A.IamfHipError) as e:
        b.render(x.data_ptr(), 3 * 1024, 3 * 1024, 1, pcm.data_ptr(), pcm.numel())
    assert e.value.code == -6  # IAMF_ERR_UNIMPLEMENTED
    b.close()
    b = A.Batch(2, mx, 2)
    x = torch.zeros((2, 16 * 1024), dtype=torch.float32, device="cuda")
    pcm = torch.zeros((2, 100), dtype=torch.uint8, device="cuda")
    with pytest.raises(A.IamfHipError) as e:
        b.render(x.data_ptr(), 16 * 1024, 16 * 1024, 1, pcm.data_ptr(), 100)
    assert e.value.code == -2  # IAMF_ERR_BUFFER_TOO_SMALL
    with pytest.raises(A.IamfHipError) as e:
        b.render(0, 16 * 1024, 16 * 1024, 1, pcm.data_ptr(), 100)
    assert e.value.code == -1
    b.close()


@pytest.mark.parametrize("fs,F,calls", [(6144, 2, [1, 1]), (2048, 3, [3]), (64, 40, [7, 33]), (17, 30, [30])])
def test_frame_size_extremes(hip, fs, F, calls):
    """max_frame_size of the reference (6144), a small aligned one, and an odd one (generic kernel)"""
    A, G, torch = hip
    x = synth.hot(300 + fs, 16, F * fs, sigma=0.2, burst_phase=100, burst_period=1500)[None]
    got = G.hip_render(A.get_h2m_matrix(3, A.SS["BINAURAL"]), 2, x, frame_size=fs, frames_per_call=calls)[0]
    want = O.stream_run(O.get_h2m(3, O.SS["BINAURAL"]), 2, x[0], fs, max_ns=max(fs, 6144))
    assert np.array_equal(got, want)


def test_widest_layouts_24_in_24_out(hip):
    A, G, torch = hip
    fs, F = 1024, 2
    x = synth.hot(9, 24, F * fs, sigma=0.25, burst_phase=100, burst_period=900)[None]
    got = G.hip_render(G.identity_matrix(24), 24, x, frame_size=fs)[0]
    z, _ = O.limiter_run(x[0], [fs] * F)
    assert np.array_equal(got, O.pack(z, 16))


@pytest.mark.parametrize("thr_db,rate", [(-3.0, 48000), (0.0, 48000), (-1.0, 44100), (-6.0, 96000), (-1.0, 16000)])
def test_limiter_settings_and_rates(hip, thr_db, rate):
    """threshold and sample rate change the limiter's constants and its table (96 kHz: 19 298 entries;
    the kernels stage the window of it a chunk can reach, so the size does not matter)"""
    A, G, torch = hip
    fs, F = 1024, 6
    x = synth.hot(1234, 16, F * fs, sigma=0.25, burst_phase=600, burst_period=2500)[None]
    got = G.hip_render(A.get_h2m_matrix(3, A.SS["BINAURAL"]), 2, x, frame_size=fs, threshold_db=thr_db,
                       sample_rate=rate)[0]
    want = O.stream_run(O.get_h2m(3, O.SS["BINAURAL"]), 2, x[0], fs, thr_db=thr_db, rate=rate)
    assert np.array_equal(got, want)


def test_many_small_streams_and_reset_idempotence(hip):
    A, G, torch = hip
    S, fs, F = 4096, 64, 8
    base = np.stack([synth.hot(50 + s, 2, F * fs, sigma=0.3, burst_phase=20 * s, burst_period=300) for s in range(4)])
    x = base[np.arange(S) % 4]
    mx = A.get_m2m_matrix(A.SS["STEREO"], A.SS["A"])
    got = G.hip_render(mx, 2, x, frame_size=fs)
    omx = O.get_m2m(O.SS["STEREO"], O.SS["A"])
    want = [O.stream_run(omx, 2, base[s], fs) for s in range(4)]
    for s in range(S):
        assert np.array_equal(got[s], want[s % 4]), s
    # reset: a batch that already rendered something gives the same PCM again after reset
    b = A.Batch(1, mx, 2, frame_size=fs)
    xin = torch.from_numpy(G.to_frames(base[:1], fs)).cuda()
    outs = []
    for _ in range(2):
        pcm = torch.zeros((F * fs * 4,), dtype=torch.uint8, device="cuda")
        n = b.render(xin.data_ptr(), F * 2 * fs, 2 * fs, F, pcm.data_ptr(), pcm.numel())
        torch.cuda.synchronize()
        outs.append(pcm.cpu().numpy()[:n * 4].copy())
        b.reset()
    b.close()
    assert np.array_equal(outs[0], outs[1])


def test_facade_error_behaviour(hip):
    A, G, torch = hip
    lib = C.CDLL(A.lib_path())
    lib.IAMF_decoder_open.restype = C.c_void_p
    for f in ("IAMF_decoder_close", "IAMF_decoder_output_layout_set_binaural"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.IAMF_decoder_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.IAMF_decoder_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_uint32), C.c_void_p]
    lib.IAMF_decoder_set_bit_depth.argtypes = [C.c_void_p, C.c_uint32]
    stream, _ = e2e_cases.build("toa_binaural_s16")
    pcm = C.create_string_buffer(1 << 16)
    rs = C.c_uint32(0)

    d = lib.IAMF_decoder_open()
    assert lib.IAMF_decoder_decode(d, stream, len(stream), C.byref(rs), pcm) == -5  # not configured
    lib.IAMF_decoder_output_layout_set_binaural(d)
    lib.IAMF_decoder_set_bit_depth(d, 16)
    # descriptors cut short: more data needed
    assert lib.IAMF_decoder_configure(d, stream[:40], 40, C.byref(rs)) == -2
    assert lib.IAMF_decoder_configure(d, stream, len(stream), C.byref(rs)) == 0
    used = rs.value
    # a temporal unit cut in the middle: nothing emitted yet, the rest completes it
    tu = stream[used:]
    n1 = lib.IAMF_decoder_decode(d, tu[:1000], 1000, C.byref(rs), pcm)
    assert n1 == 0
    n2 = lib.IAMF_decoder_decode(d, tu[rs.value:], len(tu) - rs.value, C.byref(rs), pcm)
    assert n2 == 1024 - 240
    lib.IAMF_decoder_close(d)

    # bit depth never set: the reference emits nothing useful (bit_depth 0); here configure refuses
    d = lib.IAMF_decoder_open()
    lib.IAMF_decoder_output_layout_set_binaural(d)
    assert lib.IAMF_decoder_configure(d, stream, len(stream), C.byref(rs)) == -1
    lib.IAMF_decoder_close(d)

    # an Opus codec config: upstream of this path
    import iamf_writer as W
    bad = W.sequence_header(1) + W.obu(W.OBU_CODEC_CONFIG, W.leb128(0) + b"Opus" + W.leb128(960) + b"\x00\x00" + b"\x00" * 11)
    d = lib.IAMF_decoder_open()
    lib.IAMF_decoder_output_layout_set_binaural(d)
    lib.IAMF_decoder_set_bit_depth(d, 16)
    assert lib.IAMF_decoder_configure(d, bad + b"\x20\x00", len(bad) + 2, C.byref(rs)) == -6
    lib.IAMF_decoder_close(d)


def test_pick_buffer_pair_reports_the_fastest_of_the_measured_pairs():
    """iamf_hip_pick_buffer_pair: times the no-compute traffic kernel on every (input, output)
    candidate pair and returns the indices of the smallest median; bad arguments are refused"""
    import ctypes as C

    import torch

    import iac_amd as A
    from iac_amd import hipabi
    S, chunks, rows, pieces = 64, 8, 16, 1
    in_stride, out_stride = chunks * rows * 4096 + 4096, chunks * pieces * 4096
    ins = [torch.zeros(S * in_stride, dtype=torch.uint8, device="cuda") for _ in range(3)]
    outs = [torch.zeros(S * out_stride, dtype=torch.uint8, device="cuda") for _ in range(2)]
    bi, bo, ms = hipabi.pick_buffer_pair(S, chunks, rows, pieces, [t.data_ptr() for t in ins], in_stride,
                                         [t.data_ptr() for t in outs], out_stride)
    assert ms.shape == (3, 2) and (ms > 0).all()
    assert (bi, bo) == tuple(int(v) for v in np.unravel_index(np.argmin(ms), ms.shape))
    assert outs[bo].any()   # the probe wrote into the output candidates
    L = A.lib()
    bad = L.iamf_hip_pick_buffer_pair(S, chunks, rows, pieces, None, 3, in_stride, None, 2, out_stride, None, None, None, None)
    assert bad != 0

