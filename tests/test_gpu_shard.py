"""-m gpu: the multi-device entry of the C ABI (iamf_hip_shard_*) on the one GPU a test box has.

n_devices = 1 must be the plain batch bit for bit, and the gather — ncclCommInitAll over the shard's devices, ncclSend /
ncclRecv in one group on the gather stream, RCCL loaded with dlopen — must have executed at least once before the first
8-GPU run: with one device the root sends to and receives from itself.  (The 8-GPU scaling itself cannot be measured on
this pool; NOTEBOOK.md 6 says so.)"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


def test_one_device_shard_is_the_plain_batch_and_rccl_gathers():
    import torch
    import iac_amd as A
    assert torch.cuda.is_available()
    L = A.lib()
    S, fs, F, m = 6, 1024, 3, 16
    x = np.stack([synth.hot(8100 + s, m, F * fs, burst_phase=200 + 31 * s, burst_period=2100) for s in range(S)])
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(S, m, F, fs).transpose(0, 2, 1, 3))).cuda()
    cfg = A.BatchConfig()
    cfg.n_streams, cfg.frame_size, cfg.sample_rate, cfg.out_channels, cfg.out_format = S, fs, 48000, 2, A.FMT_S16
    cfg.matrix = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    cfg.limiter_enable, cfg.limiter_threshold_db = 1, -1.0
    h = C.c_void_p()
    assert L.iamf_hip_shard_create(C.byref(cfg), None, 1, C.byref(h)) == 0
    assert L.iamf_hip_shard_devices(h) == 1
    dev, first, count = C.c_int(), C.c_int(), C.c_int()
    assert L.iamf_hip_shard_info(h, 0, C.byref(dev), C.byref(first), C.byref(count)) == 0
    assert (dev.value, first.value, count.value) == (0, 0, S)
    stride = F * fs * 2 * 2
    pcm = torch.zeros((S, stride), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((S, stride), dtype=torch.uint8, device="cuda")
    tail = torch.zeros((S, stride), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ins = (C.c_void_p * 1)(xin.data_ptr())
    outs = (C.c_void_p * 1)(pcm.data_ptr())
    n = L.iamf_hip_shard_render(h, ins, F * m * fs, m * fs, F, outs, stride)
    assert n == F * fs - 240
    ver = L.iamf_hip_shard_rccl_version()
    assert ver, "no RCCL on a ROCm box?"
    assert L.iamf_hip_shard_gather(h, 0, dst.data_ptr(), stride, outs, stride) == 0     # self send / recv through RCCL
    outs2 = (C.c_void_p * 1)(tail.data_ptr())
    n2 = L.iamf_hip_shard_flush(h, outs2, stride)                                        # overlaps the gather
    assert n2 == 240
    assert L.iamf_hip_shard_sync(h) == 0
    assert torch.equal(dst, pcm), "the gathered PCM is not what the shard rendered"
    omx = O.get_h2m(3, O.SS["BINAURAL"])
    a, b = pcm.cpu().numpy(), tail.cpu().numpy()
    for s in range(S):
        got = np.concatenate([a[s][:n * 4].view(np.int16).reshape(n, 2), b[s][:n2 * 4].view(np.int16).reshape(n2, 2)])
        assert np.array_equal(got, O.stream_run(omx, 2, x[s], fs)), s
    L.iamf_hip_shard_destroy(h)
    print("rccl", ver.decode())


def test_one_device_gather_of_rows_into_a_strided_destination_and_its_accounting():
    """iamf_hip_shard_gather_rows on the real runtime and the real RCCL: only the emitted rows travel (padded source regions
    packed by hipMemcpy2DAsync on the gather stream, a strided destination spread the same way), the flush's 240
    sample-frames per stream as 960 bytes, and iamf_hip_shard_times reports what went over the wire.  (The N > 1 form of
    the same code runs against stand-ins in tests/test_abi_and_sharding.py.)"""
    import torch
    import iac_amd as A
    L = A.lib()
    S, fs, F, m = 5, 1024, 2, 16
    x = np.stack([synth.hot(8300 + s, m, F * fs, burst_phase=100 + 17 * s, burst_period=1900) for s in range(S)])
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(S, m, F, fs).transpose(0, 2, 1, 3))).cuda()
    cfg = A.BatchConfig()
    cfg.n_streams, cfg.frame_size, cfg.sample_rate, cfg.out_channels, cfg.out_format = S, fs, 48000, 2, A.FMT_S16
    cfg.matrix = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    cfg.limiter_enable, cfg.limiter_threshold_db = 1, -1.0
    h = C.c_void_p()
    assert L.iamf_hip_shard_create(C.byref(cfg), None, 1, C.byref(h)) == 0
    stride = F * fs * 4 + 512                       # padded regions
    pcm = torch.full((S, stride), 0x11, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ins, outs = (C.c_void_p * 1)(xin.data_ptr()), (C.c_void_p * 1)(pcm.data_ptr())
    n = L.iamf_hip_shard_render(h, ins, F * m * fs, m * fs, F, outs, stride)
    assert n == F * fs - 240
    row = n * 4
    dense = torch.full((S, row), 0x22, dtype=torch.uint8, device="cuda")
    wide = torch.full((S, row + 320), 0x33, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert L.iamf_hip_shard_gather_rows(h, 0, dense.data_ptr(), row, outs, stride, row) == 0
    assert L.iamf_hip_shard_gather_rows(h, 0, wide.data_ptr(), row + 320, outs, stride, row) == 0
    n2 = L.iamf_hip_shard_flush(h, outs, stride)   # into the same regions: waits for the two gathers that read them
    assert n2 == 240
    tail = torch.full((S, 960), 0x44, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert L.iamf_hip_shard_gather_rows(h, 0, tail.data_ptr(), 960, outs, stride, 960) == 0
    assert L.iamf_hip_shard_sync(h) == 0
    ls, ts, lr = C.c_int64(), C.c_int64(), C.c_int64()
    lms, tms = C.c_double(), C.c_double()
    assert L.iamf_hip_shard_times(h, 0, C.byref(ls), C.byref(ts), C.byref(lr), C.byref(lms), C.byref(tms)) == 0
    assert (ls.value, lr.value, ts.value) == (960 * S, 960 * S, (2 * row + 960) * S)
    assert 0.0 < lms.value <= tms.value
    assert L.iamf_hip_shard_gather_rows(h, 0, tail.data_ptr(), 900, outs, stride, 960) == -1   # destination rows too short
    omx = O.get_h2m(3, O.SS["BINAURAL"])
    d, w, t = dense.cpu().numpy(), wide.cpu().numpy(), tail.cpu().numpy()
    for s in range(S):
        want = O.stream_run(omx, 2, x[s], fs)
        assert np.array_equal(d[s].view(np.int16).reshape(n, 2), want[:n]), s
        assert np.array_equal(w[s][:row], d[s]) and (w[s][row:] == 0x33).all(), s
        assert np.array_equal(t[s].view(np.int16).reshape(240, 2), want[n:]), s
    L.iamf_hip_shard_destroy(h)
    print("gather of %d-byte rows: %.3f ms" % (960, lms.value))


def test_shard_refuses_more_devices_than_the_box_has():
    import torch
    import iac_amd as A
    L = A.lib()
    cfg = A.BatchConfig()
    cfg.n_streams, cfg.frame_size, cfg.sample_rate, cfg.out_channels, cfg.out_format = 64, 1024, 48000, 2, A.FMT_S16
    cfg.matrix = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    cfg.limiter_enable, cfg.limiter_threshold_db = 1, -1.0
    h = C.c_void_p()
    nd = torch.cuda.device_count()
    devs = (C.c_int * (nd + 1))(*range(nd + 1))
    assert L.iamf_hip_shard_create(C.byref(cfg), devs, nd + 1, C.byref(h)) == -1
    devs2 = (C.c_int * 2)(0, 0)
    assert L.iamf_hip_shard_create(C.byref(cfg), devs2, 2, C.byref(h)) == -1       # the same device twice
