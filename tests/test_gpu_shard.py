"""-m gpu: the multi-device entry of the C ABI (iamf_hip_shard_*) on the one GPU a test box has.

n_devices = 1 must be the plain batch bit for bit, and the gather — ncclCommInitAll over the shard's devices, ncclSend /
ncclRecv in one group on the gather stream, RCCL loaded with dlopen — must have executed at least once before the first
8-GPU run: with one device the root sends to and receives from itself.  (The 8-GPU scaling itself cannot be measured on
this pool; DESIGN.md 6 says so.)"""
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
