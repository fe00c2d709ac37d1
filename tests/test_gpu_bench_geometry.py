"""-m gpu: the launch geometry bench.py TIMES, checked against the oracle.

Every other parity test drives tight strides and a few frames per call.  bench.py times 64-frame calls whose
input stream stride is padded (+4 KiB: frames * channels * frame size + 1024 floats), on 512-stream (headline),
3072-stream (BASELINE config 2) and 2048-stream (config 3) shards, alternating two PCM buffers.  These tests build
exactly that through bench.Workload and check: 8 streams spread over the shard against the oracle (bit-exact;
+-1 LSB for the MFMA projection of config 3), and size-independent properties over ALL streams — streams fed
identical input give identical PCM, the limiter bounds the peak, the pad between streams' PCM regions is untouched.
Reference loops stood in for: src/iamf_dec/h2m_rdr.c:1103-1150, m2m_rdr.c:1826-1837,
audio_effect_peak_limiter.c:114-204, IAMF_decoder.c:100-167."""
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import bench
    import iac_amd as A
    return torch, bench, A


def _args(streams, frames=64, pad_kb=4, pcm_pad_kb=0):
    return types.SimpleNamespace(streams=streams, frames=frames, frame_size=1024, pad_kb=pad_kb, pcm_pad_kb=pcm_pad_kb,
                                 placement_tries=1, pcm_placement_tries=2, signal="hot")


@pytest.mark.parametrize("name,streams,pcm_pad_kb", [
    ("toa_binaural_limiter_s16", 512, 0),       # the headline launch
    ("714_ssJ_limiter_s16", 3072, 0),           # BASELINE config 2 as bench.py's `configs` runs it
    ("toa_ssH_limiter_s16", 2048, 0),           # BASELINE config 3 (MFMA projection: +-1 LSB)
    ("toa_binaural_limiter_s16", 96, 4),        # padded PCM stride too
    ("toa_hrtf256_limiter_s16", 64, 0),         # HRTF form of config 4 (own specification)
    ("scalable_714_ssJ_limiter_s16", 256, 0), ("toa_ssB_lfe_limiter_s16", 128, 0), ("710_downmix_stereo_limiter_s16", 128, 0),
    ("toa_projection_binaural_limiter_s16", 64, 0), ("714_downmix_512_limiter_s16", 128, 0)])
def test_timed_geometry_against_oracle(env, name, streams, pcm_pad_kb):
    torch, bench, A = env
    dev = torch.device("cuda", 0)
    wl = bench.Workload(A, name, _args(streams, pcm_pad_kb=pcm_pad_kb), 0, dev)
    try:
        assert wl.stream_stride == wl.F * wl.in_ch * wl.fs + 1024          # the padded stride bench.py times
        # two launches into the alternating buffers first, as the timed region does: the check below must not depend on
        # the state they leave (verify() resets) nor on which buffer it lands in
        wl.render_into(wl.pcm[0])
        wl.render_into(wl.pcm[1])
        v = wl.verify(k=8)
        assert v is not None and v["ok"], v
        assert v["streams"] == 8 and v["max_lsb"] <= v["tolerance_lsb"]
        if name in ("toa_binaural_limiter_s16", "714_ssJ_limiter_s16"):
            assert v["max_lsb"] == 0
    finally:
        wl.close()


@pytest.mark.parametrize("name,streams", [("toa_binaural_limiter_s16", 512), ("714_ssJ_limiter_s16", 3072),
                                          ("toa_ssH_limiter_s16", 2048)])
def test_timed_geometry_properties_over_all_streams(env, name, streams):
    """streams s and s + 8k carry the same programme -> the same PCM, whatever workgroup / XCD / round of the launch renders
    them; the peak stays under the limiter's bound; nothing is written past a stream's own PCM run"""
    torch, bench, A = env
    dev = torch.device("cuda", 0)
    wl = bench.Workload(A, name, _args(streams, pcm_pad_kb=4), 0, dev)
    try:
        S, F, fs, oc = wl.S, wl.F, wl.fs, wl.out_ch
        n_in = F * wl.in_ch * fs
        wl.x[:, :n_in] = wl.x[:8, :n_in].repeat((S + 7) // 8, 1)[:S]      # 8 distinct programmes, tiled over the shard
        wl.batch.reset()
        for buf in wl.pcm:
            buf.fill_(0x5A)
        n = wl.render_into(wl.pcm[0])
        torch.cuda.synchronize()
        assert n == F * fs - 240
        out = wl.pcm[0]
        body = out[:, :n * oc * 2].view(torch.int16).view(S, n, oc)
        for r in range(8):
            grp = body[r::8]
            assert bool((grp == grp[0:1]).all()), (name, r)
        thr_lsb = 1.001 * 32768 * 10 ** (-1 / 20) + 1
        assert int(body.to(torch.int32).abs().max()) <= thr_lsb
        # the last 240 sample-frames of the run (withheld by the limiter) and the 4 KiB pad behind it stay as they were
        assert bool((out[:, n * oc * 2:] == 0x5A).all())
        assert bool((wl.pcm[1] == 0x5A).all())
        # and the second launch (no reset) emits the full F * fs, starting with the withheld 240
        n2 = wl.render_into(wl.pcm[1])
        torch.cuda.synchronize()
        assert n2 == F * fs
    finally:
        wl.close()
