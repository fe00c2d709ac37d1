"""-m gpu: stream sub-ranges of a batch (iamf_hip_batch_render_range / _flush_range).

The reference has one state machine per decoder handle (IAMF_decoder.c:3303-3525): nothing couples two streams, so
streams of one batch need not advance in step.  Here three groups of streams of ONE batch are driven on different
schedules — different call sizes, a trimmed frame in one group only, one group flushed while the others go on — and
every stream's PCM must be what the oracle gives for that stream alone."""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,out,m", [("TOA", "BINAURAL", 16), ("L714", "J", 12), ("TOA", "B", 16)])
def test_ranges_advance_independently(src, out, m):
    import torch
    import iac_amd as A
    fs, F, S = 1024, 6, 7
    ch = A.layout_channels(A.SS[out])
    if src == "TOA":
        mx, omx, proj = A.get_h2m_matrix(3, A.SS[out]), O.get_h2m(3, O.SS[out]), A.PROJ_EXACT
    else:
        mx, omx, proj = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out]), A.PROJ_AUTO
    x = np.stack([synth.hot(7000 + s, m, F * fs, burst_phase=300 + 17 * s, burst_period=1900) for s in range(S)])
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(S, m, F, fs).transpose(0, 2, 1, 3))).cuda()   # [S][F][m][fs]
    b = A.Batch(S, mx, ch, frame_size=fs, out_format=A.FMT_S16, limiter=True, projection=proj)
    st = torch.cuda.current_stream().cuda_stream
    cap = F * fs * ch * 2
    outs = [[] for _ in range(S)]
    pos = [0] * S      # frames consumed per stream

    def render(s0, cnt, nf, n_samples=0):
        f0 = pos[s0]
        assert all(pos[s] == f0 for s in range(s0, s0 + cnt))
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr() + 4 * f0 * m * fs, F * m * fs, m * fs
        a.n_frames, a.n_samples, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = nf, n_samples, pcm.data_ptr(), cap, st
        n = b.render_range(a, s0, cnt)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            if s0 <= s < s0 + cnt:
                outs[s].append(h[s][:n * ch * 2].view(np.int16).reshape(n, ch).copy())
                pos[s] += nf
            else:
                assert not h[s].any(), "a stream outside the range was written"
        return n

    def flush(s0, cnt):
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        n = b.flush_range(pcm.data_ptr(), cap, st, s0, cnt)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(s0, s0 + cnt):
            outs[s].append(h[s][:n * ch * 2].view(np.int16).reshape(n, ch).copy())

    # group A = streams 0..2, B = 3..4, C = 5..6
    render(0, 7, 1)            # everybody: one frame
    render(0, 3, 2)            # A runs ahead
    render(3, 2, 1)            # B one frame
    with pytest.raises(A.IamfHipError):   # the whole batch no longer stands at one position
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr(), F * m * fs, m * fs
        a.n_frames, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = 1, xin.data_ptr(), cap, st
        b.render_ex(a)
    render(5, 2, 1)            # C one frame
    render(3, 4, 1)            # B and C together (both at frame 2)
    flush(5, 2)                # C ends after 3 frames
    render(0, 5, 3)            # A and B (both at frame 3) finish
    flush(0, 5)
    for s in range(S):
        nfr = 3 if s >= 5 else 6
        want = O.stream_run(omx, ch, x[s][:, :nfr * fs], fs)
        got = np.concatenate(outs[s], axis=0)
        assert got.shape == want.shape, (s, got.shape, want.shape)
        assert np.array_equal(got, want), s
    b.close()


@pytest.mark.parametrize("src,out,m", [("TOA", "A", 16), ("L714", "J", 12), ("TOA", "H", 16), ("STEREO", "MONO", 2), ("L51", "B", 6)])
@pytest.mark.parametrize("trim", [237, 3, 1000])
def test_streams_that_stand_off_the_16_sample_grid(src, out, m, trim):
    """a first frame trimmed at its start by a number of samples that is not a multiple of 16 leaves the stream at a position
    off the grid of the limiter's 16-blocks for the rest of its life.  Until the second half of round 4 every later call
    then ran on the general kernel; render_fast_kernel / render_wide4_kernel place their ring per call and take such
    streams from 240 samples on (trim 1000: the second call still starts below 240 and stays on the general kernel).
    Calls: the trimmed frame, then 1 + 2 + 1 whole frames, the flush; against the oracle on the kept samples."""
    import torch
    import iac_amd as A
    fs, F, S = 1024, 5, 3
    ch = A.layout_channels(A.SS[out])
    if src == "TOA":
        mx, omx, proj = A.get_h2m_matrix(3, A.SS[out]), O.get_h2m(3, O.SS[out]), A.PROJ_EXACT
    else:
        mx, omx, proj = A.get_m2m_matrix(A.SS[src], A.SS[out]), O.get_m2m(O.SS[src], O.SS[out]), A.PROJ_AUTO
    x = np.stack([synth.hot(8100 + s + trim, m, F * fs, burst_phase=200 + 31 * s, burst_period=1700) for s in range(S)])
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(S, m, F, fs).transpose(0, 2, 1, 3))).cuda()   # [S][F][m][fs]
    b = A.Batch(S, mx, ch, frame_size=fs, out_format=A.FMT_S16, limiter=True, projection=proj)
    st = torch.cuda.current_stream().cuda_stream
    cap = F * fs * ch * 2
    outs = [[] for _ in range(S)]
    f0 = 0
    for nf, ns, skip in [(1, fs - trim, trim), (1, 0, 0), (2, 0, 0), (1, 0, 0)]:
        pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
        a = A.RenderArgs()
        a.d_in, a.in_stream_stride, a.in_frame_stride = xin.data_ptr() + 4 * (f0 * m * fs + skip), F * m * fs, m * fs
        a.n_frames, a.n_samples, a.d_pcm, a.pcm_stream_stride_bytes, a.stream = nf, ns, pcm.data_ptr(), cap, st
        n = b.render_ex(a)
        torch.cuda.synchronize()
        h = pcm.cpu().numpy()
        for s in range(S):
            outs[s].append(h[s][:n * ch * 2].view(np.int16).reshape(n, ch).copy())
        f0 += nf
    pcm = torch.zeros((S, cap), dtype=torch.uint8, device="cuda")
    n = b.flush(pcm.data_ptr(), cap, st)
    torch.cuda.synchronize()
    h = pcm.cpu().numpy()
    b.close()
    for s in range(S):
        outs[s].append(h[s][:n * ch * 2].view(np.int16).reshape(n, ch).copy())
        want = O.stream_run(omx, ch, np.ascontiguousarray(x[s][:, trim:]), fs)
        got = np.concatenate(outs[s], axis=0)
        assert got.shape == want.shape
        assert np.array_equal(got, want), (s, trim)
