"""-m gpu: the device LPCM unpacker (iamf_hip_lpcm_unpack, iac_amd/csrc/iamf_unpack.hip) against the reference's sample
arithmetic restated in numpy: sample = integer / 2^(bits-1) (pcm/IAMF_pcm_decoder.c:64-83, 133-149), little- and
big-endian, the reference's own byte order for big-endian 24 bit (bitstream.c:204-208: the first two bytes are assembled
little-endian), coupled sub-streams (two interleaved channels), channels nothing carries (silence), trimmed starts and
short frames.  Bit-exact: every conversion is exact or rounds as the CPU's int -> float does.  The group of decoder
handles uses it for every frame (tests/test_gpu_group.py compares whole streams with the reference's goldens)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ref_value(b, nbytes, le):
    """[..., nbytes] uint8 -> float32 as the reference's LPCM decoder computes it"""
    b = b.astype(np.int64)
    if nbytes == 2:
        v = (b[..., 0] | (b[..., 1] << 8)) if le else (b[..., 1] | (b[..., 0] << 8))
        v = np.where(v & 0x8000, v - 0x10000, v)
        return (v.astype(np.float32) / np.float32(1 << 15)).astype(np.float32)
    if nbytes == 3:
        v = (b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)) if le else (b[..., 2] | (b[..., 0] << 8) | (b[..., 1] << 16))
        v = np.where(v & 0x800000, v - 0x1000000, v)
        return (v.astype(np.float32) / np.float32(1 << 23)).astype(np.float32)
    v = (b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24)) if le else \
        (b[..., 3] | (b[..., 2] << 8) | (b[..., 1] << 16) | (b[..., 0] << 24))
    v = np.where(v & 0x80000000, v - (1 << 32), v)
    return (v.astype(np.int32).astype(np.float32) / np.float32(2.0 ** 31)).astype(np.float32)   # int32 -> f32 rounds to nearest even


@pytest.mark.parametrize("nbytes", [2, 3, 4])
@pytest.mark.parametrize("le", [1, 0])
@pytest.mark.parametrize("head", [0, 16])   # 16: {first, count} at the head of every stream's raw region, as the group keeps them
def test_unpack_matches_the_reference_arithmetic(nbytes, le, head):
    import torch
    import iac_amd as A
    fs, S = 1024, 37
    widths = [2, 2, 1, 1, 1]                       # two coupled sub-streams, three mono ones: 7 decoded channels
    # output rows: the renderer's order, two of them carried by nothing
    src_of_row = [4, 0, 1, -1, 6, 5, 2, 3, -1]
    rng = np.random.default_rng(100 * nbytes + le)
    slot, ch_off, ch_step, off = [], [], [], head
    for w in widths:
        slot.append(off)
        for k in range(w):
            ch_off.append(off + k * nbytes)
            ch_step.append(w * nbytes)
        off += w * nbytes * fs
    stride = (off + 255) & ~255
    raw = rng.integers(0, 256, size=(S, stride), dtype=np.uint8)
    # extremes in the first samples of stream 0: most negative / most positive / -1 / 0 in every channel
    for c in range(len(ch_off)):
        for i, pat in enumerate([(0x80, 0, 0, 0), (0x7f, 0xff, 0xff, 0xff), (0xff, 0xff, 0xff, 0xff), (0, 0, 0, 0)]):
            msb_first = list(pat[:nbytes])
            if le:
                by = msb_first[::-1]
            elif nbytes == 3:
                by = [msb_first[1], msb_first[0], msb_first[2]]   # the reference's 24-bit big-endian order: byte 1 on top
            else:
                by = msb_first
            raw[0, ch_off[c] + i * ch_step[c]: ch_off[c] + i * ch_step[c] + nbytes] = by
    first = rng.integers(0, 200, size=S).astype(np.int32)
    count = np.array([fs - f if s % 3 else rng.integers(0, fs - f + 1) for s, f in enumerate(first)], dtype=np.int32)
    first[0], count[0] = 0, fs
    count[5] = 0                                   # a stream that has no frame this round: its rows stay as they are
    count[6], count[7], count[8] = 1, 2, 3         # tails shorter than a lane's four samples
    lay = A.LpcmLayout()
    lay.sample_bytes, lay.little_endian, lay.channels, lay.frame_size = nbytes, le, len(src_of_row), fs
    for r, src in enumerate(src_of_row):
        lay.src_offset[r] = ch_off[src] if src >= 0 else -1
        lay.src_step[r] = ch_step[src] if src >= 0 else nbytes
    if head:
        raw[:, :8] = np.stack([first, count], 1).copy().view(np.uint8)
    d_raw = torch.from_numpy(raw).cuda()
    d_fc = torch.from_numpy(np.stack([first, count], 1).copy()).cuda()
    out = torch.full((S, len(src_of_row), fs), 7.0, dtype=torch.float32, device="cuda")
    A.lpcm_unpack(lay, d_raw.data_ptr(), stride, d_raw.data_ptr() if head else d_fc.data_ptr(), out.data_ptr(), len(src_of_row) * fs, S,
                  torch.cuda.current_stream().cuda_stream, first_count_stride=stride // 4 if head else 2)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for s in range(S):
        n = int(count[s])
        for r, src in enumerate(src_of_row):
            if src < 0:
                want = np.zeros(n, np.float32)
            else:
                idx = ch_off[src] + (int(first[s]) + np.arange(n)) * ch_step[src]
                want = ref_value(raw[s][idx[:, None] + np.arange(nbytes)[None, :]], nbytes, bool(le))
            assert np.array_equal(got[s, r, :n].view(np.uint32), want.view(np.uint32)), (s, r)
            assert np.all(got[s, r, n:] == 7.0), (s, r)          # nothing written past the frame's samples
    if nbytes == 2:
        assert got[0, 1, 0] == -1.0 and got[0, 1, 1] == np.float32(32767 / 32768) and got[0, 1, 3] == 0.0


def test_unpack_refuses_layouts_that_leave_the_raw_region():
    import torch
    import iac_amd as A
    lay = A.LpcmLayout()
    lay.sample_bytes, lay.little_endian, lay.channels, lay.frame_size = 2, 1, 2, 1024
    lay.src_offset[0], lay.src_step[0] = 0, 2
    lay.src_offset[1], lay.src_step[1] = 2048, 2
    raw = torch.zeros((2, 4096), dtype=torch.uint8, device="cuda")
    fc = torch.zeros((2, 2), dtype=torch.int32, device="cuda")
    out = torch.zeros((2, 2, 1024), dtype=torch.float32, device="cuda")
    A.lpcm_unpack(lay, raw.data_ptr(), 4096, fc.data_ptr(), out.data_ptr(), 2048, 2, None)       # fits exactly
    for bad in ("stride", "offset", "step", "fs", "out", "bytes"):
        l2 = A.LpcmLayout.from_buffer_copy(lay)
        stride, ostride = 4096, 2048
        if bad == "stride":
            stride = 4095
        elif bad == "offset":
            l2.src_offset[1] = 2050
        elif bad == "step":
            l2.src_step[0] = 1
        elif bad == "fs":
            l2.frame_size = 1022
        elif bad == "out":
            ostride = 2047
        else:
            l2.sample_bytes = 5
        with pytest.raises(A.IamfHipError):
            A.lpcm_unpack(l2, raw.data_ptr(), stride, fc.data_ptr(), out.data_ptr(), ostride, 2, None)
    torch.cuda.synchronize()
