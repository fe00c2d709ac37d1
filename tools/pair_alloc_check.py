"""What iamf_hip_pair_alloc_create delivers on the card it runs on (IAMF_HIP_PAIR_DEBUG=1 prints the kinds it found):
the cfg3 traffic shape at 2048 streams and the headline shape at 512 on the assembled buffers, next to plain allocations.
   python tools/pair_alloc_check.py        (on a GPU box)"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import iac_amd as A
from iac_amd import hipabi
S, chunks = 2048, 64
in_stride = chunks*16*4096 + 4096
out_stride = chunks*12*4096
t0 = time.time()
pa = A.PairAlloc(S*in_stride, S*out_stride, 2)
print("pair alloc: kinds", pa.kinds, "in %.1f s" % (time.time()-t0), hex(pa.d_in), [hex(p) for p in pa.d_out])
x = pa.view(pa.d_in, S*in_stride, torch.float32); x.zero_()
o = [pa.view(p, S*out_stride, torch.uint8) for p in pa.d_out]
for t in o: t.zero_()
plain_in = torch.zeros(S*in_stride//4, dtype=torch.float32, device='cuda')
plain_out = torch.zeros(S*out_stride, dtype=torch.uint8, device='cuda')
def rate(i, oo, rows, pieces, istr, ostr, n):
    _, _, ms = hipabi.pick_buffer_pair(n, chunks, rows, pieces, [i], istr, [oo], ostr)
    return n*chunks*(rows+pieces)*4096/ms[0,0]/1e6
print("cfg3 shape 2048 streams: assembled pair %.0f / %.0f GB/s, plain hipMalloc %.0f GB/s" % (
    rate(pa.d_in, pa.d_out[0], 16, 12, in_stride, out_stride, S), rate(pa.d_in, pa.d_out[1], 16, 12, in_stride, out_stride, S),
    rate(plain_in.data_ptr(), plain_out.data_ptr(), 16, 12, in_stride, out_stride, S)))
pa.close()
# headline sizes
S, in_stride, out_stride = 512, chunks*16*4096+4096, chunks*4096
pa = A.PairAlloc(S*in_stride, S*out_stride, 2)
print("headline: kinds", pa.kinds, "assembled %.0f / %.0f GB/s" % (rate(pa.d_in, pa.d_out[0], 16, 1, in_stride, out_stride, S), rate(pa.d_in, pa.d_out[1], 16, 1, in_stride, out_stride, S)))
pa.close()
print("ok")
