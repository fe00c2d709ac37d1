import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import iac_amd as A
from iac_amd import hipabi
chunks = 64
in_stride, out_stride = chunks*16*4096 + 4096, chunks*4096
CH = 2 << 30
pa = A.PairAlloc(3 * CH, CH // 2, 2)   # input of 3 chunks
print("kinds", pa.kinds)
for t in [pa.view(pa.d_in, 3*CH, torch.float32)] + [pa.view(p, CH//2, torch.uint8) for p in pa.d_out]:
    t.zero_()
def rate(i, o, n):
    _, _, ms = hipabi.pick_buffer_pair(n, chunks, 16, 1, [i], in_stride, [o], out_stride)
    return n*chunks*17*4096/ms[0,0]/1e6
for k in range(3):
    print("input chunk %d alone (480 streams): out0 %.0f  out1 %.0f GB/s" % (k, rate(pa.d_in + k*CH, pa.d_out[0], 480), rate(pa.d_in + k*CH, pa.d_out[1], 480)))
print("512 streams from the start (spans chunks 0-1): %.0f / %.0f" % (rate(pa.d_in, pa.d_out[0], 512), rate(pa.d_in, pa.d_out[1], 512)))
print("512 streams from the middle of chunk 0:        %.0f" % rate(pa.d_in + CH//2, pa.d_out[0], 512))
plain = [torch.zeros(512*in_stride//4, dtype=torch.float32, device='cuda') for _ in range(6)]
pout = torch.zeros(512*out_stride, dtype=torch.uint8, device='cuda')
print("plain hipMalloc inputs vs plain output:", [round(rate(p.data_ptr(), pout.data_ptr(), 512)) for p in plain])
print("plain hipMalloc inputs vs assembled out0:", [round(rate(p.data_ptr(), pa.d_out[0], 512)) for p in plain])
print("480 streams, plain inputs vs plain output:", [round(rate(p.data_ptr(), pout.data_ptr(), 480)) for p in plain])
