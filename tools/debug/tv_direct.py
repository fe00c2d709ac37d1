import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F
from decoder_driver import decode_stream_switching
mode = sys.argv[1]
if mode == "gpu":
    import iac_amd
    from test_gpu_fuzz_facade import _Variant
    lib = _Variant(C.CDLL(iac_amd.lib_path()), "tv")
else:
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref_tv", "libiamf_ref_tv.so"))
out = {}
stream, c = F.build(75, "tv")
for name, lays, after in (("direct10", [("ss", 10)], []), ("sw11_10", [("ss", 11), ("ss", 10)], [1]), ("sw10_11", [("ss", 10), ("ss", 11)], [1]),
                          ("sw0_10", [("ss", 0), ("ss", 10)], [2])):
    chunks, rets = decode_stream_switching(lib, stream, lays, after, bit_depth=16)
    out[name] = np.concatenate(chunks, axis=0)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "tv_direct_%s.npz" % mode), **out)
print("ok")
