import sys, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import e2e_cases
from decoder_driver import decode_stream
import iac_amd
lib = C.CDLL(iac_amd.lib_path())
g = np.load("tests/golden/e2e.npz")
for name in sys.argv[1:]:
    case = e2e_cases.CASES[name]
    stream, _ = e2e_cases.build(name)
    pcm, rets = decode_stream(lib, stream, case["layout"], bit_depth=case.get("bit_depth", 16))
    want = g[name]
    d = (pcm.astype(np.int64) - want.astype(np.int64))
    bad = np.argwhere(d != 0)
    print(name, pcm.shape, "bad", len(bad), "first", bad[:5].tolist(), "maxabs", np.abs(d).max())
    per_frame = [int((d[f*1024:(f+1)*1024] != 0).sum()) for f in range(pcm.shape[0] // 1024)]
    print(" per 1024 block:", per_frame)
    print(" per channel:", [(int((d[:, c] != 0).sum()), int(np.abs(d[:, c]).max())) for c in range(pcm.shape[1])])
