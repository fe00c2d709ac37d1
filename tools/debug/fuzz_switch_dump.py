#!/usr/bin/env python3
"""facade side of the TV layout-switch fuzz for the given seeds -> gpurun_out/fuzz_switch_dump.npz"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
import iac_amd  # noqa: E402
from decoder_driver import decode_stream_switching  # noqa: E402
from test_gpu_fuzz_facade import _Variant  # noqa: E402

lib = C.CDLL(iac_amd.lib_path())
out = {}
for seed in [int(a) for a in sys.argv[1:]]:
    vs, lays, after = F.switch_case(seed)
    stream, c = F.build(vs, "tv")
    try:
        chunks, rets = decode_stream_switching(_Variant(lib, "tv"), stream, lays, after, **F.decode_kwargs(c, "tv"))
        out["pcm_%d" % seed] = np.concatenate(chunks, axis=0)
        out["rets_%d" % seed] = np.array([r[1] if isinstance(r, tuple) else r for r in rets], dtype=np.int64)
    except AssertionError as e:
        print(seed, "error", e)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_switch_dump.npz"), **out)
print("dumped", sorted(out))
