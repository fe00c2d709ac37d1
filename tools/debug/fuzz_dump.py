#!/usr/bin/env python3
"""the facade's PCM of the given tests/e2e_fuzz.py seeds -> gpurun_out/fuzz_dump.npz (compared with the reference's in the
authoring container: tools/debug/fuzz_diff.py)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
import iac_amd  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402

lib = C.CDLL(iac_amd.lib_path())
variant = "default"
if sys.argv[1] in F.VARIANTS:   # tools/debug/fuzz_dump.py lfe 3 17 ...
    variant = sys.argv.pop(1)
from test_gpu_fuzz_facade import _Variant  # noqa: E402
dlib = lib if variant in ("default", "wide", "multi", "params", "concat", "syntax", "dparams") else _Variant(lib, variant)
out = {"variant": np.array(variant)}
for seed in [int(a) for a in sys.argv[1:]]:
    stream, c = F.build(seed, variant)
    try:
        md = dict(rows=[], owns_anchors=True, strict=False)
        pcm, rets = decode_stream(dlib, stream, c["layout"], metadata=md, **F.decode_kwargs(c, variant))
        out["meta_%d" % seed] = np.array(md["rows"], dtype=np.int64)
        out["pcm_%d" % seed] = pcm
        out["rets_%d" % seed] = np.array(rets, dtype=np.int64)
    except AssertionError as e:
        print(seed, "error", e)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_dump.npz"), **out)
print("dumped", sorted(out))
