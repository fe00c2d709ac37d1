#!/usr/bin/env python3
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
from decoder_driver import decode_stream_switching  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref_tv", "libiamf_ref_tv.so"))
d = np.load(os.path.join(ROOT, "gpurun_out", "fuzz_switch_dump.npz"))
for k in sorted([x for x in d.files if x.startswith("pcm_")], key=lambda s: int(s.split("_")[1])):
    seed = int(k[4:])
    vs, lays, after = F.switch_case(seed)
    stream, c = F.build(vs, "tv")
    chunks, rets = decode_stream_switching(ref, stream, lays, after, **F.decode_kwargs(c, "tv"))
    want = np.concatenate(chunks, axis=0)
    got = d[k]
    wr = [r[1] if isinstance(r, tuple) else r for r in rets]
    print("seed", seed, "tv", vs, c["pair"], "fs", c["fs"], "layouts", lays, "after", after, {x: c[x] for x in ("trims", "rate", "out_rate", "loudness", "limiter", "pair_ramps", "bit_depth") if x in c})
    print("   rets ref", wr, "\n   rets got", list(d["rets_%d" % seed]))
    n = min(len(got), len(want))
    if got.shape != want.shape:
        print("   SHAPE", got.shape, want.shape)
    g, w = got[:n].astype(np.int64), want[:n].astype(np.int64)
    bad = np.argwhere(g != w)
    print("   differing", len(bad), "of", g.size, "first", bad[:2].tolist(), "per column", [int((g[:, i] != w[:, i]).sum()) for i in range(min(g.shape[1], 12))])
