#!/usr/bin/env python3
"""authoring container: gpurun_out/fuzz_dump.npz (tools/debug/fuzz_dump.py) against the reference's PCM of the same seeds"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402

d = np.load(os.path.join(ROOT, "gpurun_out", "fuzz_dump.npz"))
variant = str(d["variant"]) if "variant" in d.files else "default"
REF = dict(default=("_ref", "libiamf_ref.so"), lfe=("_ref_lfe", "libiamf_ref_lfe.so"), tv=("_ref_tv", "libiamf_ref_tv.so"),
           wide=("_ref", "libiamf_ref.so"), multi=("_ref", "libiamf_ref.so"), params=("_ref", "libiamf_ref.so"), concat=("_ref", "libiamf_ref.so"), syntax=("_ref", "libiamf_ref.so"), dparams=("_ref", "libiamf_ref.so"))
ref = C.CDLL(os.path.join(ROOT, "oracle", *REF[variant]))
for k in sorted([x for x in d.files if x != "variant"], key=lambda s: int(s.split("_")[1])):
    if not k.startswith("pcm_"):
        continue
    seed = int(k[4:])
    stream, c = F.build(seed, variant)
    mdr = dict(rows=[], owns_anchors=False, strict=False)
    want, rets = decode_stream(ref, stream, c["layout"], metadata=mdr, **F.decode_kwargs(c, variant))
    if "meta_%d" % seed in d.files:
        mg, mw = d["meta_%d" % seed], np.array(mdr["rows"], dtype=np.int64)
        if mg.shape != mw.shape or not np.array_equal(mg, mw):
            print("   META shapes", mg.shape, mw.shape)
            for i in range(min(len(mg), len(mw))):
                if not np.array_equal(mg[i], mw[i]):
                    print("   row", i, "got ", [int(v) for v in mg[i] if v != -9999])
                    print("   row", i, "want", [int(v) for v in mw[i] if v != -9999])
                    break
    got = d[k]
    desc = {x: c[x] for x in ("pair", "layout", "fs", "frames", "bit_depth", "sample_size") if x in c}
    desc.update({x: c[x] for x in ("trims", "rate", "out_rate", "loudness", "limiter", "threshold", "pair_ramps") if x in c})
    print("seed", seed, desc)
    print("   rets ref", list(rets), "\n   rets got", list(d["rets_%d" % seed]))
    if got.shape != want.shape:
        print("   SHAPE", got.shape, want.shape)
        n = min(len(got), len(want))
        got, want = got[:n], want[:n]
    if got.ndim == 3:   # 24-bit: bytes -> int
        f = lambda a: (a[..., 0].astype(np.int64) | (a[..., 1].astype(np.int64) << 8) | (a[..., 2].astype(np.int8).astype(np.int64) << 16))
        got, want = f(got), f(want)
    diff = got.astype(np.int64) - want.astype(np.int64)
    bad = np.argwhere(diff != 0)
    print("   differing words", len(bad), "of", diff.size, "first", bad[:3].tolist(), "max |d|", int(np.abs(diff).max()) if diff.size else 0,
          "per channel", [int((diff[:, ch] != 0).sum()) for ch in range(diff.shape[1])])
