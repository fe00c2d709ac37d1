#!/usr/bin/env python3
"""Golden vectors of the reference's -DSAMSUNG_TV build -> tests/golden/tv.npz + manifest_tv.json.

TEST INFRASTRUCTURE.  Runs only in the authoring container: the REAL reference compiled with
-DSAMSUNG_TV (oracle/_ref_tv/libiamf_ref_tv.so, `make -C oracle ref_tv`) decodes synthetic LPCM streams
(tests/tv_cases.py) through IAMF_decoder_*; the 12-channel-stride PCM it writes is stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tv_cases as T  # noqa: E402
from decoder_driver import decode_stream, decode_stream_switching  # noqa: E402
import e2e_cases as E  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ref = C.CDLL(os.path.join(HERE, "_ref_tv", "libiamf_ref_tv.so"))


def main():
    out, manifest = {}, {}
    for name in T.CASES:
        c = T.case(name)
        pcm, rets = decode_stream(ref, T.build(name), c["layout"], **T.decode_kwargs(name))
        out[name] = pcm
        out[name + "_rets"] = np.array(rets, dtype=np.int32)
        manifest[name] = {k: v for k, v in c.items() if k != "builder"}
        print("  tv %-22s -> %s" % (name, pcm.shape))
    for name, c in T.SWITCH_CASES.items():   # the run-time layout switch (IAMF_decoder.c:3819-3881)
        stream = E.build(c["stream"])[0]
        chunks, rets = decode_stream_switching(ref, stream, c["layouts"], c["after"], bit_depth=E.CASES[c["stream"]].get("bit_depth", 16))
        out[name] = np.concatenate(chunks, axis=0)
        out[name + "_lens"] = np.array([len(x) for x in chunks], dtype=np.int32)
        out[name + "_rets"] = np.array([r[1] if isinstance(r, tuple) else r for r in rets], dtype=np.int32)
        manifest[name] = dict(stream=c["stream"], layouts=[list(l) for l in c["layouts"]], after=c["after"])
        print("  tv %-26s -> %s rets %s" % (name, out[name].shape, list(out[name + "_rets"])))
    np.savez_compressed(os.path.join(GOLD, "tv.npz"), **out)
    with open(os.path.join(GOLD, "manifest_tv.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("TV golden fixtures written")


if __name__ == "__main__":
    main()
