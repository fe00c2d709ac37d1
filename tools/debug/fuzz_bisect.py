#!/usr/bin/env python3
"""One fuzz case taken apart: variations of tests/e2e_fuzz.py case (variant, seed) — features removed one at a time —
decoded by the reference here (`gen`: tests/golden_tmp/bisect.json, hashes) and by the facade on the GPU box (`run`).
   python tools/debug/fuzz_bisect.py gen wide 7214      # authoring container
   python tools/debug/fuzz_bisect.py run                # GPU box"""
import copy
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_cases as E  # noqa: E402
import e2e_fuzz as F  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_tmp", "bisect.json")


def variations(c):
    v = {"as is": c}

    def mod(name, fn):
        d = copy.deepcopy(c)
        fn(d)
        v[name] = d
    mod("no resampling", lambda d: (d.pop("rate", None), d.pop("out_rate", None)))
    mod("no trims", lambda d: d.pop("trims", None))
    mod("threshold -1", lambda d: d.pop("threshold", None))
    mod("limiter off", lambda d: d.__setitem__("limiter", False))
    if len(c["pair"]) == 2:
        mod("first element only", lambda d: d.__setitem__("pair", (c["pair"][0],)))
        mod("second element only", lambda d: d.__setitem__("pair", (c["pair"][1],)))
        mod("elements swapped", lambda d: d.__setitem__("pair", (c["pair"][1], c["pair"][0])))
    mod("no gains", lambda d: [d.pop(k, None) for k in ("element_gain_q78", "element2_gain_q78", "output_gain_q78")])
    mod("frame size 1024", lambda d: d.__setitem__("fs", 1024))
    if c["pair"][0] == "scalable":   # the scalable element alone, taken apart further
        one = lambda d: d.__setitem__("pair", ("scalable",))
        mod("scalable alone, no layer gains", lambda d: (one(d), d.__setitem__("scalable_gains1", {})))
        for li in sorted(c.get("scalable_gains1", {})):
            mod("scalable alone, only layer %d's gain" % li, lambda d, li=li: (one(d), d.__setitem__("scalable_gains1", {li: c["scalable_gains1"][li]})))
        if 3 in c.get("scalable_gains1", {}):   # seed 7214: the 12th entry is layer 3's second; flags 27 go on to bits 3 and 4
            mod("scalable alone, layer 3 flags & 3", lambda d: (one(d), d["scalable_gains1"].__setitem__(3, (c["scalable_gains1"][3][0] & 3, c["scalable_gains1"][3][1]))))
            mod("scalable alone, layer 0 flags without bit 0", lambda d: (one(d), d["scalable_gains1"].__setitem__(0, (c["scalable_gains1"][0][0] & ~1, c["scalable_gains1"][0][1]))))
        mod("scalable alone, demixing mode constant 1", lambda d: (one(d), d.__setitem__("scalable_modes1", [1] * 32), d.__setitem__("dmx_default1", (1, 3))))
        mod("scalable alone, default modes", lambda d: (one(d), d.pop("scalable_modes1", None), d.pop("dmx_default1", None)))
        mod("scalable alone, all of it plain", lambda d: (one(d), d.pop("scalable_modes1", None), d.pop("dmx_default1", None), d.__setitem__("scalable_gains1", {}),
                                                         d.pop("trims", None), d.pop("rate", None), d.pop("out_rate", None), d.pop("threshold", None),
                                                         [d.pop(k, None) for k in ("element_gain_q78", "output_gain_q78")]))
    for lay in (9, 3, 1):
        mod("layout ss %d" % lay, lambda d, lay=lay: d.__setitem__("layout", ("ss", lay)))
    return v


def build(c):
    E.CASES["bisect_tmp"] = c
    try:
        return E.build("bisect_tmp")[0]
    finally:
        del E.CASES["bisect_tmp"]


def main():
    if sys.argv[1] == "gen":
        variant, seed = sys.argv[2], int(sys.argv[3])
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libiamf_ref.so"))
        out = dict(variant=variant, seed=seed, cases={})
        for name, c in variations(F.case(seed, variant)).items():
            try:
                pcm, rets = decode_stream(ref, build(c), c["layout"], **F.decode_kwargs(c, variant))
                out["cases"][name] = dict(case=c, sha256=F.digest(pcm), rets=[int(r) for r in rets])
            except AssertionError as e:
                out["cases"][name] = dict(case=c, error=str(e))
        json.dump(out, open(OUT, "w"), default=lambda o: list(o) if isinstance(o, tuple) else o)
        print("written", OUT, list(out["cases"]))
        return
    import iac_amd
    lib = C.CDLL(iac_amd.lib_path())
    d = json.load(open(OUT))

    def tup(c):   # JSON lists -> the tuples the generator compares against
        for k in ("pair", "layout"):
            if k in c:
                c[k] = tuple(c[k])
        if "trims" in c:
            c["trims"] = {int(k): tuple(v) for k, v in c["trims"].items()}
        for k in list(c):
            if k.startswith("scalable_gains") and isinstance(c[k], dict):
                c[k] = {int(a): tuple(b) for a, b in c[k].items()}
            if k.startswith("dmx_default"):
                c[k] = tuple(c[k])
        return c
    for name, g in d["cases"].items():
        c = tup(g["case"])
        if "error" in g:
            print("%-22s reference refused: %s" % (name, g["error"]))
            continue
        try:
            pcm, rets = decode_stream(lib, build(c), c["layout"], **F.decode_kwargs(c, d["variant"]))
            ok = F.digest(pcm) == g["sha256"]
            print("%-22s %s  rets %s" % (name, "equal" if ok else "DIFFERENT", "equal" if [int(r) for r in rets] == g["rets"] else "different"), flush=True)
        except AssertionError as e:
            print("%-22s facade: %s" % (name, e))


if __name__ == "__main__":
    main()
