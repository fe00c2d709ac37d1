#!/usr/bin/env python3
"""the facade on the seeds of tests/golden_tmp/fuzz_more.json (a hunting run of tests/e2e_fuzz.py seeds beyond the committed
set: generated in the authoring container, not committed) -> the seeds that differ"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
import iac_amd  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402

lib = C.CDLL(iac_amd.lib_path())
variant = sys.argv[1] if len(sys.argv) > 1 else "default"
gold = json.load(open(os.path.join(ROOT, "tests", "golden_tmp", "fuzz_more_%s.json" % variant)))["gold"]
from test_gpu_fuzz_facade import _Variant  # noqa: E402
dlib = lib if variant in ("default", "wide", "multi", "params", "concat", "syntax", "dparams") else _Variant(lib, variant)
bad = []
for k in sorted(gold, key=int):
    seed = int(k)
    if "sha256" not in gold[k]:
        if "error" in gold[k]:   # the reference refused the stream: the facade must refuse it the same way
            stream, c = F.build(seed, variant)
            try:
                decode_stream(dlib, stream, c["layout"], **F.decode_kwargs(c, variant))
                got = "decoded"
            except AssertionError as e:
                got = str(e)
            if got != gold[k]["error"]:
                bad.append(seed)
                print(seed, "reference error:", gold[k]["error"], "| facade:", got, {x: c[x] for x in ("syntax",) if x in c}, flush=True)
        else:
            print(seed, "reference:", gold[k])
        continue
    stream, c = F.build(seed, variant)
    try:
        pcm, rets = decode_stream(dlib, stream, c["layout"], **F.decode_kwargs(c, variant))
        ok = [int(r) for r in rets] == gold[k]["rets"] and F.digest(pcm) == gold[k]["sha256"]
        why = "" if ok else ("rets" if [int(r) for r in rets] != gold[k]["rets"] else "pcm")
    except AssertionError as e:
        ok, why = False, str(e)
    if not ok and F.reference_gain_list_overflows(c):
        print(seed, why, "(the reference overflows its output-gain arrays on this stream: undefined there, not counted)", flush=True)
        continue
    if not ok:
        bad.append(seed)
        print(seed, why, {x: c[x] for x in ("syntax", "elements", "presentations", "mix_id", "pair", "layout", "fs", "frames", "bit_depth", "sample_size", "trims", "rate", "out_rate", "loudness", "limiter", "threshold", "pair_ramps") if x in c}, flush=True)
print("checked", len(gold), "bad", bad)
