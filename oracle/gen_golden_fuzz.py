#!/usr/bin/env python3
"""tests/golden/fuzz.json: per seeded random stream of tests/e2e_fuzz.py the per-call return values and a SHA-256 of the PCM
the REAL reference (oracle/_ref/libiamf_ref.so, IAMF_decoder_*) produced.  TEST INFRASTRUCTURE; runs only in the authoring
container.  Every stream is decoded in a process of its own: a combination the reference crashes on is recorded as such
("crash") and left out of the comparison instead of ending the run.

    python oracle/gen_golden_fuzz.py            # all seeds
    python oracle/gen_golden_fuzz.py --one 17   # (internal) one stream -> a JSON line on stdout"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_fuzz as F  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402


REF = dict(default=("_ref", "libiamf_ref.so"), lfe=("_ref_lfe", "libiamf_ref_lfe.so"), tv=("_ref_tv", "libiamf_ref_tv.so"),
           wide=("_ref", "libiamf_ref.so"), multi=("_ref", "libiamf_ref.so"), params=("_ref", "libiamf_ref.so"), concat=("_ref", "libiamf_ref.so"), syntax=("_ref", "libiamf_ref.so"), dparams=("_ref", "libiamf_ref.so"))


def one(seed, variant):
    ref = C.CDLL(os.path.join(HERE, *REF[variant]))
    stream, c = F.build(seed, variant)
    md = dict(rows=[], owns_anchors=False, strict=False)   # IAMF_decoder_get_last_metadata after configure and after every delivered frame
    try:
        pcm, rets = decode_stream(ref, stream, c["layout"], metadata=md, **F.decode_kwargs(c, variant))
    except AssertionError as e:   # configure / decode refused the stream: what it said is the golden
        return dict(error=str(e))
    rets = rets[:-1]   # (the metadata run flushes a second time: decoder_driver.decode_stream; `rets` is the plain run's)
    return dict(sha256=F.digest(pcm), shape=list(pcm.shape), rets=[int(r) for r in rets], meta=F.meta_digest(md))


def one_blocks(seed):
    from decoder_driver import decode_stream_blocks
    variant, vs, block = F.blocks_case(seed)
    ref = C.CDLL(os.path.join(HERE, *REF[variant]))
    stream, c = F.build(vs, variant)
    pcm, events = decode_stream_blocks(ref, stream, c["layout"], block, **F.decode_kwargs(c, variant))
    return dict(sha256=F.digest(pcm), shape=list(pcm.shape), events=F.events_digest(events), calls=len(events),
                last=[list(e) for e in events[-2:]])


def one_switch(seed):
    import numpy as np
    from decoder_driver import decode_stream_switching
    vs, lays, after = F.switch_case(seed)
    ref = C.CDLL(os.path.join(HERE, *REF["tv"]))
    stream, c = F.build(vs, "tv")
    kw = F.decode_kwargs(c, "tv")
    try:
        chunks, rets = decode_stream_switching(ref, stream, lays, after, **kw)
    except AssertionError as e:
        return dict(error=str(e))
    pcm = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 12), np.int16)
    return dict(sha256=F.digest(pcm), shape=list(pcm.shape), rets=[list(r) if isinstance(r, tuple) else int(r) for r in rets])


def one_units(seed):
    from decoder_driver import decode_stream_units
    variant, vs, desc, units, c = F.units_case(seed)
    ref = C.CDLL(os.path.join(HERE, *REF[variant]))
    pcm, rets = decode_stream_units(ref, desc, units, c["layout"], **F.decode_kwargs(c, variant))
    return dict(sha256=F.digest(pcm), shape=list(pcm.shape), rets=[int(r) for r in rets])


def one_gmix(seed):
    variant, cases, streams = F.gmix_build(seed)
    ref = C.CDLL(os.path.join(HERE, *REF[variant]))
    out = []
    for c, st in zip(cases, streams):
        pcm, rets = decode_stream(ref, st, c["layout"], **F.decode_kwargs(c, variant))
        out.append(dict(sha256=F.digest(pcm), rets=[int(r) for r in rets]))
    return dict(handles=out)


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--gmix":
        print(json.dumps(one_gmix(int(sys.argv[2]))))
        return
    if len(sys.argv) >= 3 and sys.argv[1] == "--units":
        print(json.dumps(one_units(int(sys.argv[2]))))
        return
    if len(sys.argv) >= 3 and sys.argv[1] == "--switch":
        print(json.dumps(one_switch(int(sys.argv[2]))))
        return
    if len(sys.argv) >= 3 and sys.argv[1] == "--blocks":
        print(json.dumps(one_blocks(int(sys.argv[2]))))
        return
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        print(json.dumps(one(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "default")))
        return
    for variant, (_, n_seeds) in F.VARIANTS.items():
        out = {}
        for seed in range(n_seeds):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(seed), variant], capture_output=True, text=True)
            if r.returncode != 0:
                out[str(seed)] = dict(crash=r.returncode)
            else:
                out[str(seed)] = json.loads(r.stdout.strip().splitlines()[-1])
            c = F.case(seed, variant)
            print("  fuzz %-7s %3d %-34s %-14s fs %4d -> %s" % (variant, seed, "+".join(c["pair"]), c["layout"], c["fs"],
                                                              out[str(seed)].get("shape") or out[str(seed)]))
        name = "fuzz.json" if variant == "default" else "fuzz_%s.json" % variant
        with open(os.path.join(ROOT, "tests", "golden", name), "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        print("fuzz goldens (%s) written:" % variant, sum("sha256" in v for v in out.values()), "decoded,",
              sum("error" in v for v in out.values()), "refused,", sum("crash" in v for v in out.values()), "crashed")
    out = {}
    for seed in range(F.N_SWITCH):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--switch", str(seed)], capture_output=True, text=True)
        out[str(seed)] = dict(crash=r.returncode) if r.returncode else json.loads(r.stdout.strip().splitlines()[-1])
        print("  fuzz switch %3d %s -> %s" % (seed, F.switch_case(seed), out[str(seed)].get("shape") or out[str(seed)]))
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_switch.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("fuzz goldens (switch) written:", sum("sha256" in v for v in out.values()), "decoded,",
          sum("error" in v for v in out.values()), "refused,", sum("crash" in v for v in out.values()), "crashed")
    out = {}
    for seed in range(F.N_GMIX):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gmix", str(seed)], capture_output=True, text=True)
        out[str(seed)] = dict(crash=r.returncode) if r.returncode else json.loads(r.stdout.strip().splitlines()[-1])
        print("  fuzz gmix %3d -> %s handles" % (seed, len(out[str(seed)].get("handles", []))))
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_gmix.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("fuzz goldens (gmix) written:", sum("handles" in v for v in out.values()), "decoded,", sum("crash" in v for v in out.values()), "crashed")
    out = {}
    for seed in range(F.N_UNITS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--units", str(seed)], capture_output=True, text=True)
        out[str(seed)] = dict(crash=r.returncode) if r.returncode else json.loads(r.stdout.strip().splitlines()[-1])
        print("  fuzz units %3d -> %s" % (seed, out[str(seed)].get("shape") or out[str(seed)]))
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_units.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("fuzz goldens (units) written:", sum("sha256" in v for v in out.values()), "decoded,", sum("crash" in v for v in out.values()), "crashed")
    out = {}
    for seed in range(F.N_BLOCKS):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--blocks", str(seed)], capture_output=True, text=True)
        out[str(seed)] = dict(crash=r.returncode) if r.returncode else json.loads(r.stdout.strip().splitlines()[-1])
        print("  fuzz blocks %3d %s -> %s" % (seed, F.blocks_case(seed), out[str(seed)].get("shape") or out[str(seed)]))
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_blocks.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("fuzz goldens (blocks) written:", sum("sha256" in v for v in out.values()), "decoded,",
          sum("crash" in v for v in out.values()), "crashed")


if __name__ == "__main__":
    main()
