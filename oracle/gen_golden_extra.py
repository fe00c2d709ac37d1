"""End-to-end golden vectors: synthetic LPCM .iamf streams (tests/iamf_writer.py) decoded by the
REAL reference through its public API (IAMF_decoder_*, include/IAMF_decoder.h), the way
test/tools/iamfplayer/player/iamfplayer.c:380-431,571-600 drives it.  Called by gen_golden.py.
TEST INFRASTRUCTURE; runs only where oracle/_ref/libiamf_ref.so exists.

These cases pin what no exported stage symbol reaches: stage order, mix-gain ramps
(IAMF_decoder.c:639-664,857-982), two-element mixing (:2702-2733), loudness (:3206-3221), PCM pack
at 16/24/32 bit (:100-167), flush (:3250-3301), the down-mixer wiring (:2574-2583) and the
resampler glue (:3223-3248).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import e2e_cases  # noqa: E402


from decoder_driver import decode_stream as ref_decode  # noqa: E402


def generate_demix(ref, manifest, gold_dir):
    import demix_cases as D
    out = {}
    for name, c in D.STAGE_CASES.items():
        y = D.drive_demixer(ref, "demixer_", c, D.case_input(c))
        out[name] = y
        manifest["demix/" + name] = dict(layers=c["layers"], order=c["order"], recon=c["recon"], flags=c["flags"],
                                         fs=c["fs"], seed=c["seed"], frames=len(c["schedule"]))
        print("  demix %-16s -> %s" % (name, y.shape))
    np.savez_compressed(os.path.join(gold_dir, "demix.npz"), **out)


def generate(ref, manifest, gold_dir, synth):
    generate_demix(ref, manifest, gold_dir)
    out = {}
    for name, case in e2e_cases.CASES.items():
        stream, _ = e2e_cases.build(name)
        pcm, rets = ref_decode(ref, stream, case["layout"], bit_depth=case.get("bit_depth", 16),
                               out_rate=case.get("out_rate", 0), loudness=case.get("loudness", 0.0),
                               limiter=case.get("limiter", True), threshold=case.get("threshold", -1.0))
        out[name] = pcm
        out[name + "_rets"] = np.array(rets, dtype=np.int32)
        manifest["e2e/" + name] = {k: v for k, v in case.items() if k != "builder"}
        print("  e2e %-28s -> %s, calls %s..." % (name, pcm.shape, rets[:3]))
    np.savez_compressed(os.path.join(gold_dir, "e2e.npz"), **out)
    generate_meta(ref, manifest, gold_dir)


def generate_meta(ref, manifest, gold_dir):
    """IAMF_decoder_get_last_metadata of the reference (IAMF_decoder.c:3619-3706,4150-4168) while it decodes: one row
    after configure, one per delivered frame, one per flush call (decoder_driver.last_metadata)"""
    out = {}
    for name, mc in e2e_cases.META_CASES.items():
        case = e2e_cases.CASES[name]
        stream, _ = e2e_cases.build(name)
        md = dict(rows=[], owns_anchors=False, **{k: v for k, v in mc.items() if k != "pts"})
        ref_decode(ref, stream, case["layout"], bit_depth=case.get("bit_depth", 16), out_rate=case.get("out_rate", 0),
                   loudness=case.get("loudness", 0.0), limiter=case.get("limiter", True), threshold=case.get("threshold", -1.0),
                   metadata=md, pts=mc["pts"])
        out[name] = np.array(md["rows"], dtype=np.int64)
        manifest["meta/" + name] = dict(rows=len(md["rows"]), **{k: list(v) if isinstance(v, tuple) else v for k, v in mc.items()})
        print("  meta %-28s -> %s" % (name, out[name].shape))
    np.savez_compressed(os.path.join(gold_dir, "meta.npz"), **out)
