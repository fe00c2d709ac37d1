#!/usr/bin/env python3
"""Golden vectors of the HOA LFE generator -> tests/golden/lfe.npz + manifest_lfe.json.

TEST INFRASTRUCTURE.  Runs only in the authoring container: calls the REAL reference built with its
own switch -DDISABLE_LFE_HOA=0 (oracle/_ref_lfe/libiamf_ref_lfe.so, `make -C oracle ref_lfe`; the
default reference build compiles the generator out, ae_rdr.h:63-65).  Stage symbols:
IAMF_element_renderer_get_H2M_matrix / _render_H2M (h2m_rdr.c:1070,1088) with an lfe_filter_t
initialised by lfefilter_init (h2m_rdr.c:1192); end to end: IAMF_decoder_* on synthetic LPCM streams.
Only inputs' recipes (tests/lfe_cases.py) and the reference's OUTPUTS are stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lfe_cases as L  # noqa: E402
from decoder_driver import decode_stream  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ref = C.CDLL(os.path.join(HERE, "_ref_lfe", "libiamf_ref_lfe.so"))
FP = C.POINTER(C.c_float)


class PredefSP(C.Structure):
    _fields_ = [("system", C.c_int), ("lfe1", C.c_int), ("lfe2", C.c_int)]


class LfeFilter(C.Structure):   # lfe_filter_t, ae_rdr.h:92-96
    _fields_ = [("init", C.c_int), ("c", C.c_float), ("a1", C.c_float), ("a2", C.c_float), ("a3", C.c_float),
                ("b1", C.c_float), ("b2", C.c_float), ("ih", C.c_float * 2), ("oh", C.c_float * 2)]


class HoaLayout(C.Structure):
    _fields_ = [("order", C.c_int), ("lfe_on", C.c_int)]


class H2M(C.Structure):
    _fields_ = [("in_", C.c_int), ("out", C.c_int), ("channels", C.c_int), ("lfe1", C.c_int),
                ("lfe2", C.c_int), ("mat", FP), ("m", C.c_int), ("n", C.c_int)]


OUT_CH = {0x020: 2, 0x050: 6, 0x250: 8, 0x450: 10, 0x451: 11, 0x370: 12, 0x490: 14, 0x9A3: 24, 0x070: 8,
          0x470: 12, 0x712: 10, 0x312: 6}


def rows(a):
    arr = (FP * a.shape[0])()
    for i in range(a.shape[0]):
        arr[i] = a[i].ctypes.data_as(FP)
    return arr


def main():
    ref.lfefilter_init.argtypes = [C.POINTER(LfeFilter), C.c_float, C.c_float]
    ref.lfefilter_init.restype = None
    out, manifest = {}, {}
    for name, (order, oid, rate, sizes, seed) in L.STAGE.items():
        x = L.stage_input(name)
        hin, pout, h = HoaLayout(order, 1), PredefSP(oid, 0, 0), H2M()
        assert ref.IAMF_element_renderer_get_H2M_matrix(C.byref(hin), C.byref(pout), C.byref(h)) == 0
        f = LfeFilter()
        ref.lfefilter_init(C.byref(f), 120.0, float(rate))   # IAMF_decoder.c:2632
        ch = OUT_CH[oid]
        parts, pos = [], 0
        for ns in sizes:
            xi = np.ascontiguousarray(x[:, pos:pos + ns])
            o = np.zeros((max(ch, h.n + 2), ns), dtype=np.float32)
            ref.IAMF_element_renderer_render_H2M(C.byref(h), rows(xi), rows(o), ns, C.byref(f))
            parts.append(o[:ch].copy())
            pos += ns
        out["stage_" + name] = np.concatenate(parts, axis=1)
        out["stage_" + name + "_coef"] = np.array([f.c, f.a1, f.a2, f.a3, f.b1, f.b2], dtype=np.float32)
        manifest["stage/" + name] = dict(order=order, out_id=oid, rate=rate, sizes=sizes, seed=seed,
                                         lfe1=h.lfe1, lfe2=h.lfe2, n=h.n)
        print("  lfe stage %-18s -> %s lfe1 %d lfe2 %d" % (name, out["stage_" + name].shape, h.lfe1, h.lfe2))
    for name, c in L.E2E.items():
        stream, _ = L.build(name)
        pcm, rets = decode_stream(ref, stream, ("ss", L.SS_ENUM[c["ss"]]), bit_depth=c["bit_depth"])
        out["e2e_" + name] = pcm
        out["e2e_" + name + "_rets"] = np.array(rets, dtype=np.int32)
        manifest["e2e/" + name] = c
        print("  lfe e2e   %-18s -> %s" % (name, pcm.shape))
    np.savez_compressed(os.path.join(GOLD, "lfe.npz"), **out)
    with open(os.path.join(GOLD, "manifest_lfe.json"), "w") as fjs:
        json.dump(manifest, fjs, indent=1, sort_keys=True)
    print("LFE golden fixtures written")


if __name__ == "__main__":
    main()
