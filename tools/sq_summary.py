#!/usr/bin/env python3
"""Condense rocprofv3 --pmc SQ_* passes (one directory per workload under gpurun_out/sq/) into
profiles/<round>_sq_counters.json: per render kernel and dispatch the wave-cycle breakdown
(MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, quad-cycle units;
SQ_VALU_MFMA_BUSY_CYCLES in cycles).   python tools/sq_summary.py gpurun_out/sq profiles/r01_sq_counters.json"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    res = {}
    for d in sorted(glob.glob(os.path.join(src, "*"))):
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(files[0])):
            k = r["Kernel_Name"]
            if "render" not in k and "fir_fft" not in k:
                continue
            m = re.search(r"(render_\w+|fir_fft_kernel)<[^>]*>", k)
            k = m.group(0) if m else k[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                cnt[k] += 1
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])   # ns, under the counters (slower than unprofiled)
        for k, v in acc.items():
            n = max(cnt[k], 1)
            wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            e = {"dispatches": n, "avg_ns_under_counters": dur[k] / n}
            if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) > 0 and dur[k] > 0:
                # busy cycles of the matrix pipes, summed by the counter over the chip's 1024 SIMDs (256 CUs x 4), against the
                # dispatch's own duration at the nominal 2.4 GHz (the card holds ~2.1 under load; NOTEBOOK 4.2's "69 % busy")
                e["mfma_busy_frac_at_2p4GHz"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 2.4 * 1024), 4)
            for c, x in sorted(v.items()):
                e[c + "_per_dispatch"] = x / n
                if c != "SQ_WAVE_CYCLES" and not c.startswith("SQ_INSTS") and c != "SQ_VALU_MFMA_BUSY_CYCLES":
                    e[c + "_frac_of_wave_cycles"] = round(x / wc, 4)
            res.setdefault(os.path.basename(d), {})[k] = e
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({w: list(v) for w, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
