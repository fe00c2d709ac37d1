#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof/{trace,pmc_fetch,pmc_write}) into the
small, committed summaries under profiles/.

  python tools/prof_summary.py gpurun_out/prof profiles/r01_<tag>

writes <out>_kernel_stats.csv (the --stats table, our kernels + top others), and <out>_pmc.json
with per-launch FETCH_SIZE / WRITE_SIZE of the render kernels, corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950
FETCH_SIZE counts half of the bytes of 16-B-per-lane streaming reads, so read bytes = 2 x.
"""
import collections
import csv
import glob
import json
import os
import sys


def ours(name):
    """the kernels of the hot path (render_*_kernel, the HRTF stage's fir_fft_kernel, the LFE generator's lfe_*_kernel)"""
    return "render" in name or "fir_fft" in name or "lfe_" in name


def main():
    src, out = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    summary = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if ours(r["Name"]) or "iamf" in r["Name"]]
        keep += [r for r in rows if r not in keep][:4]
        with open(out + "_kernel_stats.csv", "w") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            for r in keep:
                r = dict(r)
                r["Name"] = r["Name"][:120]
                w.writerow(r)
        for r in keep:
            if ours(r["Name"]):
                summary.setdefault("kernels", {})[r["Name"][:80]] = {
                    "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]),
                    "max_ns": int(r["MaxNs"])}
    # The traced command spends its first launches on the placement search of bench.py's setup (other buffers, silence);
    # the TIMED region is the last `steps` dispatches of the render kernel: their durations from the kernel trace.
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    traces = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if steps and traces:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            if ours(r["Kernel_Name"]):
                per[r["Kernel_Name"][:80]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            v.sort()
            last = [d for _, d in v[-steps:]]
            summary.setdefault("timed_region", {})[k] = {"dispatches": len(last), "avg_ns": sum(last) / len(last),
                                                         "min_ns": min(last), "max_ns": max(last), "all_dispatches": len(v)}
        if os.path.exists(out + "_kernel_stats.csv"):
            with open(out + "_kernel_stats.csv", "a") as f:
                f.write("# the timed region of the traced command = the last %d dispatches of the render kernel (the earlier ones "
                        "are bench.py's placement search and warm-up):\n" % steps)
                for k, t in summary["timed_region"].items():
                    f.write("# %s: avg %.0f ns, min %d, max %d (of %d dispatches in all)\n" % (k, t["avg_ns"], t["min_ns"], t["max_ns"], t["all_dispatches"]))
    pmc = {}
    for kind, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(files[0])):
            if ours(r["Kernel_Name"]) and r["Counter_Name"] == ctr:
                agg[r["Kernel_Name"][:80]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"][:80]] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                                               "lds_block_bytes": int(r["LDS_Block_Size"]),
                                               "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
        for k, v in agg.items():
            pmc.setdefault(k, {}).update(meta[k])
            pmc[k][ctr + "_KiB_per_launch"] = sum(v) / len(v)
            pmc[k][ctr + "_launches"] = len(v)
    for k, v in pmc.items():
        rd = 2.0 * 1024 * v.get("FETCH_SIZE_KiB_per_launch", 0.0)
        wr = 1024 * v.get("WRITE_SIZE_KiB_per_launch", 0.0)
        v["hbm_read_bytes_per_launch"] = rd
        v["hbm_write_bytes_per_launch"] = wr
        v["hbm_bytes_per_launch"] = rd + wr
    summary["pmc"] = pmc
    if len(sys.argv) > 3:
        summary["command"] = sys.argv[3]
    if len(sys.argv) > 4:  # sample-frames one launch processes: gives HBM bytes per sample-frame
        sf = float(sys.argv[4])
        summary["sample_frames_per_launch"] = sf
        for v in pmc.values():
            v["hbm_bytes_per_sample_frame"] = v["hbm_bytes_per_launch"] / sf
    if len(sys.argv) > 6 and os.path.exists(sys.argv[6]):
        # the unprofiled run of the same command on the same box: its line, the card's read-out, and what the profile says
        # the roofline fraction is (algorithmic bytes / the rocprof timed-region average / 8 TB/s)
        try:
            b = json.load(open(sys.argv[6]))
            summary["device"] = b.get("device")
            summary["bench_line"] = {"value": b.get("value"), "kernel_ms": b["roofline"].get("kernel_ms"), "frac": b["roofline"].get("frac"),
                                     "streams_per_gpu": b["config"].get("streams_per_gpu"), "verified": (b.get("verified") or {}).get("ok"),
                                     "max_lsb": (b.get("verified") or {}).get("max_lsb")}
            alg = b["roofline"].get("algorithmic_bytes_per_launch")
            tags = b["roofline"].get("kernels") or [b["roofline"].get("kernel", "")]
            hit = [t for k, t in summary.get("timed_region", {}).items() if any(tag[:44] in k for tag in tags)]
            if alg and len(hit) == len(tags):
                if len(hit) == 1:
                    hit[0]["roofline_frac_from_this_profile"] = round(alg / (hit[0]["avg_ns"] * 1e-9) / 8e12, 4)
                else:   # a step of several kernels (HRTF: the stage kernel + the limiter kernel): the step = their sum
                    tot = sum(t["avg_ns"] for t in hit)
                    summary["step_of_several_kernels"] = {"kernels": tags, "sum_avg_ns": tot,
                                                          "roofline_frac_from_this_profile": round(alg / (tot * 1e-9) / 8e12, 4)}
        except Exception as e:   # noqa: BLE001
            summary["bench_line"] = {"error": str(e)}
        # the traced command's own line (HIP events under the profiler): the profiler costs the kernels 5-13 % on this pool
        try:
            tl = [x for x in open(src + ".trace.log") if x.startswith("{")][-1]
            t = json.loads(tl)
            summary["traced_run_line"] = {"value": t.get("value"), "kernel_ms": t["roofline"].get("kernel_ms"), "frac": t["roofline"].get("frac")}
        except Exception:   # noqa: BLE001
            pass
    with open(out + "_pmc.json", "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
