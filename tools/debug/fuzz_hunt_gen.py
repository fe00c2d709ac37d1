#!/usr/bin/env python3
"""Hunting run: the reference's answers (oracle/gen_golden_fuzz.py --one) for tests/e2e_fuzz.py seeds beyond the committed
sets -> tests/golden_tmp/fuzz_more_<variant>.json (git-ignored; read on the GPU box by tools/debug/fuzz_more.py).
Authoring container only (needs oracle/_ref*).   python tools/debug/fuzz_hunt_gen.py FIRST COUNT [variant ...]"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GEN = os.path.join(ROOT, "oracle", "gen_golden_fuzz.py")
ALL = ("default", "wide", "lfe", "tv", "multi", "params", "concat", "syntax", "dparams")


def one(args):
    seed, variant = args
    r = subprocess.run([sys.executable, GEN, "--one", str(seed), variant], capture_output=True, text=True)
    if r.returncode != 0:
        return seed, dict(crash=r.returncode, err="")
    return seed, json.loads(r.stdout.strip().splitlines()[-1])


def one_set(args):
    seed, name = args
    r = subprocess.run([sys.executable, GEN, "--" + name, str(seed)], capture_output=True, text=True)
    if r.returncode != 0:
        return seed, dict(crash=r.returncode)
    return seed, json.loads(r.stdout.strip().splitlines()[-1])


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[3] == "--sets":   # blocks / switch / units / gmix: tests/golden_tmp/hunt_<set>.json, read by
        os.makedirs(os.path.join(ROOT, "tests", "golden_tmp"), exist_ok=True)   # tests/test_gpu_fuzz_facade.py under IAMF_FUZZ_HUNT=first:count
        for name in sys.argv[4:] or ("blocks", "switch", "units", "gmix"):
            with ThreadPoolExecutor(max_workers=7) as ex:
                gold = {str(s): g for s, g in ex.map(one_set, [(s, name) for s in range(first, first + count)])}
            with open(os.path.join(ROOT, "tests", "golden_tmp", "hunt_%s.json" % name), "w") as f:
                json.dump(gold, f)
            print(name, len(gold), "crashed", sum("crash" in g for g in gold.values()), flush=True)
        return
    variants = sys.argv[3:] or ALL
    os.makedirs(os.path.join(ROOT, "tests", "golden_tmp"), exist_ok=True)
    for v in variants:
        with ThreadPoolExecutor(max_workers=7) as ex:
            gold = {str(s): g for s, g in ex.map(one, [(s, v) for s in range(first, first + count)])}
        with open(os.path.join(ROOT, "tests", "golden_tmp", "fuzz_more_%s.json" % v), "w") as f:
            json.dump(dict(variant=v, gold=gold), f)
        print(v, "decoded", sum("sha256" in g for g in gold.values()), "refused", sum("error" in g for g in gold.values()),
              "crashed", sum("crash" in g for g in gold.values()), flush=True)


if __name__ == "__main__":
    main()
