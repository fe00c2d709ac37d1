#!/usr/bin/env python3
"""the facade rates of bench.py alone (single handle / groups), without the rest of the line:  python tools/debug/facade_rate.py"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
assert torch.cuda.is_available()
r = b.facade_rates(1024)
print(json.dumps({k: r[k] for k in ("single_handle_msamples_s", "single_handle_us_per_call", "group_msamples_s", "group256_msamples_s")}))
