"""Probe: only bench.py's facade leg (single handle, groups of 64 and 256 handles), for a kernel trace of what a round of
the group costs on the device:  rocprofv3 --kernel-trace --stats -d out -o t --output-format csv -- python3 tools/debug/group_trace.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (initialises the GPU runtime as bench.py does)
import bench  # noqa: E402

print(json.dumps(bench.facade_rates(1024)))
