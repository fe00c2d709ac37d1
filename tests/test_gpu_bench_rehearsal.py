"""-m gpu: the N > 1 control flow of bench.py on the one GPU a test box has: `--gpus 2 --rehearse-one-gpu` makes the
launcher start two ranks that share GPU 0 and talk over gloo (RCCL refuses two ranks on one device) — barriers, max
over ranks, the final gather (whose payloads must differ between ranks) and the JSON line with n_gpus = 2.  What it
cannot show is RCCL itself; that is the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_rehearsed_on_one_gpu():
    env = dict(os.environ)
    env.pop("IAMF_BENCH_CHILD", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--streams", "16",
                        "--frames", "4", "--steps", "2", "--warmup", "1", "--repeats", "2", "--placement-tries", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launched_by"] == "iac_amd.launch"
    assert "rehearsal" in d and d["data"].startswith("REHEARSAL")
    assert d["config"]["gather"] == "final" and d["gather_bytes_per_rank"] == 16 * 4 * 1024 * 2 * 2
    assert len(d["repeats"]["ms_per_step"]) == 2 and "configs" not in d and "cpu_baseline" not in d
