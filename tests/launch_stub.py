"""Stub rank program for tests/test_launch.py: stands in for bench.py's per-GPU child.  Records the
rank environment it was given, proves it is complete by a gloo rendezvous + all-reduce, and rank 0
prints one JSON line (as bench.py's rank 0 does)."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
        "HSA_ENABLE_IPC_MODE_LEGACY", "IAMF_LAUNCHED_BY"]
rec = {k: os.environ.get(k) for k in keys}
rec["argv"] = sys.argv[1:]
rec["pid"], rec["ppid"] = os.getpid(), os.getppid()
if os.environ.get("STUB_WEDGE_RANK") == str(rank):   # a rank stuck in a collective: ignores SIGTERM, never exits
    import signal
    import time
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    with open(os.path.join(os.environ["STUB_OUT"], "wedged%d.pid" % rank), "w") as f:
        f.write(str(os.getpid()))
    time.sleep(600)
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    if os.environ.get("STUB_WEDGE_RANK"):   # fail only once the wedged rank has installed its handler
        import time
        for _ in range(200):
            if os.path.exists(os.path.join(os.environ["STUB_OUT"], "wedged%s.pid" % os.environ["STUB_WEDGE_RANK"])):
                break
            time.sleep(0.05)
    sys.exit(7)
if os.environ.get("STUB_RENDEZVOUS") == "1":
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    rec["allreduce"] = float(t.item())
    rec["ranks_seen"] = dist.get_world_size()
    dist.destroy_process_group()
with open(os.path.join(os.environ["STUB_OUT"], "rank%d.json" % rank), "w") as f:
    json.dump(rec, f)
if rank == 0:
    print(json.dumps({"n_gpus": world, "stub": True}), flush=True)
