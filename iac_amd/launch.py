"""One process per GPU: start N fresh rank processes of a script from a parent that has not
touched the GPU.

`python bench.py --gpus N` without a launcher around it lands here: the parent only spawns, waits
and relays; every child gets RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR /
MASTER_PORT and initialises its own device.  The parent never calls into HIP (no
`torch.cuda.is_available()`, no library load), so nothing that initialised a GPU is ever
replaced or forked.  This module imports neither torch nor the HIP library.
"""
import os
import socket
import subprocess
import sys
import threading


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher(env=None):
    """True when a launcher (torchrun, or launch_ranks below) already gave this process a rank"""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                "IAMF_LAUNCHED_BY": "iac_amd.launch"})
    return env


def launch_ranks(n, child_argv, timeout=None, grace=10.0):
    """Start `n` children `child_argv` (a full command line), rank r with rank_env(r, n, port).
    Rank 0's stdout is relayed to ours line by line, every child's stderr to ours; the other
    ranks' stdout is dropped (they print nothing in bench.py).  Returns 0 only if EVERY child
    exited 0; on the first failure (or on `timeout`) the remaining children are terminated (their own PIDs
    only) and, if still alive `grace` seconds later, killed."""
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(child_argv, env=rank_env(r, n, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))

    def relay(p):
        for line in p.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, args=(procs[0],), daemon=True)
    t.start()
    rc = 0
    pending = set(range(n))
    import time
    t0 = time.monotonic()
    kill_at = None   # once the others were asked to stop (SIGTERM): when to stop asking (SIGKILL)

    def stop_pending():
        nonlocal kill_at
        for q in pending:
            procs[q].terminate()
        if kill_at is None:
            kill_at = time.monotonic() + grace

    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("launch: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                stop_pending()
        if timeout is not None and time.monotonic() - t0 > timeout and pending:
            sys.stderr.write("launch: timeout after %.0f s; stopping ranks %s\n" % (timeout, sorted(pending)))
            stop_pending()
            rc = rc or 124
            timeout = None
        if kill_at is not None and pending and time.monotonic() > kill_at:
            # a rank blocked inside a collective or a HIP call may not die on SIGTERM: these are our own
            # child PIDs, so kill them and reap them — the launcher must never hang on a wedged rank
            sys.stderr.write("launch: ranks %s ignored SIGTERM for %.0f s; killing them\n" % (sorted(pending), grace))
            for q in sorted(pending):
                procs[q].kill()
            for q in sorted(pending):
                try:
                    procs[q].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pass
            pending.clear()
        time.sleep(0.05)
    t.join(timeout=10)
    return rc
