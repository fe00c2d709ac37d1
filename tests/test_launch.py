"""bench.py --gpus N must really start N rank processes: the launcher path, on CPU,
with a stub rank program in place of the GPU child."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
STUB = os.path.join(ROOT, "tests", "launch_stub.py")


def _env(tmp_path, **kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"IAMF_BENCH_CHILD": STUB, "STUB_OUT": str(tmp_path)})
    env.update(kw)
    return env


def test_gpus_n_starts_n_env_complete_ranks(tmp_path):
    n = 4
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       env=_env(tmp_path, STUB_RENDEZVOUS="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    recs = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(str(tmp_path), "rank*.json")))]
    assert len(recs) == n
    assert sorted(int(r["RANK"]) for r in recs) == list(range(n))
    assert len({r["pid"] for r in recs}) == n                    # N distinct fresh processes ...
    assert len({r["ppid"] for r in recs}) == 1                   # ... children of the one launcher
    for r in recs:
        assert r["WORLD_SIZE"] == str(n) and r["LOCAL_WORLD_SIZE"] == str(n)
        assert r["LOCAL_RANK"] == r["RANK"]
        assert r["MASTER_ADDR"] == "127.0.0.1" and r["MASTER_PORT"] == recs[0]["MASTER_PORT"]
        assert r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert r["argv"] == ["--gpus", str(n), "--steps", "2", "--warmup", "1"]   # the same command line
        assert r["ranks_seen"] == n and r["allreduce"] == n * (n + 1) / 2        # the env really rendezvouses
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == n                # rank 0's line, relayed once


def test_a_failing_rank_fails_the_launch(tmp_path):
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3"], env=_env(tmp_path, STUB_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 7
    assert "rank 1 exited with 7" in p.stderr


def test_a_rank_that_ignores_sigterm_is_killed_and_the_launch_returns(tmp_path):
    """a rank blocked in a collective may not die on SIGTERM; the launcher escalates to SIGKILL after a
    bounded grace period instead of polling forever."""
    import time
    sys.path.insert(0, ROOT)
    from iac_amd import launch
    env = _env(tmp_path, STUB_FAIL_RANK="0", STUB_WEDGE_RANK="1")
    old = dict(os.environ)
    os.environ.clear()
    os.environ.update(env)
    try:
        t0 = time.monotonic()
        rc = launch.launch_ranks(2, [sys.executable, STUB], grace=1.0)
        took = time.monotonic() - t0
    finally:
        os.environ.clear()
        os.environ.update(old)
    assert rc == 7 and took < 60
    pid = int(open(os.path.join(str(tmp_path), "wedged1.pid")).read())
    try:   # reaped: the PID is gone (or, at worst, belongs to nobody we can signal)
        os.kill(pid, 0)
        alive = True
    except OSError:
        alive = False
    assert not alive


def test_world_size_mismatch_is_refused_before_any_gpu_call(tmp_path):
    env = _env(tmp_path, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr
    assert p.stdout.strip() == ""
    # and one process asked for N>1 under a 1-rank launcher env is refused too (no silent 1-GPU line)
    env = _env(tmp_path, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""


def test_launcher_module_touches_no_gpu_library():
    code = ("import sys; import iac_amd.launch as L; import iac_amd.hipabi as H; "
            "assert H._lib is None; assert 'torch' not in sys.modules; print('ok')")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr
