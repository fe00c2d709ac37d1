"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares (no
compute without a GPU), and the N>1 plumbing (stream sharding + double-buffered gather) works
with world_size 2 over gloo."""
import ctypes
import os
import re
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(iamf_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import iac_amd
    iac_amd.build()
    lib = ctypes.CDLL(iac_amd.lib_path())
    names = _declared_functions(os.path.join(ROOT, "include", "iamf_hip.h"))
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n


def test_library_exports_only_the_two_api_prefixes():
    """A drop-in .so must not leak internal names into the host's namespace (an unprefixed
    `demix_factors` used to sit beside the 19 API symbols): every defined dynamic symbol is IAMF_* or iamf_hip_*."""
    import subprocess

    import iac_amd
    iac_amd.build()
    out = subprocess.run(["nm", "-D", "--defined-only", iac_amd.lib_path()], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert len(names) >= 30
    stray = [n for n in names if not (n.startswith("IAMF_") or n.startswith("iamf_hip_"))]
    assert stray == [], stray
    ref_api = re.findall(r"\b(IAMF_[a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "IAMF_decoder.h")).read(), flags=re.S))
    assert len(set(ref_api)) == 19
    for n in set(ref_api):
        assert n in names, n


def test_matrix_lookup_needs_no_gpu():
    import iac_amd as A
    h = A.get_h2m_matrix(3, A.SS["H"])
    assert (h.m, h.n, h.channels, h.lfe1, h.lfe2) == (16, 22, 24, 3, -1)
    with pytest.raises(KeyError):
        A.get_m2m_matrix(0x999, A.SS["A"])
    assert A.layout_channels(A.SS["H"]) == 24 and A.layout_channels(A.SS["BINAURAL"]) == 2


def test_product_matrix_blob_equals_oracle_blob():
    import numpy as np

    import iac_amd as A
    import oracle_lib as O
    for order in range(4):
        for oid in (A.SS["A"], A.SS["H"], A.SS["J"], A.SS["BINAURAL"]):
            a = A.get_h2m_matrix(order, oid)
            b = O.get_h2m(order, oid)
            assert (a.m, a.n, a.lfe1, a.lfe2) == (b.m, b.n, b.lfe1, b.lfe2)
            assert np.array_equal(np.ctypeslib.as_array(a.mat, shape=(a.m * a.n,)), b.array())


@pytest.mark.skipif(torch.cuda.is_available(), reason="no-GPU behaviour")
def test_batch_create_fails_loudly_without_gpu():
    import iac_amd as A
    with pytest.raises(A.IamfHipError):
        A.Batch(4, A.get_h2m_matrix(3, A.SS["A"]), 2)


def test_shard_streams_partitions():
    from iac_amd.sharding import shard_streams
    for n, w in ((4096, 8), (10, 3), (5, 8), (512, 1)):
        spans = [shard_streams(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_c_shard_split_is_the_python_shard_streams():
    """iamf_hip_shard_split (C, what the multi-device entry uses) and iac_amd.sharding.shard_streams (what bench.py's ranks
    use) must cut a job the same way; no GPU needed for either"""
    import ctypes as C

    import iac_amd as A
    from iac_amd.sharding import shard_streams
    L = A.lib()
    for n, w in ((4096, 8), (10, 3), (5, 8), (512, 1), (4097, 8), (8, 8)):
        for r in range(w):
            f, c = C.c_int(-1), C.c_int(-1)
            assert L.iamf_hip_shard_split(n, w, r, C.byref(f), C.byref(c)) == 0
            lo, hi = shard_streams(n, w, r)
            assert (f.value, f.value + c.value) == (lo, hi), (n, w, r)
    f, c = C.c_int(), C.c_int()
    assert L.iamf_hip_shard_split(10, 0, 0, C.byref(f), C.byref(c)) == -1
    assert L.iamf_hip_shard_split(10, 2, 2, C.byref(f), C.byref(c)) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="no-GPU behaviour")
def test_shard_create_validates_and_fails_loudly_without_gpu():
    import ctypes as C

    import iac_amd as A
    L = A.lib()
    cfg = A.BatchConfig()
    cfg.n_streams, cfg.frame_size, cfg.sample_rate, cfg.out_channels, cfg.out_format = 8, 1024, 48000, 2, 16
    cfg.matrix = A.get_h2m_matrix(3, A.SS["BINAURAL"])
    cfg.limiter_enable, cfg.limiter_threshold_db = 1, -1.0
    h = C.c_void_p()
    assert L.iamf_hip_shard_create(C.byref(cfg), None, 0, C.byref(h)) == -1      # no devices asked for
    assert L.iamf_hip_shard_create(C.byref(cfg), None, 16, C.byref(h)) == -1     # more devices than streams
    assert L.iamf_hip_shard_create(C.byref(cfg), None, 1, C.byref(h)) == -100    # IAMF_HIP_ERR_DEVICE: no GPU here
    assert not h.value
    assert L.iamf_hip_shard_devices(None) == 0
    assert isinstance(L.iamf_hip_shard_rccl_version(), bytes)                   # "" or a version; loading it needs no GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iac_amd.sharding import GatherPipeline, shard_streams
    lo, hi = shard_streams(10, world, rank)
    bufs = [torch.zeros((5, 8), dtype=torch.uint8) for _ in range(2)]
    pipe = GatherPipeline(bufs, world, rank)
    ok = True
    for i in range(steps):
        def render(buf, i=i):
            buf.fill_(100 * rank + i)  # stands in for the render kernel writing packed PCM
            return hi - lo
        n = pipe.step(render)
        ok &= n == 5
        if rank == 0 and i >= 1:
            pipe.wait_slot((i - 1) % 2)
            got = pipe.gathered(i - 1)
            ok &= all(bool((got[r] == 100 * r + (i - 1)).all()) for r in range(world))
    pipe.drain()
    if rank == 0:
        got = pipe.gathered(steps - 1)
        ok &= all(bool((got[r] == 100 * r + (steps - 1)).all()) for r in range(world))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gather_pipeline_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_generated_lfe_chain_loop_is_what_the_generator_writes(tmp_path):
    # iac_amd/csrc/lfe_chain_asm.inc (the hand-assigned-register main loop of lfe_chain_kernel) is generated code kept in
    # the tree: the committed file must be the generator's output, instruction for instruction
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "lfe_chain_asm.inc"
    env = dict(os.environ)
    env.pop("LFE_ASM_FMA", None)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_lfe_chain_asm.py"), str(out)], env=env,
                          stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(root, "iac_amd", "csrc", "lfe_chain_asm.inc")).read()


# ---- iamf_shard.hip's N > 1 path, run before any multi-GPU box does ----
# iac_amd/csrc/iamf_shard.hip is host code: it is compiled here with g++ against tests/shard_stub/fake_hip.cpp (a HIP
# runtime stand-in with 8 "devices" whose streams are real asynchronous queues and whose events have HIP's semantics) and
# loads tests/shard_stub/fake_rccl.cpp through IAMF_HIP_RCCL_LIB (send / recv matched inside the group, copies done when
# both streams have arrived).  Under AddressSanitizer + UBSan and under ThreadSanitizer.  What the driver checks: every
# byte of every step's gathered buffer (rows from the right device, stream and step — step i's gather runs beside step
# i + 1's render into the SAME PCM buffers, nothing waits in between), nothing written between the rows, the per-peer byte
# accounting, no HIP object left behind.  The stand-ins were checked to be sensitive: without the render's wait on
# `gathered`, without the gather's wait on `rendered`, or with the receive offsets indexed by device instead of by first
# stream, the driver reports MISMATCH (ASan build) / a data race (TSan build).
SHARD_STUB = os.path.join(ROOT, "tests", "shard_stub")


def _build_shard_rehearsal(kind):
    import subprocess
    b = os.path.join(SHARD_STUB, "build_" + kind)
    os.makedirs(b, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if kind == "asan" else ["-fsanitize=thread"]
    flags = ["-std=c++17", "-g", "-O1", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
             "-I/opt/rocm/include"] + san
    srcs = [os.path.join(SHARD_STUB, x) for x in ("fake_hip.cpp", "fake_rccl.cpp", "shard_driver.cpp")]
    srcs += [os.path.join(ROOT, "iac_amd", "csrc", "iamf_shard.hip"), os.path.join(ROOT, "include", "iamf_hip.h")]
    exe = os.path.join(b, "shard_driver")
    if os.path.exists(exe) and all(os.path.getmtime(x) <= os.path.getmtime(exe) for x in srcs):
        return b
    rpath = "-Wl,-rpath,$ORIGIN"
    subprocess.check_call(["g++"] + flags + ["-fPIC", "-shared", srcs[0], "-o", os.path.join(b, "libfakehip.so"), "-lpthread"])
    subprocess.check_call(["g++"] + flags + ["-fPIC", "-shared", srcs[1], "-o", os.path.join(b, "librccl_fake.so"), "-L" + b, "-lfakehip", rpath])
    subprocess.check_call(["g++"] + flags + ["-x", "c++", srcs[3], srcs[2], "-o", exe, "-L" + b, "-lfakehip", "-ldl", "-lpthread", rpath])
    return b


@pytest.fixture(scope="module", params=["asan", "tsan"])
def shard_rehearsal(request):
    return request.param, _build_shard_rehearsal(request.param)


def _run_shard(rehearsal, args, env_extra=None):
    import subprocess
    kind, b = rehearsal
    env = dict(os.environ, IAMF_HIP_RCCL_LIB=os.path.join(b, "librccl_fake.so"), ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               TSAN_OPTIONS="halt_on_error=1")
    env.update(env_extra or {})
    r = subprocess.run([os.path.join(b, "shard_driver")] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)
    if "LeakSanitizer has encountered a fatal error" in r.stderr:   # LeakSanitizer cannot always stop the world in a container
        r = subprocess.run([os.path.join(b, "shard_driver")] + [str(a) for a in args], capture_output=True, text=True, timeout=300,
                           env=dict(env, ASAN_OPTIONS="detect_leaks=0"))
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("streams,devices,root,steps,ordinals", [
    (10, 4, 0, 3, None),            # uneven split 3 / 3 / 2 / 2
    (10, 4, 2, 4, "3,1,6,0"),       # root != 0, device ordinals that are not the shard indices
    (7, 2, 1, 3, None),             # N = 2
    (67, 8, 5, 4, None),            # N = 8
    (8, 8, 7, 2, "7,6,5,4,3,2,1,0"),   # one stream per device
    (11, 3, 1, 9, None),            # more gathers in flight than the shard keeps timing event pairs for
])
def test_shard_n_devices_gather_rows_beside_the_next_render(shard_rehearsal, streams, devices, root, steps, ordinals):
    out = _run_shard(shard_rehearsal, [streams, devices, root, steps, "rows"] + ([ordinals] if ordinals else []))
    assert out[0] == "create 0 devices %d" % devices
    assert out[-2].startswith("ok %d destinations, %d streams over %d devices, root %d" % (steps + 1, streams, devices, root))
    assert out[-1] == "clean"
    # per-peer bytes of the last gather (the flush: 240 sample-frames x 2 channels x 2 bytes per stream), root receives all
    sent = [int(l.split()[4]) for l in out if l.startswith("device ")]
    recv = [int(l.split()[10]) for l in out if l.startswith("device ")]
    q, r = divmod(streams, devices)
    assert sent == [960 * (q + (1 if i < r else 0)) for i in range(devices)]
    assert recv == [960 * streams if i == root else 0 for i in range(devices)]


def test_shard_whole_region_gather_n2_and_n4(shard_rehearsal):
    assert _run_shard(shard_rehearsal, [7, 2, 1, 3, "whole"])[-1] == "clean"
    assert _run_shard(shard_rehearsal, [10, 4, 3, 2, "whole"])[-1] == "clean"


def test_shard_destroy_with_a_gather_in_flight(shard_rehearsal):
    out = _run_shard(shard_rehearsal, [10, 4, 1, 3, "destroy"])
    assert out[-1] == "clean" and out[-2].startswith("ok 3 destinations")


def test_shard_failing_peer_is_reported_and_the_shard_goes_on(shard_rehearsal):
    out = _run_shard(shard_rehearsal, [10, 4, 0, 3, "fail"], {"FAKE_RCCL_FAIL_SEND_RANK": "2", "FAKE_RCCL_FAIL_AT_GROUP": "0"})
    assert "gather step 0 under a failing peer: -100" in out   # IAMF_HIP_ERR_DEVICE
    assert out[-2].endswith("failures seen 1") and out[-1] == "clean"


def test_shard_refuses_more_devices_than_streams_and_comm_init_failure(shard_rehearsal):
    assert _run_shard(shard_rehearsal, [3, 4, 0, 1, "rows"])[0].startswith("create -1")   # IAMF_HIP_ERR_BAD_ARG
    import subprocess
    kind, b = shard_rehearsal
    env = dict(os.environ, IAMF_HIP_RCCL_LIB=os.path.join(b, "librccl_fake.so"), FAKE_RCCL_FAIL_INIT="1", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([os.path.join(b, "shard_driver"), "6", "3", "0", "1", "rows"], capture_output=True, text=True, timeout=120, env=env)
    assert "MISMATCH gather step 0 returned" in r.stdout and "ncclCommInitAll failed" in r.stderr
