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
