"""TEST HELPER (run as a child process by tests/test_gpu_nccl_one_rank.py): the collectives bench.py issues at N > 1 —
init_process_group("nccl", device_id=...), all_gather of the device indices, the per-step asynchronous gather of the packed
PCM (GatherPipeline), the final gather, all_reduce(MAX) of the timing, barriers — with ONE rank on the one GPU a test box has,
so that torch.distributed's RCCL backend has been initialised and has moved bytes through this code before the first
multi-GPU run.  Prints "ok <rccl version>"."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iac_amd.sharding import GatherPipeline  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29571")
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert (world, rank) == (1, 0)
    mine = torch.tensor([0], dtype=torch.int64, device=dev)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    assert [int(t.item()) for t in seen] == [0]
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    bufs = [torch.randint(0, 256, (64, 4096 + 128), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
    want = [b.clone() for b in bufs]
    pipe = GatherPipeline(bufs, world, rank, enabled=True, single_rank_too=True)
    for i in range(5):
        pipe.step(lambda buf: buf.add_(1))          # "render": the buffer changes every step
        want[i % 2] = want[i % 2] + 1
    pipe.drain()
    torch.cuda.synchronize()
    for b in range(2):
        assert torch.equal(pipe.gathered(b)[0], want[b]), b
    recv = [torch.empty_like(bufs[0])]
    dist.gather(bufs[0], recv, dst=0)                # the job's final gather
    torch.cuda.synchronize()
    assert torch.equal(recv[0], bufs[0])
    tmax = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    assert float(tmax.item()) == 1.25
    dist.barrier()
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    dist.destroy_process_group()
    print("ok", ver)


if __name__ == "__main__":
    main()
