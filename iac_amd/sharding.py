"""Multi-GPU plumbing for the batched renderer: how streams shard over ranks and how the packed
PCM of every step reaches rank 0.

Streams are independent (all state is per stream), so ranks never talk while rendering.  The one
exchange of a job is the gather of packed PCM to rank 0 (RCCL over xGMI on the GPU box: the "nccl"
backend; gloo in the CPU tests).  It is issued asynchronously on a double-buffered PCM tensor so
the transfer of step i overlaps the render of step i+1.
"""
import torch.distributed as dist


def shard_streams(n_total, world, rank):
    """contiguous block of stream ids [lo, hi) owned by `rank` (sizes differ by at most one)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GatherPipeline:
    """Double-buffered render -> async gather-to-rank-0.

    render_fn(buffer) fills `buffer` (one of the two tensors in `buffers`) for the next step and
    returns whatever it likes.  step() waits only for the gather that last read that buffer.
    """

    def __init__(self, buffers, world, rank, enabled=True, make_recv=None, single_rank_too=False):
        assert len(buffers) == 2
        self.buffers = buffers
        self.world, self.rank = world, rank
        # (single_rank_too: a one-rank job has nothing to gather; tests/nccl_one_rank.py runs the collective anyway)
        self.enabled = enabled and (world > 1 or single_rank_too)
        self.pending = [None, None]
        self.recv = None
        if self.enabled and rank == 0:
            mk = make_recv or (lambda t: t.new_empty(t.shape))
            self.recv = [[mk(buffers[b]) for _ in range(world)] for b in range(2)]
        self.i = 0

    def wait_slot(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def step(self, render_fn):
        b = self.i % 2
        self.wait_slot(b)
        out = render_fn(self.buffers[b])
        if self.enabled:
            self.pending[b] = dist.gather(self.buffers[b], self.recv[b] if self.rank == 0 else None,
                                          dst=0, async_op=True)
        self.i += 1
        return out

    def drain(self):
        for b in range(2):
            self.wait_slot(b)

    def gathered(self, step_index):
        """rank 0: the list (one tensor per rank) received for a finished step"""
        return self.recv[step_index % 2] if self.recv is not None else None
