"""Sharding of independent transforms / channels across the GPUs of one node.

The path shards by batches: every transform and every convolution
channel is independent, so rank g owns the contiguous block [start, start+count) of the
batch axis and runs it on its own GPU with no data-path collective.  torch.distributed
(backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests) only carries the
end-of-run reductions: max-over-ranks time and a checksum that validates the sharded run.
"""


def shard_range(total, rank, world):
    """contiguous block split: (start, count) of rank's share; ragged totals spread the
    remainder over the first ranks"""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request total=%r rank=%r world=%r" % (total, rank, world))
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class ShardedBatch:
    """one rank's view of a global batch"""

    def __init__(self, total, rank=None, world=None):
        import torch.distributed as dist
        self.active = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size() if self.active else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if self.active else 0)
        self.total = total
        self.start, self.count = shard_range(total, self.rank, self.world)

    def slice(self):
        return slice(self.start, self.start + self.count)

    def barrier(self):
        if self.active and self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def reduce_max(self, value, device="cpu"):
        """max over ranks of a scalar (elapsed time)"""
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        if self.active and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(self, value, device="cpu"):
        """sum over ranks of a scalar or tensor (checksums)"""
        import torch
        t = value.clone().to(torch.float64) if hasattr(value, "clone") else torch.tensor([float(value)], dtype=torch.float64, device=device)
        if self.active and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
