"""Multi-GPU sharding of a batch of independent MPC problems: one process per GPU.

The problems never exchange data during a solve, so the only communication
is the final gather of the control sequences to rank 0: every peer sends its [N, B_local] block
straight to the root (`torch.distributed.gather` = grouped send/recv on RCCL, all xGMI links into
the root in parallel), never a ring.  Works on any backend (tests use gloo on CPU tensors).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous split of `total` problems: rank r owns [lo, hi).  Sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ResultGather:
    """Gathers equally sized [N, B_local] result blocks to `dst`, double-buffered so that the
    gather of step i overlaps the solve of step i+1 (the collective runs on RCCL's stream)."""

    def __init__(self, n_rows, b_local, dtype, device, dst=0, depth=2, force=False, via_host=False):
        """force=True runs the collective even in a world of one (exercises the backend on a single GPU).
        via_host=True copies each block to the host first (a gloo process group next to GPU results: the
        rehearsal of several ranks on one device, which RCCL refuses)."""
        self.dst = dst
        self.via_host = via_host
        if via_host:
            device = "cpu"
        self.depth = depth
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.pending = [None] * depth
        self.recv = None
        if self.active and self.rank == dst:
            self.recv = [[torch.empty((n_rows, b_local), dtype=dtype, device=device) for _ in range(self.world)]
                         for _ in range(depth)]
        self.i = 0

    def slot(self):
        return self.i % self.depth

    def wait_slot(self, slot):
        w = self.pending[slot]
        if w is not None:
            w.wait()
            self.pending[slot] = None

    def submit(self, block):
        """Start gathering `block` (this rank's results for the current step)."""
        slot = self.slot()
        self.wait_slot(slot)
        if self.active:
            if self.via_host:
                block = block.cpu()
            self.pending[slot] = dist.gather(block, self.recv[slot] if self.rank == self.dst else None,
                                             dst=self.dst, async_op=True)
        self.i += 1
        return slot

    def finish(self):
        for s in range(self.depth):
            self.wait_slot(s)

    def assembled(self, slot):
        """On dst: [N, world * B_local] in global problem order (rank-major); else None."""
        if self.recv is None:
            return None
        return torch.cat(self.recv[slot], dim=1)


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(values, device):
    """Every rank's list of python floats, on every rank: [[rank 0's values], [rank 1's], ...] (per-rank timings for
    the benchmark line; one small all-gather outside the timed region)."""
    vals = [float(v) for v in values]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [vals]
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu().tolist()] for o in out]
