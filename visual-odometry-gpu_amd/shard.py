"""Frame-parallel sharding of a frame stream over ranks (one process per GPU).

The ORB path has no cross-frame state (src/orb.cpp:58-109 keeps only the
pyramid of the current call), so frames are independent units: rank r of W
processes frames [r*per_rank, (r+1)*per_rank) of the stream and there is NO
data-path collective.  torch.distributed (backend "nccl" == RCCL over xGMI on
the GPUs, "gloo" in the CPU tests) is used only for
  * the barrier around the timed region,
  * MAX over ranks of the elapsed time,
  * SUM of keypoint counts and an order-independent checksum of all
    descriptors, which lets an N-rank run be compared with a 1-rank run of the
    same global stream.
Backend-agnostic on purpose: the same functions run under gloo in
tests/test_sharding_gloo.py and under RCCL in bench.py.
"""
import numpy as np


def frame_range(rank, world, frames_per_rank):
    """Global frame indices owned by `rank` (weak scaling: fixed work per rank)."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad rank/world/frames_per_rank")
    first = rank * frames_per_rank
    return first, first + frames_per_rank


def split_stream(n_total, rank, world):
    """Strong-scaling split of a fixed stream of n_total frames: contiguous
    blocks, sizes differ by at most one, every frame owned exactly once."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad rank/world/n_total")
    base, extra = divmod(n_total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def descriptor_checksum(counts, desc):
    """Order-independent 64-bit checksum of the valid descriptors of a batch
    (sum of per-descriptor 64-bit folds, modulo 2^63): invariant under how
    frames are distributed over ranks."""
    counts = np.asarray(counts)
    total = 0
    for i, c in enumerate(counts):
        d = np.ascontiguousarray(desc[i][: int(c)]).reshape(-1, 4, 8).view(np.uint64).reshape(-1, 4)
        if len(d):
            fold = d[:, 0] ^ (d[:, 1] * np.uint64(3)) ^ (d[:, 2] * np.uint64(5)) ^ (d[:, 3] * np.uint64(7))
            total = (total + int(fold.astype(np.uint64).sum(dtype=np.uint64))) & 0x7FFFFFFFFFFFFFFF
    return total


class Group:
    """Thin wrapper over torch.distributed that degrades to a no-op for world == 1."""

    def __init__(self, world, device=None):
        self.world = world
        self.device = device
        if world > 1:
            import torch.distributed as dist

            self.dist = dist

    def _tensor(self, vals, dtype):
        import torch

        return torch.tensor(vals, dtype=dtype, device=self.device if self.device is not None else "cpu")

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_float(self, v):
        if self.world == 1:
            return float(v)
        import torch

        t = self._tensor([float(v)], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, v):
        if self.world == 1:
            return int(v)
        import torch

        t = self._tensor([int(v)], torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def sum_checksum(self, v):
        """Sum modulo 2^63 of per-rank checksums (kept in int64 range at every step)."""
        if self.world == 1:
            return int(v) & 0x7FFFFFFFFFFFFFFF
        import torch

        # split into two 31-bit halves + carry-safe sums so the int64 all-reduce cannot overflow
        lo, hi = int(v) & 0x7FFFFFFF, (int(v) >> 31) & 0xFFFFFFFF
        t = self._tensor([lo, hi], torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        lo, hi = int(t[0].item()), int(t[1].item())
        return (lo + (hi << 31)) & 0x7FFFFFFFFFFFFFFF
