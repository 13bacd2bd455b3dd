"""Multi-GPU host logic: one process per GPU, boards shard across ranks.

Pure host code (no GPU needed), so that the N > 1 path is testable with gloo on CPU:
  * shard_boards          contiguous board ranges per rank;
  * replicated_allreduce  the arithmetic of rs_allreduce_replicated: x = snap + allreduce_sum(x - snap),
                          wrapping i32 (order-independent) or f32.
"""
import numpy as np


def shard_boards(n_boards, rank, world):
    """Contiguous [lo, hi) of the board axis owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_boards, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def replicated_allreduce(x, snap, all_reduce_sum):
    """x, snap: numpy int32 or float32 arrays; all_reduce_sum(array) -> element-wise sum over ranks.
    Mirrors csrc/rs_comm.cpp: the (negated) own delta is reduced, then applied to the snapshot."""
    if x.dtype == np.int32:
        neg_delta = (snap.view(np.uint32) - x.view(np.uint32)).view(np.int32)           # snap - x, wrapping
        total = all_reduce_sum(neg_delta)
        return (snap.view(np.uint32) - np.asarray(total, dtype=np.int32).view(np.uint32)).view(np.int32)
    neg_delta = (snap - x).astype(np.float32)
    restored = (x + neg_delta).astype(np.float32)
    total = np.asarray(all_reduce_sum(neg_delta), dtype=np.float32)
    return (restored - total).astype(np.float32)
