"""Multi-GPU host logic: one process per GPU, boards shard across ranks.

Pure host code (no GPU needed), so that the N > 1 path is testable with gloo on CPU:
  * shard_boards          contiguous board ranges per rank;
  * deal_numbers          which deals of global batch b a rank of a data-parallel deal trainer samples (csrc/rs_trainer.cpp);
  * apply_summed_deltas   the arithmetic between sweep and apply of a data-parallel deal batch: table + allreduce_sum(delta), wrapping i32;
  * replicated_allreduce  the arithmetic of rs_allreduce_replicated: x = snap + allreduce_sum(x - snap),
                          wrapping i32 (order-independent) or f32.
  * exchange_items        the other half of a data-parallel sweep's exchange (csrc/rs_solver.cpp solver_exchange_deltas): the non-zero deltas of the rounds whose rows
                          go straight into the table travel as (cell, delta) items -- counts all-gathered, items all-gathered padded to the longest rank's, every rank's
                          items added (wrapping i32: any order, same bits).
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
    total = np.asarray(all_reduce_sum(neg_delta), dtype=np.float32)
    # every rank restarts from the snapshot ITSELF (k_delta_swap), not from x + (snap - x): that is not snap in f32 and depends on the rank's own x
    return (snap - total).astype(np.float32)


def deal_numbers(batch, rank, world, deals_per_batch):
    """(first deal number, lane base of the sampling hash) of `rank` in global batch `batch`: mirrors rs_deal_trainer_deal and
    rs_solver_params.deal_offset"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (batch * world + rank) * deals_per_batch, rank * deals_per_batch


def apply_summed_deltas(table, delta, all_reduce_sum):
    """table, delta: int32 arrays (delta = this rank's sweep result); all ranks end with table + sum of all deltas (wrapping)"""
    total = np.asarray(all_reduce_sum(np.ascontiguousarray(delta, dtype=np.int32)), dtype=np.int32)
    return (table.view(np.uint32) + total.view(np.uint32)).view(np.int32)


def exchange_items(table, delta, all_gather):
    """table, delta: int32 arrays of one shape (delta = this rank's sweep result, mostly zero); all_gather(array) -> list of every rank's array (equal shapes), rank order.
    Returns table + the non-zero deltas of ALL ranks, moved as items the way rs_iterate moves them under a communicator, and the number of item words this rank received."""
    flat = np.ascontiguousarray(delta, dtype=np.int32).ravel()
    cells = np.flatnonzero(flat).astype(np.uint32)
    counts = [int(c[0]) for c in all_gather(np.array([len(cells)], dtype=np.int64))]
    most = max(counts)
    mine = np.zeros((most, 2), dtype=np.uint32)                 # the tail beyond a rank's count is sent and never read
    mine[: len(cells), 0] = cells
    mine[: len(cells), 1] = flat[cells].view(np.uint32)
    out = np.ascontiguousarray(table, dtype=np.int32).ravel().view(np.uint32).copy()
    for r, items in enumerate(all_gather(mine)):
        items = np.asarray(items, dtype=np.uint32)[: counts[r]]
        np.add.at(out, items[:, 0].astype(np.int64), items[:, 1])   # wrapping adds
    return out.view(np.int32).reshape(np.shape(table)), most * 2 * len(counts)
