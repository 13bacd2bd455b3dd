"""CPU, world_size 2, gloo: the N > 1 host path.  (i) boards shard with NO collective: every rank runs its own
board range and the union equals the single-process result; (ii) the replicated-round all-reduce arithmetic of
rs_allreduce_replicated (x = snapshot + sum of per-rank deltas, wrapping i32) gives every rank the same table,
equal to applying all deltas on one process."""
import os
import socket

import numpy as np
import pytest

from oracle import orc
from rustsolver_amd.dist import replicated_allreduce, shard_boards

WORLD = 2
C, B_TOTAL = 6, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    rng = np.random.Generator(np.random.PCG64(2024))
    tree = orc.OracleTree(orc.options_default_river())
    init = {}
    for d in tree.as_dicts():
        if d["kind"] == orc.ACTION:
            a = len(d["children"])
            init[d["index"]] = (rng.integers(-10**6, 10**6, size=(a, B_TOTAL * C)).astype(np.int32),
                                rng.integers(0, 10**6, size=(a, B_TOTAL * C)).astype(np.int32))
    sign = rng.integers(-1, 2, size=B_TOTAL * C).astype(np.float32)
    return tree, init, sign


def _run_shard(tree, init, sign, lo, hi):
    nb = hi - lo
    tb = orc.OracleTable(tree, [nb], C)
    cols = slice(lo * C, hi * C)
    for idx, (R, S) in init.items():
        tb.set_node(idx, R[:, cols], S[:, cols])
    leaves = {d["id"]: (orc.LEAF_SIGN, sign[cols]) for d in tree.as_dicts()
              if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    sol = orc.OracleSolver(tree, tb, leaves, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    utils = []
    for it in range(2):
        for p in (0, 1):
            utils.append(sol.iterate(p))
    out = {idx: tb.get_node(idx) for idx in init}
    return out, np.stack(utils)


def _worker(rank, port, q):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    try:
        tree, init, sign = _inputs()
        # (i) sharded boards, no collective on the data path
        lo, hi = shard_boards(B_TOTAL, rank, WORLD)
        out, utils = _run_shard(tree, init, sign, lo, hi)
        gathered = [None] * WORLD
        dist.all_gather_object(gathered, (lo, hi, out, utils))          # test plumbing only
        # (ii) replicated round: same snapshot everywhere, rank-specific deltas
        rng = np.random.Generator(np.random.PCG64(7))                    # same stream on every rank
        snap = rng.integers(-2**31, 2**31 - 1, size=(3, 40)).astype(np.int32)
        deltas = [rng.integers(-2**31, 2**31 - 1, size=(3, 40)).astype(np.int32) for _ in range(WORLD)]
        x = (snap.view(np.uint32) + deltas[rank].view(np.uint32)).view(np.int32)

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()
        x_new = replicated_allreduce(x, snap, allreduce)
        xf = snap.astype(np.float32) + np.float32(rank + 1)
        xf_new = replicated_allreduce(xf, snap.astype(np.float32), allreduce)
        q.put((rank, gathered if rank == 0 else None, x_new, xf_new))
    finally:
        dist.destroy_process_group()


def test_world2_board_sharding_and_replicated_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(WORLD):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    tree, init, sign = _inputs()
    full, full_utils = _run_shard(tree, init, sign, 0, B_TOTAL)
    covered = 0
    for lo, hi, out, utils in res[0][1]:
        covered += hi - lo
        cols = slice(lo * C, hi * C)
        for idx in init:
            assert (out[idx][0] == full[idx][0][:, cols]).all() and (out[idx][1] == full[idx][1][:, cols]).all()
        assert (utils.view(np.uint32) == full_utils[:, cols].view(np.uint32)).all()
    assert covered == B_TOTAL

    rng = np.random.Generator(np.random.PCG64(7))
    snap = rng.integers(-2**31, 2**31 - 1, size=(3, 40)).astype(np.int32)
    deltas = [rng.integers(-2**31, 2**31 - 1, size=(3, 40)).astype(np.int32) for _ in range(WORLD)]
    want = snap.view(np.uint32).copy()
    for d in deltas:
        want = want + d.view(np.uint32)
    for r in range(WORLD):
        assert (res[r][2].view(np.uint32) == want).all()                # identical on every rank, = all deltas applied
        assert np.allclose(res[r][3], snap.astype(np.float32) + 3.0, rtol=1e-5)


def test_shard_boards_partitions():
    for n in (1, 7, 8, 9, 2352, 9216):
        for w in (1, 2, 4, 8):
            edges = [shard_boards(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_boards(8, 8, 8)
