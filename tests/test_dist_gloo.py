"""CPU, world_size 2, gloo: the N > 1 host path.  (i) boards shard with NO collective: every rank runs its own
board range and the union equals the single-process result; (ii) the replicated-round all-reduce arithmetic of
rs_allreduce_replicated (x = snapshot + sum of per-rank deltas, wrapping i32) gives every rank the same table,
equal to applying all deltas on one process."""
import os
import socket

import numpy as np
import pytest

from oracle import orc
from rustsolver_amd.dist import apply_summed_deltas, deal_numbers, exchange_items, replicated_allreduce, shard_boards

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
        # f32 with rank-specific deltas much larger than the snapshot: x + (snap - x) != snap there, so a restore-by-addition would differ per rank
        snap_small = rng.uniform(-1, 1, size=(3, 40)).astype(np.float32)
        big = [rng.uniform(-1e4, 1e4, size=(3, 40)).astype(np.float32) for _ in range(WORLD)]
        xg_new = replicated_allreduce((snap_small + big[rank]).astype(np.float32), snap_small, allreduce)
        q.put((rank, gathered if rank == 0 else None, x_new, xf_new, xg_new))
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
        assert res[r][4].tobytes() == res[0][4].tobytes(), "replicated f32 rounds must stay bit-identical across ranks"


# ---- data-parallel deal batches: world 2 over gloo == one process with the union batch ------------------
DP_N, DP_BATCHES, DP_SEED = 300, 3, 17
DP_SIZES = [(13, 17)]


def _dp_inputs():
    rng = np.random.Generator(np.random.PCG64(99))
    mask = 0b11111 << 10
    hands = np.array([(a, b) for a in range(52) for b in range(a) if not ((mask >> a) & 1 or (mask >> b) & 1)], dtype=np.uint8)[::9]
    tree = orc.OracleTree(orc.options_default_river())
    init = {}
    for d in tree.as_dicts():
        if d["kind"] == orc.ACTION:
            a, n = len(d["children"]), DP_SIZES[0][d["player"]]
            init[d["index"]] = (rng.integers(-10**6, 10**6, size=(a, n)).astype(np.int32), rng.integers(0, 10**6, size=(a, n)).astype(np.int32))
    return tree, init, mask, hands


def _dp_run(tree, init, mask, hands, rank, world, all_reduce_sum, all_gather=None):
    """the CPU mirror of rs_deal_trainer_train with `world` ranks: deal this rank's numbers, sweep against the replicated table, exchange the
    i32 deltas of the ranks -- nodes with an even index summed as whole cell ranges (the packed all-reduce), nodes with an odd index as all-gathered
    (cell, delta) items (the rounds whose rows go straight into the table: rs_solver.cpp solver_exchange_deltas) -- and apply"""
    n = DP_N
    tb = orc.OracleDealTable(tree, DP_SIZES)
    for idx, (R, S) in init.items():
        tb.set_node(idx, R, S)
    cidx = {(0, p): np.zeros(n, dtype=np.uint32) for p in (0, 1)}
    sign = np.zeros(n, dtype=np.float32)
    leaves = {d["id"]: (orc.LEAF_SIGN, sign) for d in tree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    _, lane_base = deal_numbers(0, rank, world, n)
    sol = orc.OracleDealSolver(tree, tb, leaves, cidx, n, lane_base=lane_base, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE,
                               base_seed=DP_SEED)
    for b in range(DP_BATCHES):
        first, _ = deal_numbers(b, rank, world, n)
        cards = orc.generate_hands(DP_SEED, first, mask, hands, hands, n)
        for p in (0, 1):   # stand-in for get_cluster: any deterministic function of the cards will do for the host-path test
            cidx[(0, p)][:] = (cards[5 + 2 * p].astype(np.uint32) * 52 + cards[6 + 2 * p]) % DP_SIZES[0][p]
        sign[:] = orc.showdown_sign(cards)
        for player in (0, 1):
            before = {idx: tb.get_node(idx) for idx in init}
            sol.iterate(player)                                   # sweep + local apply ...
            for idx in init:                                      # ... turned back into (table, delta), reduced, applied
                after = tb.get_node(idx)
                deltas = [(after[k].view(np.uint32) - before[idx][k].view(np.uint32)).view(np.int32) for k in (0, 1)]
                if all_gather is not None and idx % 2:
                    new = [exchange_items(before[idx][k], deltas[k], all_gather)[0] for k in (0, 1)]
                else:
                    new = [apply_summed_deltas(before[idx][k], deltas[k], all_reduce_sum) for k in (0, 1)]
                tb.set_node(idx, new[0], new[1])
    return {idx: tb.get_node(idx) for idx in init}


def _dp_worker(rank, port, q):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    try:
        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()
        def allgather(a):
            a = np.ascontiguousarray(a)
            as_u32 = a.dtype == np.uint32                      # gloo knows no unsigned 32-bit type: the same bits as int32
            t = torch.from_numpy((a.view(np.int32) if as_u32 else a).copy())
            outs = [torch.empty_like(t) for _ in range(WORLD)]
            dist.all_gather(outs, t)
            return [o.numpy().view(np.uint32) if as_u32 else o.numpy() for o in outs]
        tree, init, mask, hands = _dp_inputs()
        q.put((rank, _dp_run(tree, init, mask, hands, rank, WORLD, allreduce, allgather)))
    finally:
        dist.destroy_process_group()


def test_world2_data_parallel_deal_batches_equal_the_union_batch():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one process, world = 1, batches of WORLD * DP_N deals
    global DP_N
    tree, init, mask, hands = _dp_inputs()
    DP_N *= WORLD
    try:
        want = _dp_run(tree, init, mask, hands, 0, 1, lambda a: a, lambda a: [a])
    finally:
        DP_N //= WORLD
    for r in range(WORLD):
        for idx in init:
            assert (res[r][idx][0] == want[idx][0]).all() and (res[r][idx][1] == want[idx][1]).all(), (r, idx)


def test_deal_numbers_tile_the_global_batches():
    for world in (1, 2, 3, 8):
        n = 5
        seen = []
        for b in range(3):
            for r in range(world):
                first, base = deal_numbers(b, r, world, n)
                assert base == r * n
                seen += list(range(first, first + n))
        assert seen == list(range(3 * world * n))
    with pytest.raises(ValueError):
        deal_numbers(0, 2, 2, 4)


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
