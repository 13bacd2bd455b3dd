"""GPU (-m gpu): the library's multi-rank paths run by SEVERAL REAL PROCESSES on one GPU (task 3).

RCCL refuses two ranks on one device and the boxes hold one card, so until round 4 the collectives inside a sharded rs_iterate / a data-parallel trainer had only ever
executed with one rank, where an all-gather and an all-reduce are no-ops: a wrong slot offset, or a phase-1 kernel that does not wait for the collective's stream, could not
fail any test (the in-process emulations copy the slots by hand).  Here every rank is a fresh process (tests/_multiproc_worker.py) that loads tests/libstub_rccl.so through the
test-only RS_RCCL_LIB override -- the five RCCL entry points over POSIX shared memory + hipMemcpyAsync on the caller's stream, asynchronous like the real ones -- and runs the
product's own calls: rs_comm_create, rs_solver_attach_comm + rs_iterate, rs_deal_trainer_attach_comm + rs_deal_trainer_train.  The parent compares with ONE process.
At most 6 processes use the card at once (the pool's process guard): worlds 2, 3 and 5."""
import os
import subprocess
import sys

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import abstraction as ab

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_SRC = os.path.join(ROOT, "tests", "stub_rccl.c")
STUB_SO = os.path.join(ROOT, "tests", "libstub_rccl.so")
pytestmark = pytest.mark.gpu


def build_stub():
    if not os.path.exists(STUB_SO) or os.path.getmtime(STUB_SO) < os.path.getmtime(STUB_SRC):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", STUB_SRC, "-o", STUB_SO,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"])
    return STUB_SO


def run_ranks(case, workdir, world, timeout=300, extra_env=None):
    """start `world` fresh worker processes (they share GPU 0), wait, return their result files"""
    env = dict(os.environ, RS_RCCL_LIB=build_stub(), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    ident = ("/rs_stub_test_%d_%s" % (os.getpid(), os.urandom(6).hex())).encode().hex()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_multiproc_worker.py"), case, str(workdir), str(world), str(r), ident], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [np.load(os.path.join(workdir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,fuse,dtype", [(2, 1, "i32"), (3, 1, "i32"), (5, 1, "i32"), (3, 0, "i32"), (2, 1, "f32"), (3, 1, "f16")])
def test_sharded_sweep_in_real_processes_equals_one_gpu(world, fuse, dtype, tmp_path):
    """BASELINE configs[3] in small: turn and river boards sharded over `world` PROCESSES, flop replicated, one in-place all-gather of the turn roots' utility rows per sweep,
    issued by rs_iterate itself between its two phases.  Every rank must end with the unsharded run's flop table and its slice of the turn / river tables, and every root utility
    it returns must equal the single-process one -- bit for bit, float tables included (sums are never split over ranks)."""
    from tests.test_gpu_parity import setup_pair
    from rustsolver_amd.dist import shard_boards
    Cn, G = 8, [1, 5, 10]
    fan_river = G[2] // G[1]
    dt_g, dt_o = {"i32": (rs.I32, orc.T_I32), "f32": (rs.F32, orc.T_F32), "f16": (rs.F16, orc.T_F16)}[dtype]
    scale, mode = (10000.0, rs.UPD_WRAP_I32) if dtype == "i32" else (2.0 ** -6, rs.UPD_CLAMP_I64)
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), G, Cn, 99, dt_g, dt_o)
    inputs = dict(Cn=Cn, G=np.array(G), fuse=fuse, iters=2, scale=scale, mode=mode)
    for nd in tree.action_nodes():
        inputs["R%d" % nd.index], inputs["S%d" % nd.index] = table.download_node(nd.index)
    for i, nd in enumerate(tree.nodes):
        if i in lg:
            r = tree.nodes[nd.parent].round_idx
            if "sign%d" % r not in inputs:
                inputs["sign%d" % r] = table.read_lane_buffer(lg[i][1], tree.nodes[nd.parent].index)[0].copy()
    np.savez(os.path.join(tmp_path, "inputs.npz"), **inputs)
    ref = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mode, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=fuse)
    want_util = {(it, p): ref.iterate(p, want_root_util=True).copy() for it in range(2) for p in (0, 1)}
    ranks = run_ranks("sharded-" + dtype, tmp_path, world)
    for g, got in enumerate(ranks):
        tlo, thi = shard_boards(G[1], g, world)
        cols = {0: slice(0, Cn), 1: slice(tlo * Cn, thi * Cn), 2: slice(tlo * fan_river * Cn, thi * fan_river * Cn)}
        for (it, p), w in want_util.items():
            assert got["util_%d_%d" % (it, p)].view(np.uint32).tolist() == w.view(np.uint32).tolist(), "root utility, rank %d, iteration %d, player %d" % (g, it, p)
        for nd in tree.action_nodes():
            R, S = table.download_node(nd.index)
            assert got["R%d" % nd.index].tobytes() == np.ascontiguousarray(R[:, cols[nd.round_idx]]).tobytes(), "regrets of node %d on rank %d" % (nd.index, g)
            assert got["S%d" % nd.index].tobytes() == np.ascontiguousarray(S[:, cols[nd.round_idx]]).tobytes(), "strategy sums of node %d on rank %d" % (nd.index, g)


@pytest.mark.parametrize("world,streets,n", [(2, 1, 700), (3, 1, 700), (5, 1, 300), (2, 3, 700), (3, 3, 60_000), (2, 4, 30_000), (3, 4, 2_000)])
def test_data_parallel_trainer_in_real_processes_equals_one_gpu_with_the_union_batch(world, streets, n, tmp_path):
    """`world` PROCESSES x n deals on replicated tables, rs_deal_trainer_train exchanging the deltas between sweep and apply itself: cards, tables and iteration counts must
    equal ONE trainer with world * n deals per batch.  60 000 deals per rank on three streets: delta rows, ordered sweeps and the staged list walkers under the collective
    (their summing launches belong to phase 0: the delta tables must be complete when the ranks exchange them).
    Round 5: the ranks sum the traverser's delta cells alone (packed, one ncclInt32 all-reduce), and rounds whose rows go straight into the table (the ISOMORPHIC river of the
    three-street cases: > 16 384 clusters) exchange their rows as (job, row, cluster, delta) items, all-gathered and applied by every rank -- direct rows and kept records
    stay on under the communicator.  streets = 4: three streets with 200 hole-card combos per player, a river round of more than 100 000 clusters, 20 batches in
    the second case (the kept records are the working copy: the items go into them alone)."""
    if streets == 1:
        mask = ab.card_mask("4d5dAs3cKs")
        hands = ab.random_range(mask)[::3]
        n_actions, tree = rs.build_game_tree(rs.default_flop())
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)]
        extra = {}
    else:
        rng = np.random.Generator(np.random.PCG64(4))
        mask = ab.card_mask("2c9dKh")
        allh = ab.random_range(mask)
        hands = allh[rng.permutation(len(allh))[:35 if streets == 3 else 200]]
        n_actions, tree = rs.build_game_tree(rs.three_street_options())
        files = [rng.integers(0, 23, size=1286792, dtype=np.uint32), rng.integers(0, 41, size=13960050, dtype=np.uint32), None]
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
        extra = dict(file0=files[0], file1=files[1])
        if streets == 4:
            assert card_abs[2].get_size(0) >= 100_000, card_abs[2].get_size(0)
    batches = 20 if (streets, n) == (4, 2_000) else 3
    streets = min(streets, 3)
    di = 2 * world * n - 100
    np.savez(os.path.join(tmp_path, "inputs.npz"), n=n, streets=streets, batches=batches, mask=mask, hands=hands, discount_interval=di, **extra)
    single = rs.DealTrainer(tree, card_abs, [hands, hands], mask, world * n, seed=21, discount_interval=di, discount_cap=10**9)
    single.train(batches)
    ranks = run_ranks("dp-deals", tmp_path, world, timeout=600)
    assert (np.concatenate([r["cards"] for r in ranks], axis=1) == single.cards()).all()
    for g, got in enumerate(ranks):
        assert int(got["iterations"][0]) == single.iterations
        for nd in tree.action_nodes():
            want = single.infosets.download_node(nd.index)
            assert (got["R%d" % nd.index] == want[0]).all() and (got["S%d" % nd.index] == want[1]).all(), "node %d on rank %d" % (nd.index, g)
    single.status()
    # what a rank handed to the collectives per batch, against what round 4 reduced (both whole delta arrays, twice per batch): the packed traverser cells are half of that at
    # most, the rounds with direct rows send items instead of cells
    per_batch = int(ranks[0]["exchange_bytes"][0]) / batches
    whole = 2 * 2 * 4 * single.infosets.cells
    print("data-parallel exchange: %.3f MB per batch and rank (round 4: %.3f MB)" % (per_batch / 1e6, whole / 1e6))
    assert 0 < per_batch <= whole / 2 + 1


@pytest.mark.parametrize("world,dtype,layout,fuse", [(2, "i32", "plain", 1), (3, "i32", "tiled64", 1), (5, "i32", "plain", 0), (2, "f32", "plain", 1), (3, "f32", "tiled64", 1)])
def test_allreduce_replicated_in_real_processes(world, dtype, layout, fuse, tmp_path):
    """rs_replicated_begin -> rs_iterate x 2 -> rs_allreduce_replicated(round_mask = flop) with `world` PROCESSES: every rank owns B boards of the turn and the river (PASS chance
    nodes, its own leaves), the flop's rows start out the same everywhere and must END the same everywhere: snapshot + the sum of every rank's (row - snapshot), the
    reference's eight threads adding into one table (cfr.rs:195-229) made deterministic.  The parent works the expected table out with the ORACLE, rank by rank:
    i32 sums wrap and commute -> bit for bit; f32: within 1e-5 relative (the order of an RCCL sum is the library's; the stub's is rank order).  Two trips, so that the second
    snapshot is taken of the reconciled rows.  "tiled64": the ranks keep their rows in 64-lane tiles (RS_TABLE_TILE_LANES) -- the collective works on the node's cells as
    they lie, and every rank lays them out alike."""
    Cn, B, trips = 40, 3, 2
    dt_o = {"i32": orc.T_I32, "f32": orc.T_F32}[dtype]
    scale, mode, mode_o = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if dtype == "i32" else (2.0 ** -6, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    otree = orc.OracleTree(orc.options_three_street())
    rng = np.random.Generator(np.random.PCG64(1000 + world))
    lanes = B * Cn

    def rows(a):
        if dtype == "i32":
            return rng.integers(-10**6, 10**6, size=(a, lanes)).astype(np.int32), rng.integers(0, 10**6, size=(a, lanes)).astype(np.int32)
        return rng.uniform(-1000, 1000, size=(a, lanes)).astype(np.float32), rng.uniform(0, 1000, size=(a, lanes)).astype(np.float32)

    inputs = dict(Cn=Cn, B=B, trips=trips, scale=scale, mode=mode, fuse=fuse)
    flop = [nd for nd in tree.action_nodes() if nd.round_idx == 0]
    for nd in flop:
        inputs["R%d" % nd.index], inputs["S%d" % nd.index] = rows(nd.n_children)
    for g in range(world):
        for nd in tree.action_nodes():
            if nd.round_idx != 0:
                inputs["R%d_%d" % (nd.index, g)], inputs["S%d_%d" % (nd.index, g)] = rows(nd.n_children)
        for r in range(3):
            inputs["sign%d_%d" % (r, g)] = rng.integers(-1, 2, size=lanes).astype(np.float32)
    np.savez(os.path.join(tmp_path, "inputs.npz"), **inputs)
    # the oracle, one table per rank
    otabs, osols = [], []
    for g in range(world):
        otab = orc.OracleTable(otree, [B, B, B], Cn, dt_o)
        for nd in tree.action_nodes():
            who = "" if nd.round_idx == 0 else "_%d" % g
            otab.set_node(nd.index, inputs["R%d%s" % (nd.index, who)], inputs["S%d%s" % (nd.index, who)])
        lo = {}
        for i, nd in enumerate(tree.nodes):
            if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
                lo[i] = (orc.LEAF_SIGN, inputs["sign%d_%d" % (tree.nodes[nd.parent].round_idx, g)])
        otabs.append(otab)
        osols.append(orc.OracleSolver(otree, otab, lo, scale=scale, mode=mode_o, chance_mode=orc.CHANCE_PASS))
    from rustsolver_amd.dist import replicated_allreduce
    for trip in range(trips):
        snap = {nd.index: otabs[0].get_node(nd.index) for nd in flop}
        for g in range(world):
            osols[g].iterate(0), osols[g].iterate(1)
        for nd in flop:
            new = []
            for k in range(2):
                xs = [np.ascontiguousarray(otabs[g].get_node(nd.index)[k]) for g in range(world)]

                def total(neg_delta_of_rank0, xs=xs, k=k, nd=nd):   # rank order, like the stub
                    acc = neg_delta_of_rank0.copy()
                    for x in xs[1:]:
                        d = (snap[nd.index][k].view(np.uint32) - x.view(np.uint32)).view(np.int32) if dtype == "i32" else (snap[nd.index][k] - x).astype(np.float32)
                        acc = (acc.view(np.uint32) + d.view(np.uint32)).view(np.int32) if dtype == "i32" else (acc + d).astype(np.float32)
                    return acc
                new.append(replicated_allreduce(xs[0], np.ascontiguousarray(snap[nd.index][k]), total))
            for g in range(world):
                otabs[g].set_node(nd.index, new[0], new[1])
    ranks = run_ranks("replicated-" + dtype, tmp_path, world, extra_env={"RS_TABLE_TILE_LANES": "64"} if layout == "tiled64" else None)
    for g, got in enumerate(ranks):
        for nd in tree.action_nodes():
            ro, so = otabs[g].get_node(nd.index)
            if dtype == "i32":
                assert (got["R%d" % nd.index] == ro).all() and (got["S%d" % nd.index] == so).all(), "node %d (round %d) on rank %d" % (nd.index, nd.round_idx, g)
            else:
                np.testing.assert_allclose(got["R%d" % nd.index], ro, rtol=1e-5, atol=1e-3, err_msg="regrets of node %d on rank %d" % (nd.index, g))
                np.testing.assert_allclose(got["S%d" % nd.index], so, rtol=1e-5, atol=1e-3, err_msg="strategy sums of node %d on rank %d" % (nd.index, g))
        if g:   # the replicated round is the SAME on every rank, to the last bit, float tables included (every rank adds the same total to the same snapshot)
            for nd in flop:
                assert got["R%d" % nd.index].tobytes() == ranks[0]["R%d" % nd.index].tobytes() and got["S%d" % nd.index].tobytes() == ranks[0]["S%d" % nd.index].tobytes()
