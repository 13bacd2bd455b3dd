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


def run_ranks(case, workdir, world, timeout=300):
    """start `world` fresh worker processes (they share GPU 0), wait, return their result files"""
    env = dict(os.environ, RS_RCCL_LIB=build_stub(), HSA_ENABLE_IPC_MODE_LEGACY="0")
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


@pytest.mark.parametrize("world,streets,n", [(2, 1, 700), (3, 1, 700), (5, 1, 300), (2, 3, 700), (3, 3, 60_000)])
def test_data_parallel_trainer_in_real_processes_equals_one_gpu_with_the_union_batch(world, streets, n, tmp_path):
    """`world` PROCESSES x n deals on replicated tables, rs_deal_trainer_train issuing the ncclInt32 all-reduce of both delta arrays between sweep and apply itself: cards, tables and
    iteration counts must equal ONE trainer with world * n deals per batch.  60 000 deals per rank on three streets: delta rows, ordered sweeps and the staged list walkers under
    the collective (their summing launches belong to phase 0: the delta tables must be complete when the ranks exchange them)."""
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
        hands = allh[rng.permutation(len(allh))[:35]]
        n_actions, tree = rs.build_game_tree(rs.three_street_options())
        files = [rng.integers(0, 23, size=1286792, dtype=np.uint32), rng.integers(0, 41, size=13960050, dtype=np.uint32), None]
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
        extra = dict(file0=files[0], file1=files[1])
    batches = 3
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
