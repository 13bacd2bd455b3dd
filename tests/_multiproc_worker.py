"""One rank of tests/test_gpu_multiproc.py: a fresh process that shares GPU 0 with its peers and talks to them through tests/libstub_rccl.so (RS_RCCL_LIB).
    python tests/_multiproc_worker.py <case> <workdir> <world> <rank> <id-hex>
Reads <workdir>/inputs.npz (written by the test), runs the library's OWN multi-rank path -- rs_solver_attach_comm + rs_iterate, rs_deal_trainer_attach_comm + train, or rs_replicated_begin + rs_iterate + rs_allreduce_replicated --
and writes what it ended up with to <workdir>/rank<rank>.npz."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

case, workdir, world, rank, id_hex = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
assert os.environ.get("RS_RCCL_LIB"), "the worker must run with RS_RCCL_LIB set"
import rustsolver_amd as rs  # noqa: E402
from rustsolver_amd import _lib as L  # noqa: E402
from rustsolver_amd import abstraction as ab  # noqa: E402

lib = L.load()
ident = (C.c_char * L.COMM_ID_BYTES).from_buffer_copy(bytes.fromhex(id_hex).ljust(L.COMM_ID_BYTES, b"\0"))
inp = np.load(os.path.join(workdir, "inputs.npz"))
out = {}

if case.startswith("sharded"):
    from rustsolver_amd.dist import shard_boards
    dtype = {"i32": rs.I32, "f32": rs.F32, "f16": rs.F16}[case.split("-")[1]]
    Cn, G, fuse, iters = int(inp["Cn"]), [int(x) for x in inp["G"]], int(inp["fuse"]), int(inp["iters"])
    scale, mode = float(inp["scale"]), int(inp["mode"])
    fan_river = G[2] // G[1]
    tlo, thi = shard_boards(G[1], rank, world)
    boards = [1, thi - tlo, (thi - tlo) * fan_river]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    tb = rs.create_infosets(n_actions, tree, [Cn], boards, dtype)
    cols = {0: slice(0, Cn), 1: slice(tlo * Cn, thi * Cn), 2: slice(tlo * fan_river * Cn, thi * fan_river * Cn)}
    for nd in tree.action_nodes():
        tb.upload_node(nd.index, np.ascontiguousarray(inp["R%d" % nd.index][:, cols[nd.round_idx]]), np.ascontiguousarray(inp["S%d" % nd.index][:, cols[nd.round_idx]]))
    leaves, bufs = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in bufs:
                bufs[r] = tb.lane_buffer(parent.index, 1, np.ascontiguousarray(inp["sign%d" % r][cols[r]]))
            leaves[i] = (rs.LEAF_SIGN, bufs[r])
    sv = rs.MCCFRTrainer(tree, tb, leaves, scale=scale, mode=mode, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=fuse, shard=(world, rank, 1, G[1]))
    comm = C.c_void_p()
    L.check(lib.rs_comm_create(tb._h, ident, rank, world, C.byref(comm)))
    sv.attach_comm(comm)
    for it in range(iters):
        for player in (0, 1):
            out["util_%d_%d" % (it, player)] = sv.iterate(player, want_root_util=True)   # phase 0, ncclAllGather, phase 1: all inside rs_iterate
    for nd in tree.action_nodes():
        r, s2 = tb.download_node(nd.index)
        out["R%d" % nd.index], out["S%d" % nd.index] = r, s2
    sv.attach_comm(None)
    lib.rs_comm_destroy(comm)
elif case == "dp-deals":
    n, streets, batches = int(inp["n"]), int(inp["streets"]), int(inp["batches"])
    mask = int(inp["mask"])
    hands = inp["hands"]
    if streets == 1:
        n_actions, tree = rs.build_game_tree(rs.default_flop())
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)]
    else:
        n_actions, tree = rs.build_game_tree(rs.three_street_options())
        files = [inp["file0"], inp["file1"], None]
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, world=world, rank=rank, seed=21, discount_interval=int(inp["discount_interval"]), discount_cap=10**9)
    comm = C.c_void_p()
    L.check(lib.rs_comm_create(tr.infosets._h, ident, rank, world, C.byref(comm)))
    tr.attach_comm(comm)
    tr.train(batches)                                   # deal, sweep, the ranks' deltas exchanged (packed all-reduce + all-gathered items), apply, discount ticks: all inside rs_deal_trainer_train
    out["exchange_bytes"] = np.array([tr.exchange_bytes()], dtype=np.uint64)
    out["cards"] = tr.cards()
    out["iterations"] = np.array([tr.iterations])
    for nd in tree.action_nodes():
        r, s2 = tr.infosets.download_node(nd.index)
        out["R%d" % nd.index], out["S%d" % nd.index] = r, s2
    tr.status()
    tr.attach_comm(None)
    lib.rs_comm_destroy(comm)
elif case.startswith("replicated"):
    # rs_replicated_begin / rs_allreduce_replicated (the collective north_star names: "RCCL all-reduce ... for the average-strategy accumulator"): every rank sweeps its OWN
    # boards of the turn and the river without exchanging anything (PASS chance nodes: a lane keeps its board through the rounds), then the replicated flop round is
    # reconciled: x = snapshot + sum over ranks of (x - snapshot), regrets and strategy sums alike
    dtype = {"i32": rs.I32, "f32": rs.F32}[case.split("-")[1]]
    Cn, B, trips = int(inp["Cn"]), int(inp["B"]), int(inp["trips"])
    scale, mode = float(inp["scale"]), int(inp["mode"])
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    tb = rs.create_infosets(n_actions, tree, [Cn], [B, B, B], dtype)
    for nd in tree.action_nodes():
        who = "" if nd.round_idx == 0 else "_%d" % rank          # the flop's rows are the same on every rank, the later rounds' are the rank's own
        tb.upload_node(nd.index, inp["R%d%s" % (nd.index, who)], inp["S%d%s" % (nd.index, who)])
    leaves, bufs = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in bufs:
                bufs[r] = tb.lane_buffer(parent.index, 1, inp["sign%d_%d" % (r, rank)])
            leaves[i] = (rs.LEAF_SIGN, bufs[r])
    sv = rs.MCCFRTrainer(tree, tb, leaves, scale=scale, mode=mode, chance_mode=rs.CHANCE_PASS, fuse_subtrees=int(inp["fuse"]))
    comm = C.c_void_p()
    L.check(lib.rs_comm_create(tb._h, ident, rank, world, C.byref(comm)))
    for trip in range(trips):
        L.check(lib.rs_replicated_begin(tb._h, 0b001))
        sv.iterate(0), sv.iterate(1)
        L.check(lib.rs_allreduce_replicated(tb._h, comm, 0b001))
    for nd in tree.action_nodes():
        r, s2 = tb.download_node(nd.index)
        out["R%d" % nd.index], out["S%d" % nd.index] = r, s2
    lib.rs_comm_destroy(comm)
else:
    raise SystemExit("unknown case " + case)
np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
print("rank %d of %d done" % (rank, world))
