#!/usr/bin/env python3
"""bench.py -- CFR iterations/sec of the MI355X regret/strategy-update engine + the roofline of its
dominant kernel (the river regret-update kernel), next to the CPU restatement of the reference.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B] [--clusters C]

Workload (BASELINE.json configs[1]): the 14-action-node river tree of
options::default_flop() (options.rs:52-81), C = 1000 clusters, A in {2,3}, i32 tables, with the board axis
replicated to `--boards` per GPU so that one sweep streams >= 8 GB (>> the 256 MB Infinity Cache).
One STEP = one CFR iteration = both traversers swept over every (board, cluster) lane of the tree
(rs_iterate x 2).  `value` = board-iterations / second summed over all ranks, inputs resident in HBM.
N > 1: boards shard across ranks (one process per GPU, no data-path collective for this river-only
workload -- nothing is replicated), weak scaling.

The oracle (oracle/) is used ONLY for the `cpu_baseline` leg.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--boards", type=int, default=9216, help="boards per GPU (9216 x 1000 lanes = 8.0 GB update traffic/iteration)")
    ap.add_argument("--clusters", type=int, default=1000)
    ap.add_argument("--mode", choices=["clamp", "wrap"], default="clamp",
                    help="clamp: cfr.rs:413-464 scale 100 (the live mccfr update); wrap: cfr.rs:612-621 scale 10000")
    ap.add_argument("--graph", type=int, default=0, help="replay each traverser sweep as one hipGraph")
    ap.add_argument("--fuse", type=int, default=-1,
                    help="1: one tree-specialised (hipRTC) kernel per traverser sweep; 0: level-by-level node kernels; "
                         "-1: 1 if libhiprtc can be loaded")
    ap.add_argument("--tree", choices=["river", "three-street"], default="river",
                    help="river: options::default_flop() (BASELINE configs[1]); three-street: configs[2] (706 action nodes)")
    ap.add_argument("--boards3", default="1,49,2352", help="--tree three-street: boards per round (flop,turn,river)")
    ap.add_argument("--dtype", choices=["i32", "f32", "f16"], default="i32", help="table element type (f16 = BASELINE configs[4])")
    ap.add_argument("--opp", choices=["full", "sample"], default="full", help="opponent nodes: cfr() full width or mccfr() sampled")
    ap.add_argument("--dp-deals", type=int, default=0,
                    help="1: instead of the board-sharded sweep run the data-parallel deal trainer (replicated table, one ncclInt32 all-reduce of the "
                         "delta tables per traverser sweep over RCCL); needs torch.distributed.run, works with one rank too")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the auxiliary legs (single board, deal batches, trainers, solve, k-means): "
                    "the rocprofv3 --pmc passes only need the headline kernels, and counter collection serialises every dispatch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_trainer(rs, n_boards, n_clusters, mode, graph, device, seed, fuse=1, tree_kind="river", dtype="i32", opp="full", shard=None):
    """n_boards: int (river tree) or [flop, turn, river] (three-street tree)"""
    from rustsolver_amd import _lib as L
    three = tree_kind == "three-street"
    options = rs.three_street_options() if three else rs.default_flop()
    boards = list(n_boards) if three else [n_boards]
    dt = {"i32": rs.I32, "f32": rs.F32, "f16": rs.F16}[dtype]
    n_actions, tree = rs.build_game_tree(options)
    table = rs.create_infosets(n_actions, tree, [n_clusters], boards, dt, device)
    # synthetic inputs generated on the device (SURVEY.md 8(d) config 2 / 3 distributions)
    if dtype == "f16":
        table.fill_random(seed, (-2000, 2000), (0, 2000))       # binary16 holds integers exactly up to 2048
    else:
        table.fill_random(seed, (-10**6, 10**6), (0, 10**6))
    signs, leaves = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                signs[r] = table.lane_buffer(parent.index, 1)
                L.check(L.load().rs_fill_uniform_f32(table._h, signs[r].ptr, table.pitch(parent.index), seed + 17 + r, -1.0, 1.0))
            leaves[i] = (rs.LEAF_SIGN, signs[r])
    if dtype == "i32":
        scale, m = (100.0, rs.UPD_CLAMP_I64) if mode == "clamp" else (10000.0, rs.UPD_WRAP_I32)
    else:
        scale, m = 1.0, rs.UPD_CLAMP_I64
    sampled = opp == "sample"
    chance = rs.CHANCE_PASS if (not three or sampled) else rs.CHANCE_ENUM
    trainer = rs.MCCFRTrainer(tree, table, leaves, scale=scale, mode=m, chance_mode=chance, use_graph=bool(graph),
                              fuse_subtrees=(None if int(fuse) < 0 else int(fuse)), opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=seed,
                              shard=shard)
    table.sync()
    return trainer


def run_steps(trainer, k):
    from rustsolver_amd import _lib as L
    lib, h = L.load(), trainer._h
    for _ in range(k):
        L.check(lib.rs_iterate(h, 0, None))
        L.check(lib.rs_iterate(h, 1, None))


def cpu_baseline(n_clusters, mode, seconds):
    """The C restatement in REFERENCE LAYOUT (boxed AoS, scalar recursion per lane, per-visit allocations,
    infoset.rs:85 / cfr.rs:372-373), 8 threads like the reference's N_THREADS (cfr.rs:195) or all cores
    if fewer, on a bounded sample of the same workload."""
    import numpy as np
    from oracle import orc
    threads = min(8, os.cpu_count() or 1)
    boards = 64
    rng = np.random.Generator(np.random.PCG64(1235))
    tree = orc.OracleTree(orc.options_default_river())
    tb = orc.OracleTable(tree, [boards], n_clusters)
    for d in tree.as_dicts():
        if d["kind"] == orc.ACTION:
            a, n = tb.node_shape(d["index"])
            tb.set_node(d["index"], rng.integers(-10**6, 10**6, size=(a, n)).astype(np.int32),
                        rng.integers(0, 10**6, size=(a, n)).astype(np.int32))
    sign = np.sign(rng.uniform(-1, 1, size=boards * n_clusters)).astype(np.float32)
    leaves = {d["id"]: (orc.LEAF_SIGN, sign) for d in tree.as_dicts()
              if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    scale, m = (100.0, orc.UPD_CLAMP_I64) if mode == "clamp" else (10000.0, orc.UPD_WRAP_I32)
    sol = orc.OracleSolver(tree, tb, leaves, scale=scale, mode=m, chance_mode=orc.CHANCE_PASS, ref_alloc=True)
    t0 = time.perf_counter()
    sol.run_iterations(1, threads)
    t1 = time.perf_counter() - t0
    iters = max(1, int(seconds / max(t1, 1e-6)))
    t0 = time.perf_counter()
    sol.run_iterations(iters, threads)
    dt = time.perf_counter() - t0
    return {
        "value": boards * iters / dt, "unit": "board-iterations/s", "cores": threads, "kind": "port",
        "host_cores": os.cpu_count(),
        "sample": "%d iterations x %d boards x %d clusters of the same river tree, reference layout "
                  "(boxed AoS, per-visit allocs), %d threads, %.1f s" % (iters, boards, n_clusters, threads, dt),
    }


def cpu_baseline_tuned(n_clusters, mode, seconds):
    """Non-strawman CPU number: the same per-lane arithmetic on a contiguous pooled layout, no per-visit
    allocations, ALL host cores."""
    import numpy as np
    from oracle import orc
    threads = os.cpu_count() or 1
    boards = max(64, 4 * threads)
    tree = orc.OracleTree(orc.options_default_river())
    tb = orc.OracleFlatTable(tree, [boards], n_clusters, seed=7)
    rng = np.random.Generator(np.random.PCG64(77))
    sign = np.sign(rng.uniform(-1, 1, size=boards * n_clusters)).astype(np.float32)
    leaves = {d["id"]: (orc.LEAF_SIGN, sign) for d in tree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    scale, m = (100.0, orc.UPD_CLAMP_I64) if mode == "clamp" else (10000.0, orc.UPD_WRAP_I32)
    sol = orc.OracleSolver(tree, tb, leaves, scale=scale, mode=m, chance_mode=orc.CHANCE_PASS, ref_alloc=False)
    t0 = time.perf_counter()
    sol.run_iterations(2, threads)
    t1 = (time.perf_counter() - t0) / 2
    iters = max(2, int(seconds / max(t1, 1e-6)))
    t0 = time.perf_counter()
    sol.run_iterations(iters, threads)
    dt = time.perf_counter() - t0
    return {"value": boards * iters / dt, "unit": "board-iterations/s", "cores": threads, "kind": "port",
            "sample": "%d iterations x %d boards x %d clusters, pooled contiguous layout, no per-visit allocations, %d threads, %.1f s"
                      % (iters, boards, n_clusters, threads, dt)}


def deal_batch_leg(rs, device, n_deals, n_clusters, with_cpu, cpu_seconds):
    """SURVEY N2 measured: the reference's own train() loop, batched.  One DEAL-ITERATION = one sampled deal traversed by
    mccfr for both players (cfr.rs:209-226): sampled opponents (cfr.rs:467-476), clamp update scale 100 (cfr.rs:413-464),
    real get_cluster addressing into the reference-shaped table [action_node][cluster] (1000 clusters, river tree)."""
    import numpy as np
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n_actions, tree, [n_clusters], [1], rs.I32, device)
    table.fill_random(4321, (-10**6, 10**6), (0, 10**6))
    rng = np.random.Generator(np.random.PCG64(4321))
    cidx = {(0, p): rng.integers(0, n_clusters, size=n_deals).astype(np.uint32) for p in (0, 1)}
    sign = np.sign(rng.uniform(-1, 1, size=n_deals)).astype(np.float32)
    sbuf = rs.deal_buffer(table, n_deals, sign)
    leaves = {i: (rs.LEAF_SIGN, sbuf) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    tr = rs.MCCFRTrainer(tree, table, leaves, scale=100.0, mode=rs.UPD_CLAMP_I64, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=1,
                         use_graph=True)
    run_steps(tr, 5)
    table.sync()
    k = 50
    t0 = time.perf_counter()
    run_steps(tr, k)
    table.sync()
    dt = time.perf_counter() - t0
    out = {"what": "batched mccfr over sampled deals (reference iteration = 1 deal x both players), river tree, %d clusters, "
                   "%d deals per batch, sampled opponents, i32 clamp update, cluster-id gathers + atomic deltas" % (n_clusters, n_deals),
           "value": n_deals * k / dt, "unit": "deal-iterations/s", "ms_per_batch": dt / k * 1e3, "n_deals": n_deals,
           "launches_per_batch": tr.n_launches(0) + tr.n_launches(1)}
    tr.destroy()
    table.destroy()
    if with_cpu:
        from oracle import orc
        threads = min(8, os.cpu_count() or 1)
        nd_cpu = 200_000
        otree = orc.OracleTree(orc.options_default_river())
        otab = orc.OracleDealTable(otree, [(n_clusters, n_clusters)])
        for d in otree.as_dicts():
            if d["kind"] == orc.ACTION:
                a_, n_ = otab.node_shape(d["index"])
                otab.set_node(d["index"], rng.integers(-10**6, 10**6, size=(a_, n_)).astype(np.int32),
                              rng.integers(0, 10**6, size=(a_, n_)).astype(np.int32))
        oc = {(0, p): cidx[(0, p)][:nd_cpu] for p in (0, 1)}
        ol = {d["id"]: (orc.LEAF_SIGN, sign[:nd_cpu]) for d in otree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
        osol = orc.OracleDealSolver(otree, otab, ol, oc, nd_cpu, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=1)
        t0 = time.perf_counter()
        osol.run_sweeps(1, threads)
        t1 = time.perf_counter() - t0
        sweeps = max(1, int(cpu_seconds / max(t1, 1e-6)))
        t0 = time.perf_counter()
        osol.run_sweeps(sweeps, threads)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nd_cpu * sweeps / dtc, "unit": "deal-iterations/s", "cores": threads, "kind": "port",
                               "sample": "%d sweeps x %d deals, reference layout + per-visit allocations, %d threads, %.1f s"
                                         % (sweeps, nd_cpu, threads, dtc)}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def deal_trainer_leg(rs, device, n_deals, with_cpu, cpu_seconds):
    """The reference's train() as coded, end to end on the device: options::default_flop() (board 4d5dAs3cKs, random ranges, ISOMORPHIC river
    abstraction = 1081 clusters per player, cfr.rs:159-184).  Per batch: generate_hand (cfr.rs:100-143) -> get_cluster for both players
    (canonical hand index + dense id, cfr.rs:357-365) -> showdown comparison (cfr.rs:323-347) -> sampled mccfr sweep for both players
    (cfr.rs:299-479).  One DEAL-ITERATION = one sampled deal traversed for both players (cfr.rs:209-226)."""
    import numpy as np
    from rustsolver_amd import abstraction as ab
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    tr = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, n_deals, seed=7, discount_interval=0, use_graph=True, device=device)
    tr.infosets.fill_random(4321, (-10**6, 10**6), (0, 10**6))
    tr.train(3)
    tr.status()
    k = 30
    t0 = time.perf_counter()
    tr.train(k)
    tr.infosets.sync()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(k):
        tr.deal()
    tr.infosets.sync()
    dt_deal = time.perf_counter() - t0
    tr.status()
    out = {"what": "MCCFRTrainer::train as coded (default_flop board, random ranges, ISOMORPHIC river abstraction, %d clusters): deal sampling, "
                   "hand indexing + dense ids, showdown evaluation and the sampled mccfr sweep of both players, all on the device, %d deals per batch"
                   % (card_abs.get_size(0), n_deals),
           "value": n_deals * k / dt, "unit": "deal-iterations/s", "ms_per_batch": dt / k * 1e3, "ms_dealing_only": dt_deal / k * 1e3,
           "n_deals": n_deals}
    tr.destroy()
    if with_cpu:
        from oracle import orc
        threads = min(8, os.cpu_count() or 1)
        nd_cpu = 100_000
        rng = np.random.Generator(np.random.PCG64(4321))
        otree = orc.OracleTree(orc.options_default_river())
        sizes = [(card_abs.get_size(0), card_abs.get_size(1))]
        otab = orc.OracleDealTable(otree, sizes)
        for d in otree.as_dicts():
            if d["kind"] == orc.ACTION:
                a_, n_ = otab.node_shape(d["index"])
                otab.set_node(d["index"], rng.integers(-10**6, 10**6, size=(a_, n_)).astype(np.int32),
                              rng.integers(0, 10**6, size=(a_, n_)).astype(np.int32))
        cidx = {(0, p): np.zeros(nd_cpu, dtype=np.uint32) for p in (0, 1)}
        sign = np.zeros(nd_cpu, dtype=np.float32)
        ol = {d["id"]: (orc.LEAF_SIGN, sign) for d in otree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
        osol = orc.OracleDealSolver(otree, otab, ol, cidx, nd_cpu, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=1)
        t0 = time.perf_counter()
        orc.run_train_from_cards(osol, cidx, sign, mask, [hands, hands], 7, 1, threads)
        t1 = time.perf_counter() - t0
        sweeps = max(1, int(cpu_seconds / max(t1, 1e-6)))
        t0 = time.perf_counter()
        orc.run_train_from_cards(osol, cidx, sign, mask, [hands, hands], 7, sweeps, threads)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nd_cpu * sweeps / dtc, "unit": "deal-iterations/s", "cores": threads, "kind": "port",
                               "sample": "%d sweeps x %d deals from cards (generate_hand, hand index + dense id ONCE per deal and player, "
                                         "brute-force showdown once per deal, sampled mccfr with reference-style allocations), %d threads, %.1f s"
                                         % (sweeps, nd_cpu, threads, dtc)}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def solve_leg(rs, device):
    """Does the batched trainer SOLVE the game?  The reference's own configuration from a zero table, 65 536 deals per batch (the
    time-to-quality sweet spot), the reference's discount and prune schedules; then the exploitability of the
    average strategy (rs_best_response, both players) and calc_br as coded (cfr.rs:629-745) at the last tick."""
    from rustsolver_amd import _lib as L
    from rustsolver_amd import abstraction as ab
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    n, batches = 1 << 16, 1024
    tr = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, n, seed=1, use_graph=True, device=device)
    tr.set_tick_br(True)
    e0 = tr.exploitability()
    tr.train(2)
    tr.status()
    t0 = time.perf_counter()
    tr.train(batches - 2)
    tr.status()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    br = tr.best_response(L.BR_MAX)
    dt_br = time.perf_counter() - t0
    ev = tr.best_response(L.BR_AVERAGE)
    pair, tick = tr.last_br()
    out = {"what": "options::default_flop() from a zero table: %d batches x %d deals (%.3g iterations), discount ticks and prune schedule as coded; "
                   "exploitability = (BR value of player 0 + BR value of player 1) / 2 per deal against the average strategies, pot 35" % (batches, n, batches * n),
           "exploitability_before": e0, "exploitability_after": float(br.sum() / 2), "seconds_training": dt * batches / (batches - 2),
           "best_response_values": [float(x) for x in br], "average_profile_values": [float(x) for x in ev],
           "ms_best_response_both_players": dt_br * 1e3, "calc_br_as_coded_last_tick": [float(x) for x in pair], "last_tick_iteration": tick}
    tr.destroy()
    return out


def kmeans_leg(rs, device, with_cpu, cpu_seconds):
    """SURVEY N4 measured at the reference's own size: gen_emd(1, 500, 250, 20) (gen_abstraction/main.rs:384) = Kmeans::predict of the
    1 286 792 canonical flop histograms (20 bins, counts out of 250 samples) against 500 centers with emd_1d: the sweep whose output is
    the round_1_emd.dat bucket file."""
    import numpy as np
    from rustsolver_amd import abstraction as ab
    n, k, bins = 1286792, 500, 20
    rng = np.random.Generator(np.random.PCG64(2024))
    centre = rng.random(n)[:, None] * bins
    width = (0.5 + 6 * rng.random(n))[:, None]
    x = np.exp(-0.5 * ((np.arange(bins)[None, :] - centre) / width) ** 2)
    data = (np.floor(x / x.sum(axis=1, keepdims=True) * 250) / 250.0).astype(np.float32)
    centers = data[rng.choice(n, size=k, replace=False)]
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n_actions, tree, [4], [1], rs.I32, device)
    km = ab.Kmeans(table, data)
    km.predict(centers[:8])
    t0 = time.perf_counter()
    cl, md = km.predict(centers)
    dt = time.perf_counter() - t0
    out = {"what": "Kmeans::predict, emd_1d, %d histograms x %d centers x %d bins (gen_emd(1, 500, 250, 20)); time includes the result download" % (n, k, bins),
           "value": n * k / dt, "unit": "distance evaluations/s", "seconds_per_sweep": dt}
    if with_cpu:
        from oracle import orc
        threads = min(16, os.cpu_count() or 1)   # kmeans.rs:19 N_THREADS = 16 (rayon's pool in the reference)
        ns = 4000
        t0 = time.perf_counter()
        ocl, omd = orc.kmeans_predict(data[:ns], centers, orc.DIST_EMD, threads=threads)
        t1 = time.perf_counter() - t0
        reps = max(1, int(cpu_seconds / max(t1, 1e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            ocl, omd = orc.kmeans_predict(data[:ns], centers, orc.DIST_EMD, threads=threads)
        dtc = time.perf_counter() - t0
        same = bool((ocl == cl[:ns]).all() and (omd.view(np.uint32) == md[:ns].view(np.uint32)).all())
        out["cpu_baseline"] = {"value": ns * k * reps / dtc, "unit": "distance evaluations/s", "cores": threads, "kind": "port",
                               "sample": "%d x (%d histograms x %d centers), literal emd.rs:53-113, %d threads, %.1f s; identical to the GPU result: %s"
                                         % (reps, ns, k, threads, dtc, same)}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    table.destroy()
    return out


def three_street_leg(rs, device, with_cpu=False, cpu_seconds=6.0):
    """The reference's commented-out "real" configuration (options.rs:68-77): flop start, three betting rounds (706 action nodes), 5 000-bucket
    files on every street (EMD / OCHS shape: index -> bucket file -> dense id), sampled mccfr over 1 M deals per batch, everything on the device.
    Round subtrees with reach-down / walk-up kernels, live-deal lists, cluster-partitioned LDS tiles."""
    import numpy as np
    from rustsolver_amd import abstraction as ab
    rng = np.random.Generator(np.random.PCG64(1))
    mask = ab.card_mask("7h8hQc")
    hands = ab.random_range(mask)
    k, n = 5000, 1 << 20
    files = [rng.integers(0, k, size=size, dtype=np.uint32) for size in (1286792, 13960050, 123156254)]   # hand_indexer sizes of [2,3] [2,4] [2,5]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
    t0 = time.perf_counter()
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=7, discount_interval=0, device=device)
    create_s = time.perf_counter() - t0
    tr.train(2)
    tr.status()
    reps = 5
    t0 = time.perf_counter()
    tr.train(reps)
    tr.infosets.sync()
    dt = (time.perf_counter() - t0) / reps
    out = {"what": "MCCFRTrainer::train on a flop-start three-street tree (%d action nodes), %d-bucket files on flop / turn / river, %d deals per batch, sampled "
                   "opponents: deal sampling, hand indexing through the bucket files, showdowns and the sweep on the device" % (n_actions, k, n),
           "value": n / dt, "unit": "deal-iterations/s", "ms_per_batch": dt * 1e3, "n_deals": n, "clusters": [a_.get_size(0) for a_ in card_abs],
           "table_bytes": int(tr.infosets.nbytes if not callable(tr.infosets.nbytes) else tr.infosets.nbytes()), "trainer_create_s": create_s}
    sizes = [(a_.get_size(0), a_.get_size(1)) for a_ in card_abs]
    tr.destroy()
    if with_cpu:   # the same loop on the host cores: the oracle's train-from-cards (reference layout, per-visit allocations), 8 threads
        from oracle import orc
        threads = min(8, os.cpu_count() or 1)
        nd_cpu = 50_000
        otree = orc.OracleTree(orc.options_three_street())
        otab = orc.OracleDealTable(otree, sizes)
        cidx = {(r, p): np.zeros(nd_cpu, dtype=np.uint32) for r in range(3) for p in (0, 1)}
        sign = np.zeros(nd_cpu, dtype=np.float32)
        ol = {d["id"]: (orc.LEAF_SIGN, sign) for d in otree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
        osol = orc.OracleDealSolver(otree, otab, ol, cidx, nd_cpu, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=1)
        def run(k_):   # every call rebuilds the dense-id maps of all rounds (seconds): time DIFFERENCES between sweep counts
            t0_ = time.perf_counter()
            orc.run_train_from_cards(osol, cidx, sign, mask, [hands, hands], 7, k_, threads, bucket_files=files)
            return time.perf_counter() - t0_
        t_one, t_five = run(1), run(5)
        per = max((t_five - t_one) / 4.0, 1e-4)
        sweeps = max(4, int(cpu_seconds / per))
        dtc = max(run(1 + sweeps) - t_one, 1e-6)
        out["cpu_baseline"] = {"value": nd_cpu * sweeps / dtc, "unit": "deal-iterations/s", "cores": threads, "kind": "port",
                               "sample": "%d sweeps x %d deals from cards through the same bucket files (generate_hand, hand index -> bucket -> dense id per round and "
                                         "player, brute-force showdown, sampled mccfr with reference-style allocations), %d threads, %.1f s" % (sweeps, nd_cpu, threads, dtc)}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def dp_deals_main(a, rs, dist, rank, n_gpus, device, real_stdout):
    """--dp-deals 1: MCCFRTrainer::train as coded (see deal_trainer_leg), data-parallel: every rank deals and sweeps 4 M deals of each global
    batch against its replica of the table; per traverser sweep the two i32 delta arrays are all-reduced over RCCL (xGMI) and every rank
    applies the union.  WEAK scaling; `value` = deal-iterations/s of the whole job."""
    import ctypes as C2
    import torch
    from rustsolver_amd import _lib as L
    from rustsolver_amd import abstraction as ab
    if dist is None:
        raise RuntimeError("--dp-deals needs a process group: launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --dp-deals 1")
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    n = 1 << 22
    tr = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, n, seed=7, discount_interval=0, device=device, world=n_gpus, rank=rank)
    tr.infosets.fill_random(4321, (-10**6, 10**6), (0, 10**6))   # same seed on every rank: the replicas start identical
    ident = (C2.c_char * L.COMM_ID_BYTES)()
    if rank == 0:
        L.check(L.load().rs_comm_unique_id(ident))
    t_id = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t_id, src=0)
    ident = (C2.c_char * L.COMM_ID_BYTES).from_buffer_copy(bytes(t_id.cpu().tolist()))
    comm = C2.c_void_p()
    L.check(L.load().rs_comm_create(tr.infosets._h, ident, rank, n_gpus, C2.byref(comm)))
    tr.attach_comm(comm)

    def barrier():
        tr.infosets.sync()
        dist.barrier()
        torch.cuda.synchronize()

    tr.train(a.warmup)
    barrier()
    t0 = time.perf_counter()
    tr.train(a.steps)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tr.status()
    # the replicas must still be identical: compare a checksum of rank 0's table with everybody's
    chk = 0
    for nd in tree.action_nodes():
        r_, s_ = tr.infosets.download_node(nd.index)
        chk = (chk * 1000003 + int(r_.astype("int64").sum()) * 31 + int(s_.astype("int64").sum())) % (1 << 61)
    c = torch.tensor([chk], dtype=torch.int64, device="cuda")
    cmin, cmax = c.clone(), c.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    tr.attach_comm(None)
    L.load().rs_comm_destroy(comm)
    if rank == 0:
        out = {"metric": "mccfr_deal_iterations_per_sec", "value": n * n_gpus * a.steps / elapsed, "unit": "deal-iterations/s", "n_gpus": n_gpus,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "i32", "data": "synthetic",
               "config": {"workload": "MCCFRTrainer::train as coded (default_flop board, random ranges, ISOMORPHIC river, 1081 clusters), data-parallel: "
                                      "%d deals per rank and batch, 1 step = 1 global batch (both traversers), deltas all-reduced as ncclInt32" % n,
                          "parallelism": "dp%d: replicated table, 2 all-reduces of %d i32 cells per traverser sweep" % (n_gpus, tr.infosets.cells),
                          "replicas_identical": bool(cmin.item() == cmax.item())}}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out))
        sys.stdout.flush()
        os.dup2(2, 1)
    tr.destroy()
    dist.destroy_process_group()


def pmc_traffic(a, kernel):
    """HBM bytes per update launch from the committed rocprofv3 PMC passes (profiles/), if they were taken on
    this exact workload; PMC counters cannot be read from inside the process."""
    import glob
    if (a.boards, a.clusters, a.mode, a.tree, a.dtype, a.opp) != (9216, 1000, "clamp", "river", "i32", "full"):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline_traffic_%s.json" % kernel)))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: libraries that print banners there (RCCL's version block, hipRTC) are
    # sent to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # under torch.distributed.run (RANK / MASTER_ADDR set) use the process group even for one rank, so that the
    # N > 1 code path can be exercised on a single-GPU box
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if a.gpus != n_gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (a.gpus, world, n_gpus), file=sys.stderr)

    import rustsolver_amd as rs  # raises if the HIP library is missing (no fallback)
    if rs.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU: the engine has no CPU fallback")
    device = local_rank % max(1, rs.device_count())

    def barrier():
        trainer.infosets.sync()
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    if a.dp_deals:
        dp_deals_main(a, rs, dist, rank, n_gpus, device, real_stdout)
        return

    three = a.tree == "three-street"
    boards3 = [int(x) for x in a.boards3.split(",")]
    if three and a.opp == "sample":
        boards3 = [boards3[-1]] * 3          # mccfr(): every lane is one full run-out (pass-through chance nodes)
    shard = None
    if three and n_gpus > 1 and a.opp == "full":
        # BASELINE configs[3]: turn and river boards sharded over the ranks, flop replicated, one RCCL all-gather per sweep
        # (STRONG scaling: the global problem is fixed).  Not yet run on more than one physical GPU.
        from rustsolver_amd.dist import shard_boards
        tlo, thi = shard_boards(boards3[1], rank, n_gpus)
        fan = boards3[2] // boards3[1]
        shard = (n_gpus, rank, 1, boards3[1])
        global_river = boards3[2]
        boards3 = [boards3[0], thi - tlo, (thi - tlo) * fan]
    trainer = make_trainer(rs, boards3 if three else a.boards, a.clusters, a.mode, a.graph, device, 1234 + 1 + rank, a.fuse,
                           a.tree, a.dtype, a.opp, shard)
    if shard is not None:
        import ctypes as C2
        import torch
        from rustsolver_amd import _lib as L
        ident = (C2.c_char * L.COMM_ID_BYTES)()
        if rank == 0:
            L.check(L.load().rs_comm_unique_id(ident))
        t_id = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device="cuda")
        dist.broadcast(t_id, src=0)
        ident = (C2.c_char * L.COMM_ID_BYTES).from_buffer_copy(bytes(t_id.cpu().tolist()))
        comm = C2.c_void_p()
        L.check(L.load().rs_comm_create(trainer.infosets._h, ident, rank, n_gpus, C2.byref(comm)))
        trainer.attach_comm(comm)
    if three:
        a.boards = boards3[-1] if shard is None else global_river / n_gpus   # `value` counts river boards (global when sharded)
    table = trainer.infosets

    # ---- warmup, then the timed region: exactly K steps between barrier+sync on both sides -------------
    run_steps(trainer, a.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(trainer, a.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: the same K steps again with every launch bracketed by HIP events on the table's
    # stream (graph replay off so that single launches can be timed) ---------------------------------------
    table.profile_reset()
    table.profile_enable(True)
    t0 = time.perf_counter()
    run_steps(trainer, a.steps)
    table.sync()
    elapsed_prof = time.perf_counter() - t0
    prof = table.profile_read()
    table.profile_enable(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj))
        sys.stdout.flush()

    dom_name = "tree" if prof["tree"]["launches"] else "update"
    upd = prof[dom_name]
    achieved = upd["algo_bytes"] / (upd["ms"] * 1e-3) / 1e9 if upd["ms"] > 0 else 0.0
    kernels = {k: {"launches": v["launches"], "ms_per_step": v["ms"] / a.steps,
                   "algo_GBps": (v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
               for k, v in prof.items() if v["launches"]}

    # ---- the per-node river regret-update kernel on its own (20A+8 bytes per lane, all inputs
    # buffers): rs_update_node on the root node (A = 3), HIP-event timed ---------------------------------------
    upd_node = None
    try:
        import numpy as np  # noqa: F401
        from rustsolver_amd import _lib as L
        lib = L.load()
        ubuf = table.lane_buffer(0, 3)
        rbuf = table.lane_buffer(0, 1)
        obuf = table.lane_buffer(0, 1)
        L.check(lib.rs_fill_uniform_f32(table._h, ubuf.ptr, 3 * table.pitch(0), 5, -1035.0, 1035.0))
        L.check(lib.rs_fill_uniform_f32(table._h, rbuf.ptr, table.pitch(0), 6, 0.0, 1.0))
        mode_flag = rs.UPD_CLAMP_I64 if (a.mode == "clamp" or a.dtype != "i32") else rs.UPD_WRAP_I32
        scale = (100.0 if a.mode == "clamp" else 10000.0) if a.dtype == "i32" else 1.0
        for _ in range(3):
            L.check(lib.rs_update_node(table._h, 0, ubuf.ptr, rbuf.ptr, scale, mode_flag, obuf.ptr))
        table.profile_reset()
        table.profile_enable(True)
        for _ in range(20):
            L.check(lib.rs_update_node(table._h, 0, ubuf.ptr, rbuf.ptr, scale, mode_flag, obuf.ptr))
        pu = table.profile_read()["update"]
        table.profile_enable(False)
        gbs = pu["algo_bytes"] / (pu["ms"] * 1e-3) / 1e9
        upd_node = {"kernel": "rs::k_update<3> via rs_update_node (%s per lane, all inputs buffers)" %
                              ("20A+8 = 68 B" if a.dtype != "f16" else "12A+8 = 44 B"),
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "launches": pu["launches"], "avg_launch_ms": pu["ms"] / max(1, pu["launches"]),
                    "algo_bytes_per_launch": pu["algo_bytes"] / max(1, pu["launches"])}
        for b in (ubuf, rbuf, obuf):
            b.free()
    except Exception as e:
        upd_node = {"error": str(e)}

    # ---- discount sweep (cfr.rs:250-261, row a8): 16 bytes per cell, whole table --------------------------------------
    disc = None
    try:
        table.profile_reset()
        table.profile_enable(True)
        for _ in range(10):
            table.discount(0.999)
        pd = table.profile_read()["discount"]
        table.profile_enable(False)
        gbs = pd["algo_bytes"] / (pd["ms"] * 1e-3) / 1e9
        disc = {"kernel": "rs::k_discount (16 B per cell)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": pd["ms"] / max(1, pd["launches"])}
    except Exception as e:
        disc = {"error": str(e)}

    dom_kernel = ("rs_tree_p{0,1}_lanes (tree-specialised, hipRTC: regret matching, reach, utilities and the regret / strategy_sum "
                  "update of all 14 river nodes in one launch per traverser)") if dom_name == "tree" else \
                 "rs::k_update (river regret/strategy_sum update, all action counts)"
    out = {
        "metric": "cfr_iterations_per_sec",
        "value": a.boards * n_gpus * a.steps / elapsed,
        "unit": "board-iterations/s",
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if shard is not None else "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {
            "workload": ("config2 river-only: 14-action-node tree of options::default_flop(), %d clusters, A in {2,3}, "
                         "%d boards per GPU" % (a.clusters, a.boards) if not three else
                         "config3 flop+turn+river: 706-action-node tree, %d clusters per round, boards %s per GPU"
                         % (a.clusters, "/".join(str(b) for b in boards3))) +
                        ", %s tables, %s update, opponent %s; 1 step = 1 CFR iteration (both traversers, all lanes)"
                        % (a.dtype, "cfr.rs:413-464 clamp scale 100" if a.mode == "clamp" else "cfr.rs:612-621 wrap scale 10000",
                           "full width (cfr.rs:576-589)" if a.opp == "full" else "sampled (mccfr, cfr.rs:467-476)"),
            "n_boards_per_gpu": a.boards, "n_clusters": a.clusters, "lanes_per_gpu": a.boards * a.clusters,
            "lane_pitch": int(table.pitch(0)),
            "table_bytes_per_gpu": table.nbytes, "workspace_bytes_per_gpu": trainer.workspace_bytes,
            "launches_per_step": trainer.n_launches(0) + trainer.n_launches(1), "hip_graph": bool(a.graph),
            "fused_subtrees": bool(trainer.fused),
            "parallelism": "boards sharded x%d, no collective (nothing replicated in a river-only tree)" % n_gpus,
        },
        "roofline": {
            "kernel": dom_kernel,
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(a, dom_name),
            "traffic_source": "profiles/r*_roofline_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % dom_name,
            "launches": upd["launches"], "avg_launch_ms": upd["ms"] / max(1, upd["launches"]),
            "algo_bytes_per_launch": upd["algo_bytes"] / max(1, upd["launches"]),
            "note": "achieved = algorithmic bytes (DESIGN.md) / HIP-event duration of every launch of this kernel in a second, "
                    "event-bracketed pass over the same K steps (ms_per_step there: %.3f)" % (elapsed_prof / a.steps * 1e3),
        },
        "roofline_update_node": upd_node,
        "roofline_discount": disc,
        "kernels": kernels,
        "lane_updates_per_sec": sum(table.lanes(n) for n in range(table.n_nodes)) * n_gpus * a.steps / elapsed,
    }

    try:   # the card's own streaming ceiling (cards of one pool differ by more than 10 %): a plain float4 copy, 2 x 2 GiB, on the same stream
        import ctypes as C3
        from rustsolver_amd import _lib as L3
        g = C3.c_double()
        L3.check(L3.load().rs_stream_probe(table._h, 2 << 30, 10, C3.byref(g)))
        out["stream_probe"] = {"copy_GBps": g.value, "what": "nt float4 copy, 2 GiB read + 2 GiB written per launch, best of 1 024 / 4 096 / 16 384 workgroups, HIP events",
                               "tree_kernel_over_copy": achieved / g.value if g.value > 0 else None}
        out["roofline"]["frac_of_copy_on_this_card"] = out["stream_probe"]["tree_kernel_over_copy"]
    except Exception as e:
        out["stream_probe"] = {"error": str(e)}

    if n_gpus > 1 or a.no_extra:   # the extra legs (single board, deal batches, CPU baselines) are N = 1 material
        emit(out)
        os.dup2(2, 1)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- single-board latency (the reference-as-coded shape: n_boards = 1), hipGraph replay ------------------
    try:
        small = make_trainer(rs, [1, 1, 1] if three else 1, a.clusters, a.mode, 1, device, 99, a.fuse, a.tree, a.dtype, a.opp)
        run_steps(small, 20)
        small.infosets.sync()
        t0 = time.perf_counter()
        run_steps(small, 200)
        small.infosets.sync()
        dt = (time.perf_counter() - t0) / 200
        out["single_board"] = {"us_per_iteration": dt * 1e6, "iterations_per_s": 1.0 / dt, "hip_graph": True,
                               "launches_per_iteration": small.n_launches(0) + small.n_launches(1)}
        small.destroy()
    except Exception as e:  # the headline number must still be reported
        out["single_board"] = {"error": str(e)}

    try:
        out["deal_batch"] = deal_batch_leg(rs, device, 1 << 22, a.clusters, not a.no_cpu, min(a.cpu_seconds, 6.0))
    except Exception as e:
        out["deal_batch"] = {"error": str(e)}

    try:
        out["deal_trainer"] = deal_trainer_leg(rs, device, 1 << 22, not a.no_cpu, min(a.cpu_seconds, 6.0))
    except Exception as e:
        out["deal_trainer"] = {"error": str(e)}

    try:
        out["deal_trainer_three_street"] = three_street_leg(rs, device, not a.no_cpu, min(a.cpu_seconds, 5.0))
    except Exception as e:
        out["deal_trainer_three_street"] = {"error": str(e)}

    try:
        out["solve"] = solve_leg(rs, device)
    except Exception as e:
        out["solve"] = {"error": str(e)}

    try:
        out["kmeans_predict"] = kmeans_leg(rs, device, not a.no_cpu, min(a.cpu_seconds, 5.0))
    except Exception as e:
        out["kmeans_predict"] = {"error": str(e)}

    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(a.clusters, a.mode, a.cpu_seconds)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        try:
            out["cpu_baseline_tuned"] = cpu_baseline_tuned(a.clusters, a.mode, min(a.cpu_seconds, 8.0))
            out["gpu_over_cpu_tuned"] = out["value"] / out["cpu_baseline_tuned"]["value"]
        except Exception as e:
            out["cpu_baseline_tuned"] = {"error": str(e)}
    emit(out)
    if dist is not None:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
