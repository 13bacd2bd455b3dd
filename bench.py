#!/usr/bin/env python3
"""bench.py -- CFR iterations/sec of the MI355X regret/strategy-update engine + the roofline of its
dominant kernel (the river regret-update kernel), next to the CPU restatement of the reference.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B] [--clusters C]

Workload (BASELINE.json configs[1]): the 14-action-node river tree of
options::default_flop() (options.rs:52-81), C = 1000 clusters, A in {2,3}, i32 tables, with the board axis
replicated to `--boards` per GPU so that one sweep streams >= 8 GB (>> the 256 MB Infinity Cache).
One STEP = one CFR iteration = both traversers swept over every (board, cluster) lane of the tree
(rs_iterate x 2).  `value` = board-iterations / second summed over all ranks, inputs resident in HBM.
N > 1: `python bench.py --gpus N` STARTS ITS OWN N RANKS (one process per GPU, torch.distributed over RCCL) when it was not launched by
torch.distributed.run -- the parent touches no GPU API, spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, forwards
rank 0's single JSON line and fails if any child fails (the reference's train() spawns its own 8 workers the same way, cfr.rs:195-229).
The headline stays config 2 with boards sharded across ranks (no data-path collective: nothing is replicated in a river-only tree), weak
scaling; beside it the N > 1 line carries `config4` (the 706-node flop+turn+river sweep with turn / river boards sharded over the ranks,
one RCCL all-gather per traverser sweep, STRONG scaling, all-gather time split out) and `dp_deals` (data-parallel sampled training, one
ncclInt32 all-reduce of the delta tables per sweep).  N = 1 carries `config3` (the same sweep on one GPU, 135 GB table) instead.

The oracle (oracle/) is used ONLY for the `cpu_baseline` legs.
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
    ap.add_argument("--skip", default="", help="comma-separated auxiliary legs to leave out (e.g. solve_three_street under rocprofv3, whose 10^5 small dispatches the tracer does not survive)")
    ap.add_argument("--config3-steps", type=int, default=5, help="timed iterations of the config 3 / config 4 leg (0 = skip the leg)")
    ap.add_argument("--saturating", type=int, default=100, help="i32 tables: one regret cell in N gets |regret| > 2.1e9 (SURVEY.md 8(d): at least 1 %% of the lanes exercise the "
                    "saturating adds); 0 = the plain +-10^6 fill of rounds 1 and 2")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full result (every leg, every per-kernel table, the descriptive strings) is "
                    "written; stdout carries the compact line")
    return ap.parse_args()


# ---- self-launch: python bench.py --gpus N starts its own N ranks -----------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run.  MUST NOT touch the GPU (no HIP call, no torch.cuda, no librustsolver_amd): it only starts N
    fresh python processes of this file with the torch.distributed environment set, forwards rank 0's stdout (the single JSON line) and
    returns non-zero as soon as any rank fails, ending the others (exactly the PIDs it started)."""
    import subprocess
    import threading
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RS_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
        time.sleep(0.05)
    for r in live:          # a rank failed: end exactly the children this process started
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    reader.join(timeout=20)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.strip()]
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return rc


def probe_main(kind):
    """RS_BENCH_PROBE (CPU tests of the launcher, no GPU): a child only proves the rendezvous -- gloo process group over the environment the
    launcher set, all ranks gathered -- and rank 0 prints them.  'fail': rank 1 exits with code 3 before the rendezvous."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if kind == "fail" and rank == 1:
        sys.exit(3)
    import torch.distributed as dist
    dist.init_process_group("gloo")
    seen = [None] * world
    dist.all_gather_object(seen, (rank, int(os.environ["LOCAL_RANK"]), os.getpid()))
    if rank == 0:
        print(json.dumps({"probe": True, "n_gpus": world, "ranks": sorted(x[0] for x in seen), "local_ranks": sorted(x[1] for x in seen),
                          "distinct_pids": len({x[2] for x in seen}), "master": os.environ["MASTER_ADDR"]}))
    dist.destroy_process_group()


def make_trainer(rs, n_boards, n_clusters, mode, graph, device, seed, fuse=1, tree_kind="river", dtype="i32", opp="full", shard=None, saturating=0, outlier_leaves=0):
    """n_boards: int (river tree) or [flop, turn, river] (three-street tree).  saturating: one regret cell in N beyond +-2.1e9 (i32 tables).  outlier_leaves: the
    showdown leaves become RS_LEAF_UTIL rows ~ U(-pot, pot) in which one lane in N holds +-1e9, so that (scale * reach) * (u - util) passes 2^31 there and the wave takes
    the exact i64 branch of the clamp update"""
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
        if saturating and dtype == "i32":
            L.check(L.load().rs_table_plant_saturating(table._h, seed, saturating))
    signs, leaves = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                signs[r] = table.lane_buffer(parent.index, 1)
                if outlier_leaves:
                    L.check(L.load().rs_fill_uniform_f32(table._h, signs[r].ptr, table.pitch(parent.index), seed + 17 + r, -1035.0, 1035.0))
                    L.check(L.load().rs_plant_outliers_f32(table._h, signs[r].ptr, table.pitch(parent.index), seed + 29 + r, outlier_leaves, 1.0e9))
                else:
                    L.check(L.load().rs_fill_uniform_f32(table._h, signs[r].ptr, table.pitch(parent.index), seed + 17 + r, -1.0, 1.0))
            leaves[i] = (rs.LEAF_UTIL if outlier_leaves else rs.LEAF_SIGN, signs[r])
    if dtype == "i32":
        scale, m = (100.0, rs.UPD_CLAMP_I64) if mode.startswith("clamp") else (10000.0, rs.UPD_WRAP_I32)
        if mode.endswith("+prune"):   # cfr() with prune = true (cfr.rs:379-386); the synthetic regrets stay above the threshold, so this prices the pruned kernels, not skipped work
            m |= rs.UPD_PRUNE
    else:
        # float tables: the utility of an ENUM chance node is a SUM over its deals (cfr.rs:519), so a flop regret delta reaches millions on the three-street tree and
        # overflows binary16 at scale 1 (and a NaN reach marks an inactive lane on the device: an overflowing run would measure kernels that skip their stores)
        scale, m = (2.0 ** -12 if three else 1.0), rs.UPD_CLAMP_I64
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


def _timed_iterations(run, seconds):
    """run(iterations) -> None; one probing iteration, then as many as fit `seconds`; returns (iterations, seconds taken)"""
    t0 = time.perf_counter()
    run(1)
    t1 = time.perf_counter() - t0
    iters = max(1, int(seconds / max(t1, 1e-6)))
    t0 = time.perf_counter()
    run(iters)
    return iters, time.perf_counter() - t0


def cpu_baseline(n_clusters, mode, seconds):
    """The C restatement in REFERENCE LAYOUT (boxed AoS, scalar recursion per lane, per-visit allocations,
    infoset.rs:85 / cfr.rs:372-373) on a bounded sample of the same workload: with 8 threads like the reference's
    N_THREADS (cfr.rs:195) -- `value` -- and with every host core -- `all_cores` (BASELINE.md section 2 asks for both)."""
    import numpy as np
    from oracle import orc
    threads = min(8, os.cpu_count() or 1)
    boards = max(64, 2 * (os.cpu_count() or 1))
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
    iters, dt = _timed_iterations(lambda k: sol.run_iterations(k, threads), seconds * 0.5)
    out = {
        "value": boards * iters / dt, "unit": "board-iterations/s", "cores": threads, "kind": "port",
        "host_cores": os.cpu_count(),
        "sample": "%d iterations x %d boards x %d clusters of the same river tree, reference layout "
                  "(boxed AoS, per-visit allocs), %d threads, %.1f s" % (iters, boards, n_clusters, threads, dt),
    }
    allc = os.cpu_count() or 1
    if allc > threads:
        iters2, dt2 = _timed_iterations(lambda k: sol.run_iterations(k, allc), seconds * 0.5)
        out["all_cores"] = {"value": boards * iters2 / dt2, "unit": "board-iterations/s", "cores": allc,
                            "sample": "%d iterations of the same sample on %d threads, %.1f s" % (iters2, allc, dt2)}
    return out


def cpu_config3(mode, seconds):
    """The CPU side of config 3 (>= 6x node-vs-host on the 706-node, 5 000-bucket tree): the enumerating cfr() (cfr.rs:481-627, boards
    enumerated at the public chance nodes, :502-522) on the three-street tree with 5 000 clusters per round and a board SUBSET that still leaves the caches,
    (a) the literal per-lane restatement in reference layout with 8 threads (cfr.rs:195) and with every core, (b) `cpu_soa`: block-major SoA, 20 clusters per
    unit, the sibling river boards of a unit walked together, vectorised (oracle/cpu_soa.c soae_*), compared bit for bit with (a)'s code on a small table before it is timed.  Rates are river-board-iterations/s:
    the unit the GPU leg reports, and per river board the work is the same whatever the fan."""
    import numpy as np
    from oracle import orc
    C_ = 5000
    scale, m = (100.0, orc.UPD_CLAMP_I64) if mode == "clamp" else (10000.0, orc.UPD_WRAP_I32)
    tree = orc.OracleTree(orc.options_three_street())
    dd = tree.as_dicts()
    acts = [(d["index"], len(d["children"]), d["round_idx"]) for d in dd if d["kind"] == orc.ACTION]
    showdowns = [(d["id"], dd[d["parent"]]["round_idx"]) for d in dd if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED]
    rng = np.random.Generator(np.random.PCG64(1237))
    allc = os.cpu_count() or 1
    out = {}
    # ---- identical outputs first: soae against the per-lane oracle, boards 1 / 2 / 6, 37 clusters, saturating ranges, 2 iterations ----------------------
    vb, vc = [1, 2, 6], 37
    vs = [np.sign(rng.uniform(-1, 1, size=b * vc)).astype(np.float32) for b in vb]
    for x in vs:
        x[::7] = 0.0
    chk = orc.SoaEnumSolver(tree, vb, vc, vs, scale, m)
    chk.fill(11, (-2**31, 2**31 - 1), (0, 2**31 - 1), threads=3)
    tb = orc.OracleTable(tree, vb, vc)
    for idx, na, r in acts:
        r_, s_ = chk.get_node(idx, na, r)
        tb.set_node(idx, r_, s_)
    osol = orc.OracleSolver(tree, tb, {i: (orc.LEAF_SIGN, vs[r]) for i, r in showdowns}, scale=scale, mode=m, chance_mode=orc.CHANCE_ENUM)
    osol.run_iterations(2, min(allc, 4))
    chk.run(2, 3)
    same = True
    for idx, na, r in acts:
        r_, s_ = chk.get_node(idx, na, r)
        ro, so = tb.get_node(idx)
        same = same and bool((r_ == ro).all() and (s_ == so).all())
    chk.destroy()
    del tb, osol
    if not same:
        raise RuntimeError("cpu_soa (enumerating) and the per-lane oracle disagree: refusing to time it")
    # ---- (a) reference layout: boards 1 / 2 / 8 = 23 M boxed info sets, about 2 GB of heap ---------------------------------------------------------------
    rb = [1, 2, 8]
    t0 = time.perf_counter()
    tb = orc.OracleTable(tree, rb, C_)
    for idx, na, r in acts:
        a_, n_ = tb.node_shape(idx)
        tb.set_node(idx, rng.integers(-10**6, 10**6, size=(a_, n_)).astype(np.int32), rng.integers(0, 10**6, size=(a_, n_)).astype(np.int32))
    rs_ = [np.sign(rng.uniform(-1, 1, size=b * C_)).astype(np.float32) for b in rb]
    sol = orc.OracleSolver(tree, tb, {i: (orc.LEAF_SIGN, rs_[r]) for i, r in showdowns}, scale=scale, mode=m, chance_mode=orc.CHANCE_ENUM, ref_alloc=True)
    build_s = time.perf_counter() - t0
    t8 = min(8, allc)
    iters, dt = _timed_iterations(lambda k: sol.run_iterations(k, t8), seconds * 0.3)
    out["cpu_baseline"] = {"value": rb[2] * iters / dt, "unit": "river-board-iterations/s", "cores": t8, "host_cores": allc, "kind": "port",
                           "sample": "%d iterations of the 706-node tree, %d clusters per round, boards %s, ENUM chance, reference layout (boxed AoS, per-visit allocs), "
                                     "%d threads, %.1f s (+ %.1f s building the boxed table)" % (iters, C_, "/".join(map(str, rb)), t8, dt, build_s)}
    if allc > t8:
        iters2, dt2 = _timed_iterations(lambda k: sol.run_iterations(k, allc), seconds * 0.3)
        out["cpu_baseline"]["all_cores"] = {"value": rb[2] * iters2 / dt2, "unit": "river-board-iterations/s", "cores": allc,
                                            "sample": "%d iterations of the same sample on %d threads, %.1f s" % (iters2, allc, dt2)}
    del sol, tb
    # ---- (b) cpu_soa: boards 1 / 7 / 336 (19 GB) where the host has the memory and the threads to fill it, else 1 / 4 / 48 (2.8 GB) -------------------------
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    sb = [1, 7, 336] if (allc >= 64 and avail > 48 * 2**30) else [1, 4, 48]
    ss = [np.sign(rng.uniform(-1, 1, size=b * C_)).astype(np.float32) for b in sb]
    soa = orc.SoaEnumSolver(tree, sb, C_, ss, scale, m)
    used = soa.fill(7, (-10**6, 10**6), (0, 10**6), threads=allc)
    iters3, dt3 = _timed_iterations(lambda k: soa.run(k, allc), seconds * 0.4)
    out["cpu_soa"] = {"value": sb[2] * iters3 / dt3, "unit": "river-board-iterations/s", "cores": used, "host_cores": allc, "kind": "port",
                      "table_bytes": soa.table_bytes, "identical_to_per_lane_oracle": same,
                      "sample": "%d iterations of the 706-node tree, %d clusters per round, boards %s, ENUM chance, block-major SoA, units of 20 clusters walked through the whole "
                                "tree (sibling river boards together), gcc -O3 -march=native, %d threads (one per unit at most: 250 units), %.1f s" % (iters3, C_, "/".join(map(str, sb)), used, dt3)}
    soa.destroy()
    return out


def cpu_soa(n_clusters, mode, seconds, boards=None, threads=None):
    """The NON-strawman CPU number (BASELINE.md section 2 `cpu_soa`): the GPU's own SoA layout, the tree walked over blocks of 256 contiguous
    lanes so that every inner loop vectorises (gcc -O3 -march=native, compiled on this host), persistent threads on ALL host cores with
    first-touch placement (oracle/cpu_soa.c).  Before anything is timed the same code and the literal per-lane restatement (rs_oracle.c) are
    run on the same seeded inputs and their tables compared bit for bit."""
    import numpy as np
    from oracle import orc
    threads = threads or (os.cpu_count() or 1)
    scale, m = (100.0, orc.UPD_CLAMP_I64) if mode == "clamp" else (10000.0, orc.UPD_WRAP_I32)
    tree = orc.OracleTree(orc.options_default_river())
    rng = np.random.Generator(np.random.PCG64(1236))
    # ---- identical outputs first (3 boards, 2 iterations, saturating ranges included) ----------------------------------------------
    vb = 3
    vsign = np.sign(rng.uniform(-1, 1, size=vb * n_clusters)).astype(np.float32)
    vsign[::9] = 0.0
    chk = orc.SoaSolver(tree, vb * n_clusters, vsign, scale, m)
    chk.fill(11, (-2**31, 2**31 - 1), (0, 2**31 - 1), threads=min(threads, 4))
    tb = orc.OracleTable(tree, [vb], n_clusters)
    acts = [(d["index"], len(d["children"])) for d in tree.as_dicts() if d["kind"] == orc.ACTION]
    for idx, na in acts:
        r_, s_ = chk.get_node(idx, na)
        tb.set_node(idx, r_, s_)
    leaves = {d["id"]: (orc.LEAF_SIGN, vsign) for d in tree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    osol = orc.OracleSolver(tree, tb, leaves, scale=scale, mode=m, chance_mode=orc.CHANCE_PASS)
    osol.run_iterations(2, min(threads, 4))
    chk.run(2, min(threads, 4))
    same = True
    for idx, na in acts:
        r_, s_ = chk.get_node(idx, na)
        ro, so = tb.get_node(idx)
        same = same and bool((r_ == ro).all() and (s_ == so).all())
    chk.destroy()
    if not same:
        raise RuntimeError("cpu_soa and the per-lane oracle disagree: refusing to time it")
    # ---- the timed sample: big enough to leave every cache (>= 1.2 GB of table for 1000 clusters) ----------------------------------------
    boards = boards or max(4096, 16 * threads)
    lanes = boards * n_clusters
    sign = np.sign(rng.uniform(-1, 1, size=lanes)).astype(np.float32)
    sol = orc.SoaSolver(tree, lanes, sign, scale, m)
    sol.fill(7, (-10**6, 10**6), (0, 10**6), threads=threads)
    t0 = time.perf_counter()
    sol.run(1, threads)
    t1 = time.perf_counter() - t0
    iters = max(2, int(seconds / max(t1, 1e-6)))
    t0 = time.perf_counter()
    sol.run(iters, threads)
    dt = time.perf_counter() - t0
    # algorithmic bytes per lane and traverser as for the GPU tree kernel: own nodes 16 A, opponent nodes 4 A, sign row 4
    own = {p_: sum(len(d["children"]) for d in tree.as_dicts() if d["kind"] == orc.ACTION and d["player"] == p_) for p_ in (0, 1)}
    per_lane = sum(16 * own[p_] + 4 * own[1 - p_] + 4 for p_ in (0, 1))
    out = {"value": boards * iters / dt, "unit": "board-iterations/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
           "algo_GBps": lanes * per_lane * iters / dt / 1e9, "table_bytes": sol.table_bytes, "identical_to_per_lane_oracle": same,
           "sample": "%d iterations x %d boards x %d clusters of the same river tree, SoA [A][lanes] rows, blocks of 256 lanes, gcc -O3 -march=native "
                     "auto-vectorised, %d persistent threads (first touch), %.1f s" % (iters, boards, n_clusters, threads, dt)}
    sol.destroy()
    return out


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


def roofline_deals(workload, ms_per_batch_now):
    """The deal kernels are not HBM-streaming kernels: they gather a cache-resident table and spend their time issuing vector instructions and waiting on gathers.
    Bound: VALU issue (a wave64 instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz).  The instruction counts and the issue / stall / wait shares
    of the waves' lifetime come from the committed rocprofv3 --pmc passes of the same workload (profiles/r*_deals.json: CANNED, counters cannot be read in-process);
    the time is this run's."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_deals.json")))
    if not files:
        return None
    try:
        prof = json.load(open(files[-1]))[workload]
    except Exception:
        return None
    simd_issue_per_s = 256 * 4 * 2.4e9 / 4.0          # wave-instructions per second the chip can issue
    tot_valu = tot_ms = 0.0
    kernels = {}
    batches = None
    for k, v in prof["kernels"].items():
        if not k.startswith("rs_tree_") or not v.get("valu_insts_per_wave") or not v.get("waves_per_dispatch"):
            continue
        if batches is None:
            batches = v["dispatches"]                 # the first-round kernel of a traverser runs once per batch
        batches = min(batches, v["dispatches"])
    if not batches:
        return None
    for k, v in prof["kernels"].items():
        if not k.startswith("rs_tree_") or not v.get("valu_insts_per_wave") or not v.get("waves_per_dispatch"):
            continue
        per_batch = v["dispatches"] / batches
        valu = v["valu_insts_per_wave"] * v["waves_per_dispatch"] * per_batch
        tot_valu += valu
        tot_ms += v["avg_us"] * per_batch / 1e3
        kernels[k] = {"dispatches_per_batch": per_batch, "avg_us": v["avg_us"], "valu_wave_insts_per_batch": valu, "issue_share": v["issue_share"],
                      "issue_stall_share": v["issue_stall_share"], "wait_share": v["wait_share"], "hbm_bytes_per_dispatch": v["hbm_bytes_per_dispatch"]}
    floor_ms = tot_valu / simd_issue_per_s * 1e3
    return {"bound": "valu", "unit": "fraction of the VALU issue slots of 1 024 SIMDs at 2.4 GHz", "peak": 1.0,
            "achieved": floor_ms / ms_per_batch_now if ms_per_batch_now else None, "frac": floor_ms / ms_per_batch_now if ms_per_batch_now else None,
            "valu_issue_floor_ms_per_batch": floor_ms, "ms_per_batch_this_run": ms_per_batch_now, "tree_kernel_ms_per_batch_in_profile": tot_ms,
            "canned": True, "source": os.path.relpath(files[-1], ROOT) + " (rocprofv3 --pmc SQ_* passes of tools/time_deal_trainer.py / time_three_street.py, committed)",
            "kernels": kernels}


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
    if n_deals == 1 << 22:
        out["roofline_deals"] = roofline_deals("river_4m", out["ms_per_batch"])
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


def solve_three_street_leg(rs, device):
    """Does the batched trainer solve a MULTI-ROUND game?  Flop start 7h8hQc, the reference's three-street tree (706 action nodes), lossless (ISOMORPHIC) abstractions
    on flop, turn and river, 200-combo ranges (a seeded subset of the random range: the best response is O(hands^2) per showdown leaf and run-out), 65 536 deals per
    batch.  Exploitability = (BR value of player 0 + BR value of player 1) / 2 per deal against the average strategies, best response over all 2 352 run-outs
    (rs_best_response_rounds), before training and along the way."""
    import numpy as np
    from rustsolver_amd import _lib as L
    from rustsolver_amd import abstraction as ab
    mask = ab.card_mask("7h8hQc")
    rng = np.random.Generator(np.random.PCG64(2))
    hands = ab.random_range(mask)
    hands = hands[np.sort(rng.choice(len(hands), 200, replace=False))]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
    n = 1 << 16
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=1, use_graph=True, device=device)
    t0 = time.perf_counter()
    e0 = tr.exploitability()
    first_br_s = time.perf_counter() - t0
    curve, done, train_s = [[0, e0]], 0, 0.0
    for upto in (64, 256, 1024):
        t0 = time.perf_counter()
        tr.train(upto - done)
        tr.status()
        train_s += time.perf_counter() - t0
        done = upto
        t0 = time.perf_counter()
        curve.append([done * n, tr.exploitability()])
        br_s = time.perf_counter() - t0
    ev = tr.best_response(L.BR_AVERAGE)
    out = {"what": "flop-start three-street game (%d action nodes), ISOMORPHIC abstractions on all three streets (%s clusters per player), 200-combo ranges, %d deals per "
                   "batch, discount and prune schedules as coded; exploitability per deal (pot 35) from a best response over all 2 352 run-outs"
                   % (n_actions, "/".join(str(a_.get_size(0)) for a_ in card_abs), n),
           "exploitability_curve": curve, "seconds_training": train_s, "seconds_first_best_response": first_br_s, "seconds_best_response_both_players": br_s,
           "average_profile_values": [float(x) for x in ev], "table_bytes": int(tr.infosets.nbytes)}
    tr.destroy()
    # the same game with the FULL 1 176-combo random ranges (lossless abstractions: 7.4 GB of table): what the rank-order showdowns (RS_BR_SORTED) are for -- the pair loop
    # of cfr.rs:323-347 took 3.1 s per exploitability call at this size
    try:
        hands_f = ab.random_range(mask)
        abs_f = [ab.CardAbstraction.init([hands_f, hands_f], mask, r, None) for r in range(3)]
        trf = rs.DealTrainer(tree, abs_f, [hands_f, hands_f], mask, n, seed=1, use_graph=True, device=device)
        trf.exploitability()                       # the first call computes and caches the cluster ids of every (board prefix, hand)
        t0 = time.perf_counter()
        ef0 = trf.exploitability()
        brf_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        trf.train(1024)
        trf.status()
        trainf_s = time.perf_counter() - t0
        out["full_ranges"] = {"what": "the same game with both full %d-combo ranges (%s clusters per player), 1 024 batches of %d deals"
                                      % (len(hands_f), "/".join(str(a_.get_size(0)) for a_ in abs_f), n),
                              "exploitability_curve": [[0, ef0], [1024 * n, trf.exploitability()]], "seconds_training": trainf_s,
                              "seconds_best_response_both_players": brf_s, "table_bytes": int(trf.infosets.nbytes)}
        trf.destroy()
    except Exception as e:
        out["full_ranges"] = {"error": str(e)}
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
    # ---- the training loop that produces those centers: Kmeans::fit_regular (kmeans.rs:497-600), ten Hamerly-bounded rounds, same data ----------------------
    try:
        t0 = time.perf_counter()
        fc, fcent, fb, finertia = km.fit_regular(centers, ab.DIST_EMD, 10)
        dtf = time.perf_counter() - t0
        fit = {"what": "Kmeans::fit_regular, emd_1d, 10 rounds of init_s -> reassign_clusters (Hamerly bounds) -> means -> bound shifts over the same %d histograms and %d "
                       "centers; time includes the downloads of clusters and bounds" % (n, k),
               "seconds": dtf, "inertia": float(finertia), "clusters_used": int(len(np.unique(fc)))}
        if with_cpu:
            from oracle import orc
            ns = 20000   # the literal port on a sample, and the device on the same sample: identical bits or the comparison is void
            t0 = time.perf_counter()
            oc, ocent, ob, oin = orc.kmeans_fit_regular(data[:ns], centers, orc.DIST_EMD, 10)
            dtc = time.perf_counter() - t0
            kms = ab.Kmeans(table, data[:ns])
            gc, gcent, gb, gin = kms.fit_regular(centers, ab.DIST_EMD, 10)
            same = bool((gc == oc).all() and (gcent.view(np.uint32) == ocent.view(np.uint32)).all() and (gb.view(np.uint32) == ob.view(np.uint32)).all())
            fit["cpu_baseline"] = {"value": ns * 10 / dtc, "unit": "datum-rounds/s", "cores": 1, "kind": "port",
                                   "sample": "10 rounds over %d histograms, literal kmeans.rs:497-600 (oracle/kmeans_fit.c), 1 thread, %.1f s; identical to the GPU result on the "
                                             "same sample: %s" % (ns, dtc, same)}
            fit["value"] = n * 10 / dtf
            fit["unit"] = "datum-rounds/s"
            # no gpu_over_cpu here: the port is one thread on a sample (Hamerly's skip rates depend on n), the reference runs these loops under rayon with 16 threads
            # (kmeans.rs:19) -- the two rates are not the same workload; the leg's figure is `seconds`
        out["fit_regular"] = fit
    except Exception as e:
        out["fit_regular"] = {"error": str(e)}
    table.destroy()
    return out


def three_street_leg(rs, device, with_cpu=False, cpu_seconds=6.0):
    """The reference's commented-out "real" configuration (options.rs:68-77): flop start, three betting rounds (706 action nodes), 5 000-bucket
    files on every street (EMD / OCHS shape: index -> bucket file -> dense id), sampled mccfr over 4 M deals per batch, everything on the device.
    Round subtrees with reach-down / walk-up kernels, live-deal lists, cluster-partitioned LDS tiles."""
    import numpy as np
    from rustsolver_amd import abstraction as ab
    rng = np.random.Generator(np.random.PCG64(1))
    mask = ab.card_mask("7h8hQc")
    hands = ab.random_range(mask)
    k, n = 5000, 1 << 22
    files = [rng.integers(0, k, size=size, dtype=np.uint32) for size in (1286792, 13960050, 123156254)]   # hand_indexer sizes of [2,3] [2,4] [2,5]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
    t0 = time.perf_counter()
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=7, discount_interval=0, use_graph=True, device=device)   # graph replay: 10-17 % on this tree (A/B: DESIGN.md 8a)
    create_s = time.perf_counter() - t0
    tr.train(2)
    tr.status()
    reps = 5    # batches 3-7 of a fresh trainer, as in every round so far (the work per batch FALLS as training proceeds -- fewer live subtrees per deal -- so the window is part of the number)
    t0 = time.perf_counter()
    tr.train(reps)
    tr.infosets.sync()
    dt = (time.perf_counter() - t0) / reps
    tr.train(40)                                # ... and batches 48-67, one call of 20
    tr.infosets.sync()
    t0 = time.perf_counter()
    tr.train(20)
    tr.infosets.sync()
    dt_later = (time.perf_counter() - t0) / 20
    walks_later = [tr.walk_counts(0), tr.walk_counts(1)]
    out = {"what": "MCCFRTrainer::train on a flop-start three-street tree (%d action nodes), %d-bucket files on flop / turn / river, %d deals per batch, sampled "
                   "opponents: deal sampling, hand indexing through the bucket files, showdowns and the sweep on the device" % (n_actions, k, n),
           "value": n / dt, "unit": "deal-iterations/s", "ms_per_batch": dt * 1e3, "n_deals": n, "clusters": [a_.get_size(0) for a_ in card_abs],
           "table_bytes": int(tr.infosets.nbytes if not callable(tr.infosets.nbytes) else tr.infosets.nbytes()), "trainer_create_s": create_s,
           "ms_per_batch_batches_48_67": dt_later * 1e3, "walks_per_round_batch_67": walks_later}
    out["roofline_deals"] = roofline_deals("three_street_4m", out["ms_per_batch"])
    sizes = [(a_.get_size(0), a_.get_size(1)) for a_ in card_abs]
    tr.destroy()
    # the same game at the batch sizes that decide time-to-exploitability: 64 K and 4 K deals per batch, batches 4-43 of a fresh trainer
    for nn, key in ((1 << 16, "ms_per_batch_64k"), (1 << 12, "ms_per_batch_4k")):
        try:
            ts = rs.DealTrainer(tree, card_abs, [hands, hands], mask, nn, seed=7, discount_interval=0, use_graph=True, device=device)
            ts.train(3)
            ts.status()
            t0 = time.perf_counter()
            ts.train(40)
            ts.infosets.sync()
            out[key] = (time.perf_counter() - t0) / 40 * 1e3
            ts.destroy()
        except Exception as e:
            out[key] = None
            out[key + "_error"] = str(e)
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


def jit_leg(device):
    """Cold start of the deal path's generated kernels: creating the three-street deal trainer (the 706-node tree's round subtrees: several dozen
    hipRTC kernels) in a FRESH process with an EMPTY kernel cache and comgr's own cache switched off, then once more with the cache the first run left."""
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    prog = ("import sys, time; sys.path.insert(0, %r)\nimport numpy as np\nimport rustsolver_amd as rs\nfrom rustsolver_amd import abstraction as ab\n"
            "rng = np.random.Generator(np.random.PCG64(1)); mask = ab.card_mask('7h8hQc'); hands = ab.random_range(mask)\n"
            "files = [rng.integers(0, 5000, size=s, dtype=np.uint32) for s in (1286792, 13960050, 123156254)]\n"
            "n_actions, tree = rs.build_game_tree(rs.three_street_options())\n"
            "card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]\n"
            "t0 = time.perf_counter(); tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, 1 << 22, seed=7, discount_interval=0, use_graph=True, device=%d)\n"
            "print('CREATE_S', time.perf_counter() - t0)\n") % (root, device)
    cache = tempfile.mkdtemp(prefix="rs_jit_bench_")
    out = {"what": "seconds rs.DealTrainer(...) takes on the three-street tree (5 000-bucket files, 4 M deals per batch) in a fresh process: empty kernel cache "
                   "(and AMD_COMGR_CACHE=0), then with the cache that run left"}
    try:
        for key in ("cold_s", "warm_s"):
            env = dict(os.environ, RS_JIT_CACHE=cache, AMD_COMGR_CACHE="0")
            r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("CREATE_S")]
            if r.returncode != 0 or not line:
                out["error"] = (r.stderr or r.stdout)[-300:]
                break
            out[key] = float(line[0].split()[1])
            if key == "cold_s":
                out["kernels"] = len([f for f in os.listdir(cache) if f.endswith(".hsaco")])
    finally:
        shutil.rmtree(cache, ignore_errors=True)
    return out


def make_comm(table, dist, rank, n_gpus):
    """an RCCL communicator over the ranks of the torch process group (its unique id travels through one broadcast)"""
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
    L.check(L.load().rs_comm_create(table._h, ident, rank, n_gpus, C2.byref(comm)))
    return comm


def _step_stats(ms):
    ms = sorted(ms)
    if not ms:
        return None
    return {"min": ms[0], "median": ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]), "max": ms[-1], "n": len(ms)}


def _max_over_ranks(dist, x):
    if dist is None:
        return x
    import torch
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class SetupFailed(RuntimeError):
    """raised by _agree on EVERY rank alike: the leg can be skipped without leaving anybody in a collective"""


def _agree(dist, make, what):
    """Runs `make()` (a rank-local setup step that may fail: allocation, kernel compilation) and lets every rank learn whether it worked EVERYWHERE before
    anybody enters a collective: one failed rank would otherwise leave the others waiting in RCCL until the launcher's timeout."""
    err, result = None, None
    try:
        result = make()
    except Exception as e:   # reported on every rank below
        err = e
    if dist is not None:
        import torch
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and err is None:
            err = RuntimeError("%s failed on another rank" % what)
    if err is not None:
        raise SetupFailed("%s: %s" % (what, err))
    return result


def dp_deals_leg(rs, dist, rank, n_gpus, device, steps, warmup, n=1 << 22):
    """MCCFRTrainer::train as coded (see deal_trainer_leg), data-parallel: every rank deals and sweeps n deals of each global batch against
    its replica of the table; per traverser sweep the two i32 delta arrays are all-reduced over RCCL (xGMI) and every rank applies the union.
    WEAK scaling; `value` = deal-iterations/s of the whole job."""
    import torch
    from rustsolver_amd import _lib as L
    from rustsolver_amd import abstraction as ab
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    def make():
        t = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, n, seed=7, discount_interval=0, device=device, world=n_gpus, rank=rank)
        t.infosets.fill_random(4321, (-10**6, 10**6), (0, 10**6))   # same seed on every rank: the replicas start identical
        t.infosets.sync()
        return t
    tr = _agree(dist, make, "dp_deals trainer")
    comm = make_comm(tr.infosets, dist, rank, n_gpus)
    tr.attach_comm(comm)

    def barrier():
        tr.infosets.sync()
        dist.barrier()
        torch.cuda.synchronize()

    tr.train(warmup)
    barrier()
    t0 = time.perf_counter()
    tr.train(steps)
    barrier()
    elapsed = _max_over_ranks(dist, time.perf_counter() - t0)
    tr.status()
    # the replicas must still be identical: compare a checksum of rank 0's table with everybody's
    cr, cs = tr.infosets.checksum()
    c = torch.tensor([(cr ^ (cs * 1000003)) & ((1 << 62) - 1)], dtype=torch.int64, device="cuda")
    cmin, cmax = c.clone(), c.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    cells = tr.infosets.cells
    exchanged = tr.exchange_bytes() / max(1, warmup + steps)   # what this rank handed to the collectives per global batch (rs_solver_exchange_bytes)
    tr.attach_comm(None)
    L.load().rs_comm_destroy(comm)
    tr.destroy()
    return {"metric": "mccfr_deal_iterations_per_sec", "value": n * n_gpus * steps / elapsed, "unit": "deal-iterations/s", "n_gpus": n_gpus,
            "rccl_ranks": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "dtype": "i32",
            "workload": "MCCFRTrainer::train as coded (default_flop board, random ranges, ISOMORPHIC river, 1081 clusters), data-parallel: %d deals "
                        "per rank and batch, 1 step = 1 global batch (both traversers), the traverser's delta cells all-reduced as ncclInt32" % n,
            "parallelism": "dp%d: replicated table of %d i32 cells per array; per traverser sweep ONE all-reduce of that traverser's packed delta cells (rounds with direct "
                           "rows: their rows all-gathered as 12-byte items instead)" % (n_gpus, cells),
            "exchange_bytes_per_batch_and_rank": exchanged, "exchange_bytes_per_batch_round4": 2 * 2 * 4 * cells,
            "replicas_identical": bool(cmin.item() == cmax.item())}


def three_street_sweep_leg(rs, dist, rank, n_gpus, device, a, steps, dtype="i32"):
    """BASELINE configs[2] (N = 1: `config3`) and configs[3] (N > 1: `config4`): the 706-action-node flop+turn+river tree, 5 000 clusters on every
    round, boards 1 / 49 / 2 352, i32 tables (135 GB in all), full-width cfr() with ENUM chance nodes (cfr.rs:502-522).  N > 1: turn and river
    boards sharded over the ranks, flop replicated, ONE RCCL all-gather of the turn-root utility rows per traverser sweep between phase 0
    (> 99 % of the bytes) and phase 1: STRONG scaling, bit-identical to the single-GPU sweep for any rank count."""
    from rustsolver_amd import _lib as L
    from rustsolver_amd.dist import shard_boards
    C_, G = 5000, ([1, 49, 2352] if dtype != "f16" else [1, 98, 4704])   # BASELINE configs[4]: binary16 tables hold twice the boards in the same 135 GB
    cfg_no = (3 if n_gpus == 1 else 4) if dtype != "f16" else 5
    boards3, shard = list(G), None
    if n_gpus > 1:
        tlo, thi = shard_boards(G[1], rank, n_gpus)
        shard = (n_gpus, rank, 1, G[1])
        boards3 = [G[0], thi - tlo, (thi - tlo) * (G[2] // G[1])]
    t0 = time.perf_counter()
    tr = _agree(dist if n_gpus > 1 else None, lambda: make_trainer(rs, boards3, C_, a.mode, 0, device, 1234 + 2 + rank, a.fuse, "three-street", dtype, "full", shard),
                "config%d tables and launch plan" % cfg_no)
    create_s = time.perf_counter() - t0
    table = tr.infosets
    comm = None
    if shard is not None:
        comm = make_comm(table, dist, rank, n_gpus)
        tr.attach_comm(comm)
    lib = L.load()

    def barrier():
        table.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(tr, 2)
    barrier()
    table.profile_mark()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_steps(tr, 1)
        table.profile_mark()
    barrier()
    elapsed = _max_over_ranks(dist, time.perf_counter() - t0)
    step_ms = table.profile_marks()
    out = {"workload": "config%d flop+turn+river: 706-action-node tree, %d clusters per round, boards %s%s, %s tables, %s, "
                       "full-width opponents, ENUM chance; 1 step = 1 CFR iteration (both traversers, all lanes)"
                       % (cfg_no, C_, "/".join(str(b) for b in G), "" if n_gpus == 1 else " (global; turn and river boards sharded x%d)" % n_gpus,
                          {"i32": "i32", "f32": "f32", "f16": "binary16 (f32 arithmetic in registers)"}[dtype],
                          ("cfr.rs:413-464 clamp update scale 100" if a.mode == "clamp" else "cfr.rs:612-621 wrap update scale 10000") if dtype == "i32" else
                          "r += (scale * reach) * (u - util), scale 2^-12"),
           "n_gpus": n_gpus, "scaling": "strong", "steps": steps, "ms_per_iteration": elapsed / steps * 1e3, "step_ms_hip_events_rank0": _step_stats(step_ms),
           "value": G[2] * steps / elapsed, "unit": "river-board-iterations/s (global)",
           "table_bytes_per_gpu": table.nbytes, "workspace_bytes_per_gpu": tr.workspace_bytes, "launches_per_iteration": tr.n_launches(0) + tr.n_launches(1),
           "fused_subtrees": bool(tr.fused), "trainer_create_s": create_s}
    if shard is not None:   # the same iterations driven phase by phase, with the all-gather bracketed by events (rank 0's stream)
        for _ in range(steps):
            for p_ in (0, 1):
                buf, nbytes = tr.exchange_info(p_)
                table.profile_mark()
                L.check(lib.rs_iterate_phase(tr._h, p_, 0, None))
                table.profile_mark()
                L.check(lib.rs_comm_allgather(comm, table._h, buf, nbytes))
                table.profile_mark()
                L.check(lib.rs_iterate_phase(tr._h, p_, 1, None))
        table.profile_mark()
        ms = table.profile_marks()
        ph = [sum(ms[k::3]) / steps for k in range(3)]
        out["phases_ms_per_iteration_rank0"] = {"phase0_sharded_rounds": ph[0], "allgather": ph[1], "phase1_replicated_rounds": ph[2]}
        out["allgather_bytes_per_rank_and_sweep"] = int(nbytes)
        out["rccl_ranks"] = n_gpus
    # per-kernel rates: the same iterations with every launch bracketed by HIP events
    barrier()
    table.profile_reset()
    table.profile_enable(True)
    run_steps(tr, min(steps, 3))
    prof = table.profile_read()
    table.profile_enable(False)
    k_ = min(steps, 3)
    out["kernels"] = {k: {"launches_per_iteration": v["launches"] / k_, "ms_per_iteration": v["ms"] / k_,
                          "algo_GBps": (v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None,
                          "frac_of_8TBps": (v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if v["ms"] > 0 else None}
                      for k, v in prof.items() if v["launches"]}
    tot_b, tot_ms = sum(v["algo_bytes"] for v in prof.values()), sum(v["ms"] for v in prof.values())
    out["algo_GBps_all_kernels"] = tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else None
    out["table_checksum"] = ["%016x" % x for x in table.checksum()]
    if comm is not None:
        tr.attach_comm(None)
        lib.rs_comm_destroy(comm)
    tr.destroy()
    table.destroy()
    return out


def pmc_traffic(a, kernel):
    """(HBM bytes per launch, file) from the newest committed rocprofv3 PMC passes (profiles/), if they were taken on this exact workload.  PMC
    counters cannot be read from inside the process, so this figure is CANNED: it comes from a tracked file, not from this run."""
    import glob
    if (a.boards, a.clusters, a.mode, a.tree, a.dtype, a.opp) != (9216, 1000, "clamp", "river", "i32", "full"):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline_traffic_%s.json" % kernel)))
    if not files:
        return None, None
    try:
        rec = json.load(open(files[-1]))
        import hashlib
        h = hashlib.sha256()
        for f in ("rs_device.hpp", "rs_jit.cpp", "rs_kernels.hip"):
            h.update(open(os.path.join(ROOT, "rustsolver_amd", "csrc", f), "rb").read())
        pmc_traffic.info = {"commit": rec.get("commit"), "kernel_source_sha16": rec.get("kernel_source_sha16"),
                            "stale": (rec.get("kernel_source_sha16") != h.hexdigest()[:16]) if rec.get("kernel_source_sha16") else None}
        return float(rec["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


pmc_traffic.info = {}


def compact_line(out):
    """The one line stdout carries: the contract keys, `roofline` and `cpu_baseline` as the task prescribes them, and ONE flat number per auxiliary leg."""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    def r3(x):
        return float("%.4g" % x) if isinstance(x, (int, float)) and not isinstance(x, bool) else x

    if "metric" not in out or "roofline" not in out:   # the dp-deals line and the probes are short already
        return out
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "n_boards_per_gpu", "n_clusters", "regret_fill", "table_bytes_per_gpu", "launches_per_step", "parallelism",
                                          "process_group_ranks") if k in cfg}
    rf = out["roofline"]
    line["roofline"] = {k: r3(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_canned", "traffic_commit", "traffic_stale", "launches", "avg_launch_ms",
                                               "algo_bytes_per_launch", "copy_GBps", "frac_of_copy_on_this_card") if k in rf}
    line["roofline"]["kernel"] = "rs_tree_p{0,1}_lanes" if "rs_tree" in rf.get("kernel", "") else "rs::k_update"
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        line["cpu_baseline"] = {"value": r3(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:160]}
        if "all_cores" in cb:
            line["cpu_baseline"]["all_cores"] = {"value": r3(cb["all_cores"]["value"]), "cores": cb["all_cores"]["cores"]}
    flat = {
        "step_ms_median": g(out, "step_ms_hip_events", "median"),
        "exact_i64_value": g(out, "exact_i64_pass", "value"),
        "gpu_over_cpu": out.get("gpu_over_cpu"), "gpu_over_cpu_all_cores": out.get("gpu_over_cpu_all_cores"),
        "cpu_soa_value": g(out, "cpu_soa", "value"), "cpu_soa_cores": g(out, "cpu_soa", "cores"), "cpu_soa_GBps": g(out, "cpu_soa", "algo_GBps"),
        "gpu_over_cpu_soa": out.get("gpu_over_cpu_soa"),
        "update_node_frac": g(out, "roofline_update_node", "frac"), "discount_frac": g(out, "roofline_discount", "frac"),
        "stream_copy_GBps": g(out, "stream_probe", "copy_GBps"),
        "single_board_us": g(out, "single_board", "us_per_iteration"),
        "deal_batch_Mps": (g(out, "deal_batch", "value") or 0) / 1e6 or None, "deal_batch_ms": g(out, "deal_batch", "ms_per_batch"),
        "deal_trainer_Mps": (g(out, "deal_trainer", "value") or 0) / 1e6 or None, "deal_trainer_ms": g(out, "deal_trainer", "ms_per_batch"),
        "deal_trainer_valu_frac": g(out, "deal_trainer", "roofline_deals", "frac"),
        "deal_trainer_3s_Mps": (g(out, "deal_trainer_three_street", "value") or 0) / 1e6 or None, "deal_trainer_3s_ms": g(out, "deal_trainer_three_street", "ms_per_batch"), "deal_trainer_3s_ms_later": g(out, "deal_trainer_three_street", "ms_per_batch_batches_48_67"),
        "deal_trainer_3s_64k_ms": g(out, "deal_trainer_three_street", "ms_per_batch_64k"), "deal_trainer_3s_4k_ms": g(out, "deal_trainer_three_street", "ms_per_batch_4k"),
        "deal_trainer_3s_valu_frac": g(out, "deal_trainer_three_street", "roofline_deals", "frac"),
        "solve_expl": [g(out, "solve", "exploitability_before"), g(out, "solve", "exploitability_after")], "solve_s": g(out, "solve", "seconds_training"),
        "solve_3s_expl": [x[1] for x in (g(out, "solve_three_street", "exploitability_curve") or [])] or None, "solve_3s_s": g(out, "solve_three_street", "seconds_training"),
        "solve_3s_br_s": g(out, "solve_three_street", "seconds_best_response_both_players"),
        "solve_3s_full_expl": [x[1] for x in (g(out, "solve_three_street", "full_ranges", "exploitability_curve") or [])] or None,
        "solve_3s_full_br_s": g(out, "solve_three_street", "full_ranges", "seconds_best_response_both_players"),
        "solve_3s_full_s": g(out, "solve_three_street", "full_ranges", "seconds_training"),
        "kmeans_predict_ms": (g(out, "kmeans_predict", "seconds_per_sweep") or 0) * 1e3 or None, "kmeans_fit_regular_s": g(out, "kmeans_predict", "fit_regular", "seconds"),
        "dp_deals_Mps": (g(out, "dp_deals", "value") or 0) / 1e6 or None,
        "jit_cold_s": g(out, "jit", "cold_s"), "jit_warm_s": g(out, "jit", "warm_s"), "jit_kernels": g(out, "jit", "kernels"),
    }
    for key in ("config3", "config4", "config5"):
        c = out.get(key)
        if not isinstance(c, dict):
            continue
        if "error" in c and "value" not in c:
            flat[key + "_error"] = str(c["error"])[:120]
            continue
        flat[key + "_ms"] = c.get("ms_per_iteration")
        flat[key + "_value"] = c.get("value")
        flat[key + "_launches"] = c.get("launches_per_iteration")
        flat[key + "_frac"] = (c["algo_GBps_all_kernels"] / HBM_PEAK_GBS) if c.get("algo_GBps_all_kernels") else None
        flat[key + "_tree_kernel_frac"] = g(c, "kernels", "tree", "frac_of_8TBps")
        flat[key + "_chance_kernel_frac"] = g(c, "kernels", "chance", "frac_of_8TBps")
        flat[key + "_cpu8"] = g(c, "cpu_baseline", "value")
        flat[key + "_cpu_all"] = g(c, "cpu_baseline", "all_cores", "value")
        flat[key + "_cpu_all_cores"] = g(c, "cpu_baseline", "all_cores", "cores")
        flat[key + "_cpu_soa"] = g(c, "cpu_soa", "value")
        flat[key + "_cpu_soa_cores"] = g(c, "cpu_soa", "cores")
        flat[key + "_gpu_over_cpu8"] = c.get("gpu_over_cpu")
        flat[key + "_gpu_over_cpu_all"] = c.get("gpu_over_cpu_all_cores")
        flat[key + "_gpu_over_cpu_soa"] = c.get("gpu_over_cpu_soa")
        flat[key + "_allgather_ms"] = g(c, "phases_ms_per_iteration_rank0", "allgather")
        if isinstance(c.get("cpu_baseline"), dict) and "error" in c["cpu_baseline"]:
            flat[key + "_cpu_error"] = str(c["cpu_baseline"]["error"])[:120]
    for k, v in flat.items():
        if v is None or v == [None, None]:
            continue
        line[k] = [r3(x) for x in v] if isinstance(v, list) else r3(v)
    for leg in ("single_board", "deal_batch", "deal_trainer", "deal_trainer_three_street", "solve", "solve_three_street", "kmeans_predict", "dp_deals", "exact_i64_pass"):
        if isinstance(out.get(leg), dict) and "error" in out[leg]:
            line[leg + "_error"] = str(out[leg]["error"])[:120]
    return line


def main():
    a = parse()
    if os.environ.get("RS_BENCH_PROBE") and "RANK" in os.environ:
        probe_main(os.environ["RS_BENCH_PROBE"])
        return
    if a.gpus > 1 and "RANK" not in os.environ:
        # not under torch.distributed.run: start the N ranks ourselves, BEFORE anything in this process touches the GPU
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
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
    if a.gpus != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with matching values (or plain `python bench.py --gpus N`, which starts its own ranks)"
                         % (a.gpus, world))

    import rustsolver_amd as rs  # raises if the HIP library is missing (no fallback)
    if rs.device_count() < 1:
        raise RuntimeError("bench.py needs a GPU: the engine has no CPU fallback")
    device = local_rank % max(1, rs.device_count())

    def emit(obj):
        """stdout: ONE compact JSON line (every contract key, `roofline`, `cpu_baseline`, one flat number per leg -- below 4 KB, so that a driver that keeps only a
        tail of the output still holds all of it); the full result with every per-kernel table and descriptive string goes to --detail (bench_detail.json)"""
        line = compact_line(obj)
        try:
            with open(a.detail, "w") as f:
                json.dump(obj, f, indent=1)
            line["detail"] = os.path.relpath(a.detail, ROOT)
        except OSError as e:
            line["detail"] = "not written: %s" % e
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line, separators=(",", ":")))
        sys.stdout.flush()
        os.dup2(2, 1)

    def barrier():
        trainer.infosets.sync()
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    if a.dp_deals:
        if dist is None:
            raise RuntimeError("--dp-deals needs a process group: python bench.py --gpus N --dp-deals 1 (N > 1 starts its own ranks), or torch.distributed.run")
        leg = dp_deals_leg(rs, dist, rank, n_gpus, device, a.steps, a.warmup)
        if rank == 0:
            emit(dict(leg, higher_is_better=True, vs_baseline=None, data="synthetic",
                      config={"workload": leg["workload"], "parallelism": leg["parallelism"], "replicas_identical": leg["replicas_identical"]}))
        dist.destroy_process_group()
        return

    three = a.tree == "three-street"
    boards3 = [int(x) for x in a.boards3.split(",")]
    if three and a.opp == "sample":
        boards3 = [boards3[-1]] * 3          # mccfr(): every lane is one full run-out (pass-through chance nodes)
    shard = None
    if three and n_gpus > 1 and a.opp == "full":
        # BASELINE configs[3]: turn and river boards sharded over the ranks, flop replicated, one RCCL all-gather per sweep
        # (STRONG scaling: the global problem is fixed)
        from rustsolver_amd.dist import shard_boards
        tlo, thi = shard_boards(boards3[1], rank, n_gpus)
        fan = boards3[2] // boards3[1]
        shard = (n_gpus, rank, 1, boards3[1])
        global_river = boards3[2]
        boards3 = [boards3[0], thi - tlo, (thi - tlo) * fan]
    trainer = make_trainer(rs, boards3 if three else a.boards, a.clusters, a.mode, a.graph, device, 1234 + 1 + rank, a.fuse,
                           a.tree, a.dtype, a.opp, shard, saturating=a.saturating)
    comm = None
    if shard is not None:
        comm = make_comm(trainer.infosets, dist, rank, n_gpus)
        trainer.attach_comm(comm)
    if three:
        a.boards = boards3[-1] if shard is None else global_river / n_gpus   # `value` counts river boards (global when sharded)
    table = trainer.infosets

    # ---- warmup, then the timed region: exactly K steps between barrier+sync on both sides.  One event per step boundary on the table's
    # stream (rs_profile_mark: a hipEventRecord, no synchronisation) gives the per-step times of THIS pass ---------------------------------
    run_steps(trainer, a.warmup)
    barrier()
    table.profile_mark()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_steps(trainer, 1)
        table.profile_mark()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = table.profile_marks()
    elapsed = _max_over_ranks(dist, elapsed)

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

    dom_name = "tree" if prof["tree"]["launches"] else "update"
    upd = prof[dom_name]
    achieved = upd["algo_bytes"] / (upd["ms"] * 1e-3) / 1e9 if upd["ms"] > 0 else 0.0
    kernels = {k: {"launches": v["launches"], "ms_per_step": v["ms"] / a.steps,
                   "algo_GBps": (v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
               for k, v in prof.items() if v["launches"]}

    # ---- the per-node river regret-update kernel on its own (20A+8 bytes per lane, all inputs
    # buffers): rs_update_node on the root node (A = 3), HIP-event timed ---------------------------------------
    upd_node = None
    if rank == 0:
        try:
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
    if rank == 0:
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
    traffic, traffic_file = pmc_traffic(a, dom_name)
    out = {
        "metric": "cfr_iterations_per_sec",
        "value": a.boards * n_gpus * a.steps / elapsed,
        "unit": "board-iterations/s",
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if shard is not None else "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "step_ms_hip_events": dict(_step_stats(step_ms) or {}, what="per-step durations of the TIMED pass itself, one event per step boundary on the table's stream (rank 0)"),
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
            "parallelism": ("boards sharded x%d, no collective (nothing replicated in a river-only tree)" % n_gpus) if shard is None else
                           ("turn / river boards sharded x%d, flop replicated, one RCCL all-gather per traverser sweep" % n_gpus),
            "process_group_ranks": (dist.get_world_size() if dist is not None else 1),
            "self_launched": bool(os.environ.get("RS_BENCH_SELF_LAUNCHED")),
        },
        "roofline": {
            "kernel": dom_kernel,
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_canned": traffic is not None,
            # the canned figure's provenance: the commit its PMC passes ran at, and whether the kernel sources (rs_device.hpp, rs_jit.cpp, rs_kernels.hip) have changed since
            "traffic_commit": pmc_traffic.info.get("commit"), "traffic_stale": pmc_traffic.info.get("stale"),
            "traffic_source": ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; NOT measured in this run" % traffic_file) if traffic_file else None,
            "launches": upd["launches"], "avg_launch_ms": upd["ms"] / max(1, upd["launches"]),
            "algo_bytes_per_launch": upd["algo_bytes"] / max(1, upd["launches"]),
            "ms_per_step_timed_pass": elapsed / a.steps * 1e3, "ms_per_step_event_pass": elapsed_prof / a.steps * 1e3,
            "note": "achieved = algorithmic bytes (DESIGN.md section 4) / HIP-event duration of every launch of this kernel in a second, event-bracketed pass "
                    "over the same K steps right after the timed one; both passes' step times are given",
        },
        "roofline_update_node": upd_node,
        "roofline_discount": disc,
        "kernels": kernels,
        "lane_updates_per_sec": sum(table.lanes(n) for n in range(table.n_nodes)) * n_gpus * a.steps / elapsed,
    }

    if rank == 0:
        try:   # the card's own streaming ceiling (cards of one pool differ by more than 10 %): a plain float4 copy, 2 x 2 GiB, on the same stream
            import ctypes as C3
            from rustsolver_amd import _lib as L3
            g = C3.c_double()
            L3.check(L3.load().rs_stream_probe(table._h, 2 << 30, 10, C3.byref(g)))
            out["stream_probe"] = {"copy_GBps": g.value, "what": "nt float4 copy, 2 GiB read + 2 GiB written per launch, best of 1 024 / 4 096 / 16 384 workgroups, HIP events",
                                   "tree_kernel_over_copy": achieved / g.value if g.value > 0 else None}
            out["roofline"]["frac_of_copy_on_this_card"] = out["stream_probe"]["tree_kernel_over_copy"]
            out["roofline"]["copy_GBps"] = g.value   # the card's own ceiling next to `frac`: boxes of this pool differ by several percent on identical code (profiles/r04_headline_ab.md)
        except Exception as e:
            out["stream_probe"] = {"error": str(e)}

    out["config"]["regret_fill"] = ("U{-10^6..10^6}, one cell in %d beyond +-2.1e9 (SURVEY 8(d) saturation note)" % a.saturating) if (a.saturating and a.dtype == "i32") \
        else "U{-10^6..10^6}"
    # ---- the same sweep with utilities that force |delta| >= 2^31 on one lane in 100 (RS_LEAF_UTIL rows with +-1e9 outliers): about half of the waves then take the exact
    # i64 branch of the clamp update (the test is made per wave, rs_device.hpp visit_i32) -- the slow path priced once, beside the headline that stays on the fast one
    if rank == 0 and n_gpus == 1 and not three and a.dtype == "i32" and a.mode == "clamp" and not a.no_extra:
        try:
            t2 = make_trainer(rs, a.boards, a.clusters, a.mode, a.graph, device, 1234 + 77, a.fuse, a.tree, a.dtype, a.opp, None, saturating=a.saturating, outlier_leaves=100)
            run_steps(t2, 3)
            t2.infosets.sync()
            k2 = min(a.steps, 20)
            t0 = time.perf_counter()
            run_steps(t2, k2)
            t2.infosets.sync()
            dt2 = time.perf_counter() - t0
            out["exact_i64_pass"] = {"value": a.boards * k2 / dt2, "unit": "board-iterations/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2,
                                     "what": "the headline sweep with RS_LEAF_UTIL rows ~ U(-1035, 1035) in which one lane in 100 holds +-1e9: (100 * reach) * (u - util) passes 2^31 "
                                             "there and the lane's whole wave takes the exact i64 add-and-clamp (cfr.rs:445-461) instead of the saturating-add fast path"}
            t2.destroy()
            t2.infosets.destroy()
        except Exception as e:
            out["exact_i64_pass"] = {"error": str(e)}

    # the headline's table is no longer needed: the legs below want the memory (config 3 / 4 is 135 GB on one GPU)
    if comm is not None:
        from rustsolver_amd import _lib as L4
        trainer.attach_comm(None)
        L4.load().rs_comm_destroy(comm)
    trainer.destroy()
    table.destroy()

    if a.no_extra or three:
        if rank == 0:
            emit(out)
        if dist is not None:
            dist.destroy_process_group()
        return

    if n_gpus > 1 or dist is not None:
        # ---- N ranks: the two RCCL paths beside the headline (every rank takes part, rank 0 reports) ------------------------------------
        if a.config3_steps > 0:
            try:
                out["config4" if n_gpus > 1 else "config3"] = three_street_sweep_leg(rs, dist, rank, n_gpus, device, a, a.config3_steps)
            except SetupFailed as e:     # every rank saw it before the leg's first collective: skip the leg, keep the rest
                out["config4" if n_gpus > 1 else "config3"] = {"error": str(e)}
            except Exception as e:
                out["config4" if n_gpus > 1 else "config3"] = {"error": str(e)}
                if n_gpus > 1:   # a rank that drops out of a collective leaves the others hanging: report what was measured, then fail the whole job loudly
                    if rank == 0:
                        emit(out)
                    raise
        try:
            out["dp_deals"] = dp_deals_leg(rs, dist, rank, n_gpus, device, 10, 3)
        except SetupFailed as e:
            out["dp_deals"] = {"error": str(e)}
        except Exception as e:
            out["dp_deals"] = {"error": str(e)}
            if n_gpus > 1:
                if rank == 0:
                    emit(out)
                raise
        if rank == 0:
            emit(out)
        dist.destroy_process_group()
        return

    # ---- one GPU: config 3 at size, then the rows either side of the path ---------------------------------------------------------------
    if a.config3_steps > 0:
        try:
            out["config3"] = three_street_sweep_leg(rs, None, 0, 1, device, a, a.config3_steps)
        except Exception as e:
            out["config3"] = {"error": str(e)}
        if not a.no_cpu and "value" in out["config3"]:   # the denominator of BASELINE.md section 3's ">= 6x node-vs-host" on this tree
            try:
                c3 = cpu_config3(a.mode, max(10.0, a.cpu_seconds * 1.5))
                out["config3"].update(c3)
                g = out["config3"]["value"]
                out["config3"]["gpu_over_cpu"] = g / c3["cpu_baseline"]["value"]
                if "all_cores" in c3["cpu_baseline"]:
                    out["config3"]["gpu_over_cpu_all_cores"] = g / c3["cpu_baseline"]["all_cores"]["value"]
                out["config3"]["gpu_over_cpu_soa"] = g / c3["cpu_soa"]["value"]
            except Exception as e:
                out["config3"]["cpu_baseline"] = {"error": str(e)}

    # ---- BASELINE configs[4] on one GPU: binary16 tables, twice the boards, the same 135 GB -----------------------------------------------------------
    if a.config3_steps > 0 and "config5" not in a.skip.split(","):
        try:
            out["config5"] = three_street_sweep_leg(rs, None, 0, 1, device, a, a.config3_steps, dtype="f16")
        except Exception as e:
            out["config5"] = {"error": str(e)}

    # ---- single-board latency (the reference-as-coded shape: n_boards = 1), hipGraph replay ------------------
    try:
        small = make_trainer(rs, 1, a.clusters, a.mode, 1, device, 99, a.fuse, a.tree, a.dtype, a.opp)
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

    try:   # before the legs that would leave these kernels in comgr's cache
        out["jit"] = jit_leg(device)
    except Exception as e:
        out["jit"] = {"error": str(e)}

    try:
        out["deal_trainer_three_street"] = three_street_leg(rs, device, not a.no_cpu, min(a.cpu_seconds, 5.0))
    except Exception as e:
        out["deal_trainer_three_street"] = {"error": str(e)}

    try:
        out["solve"] = solve_leg(rs, device)
    except Exception as e:
        out["solve"] = {"error": str(e)}

    if "solve_three_street" not in a.skip.split(","):
        try:
            out["solve_three_street"] = solve_three_street_leg(rs, device)
        except Exception as e:
            out["solve_three_street"] = {"error": str(e)}

    try:
        out["kmeans_predict"] = kmeans_leg(rs, device, not a.no_cpu, min(a.cpu_seconds, 5.0))
    except Exception as e:
        out["kmeans_predict"] = {"error": str(e)}

    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(a.clusters, a.mode, a.cpu_seconds)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if "all_cores" in out["cpu_baseline"]:
            out["gpu_over_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
        try:
            out["cpu_soa"] = cpu_soa(a.clusters, a.mode, min(a.cpu_seconds, 10.0))
            out["gpu_over_cpu_soa"] = out["value"] / out["cpu_soa"]["value"]
        except Exception as e:
            out["cpu_soa"] = {"error": str(e)}
    emit(out)


if __name__ == "__main__":
    main()
