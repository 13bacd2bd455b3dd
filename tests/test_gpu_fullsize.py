"""GPU: BASELINE configs[2] and configs[4] AT THEIR STATED SIZE -- the 706-action-node flop+turn+river tree, 5 000 clusters on every round, boards
1 / 49 / 2 352 (i32, 135 GB) and 1 / 98 / 4 704 (binary16 tables, the same 135 GB = 2x the cells) -- through the C ABI against the CPU oracle.

How a 135 GB table is compared with a scalar oracle: in the lane model a lane is (board, cluster) and cluster c of a flop lane feeds cluster c
of its 49 turn boards and of their 48 river boards each (cfr.rs:502-522 enumerates boards, never clusters), so the sweep DECOMPOSES BY CLUSTER:
the cells of cluster c on all 1 + 49 + 2 352 boards depend on nothing but themselves and the leaf inputs of those lanes.  The oracle therefore
runs the whole three-street tree for a few sampled clusters (an OracleTable with n_clusters = len(sample), its cells initialised from the host
mirror of the device-side fill, rustsolver_amd/synth.py) and every cell of those clusters -- 1.35 M info sets per cluster -- and the root
utilities are compared bit for bit; the fused plan and the level plan must additionally agree on a checksum of the WHOLE table.
These are the shapes where 64-bit offsets, 16 384-lane tiles on the turn / river nodes and the 16-byte chance kernels are actually exercised."""
import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import _lib as L
from rustsolver_amd import synth

pytestmark = pytest.mark.gpu

SEED = 1234 + 2


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if rs.device_count() < 1:
        pytest.fail("no HIP device visible: GPU parity tests need a real MI355X (there is no CPU fallback)")


def build_gpu(G, C, dtype, fill):
    n, tree = rs.build_game_tree(rs.three_street_options())
    table = rs.create_infosets(n, tree, [C], G, dtype)
    table.fill_random(SEED, *fill)
    signs, leaves = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                signs[r] = (table.lane_buffer(parent.index, 1), table.pitch(parent.index))
                L.check(L.load().rs_fill_uniform_f32(table._h, signs[r][0].ptr, signs[r][1], SEED + 17 + r, -1.0, 1.0))
            leaves[i] = (rs.LEAF_SIGN, signs[r][0])
    return n, tree, table, signs, leaves


def build_oracle(tree, table, signs, G, C, sample, odtype, fill):
    """the oracle's view of the sampled clusters: lane (b, j) of the oracle = lane (b, sample[j]) of the GPU table"""
    k = len(sample)
    otree = orc.OracleTree(orc.options_three_street())
    otab = orc.OracleTable(otree, G, k, odtype)
    lanes_of_round = {r: (np.arange(G[r], dtype=np.int64)[:, None] * C + np.asarray(sample, dtype=np.int64)[None, :]).reshape(-1) for r in range(3)}
    for nd in tree.action_nodes():
        if nd.n_children == 0:
            continue
        ln = lanes_of_round[nd.round_idx]
        R = synth.table_lane_values(table, nd.index, SEED, fill[0][0], fill[0][1], ln)
        S = synth.table_lane_values(table, nd.index, SEED, fill[1][0], fill[1][1], ln, ssum=True)
        if odtype == orc.T_I32:
            otab.set_node(nd.index, R.astype(np.int32), S.astype(np.int32))
        else:
            otab.set_node(nd.index, R.astype(np.float32), S.astype(np.float32))
    sign_host = {r: synth.uniform_f32(SEED + 17 + r, signs[r][1], -1.0, 1.0)[lanes_of_round[r]] for r in signs}
    dicts = otree.as_dicts()
    leaves_o = {d["id"]: (orc.LEAF_SIGN, sign_host[dicts[d["parent"]]["round_idx"]]) for d in dicts
                if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    return otree, otab, leaves_o, lanes_of_round


def compare_sample(tree, table, otab, lanes_of_round):
    n_cells = 0
    for nd in tree.action_nodes():
        if nd.n_children == 0:
            continue
        r, s = table.get_infosets(nd.index, lanes_of_round[nd.round_idx])
        ro, so = otab.get_node(nd.index)
        if r.dtype == np.int32:
            assert (r == ro).all() and (s == so).all(), "table differs from the oracle at node %d (round %d)" % (nd.index, nd.round_idx)
        else:
            assert (bits(r) == bits(ro)).all() and (bits(s) == bits(so)).all(), "table differs from the oracle at node %d" % nd.index
        n_cells += r.size
    return n_cells


def run_case(G, C, sample, dtype, odtype, fill, scale, mode_g, mode_o, check_level_plan):
    n, tree, table, signs, leaves = build_gpu(G, C, dtype, fill)
    assert table.nbytes > 130e9, "this test is about the full-size table"
    assert sum(table.tile_lanes(nd.index) != table.pitch(nd.index) for nd in tree.action_nodes() if nd.n_children) == 574, "the 574 river nodes (11.76 M lanes) are tiled at this size, flop and turn nodes (< 2^20 lanes) are plain"
    otree, otab, leaves_o, lanes_of_round = build_oracle(tree, table, signs, G, C, sample, odtype, fill)
    osol = orc.OracleSolver(otree, otab, leaves_o, scale=scale, mode=mode_o, chance_mode=orc.CHANCE_ENUM)
    tr = rs.MCCFRTrainer(tree, table, leaves, scale=scale, mode=mode_g, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=1)
    iters = 2
    for it in range(iters):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=8)
            assert (bits(got[np.asarray(sample)]) == bits(want)).all(), "root utilities of the sampled clusters, it=%d p=%d" % (it, player)
    cells = compare_sample(tree, table, otab, lanes_of_round)
    assert cells == len(sample) * (38 + G[1] * 310 + G[2] * 1430)
    fused_sum = table.checksum()
    launches_fused = tr.n_launches(0) + tr.n_launches(1)
    tr.destroy()
    if check_level_plan:   # the level-by-level plan on the same inputs must leave the same 135 GB, bit for bit
        table.fill_random(SEED, *fill)
        lv = rs.MCCFRTrainer(tree, table, leaves, scale=scale, mode=mode_g, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=0)
        for it in range(iters):
            for player in (0, 1):
                lv.iterate(player)
        assert table.checksum() == fused_sum, "fused and level plans disagree somewhere in the full-size table"
        assert launches_fused < lv.n_launches(0) + lv.n_launches(1)
        lv.destroy()
    # the discount sweep at this size (cfr.rs:250-261): sampled clusters again
    d = rs.discount_factor(300001)
    if check_level_plan:
        table.discount(d)
        otab.discount(d)
        compare_sample(tree, table, otab, lanes_of_round)
    table.destroy()


def test_full_size_config3_sampled_clusters_match_oracle():
    """configs[2]: i32 tables, boards 1 / 49 / 2 352, 5 000 clusters: 135 GB, cfr() with ENUM chance (cfr.rs:502-522, :559-625)"""
    run_case([1, 49, 2352], 5000, [0, 1234, 2500, 4999], rs.I32, orc.T_I32, ((-10**6, 10**6), (0, 10**6)), 10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32, True)


def test_full_size_config5_f16_tables_twice_the_boards():
    """configs[4] on one GPU: binary16 regret / strategy tables with f32 arithmetic, 2x the boards in the same 135 GB"""
    run_case([1, 98, 4704], 5000, [7, 4321], rs.F16, orc.T_F16, ((-2000, 2000), (0, 2000)), 2.0 ** -12, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64, False)


@pytest.mark.parametrize("fuse", [1, 0])
def test_config3_cluster_count_whole_table_on_few_boards(fuse):
    """5 000 clusters (16-byte chance kernels, n_clusters % 4 == 0, rows of 20 KB) on boards 1 / 2 / 6: small enough for the oracle to hold the WHOLE
    table, so every cell is compared, clamp update with saturating and prunable values planted"""
    G, C = [1, 2, 6], 5000
    n, tree = rs.build_game_tree(rs.three_street_options())
    table = rs.create_infosets(n, tree, [C], G)
    otree = orc.OracleTree(orc.options_three_street())
    otab = orc.OracleTable(otree, G, C)
    rng = np.random.Generator(np.random.PCG64(33))
    for nd in tree.action_nodes():
        a, lanes = nd.n_children, table.lanes(nd.index)
        R = rng.integers(-10**6, 10**6, size=(a, lanes)).astype(np.int32)
        S = rng.integers(0, 10**6, size=(a, lanes)).astype(np.int32)
        if a:
            R[a - 1, ::13] = 2_147_000_000
            S[0, ::17] = 2_147_400_000
        table.upload_node(nd.index, R, S)
        otab.set_node(nd.index, R, S)
    signs, lg, lo = {}, {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                sv = rng.integers(-1, 2, size=G[r] * C).astype(np.float32)
                signs[r] = (sv, table.lane_buffer(parent.index, 1, sv))
            lg[i] = (rs.LEAF_SIGN, signs[r][1])
            lo[i] = (orc.LEAF_SIGN, signs[r][0])
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=fuse)
    osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_ENUM)
    for player in (0, 1):
        got = tr.iterate(player, want_root_util=True)
        want = osol.iterate(player, threads=8)
        assert (bits(got) == bits(want)).all()
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index
    table.destroy()


def test_batched_get_infosets_equals_single_gets():
    """rs_get_infosets on plain and tiled node blocks, i32 and binary16, against rs_get_infoset and the uploaded values; out-of-range lanes are refused"""
    import os
    for dtype in (rs.I32, rs.F16):
        for tile in (None, "64"):
            if tile:
                os.environ["RS_TABLE_TILE_LANES"] = tile
            try:
                n, tree = rs.build_game_tree(rs.default_flop())
                table = rs.create_infosets(n, tree, [50], [3], dtype)
            finally:
                os.environ.pop("RS_TABLE_TILE_LANES", None)
            rng = np.random.Generator(np.random.PCG64(3))
            nd = tree.action_nodes()[1]
            lanes = table.lanes(nd.index)
            assert (table.tile_lanes(nd.index) != table.pitch(nd.index)) == bool(tile)
            if dtype == rs.I32:
                R = rng.integers(-2**31, 2**31 - 1, size=(nd.n_children, lanes)).astype(np.int32)
                S = rng.integers(0, 2**31 - 1, size=(nd.n_children, lanes)).astype(np.int32)
            else:
                R = rng.uniform(-100, 100, size=(nd.n_children, lanes)).astype(np.float16).astype(np.float32)
                S = rng.uniform(0, 100, size=(nd.n_children, lanes)).astype(np.float16).astype(np.float32)
            table.upload_node(nd.index, R, S)
            pick = rng.permutation(lanes)[:40]
            r, s = table.get_infosets(nd.index, pick)
            assert r.tobytes() == R[:, pick].tobytes() and s.tobytes() == S[:, pick].tobytes()
            with pytest.raises(rs.RsError) as e:
                table.get_infosets(nd.index, [0, lanes])
            assert e.value.code == L.ERR_OOB
            c0 = table.checksum()
            one = table.download_node(nd.index)
            one[0][0, 0] += 1
            table.upload_node(nd.index, one[0], one[1])
            c1 = table.checksum()
            assert c0[0] != c1[0] and c0[1] == c1[1], "the checksum sees a one-cell change in exactly the array that changed"
            table.destroy()


def test_stale_jit_cache_is_recompiled(tmp_path):
    """a corrupt or foreign code object in the on-disk kernel cache must be thrown away and compiled again, not fail solver creation (fresh processes: the
    in-process cache would hide the file)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys; sys.path.insert(0, %r); import numpy as np; import rustsolver_amd as rs\n"
            "n, tree = rs.build_game_tree(rs.default_flop()); t = rs.create_infosets(n, tree, [8], [2])\n"
            "sv = np.ones(16, dtype=np.float32); b = t.lane_buffer(0, 1, sv)\n"
            "lv = {i: (rs.LEAF_SIGN, b) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}\n"
            "tr = rs.MCCFRTrainer(tree, t, lv, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=1)\n"
            "u = tr.iterate(0, want_root_util=True); print('UTIL', u[:16].view(np.uint32).tolist())\n") % root
    env = dict(os.environ, RS_JIT_CACHE=str(tmp_path))
    first = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert first.returncode == 0, first.stderr[-2000:]
    blobs = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert blobs, "the first run must have written its kernels to the cache"
    for f in blobs:
        with open(os.path.join(tmp_path, f), "wb") as fh:
            fh.write(b"not a code object" * 10)
    second = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert second.returncode == 0, second.stderr[-2000:]
    assert [ln for ln in first.stdout.splitlines() if ln.startswith("UTIL")] == [ln for ln in second.stdout.splitlines() if ln.startswith("UTIL")]
    for f in blobs:
        assert os.path.getsize(os.path.join(tmp_path, f)) > 1000, "the unusable blob must have been replaced by a fresh compile"


def test_short_division_of_regret_matching_equals_the_compilers():
    """regret matching on i32 tables divides with div_exact_pos (rs_device.hpp): the compiler's f32 division minus the operand scaling and the special-case fix-up,
    valid for positive normal operands -- 2^28 hashed (regret, sum of positive regrets) pairs must give the same bits as `a / b` on the device"""
    import ctypes as C
    n, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n, tree, [4], [1])
    bad, first = C.c_uint64(123), (C.c_float * 2)()
    for seed in (1, 2):
        L.check(L.load().rs_selftest_division(table._h, 1 << 28, seed, C.byref(bad), first))
        assert bad.value == 0, "div_exact_pos differs from a / b, first at a = %r b = %r" % (first[0], first[1])
    table.destroy()


def _logical_setup(boards, lane_off, C, seed, dtype=rs.I32, shard=None, scale=10000.0, mode=None):
    """three-street table + sign rows + solver whose contents are a function of the LOGICAL cell (node, action, global lane): rs_table_fill_random_logical /
    rs_fill_uniform_f32_at with the slice's global lane offsets"""
    from rustsolver_amd import _lib as L
    import ctypes as C_
    lib = L.load()
    n, tree = rs.build_game_tree(rs.three_street_options())
    table = rs.create_infosets(n, tree, [C], boards, dtype)
    off = (C_.c_uint64 * 3)(*lane_off)
    rng_r, rng_s = ((-10**6, 10**6), (0, 10**6)) if dtype == rs.I32 else ((-2000, 2000), (0, 2000))
    L.check(lib.rs_table_fill_random_logical(table._h, seed, rng_r[0], rng_r[1], rng_s[0], rng_s[1], off))
    signs, leaves = {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                signs[r] = table.lane_buffer(parent.index, 1)
                L.check(lib.rs_fill_uniform_f32_at(table._h, signs[r].ptr, boards[r] * C, seed + 17 + r, -1.0, 1.0, lane_off[r]))
            leaves[i] = (rs.LEAF_SIGN, signs[r])
    tr = rs.MCCFRTrainer(tree, table, leaves, scale=scale, mode=rs.UPD_WRAP_I32 if mode is None else mode, chance_mode=rs.CHANCE_ENUM, shard=shard)
    return tree, table, tr, off


def _logical_checksum(table, off):
    from rustsolver_amd import _lib as L
    import ctypes as C_
    out = (C_.c_uint64 * 6)()
    L.check(L.load().rs_table_checksum_logical(table._h, off, out))
    return [int(x) for x in out]


@pytest.mark.parametrize("dtype", ["i32", "f16"])
def test_config4_shape_eight_emulated_ranks_equal_the_single_gpu_table(dtype):
    """BASELINE configs[3] (and the 8-GPU half of configs[4]) at its real shape on ONE GPU: the 706-node tree, 5 000 clusters, boards 1 / 49 / 2 352 (i32; binary16: 1 / 98 /
    4 704), first unsharded (135 GB), then as EIGHT ranks -- 6 or 7 turn boards and 288 or 336 river boards each (12 / 13 and 576 / 624 for binary16), flop replicated -- that
    live side by side in the same GPU's memory and are driven phase by phase, the all-gather of the turn-root utility slots done by hand between phase 0 and phase 1 (in
    production: one ncclAllGather over xGMI).  Contents are a function of the logical cell, so the union of the ranks' turn / river checksums must equal the unsharded table's
    and every rank's replicated flop tables must equal its flop tables.  What this cannot show is the RCCL call itself on more than one physical GPU."""
    from rustsolver_amd import _lib as L
    from rustsolver_amd.dist import shard_boards
    lib = L.load()
    Cn, W, iters = 5000, 8, 2
    G = [1, 49, 2352] if dtype == "i32" else [1, 98, 4704]
    dt = rs.I32 if dtype == "i32" else rs.F16
    scale, mode = (10000.0, rs.UPD_WRAP_I32) if dtype == "i32" else (2.0 ** -12, rs.UPD_CLAMP_I64)
    fan = G[2] // G[1]
    tree, table, tr, off = _logical_setup(G, [0, 0, 0], Cn, SEED, dt, None, scale, mode)
    roots = []
    for it in range(iters):
        for player in (0, 1):
            roots.append(tr.iterate(player, want_root_util=True).copy())
    want = _logical_checksum(table, off)
    tr.destroy()
    table.destroy()
    ranks = []
    for g in range(W):
        tlo, thi = shard_boards(G[1], g, W)
        boards = [1, thi - tlo, (thi - tlo) * fan]
        assert boards[1] in ((6, 7) if dtype == "i32" else (12, 13))
        ranks.append(_logical_setup(boards, [0, tlo * Cn, tlo * fan * Cn], Cn, SEED, dt, (W, g, 1, G[1]), scale, mode))
    k = 0
    for it in range(iters):
        for player in (0, 1):
            for (_, tb, sv, _) in ranks:
                sv.iterate_phase(player, 0)
            slots = []
            for g, (_, tb, sv, _) in enumerate(ranks):   # the all-gather by hand: rank g's slot goes to everybody
                ptr, nbytes = sv.exchange_info(player)
                host = np.empty(nbytes // 4, dtype=np.float32)
                L.check(lib.rs_d2h(tb._h, host.ctypes.data, ptr + g * nbytes, nbytes))
                slots.append(host)
            for (_, tb, sv, _) in ranks:
                ptr, nbytes = sv.exchange_info(player)
                for g, host in enumerate(slots):
                    L.check(lib.rs_h2d(tb._h, ptr + g * nbytes, host.ctypes.data, nbytes))
            for (_, tb, sv, _) in ranks:
                got = sv.iterate_phase(player, 1, want_root_util=True)
                assert (bits(got) == bits(roots[k])).all(), "root utilities of a sharded rank, it=%d p=%d" % (it, player)
            k += 1
    union = [0] * 6
    for g, (_, tb, sv, off_g) in enumerate(ranks):
        cs = _logical_checksum(tb, off_g)
        assert cs[0:2] == want[0:2], "rank %d: the replicated flop tables differ from the single-GPU run's" % g
        for j in range(2, 6):
            union[j] = (union[j] + cs[j]) & (2**64 - 1)
    assert union[2:6] == want[2:6], "the union of the ranks' turn / river tables differs from the single-GPU table"
    for (_, tb, sv, _) in ranks:
        sv.destroy()
        tb.destroy()


def test_three_street_trainer_kernel_forms_agree_at_size(monkeypatch):
    """The deal path at the size bench.py times it (flop start, 706 action nodes, 5 000-bucket files, a million deals per batch) under the two families of kernel forms: LDS delta
    tiles + one compaction job per root (round 2), and what the engine picks at this size (delta rows in the list walkers, summed per round; sibling roots compacted in one scan;
    strategy records for the opponent's nodes).  The oracle cannot follow at this size; integer deltas commute, so every form must leave the SAME table, cell for cell: whole-table
    checksums after three batches (dealing, indexing, showdowns, both sweeps, prune flags from the second batch on)."""
    from rustsolver_amd import abstraction as ab
    rng = np.random.Generator(np.random.PCG64(1))
    mask = ab.card_mask("7h8hQc")
    hands = ab.random_range(mask)
    K = 5000
    files = [rng.integers(0, K, size=1286792, dtype=np.uint32), rng.integers(0, K, size=13960050, dtype=np.uint32), rng.integers(0, K, size=123156254, dtype=np.uint32)]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
    sums = {}
    for form, env in (("tiles", {"RS_JIT_ROWS": "0", "RS_JIT_ORDERED": "0", "RS_JIT_NO_SIBLINGS": "1", "RS_JIT_NO_STAGE": "1"}), ("engine", {})):
        for k in ("RS_JIT_ROWS", "RS_JIT_ORDERED", "RS_JIT_NO_SIBLINGS", "RS_JIT_NO_STAGE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, 1 << 20, seed=7, discount_interval=0, prune_threshold=1 << 20, use_graph=True)
        tr.train(3)
        tr.status()
        sums[form] = tr.infosets.checksum()
        assert sum(tr.walk_counts(0)) > 3 << 20   # every deal walked its flop subtree and a few turn / river subtrees
        del tr
    assert sums["tiles"] == sums["engine"], sums
