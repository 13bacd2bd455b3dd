"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the committed golden
fixtures.  Bar: BIT-EXACT for RS_I32 tables (regrets, strategy_sum, and every f32 strategy / utility the
reference would compute); RS_F32 / RS_F16 extension modes are also compared bit for bit (same IEEE f32
operations, no FMA contraction) with a documented fallback tolerance of 1e-5 relative."""
import json
import os

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import _lib as L

pytestmark = pytest.mark.gpu


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).reshape(-1).view(np.uint32)


def f32(b):
    return np.array(b, dtype=np.uint32).view(np.float32)


def assert_bits(a, b, what=""):
    a, b = bits(a), bits(b)
    bad = np.nonzero(a != b)[0]
    assert bad.size == 0, "%s: %d/%d f32 values differ, first at %d: %08x vs %08x" % (what, bad.size, a.size, bad[0], a[bad[0]], b[bad[0]])


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if rs.device_count() < 1:
        pytest.fail("no HIP device visible: GPU parity tests need a real MI355X (there is no CPU fallback)")


@pytest.fixture(scope="module")
def known(golden_dir):
    return json.load(open(os.path.join(golden_dir, "known_answers.json")))


@pytest.fixture(scope="module")
def rand(golden_dir):
    return json.load(open(os.path.join(golden_dir, "random_cases.json")))


def one_node_table(A, C, dtype=rs.I32, n_boards=1):
    return rs.InfosetTable.create([(A, C, n_boards, 0, 0)], dtype)


# ---- golden vectors through the ABI -------------------------------------------------------------------

def test_known_answers_update_and_strategy(known):
    # infoset.rs:83-102 + cfr.rs:413-464, scale 100, one info set per lane
    for A in (2, 3):
        kas = [k for k in known["update"] if len(k["regrets"]) == A]
        n = len(kas)
        t = one_node_table(A, n)
        R = np.array([k["regrets"] for k in kas], dtype=np.int32).T
        S = np.array([k["ssum"] for k in kas], dtype=np.int32).T
        U = np.array([k["utils"] for k in kas], dtype=np.float32).T
        reach = np.array([k["reach"] for k in kas], dtype=np.float32)
        t.upload_node(0, R, S)
        sig = t.regret_match_node(0)
        for i, k in enumerate(kas):
            assert bits(sig[:, i]).tolist() == k["strategy_bits"]
            assert bits(t[0][i].get_strategy()).tolist() == k["strategy_bits"]      # Infoset::get_strategy
            assert t[0][i].regrets.tolist() == k["regrets"]                          # get-infoset
        util = t.update_node(0, U, reach, 100.0, rs.UPD_CLAMP_I64)
        r, s = t.download_node(0)
        for i, k in enumerate(kas):
            assert bits(util[i])[0] == k["util_bits"]
            assert r[:, i].tolist() == k["new_regrets"] and s[:, i].tolist() == k["new_ssum"]


def test_known_answers_discount(known):
    x = np.array(known["discount_in"], dtype=np.int32)
    for da in known["discount"]:
        t = one_node_table(1, len(x))
        t.upload_node(0, x[None, :], x[None, :])
        d = rs.discount_factor(da["tc"])
        assert bits(d)[0] == da["d_bits"]
        t.discount(d)                                        # cfr.rs:250-261
        r, s = t.download_node(0)
        assert r[0].tolist() == da["out"] and s[0].tolist() == da["out"]


def test_golden_random_update_cases(rand):
    for c in rand["update"]:
        A, n = c["A"], c["n"]
        R = np.array(c["regrets"], dtype=np.int32).reshape(A, n)
        S = np.array(c["ssum"], dtype=np.int32).reshape(A, n)
        U, reach = f32(c["utils_bits"]).reshape(A, n), f32(c["reach_bits"])
        mode = (rs.UPD_CLAMP_I64 if c["mode"] == "clamp" else rs.UPD_WRAP_I32) | (rs.UPD_PRUNE if c["prune"] else 0)
        t = one_node_table(A, n)
        t.upload_node(0, R, S)
        assert_bits(t.regret_match_node(0), f32(c["strategy_bits"]), "strategy")
        util = t.update_node(0, U, reach, c["scale"], mode)
        r, s = t.download_node(0)
        tag = "A=%d %s prune=%s" % (A, c["mode"], c["prune"])
        assert_bits(util, f32(c["util_bits"]), tag)
        assert (r.reshape(-1) == np.array(c["new_regrets"], dtype=np.int32)).all(), tag
        assert (s.reshape(-1) == np.array(c["new_ssum"], dtype=np.int32)).all(), tag


def test_golden_random_discount_cases(rand):
    for c in rand["discount"]:
        x = np.array(c["x"], dtype=np.int32)
        t = one_node_table(1, len(x))
        t.upload_node(0, x[None, :], x[None, :])
        t.discount(f32([c["d_bits"]])[0])
        r, s = t.download_node(0)
        assert r[0].tolist() == c["out"] and s[0].tolist() == c["out"]


def test_golden_extension_cases(rand):
    for c in rand["extension"]:
        A, n = c["A"], c["n"]
        t = one_node_table(A, n, rs.F16 if c["f16"] else rs.F32)
        R, S = f32(c["regrets_bits"]).reshape(A, n), f32(c["ssum_bits"]).reshape(A, n)
        t.upload_node(0, R, S)
        r0, s0 = t.download_node(0)
        assert_bits(r0, R, "upload roundtrip"), assert_bits(s0, S, "upload roundtrip")
        mode = rs.UPD_CLAMP_I64 | (rs.UPD_RMPLUS if c["rmplus"] else 0)
        util = t.update_node(0, f32(c["utils_bits"]).reshape(A, n), f32(c["reach_bits"]), c["scale"], mode)
        r, s = t.download_node(0)
        assert_bits(util, f32(c["util_bits"]), "util")                           # tolerance if this ever fails: 1e-5 rel
        assert_bits(r, f32(c["new_regrets_bits"]), "regrets")
        assert_bits(s, f32(c["new_ssum_bits"]), "ssum")


# ---- per-node kernels vs the oracle on seeded inputs ------------------------------------------------------

@pytest.mark.parametrize("A", [2, 3, 6])
def test_update_clamp_paths_agree_with_the_oracle_around_2_pow_31(A):
    """The clamp update has two paths on the device: one conversion + one saturating add while every |scale * reach * (u - util)| and
    |scale * reach * sigma| of a WAVE stays below 2^31, the i64 form otherwise (rs_device.hpp visit_i32).  Deltas here straddle 2^31 and
    2^32, mixed per lane with ordinary ones, with NaN / inf utilities and regrets at the i32 limits: both paths and their boundary."""
    rng = np.random.Generator(np.random.PCG64(77 + A))
    n = 64 * 40                                                    # 40 waves: some all-small, some mixed, some all-large
    R = rng.integers(-10**6, 10**6, size=(A, n)).astype(np.int32)
    S = rng.integers(0, 10**6, size=(A, n)).astype(np.int32)
    U = rng.uniform(-1035, 1035, size=(A, n)).astype(np.float32)
    reach = rng.uniform(0, 1, size=n).astype(np.float32)
    big = np.zeros(n, dtype=bool)
    big[64 * 10: 64 * 20] = rng.random(640) < 0.1                  # mixed waves
    big[64 * 20: 64 * 30] = True                                    # whole waves on the i64 path
    nb = int(big.sum())
    mag = rng.choice([2.0**31 / 100, 2.0**31 / 100 * 1.0000001, 2.0**32 / 100, 2.0**32 / 100 * 1.001, 1e9, 3e7, 2.1e7], size=(A, nb))
    U[:, big] = (mag * rng.choice([-1.0, 1.0], size=(A, nb))).astype(np.float32)
    reach[big] = 1.0
    edge = np.flatnonzero(big)[:: 7]
    R[:, edge] = rng.choice([2**31 - 1, -2**31, 2**31 - 2, -2**31 + 1, 0], size=(A, len(edge)))
    S[:, edge] = rng.choice([2**31 - 1, 0, 2**31 - 5], size=(A, len(edge)))
    U[0, 64 * 30 + 3] = np.nan                                      # Rust: NaN as i64 = 0
    U[A - 1, 64 * 31 + 9] = np.inf
    U[0, 64 * 32 + 1] = -np.inf
    for rmplus in (False, True):
        t = one_node_table(A, n, rs.I32, 1)
        t.upload_node(0, R, S)
        util = t.update_node(0, U, reach, 100.0, rs.UPD_CLAMP_I64 | (rs.UPD_RMPLUS if rmplus else 0))
        r, s = t.download_node(0)
        for k in range(n):
            if rmplus:
                wu, rk, sk = orc.update_infoset_rmplus(R[:, k], S[:, k], U[:, k], reach[k], 100.0)
            else:
                wu, rk, sk = orc.update_infoset(R[:, k], S[:, k], U[:, k], reach[k], 100.0, orc.UPD_CLAMP_I64, False)
            assert (r[:, k] == rk).all() and (s[:, k] == sk).all(), (A, rmplus, k, U[:, k], R[:, k], r[:, k], rk)
            assert np.float32(util[k]).view(np.uint32) == np.float32(wu).view(np.uint32) or (np.isnan(util[k]) and np.isnan(wu))


@pytest.mark.parametrize("A", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("C,B", [(1, 1), (5, 3), (169, 1), (1000, 2), (1081, 1)])
def test_update_node_vs_oracle(A, C, B):
    rng = np.random.Generator(np.random.PCG64(1000 * A + C + B))
    n = C * B
    R = rng.integers(-10**6, 10**6, size=(A, n)).astype(np.int32)
    S = rng.integers(0, 10**6, size=(A, n)).astype(np.int32)
    sat = rng.random(n) < 0.02                                                    # >= 1% saturating lanes
    R[:, sat] = rng.integers(2_100_000_000, 2_147_483_647, size=(A, int(sat.sum()))) * rng.choice([-1, 1], size=(A, int(sat.sum())))
    S[:, sat] = 2_147_483_000
    U = rng.uniform(-1035, 1035, size=(A, n)).astype(np.float32)
    reach = rng.uniform(0, 1, size=n).astype(np.float32)
    for mode_o, mode_g, scale in ((orc.UPD_CLAMP_I64, rs.UPD_CLAMP_I64, 100.0), (orc.UPD_WRAP_I32, rs.UPD_WRAP_I32, 10000.0)):
        for prune in (False, True):
            t = one_node_table(A, C, rs.I32, B)
            t.upload_node(0, R, S)
            util = t.update_node(0, U, reach, scale, mode_g | (rs.UPD_PRUNE if prune else 0))
            r, s = t.download_node(0)
            want_u = np.zeros(n, dtype=np.float32)
            for k in range(n):
                want_u[k], rk, sk = orc.update_infoset(R[:, k], S[:, k], U[:, k], reach[k], scale, mode_o, prune)
                assert (r[:, k] == rk).all() and (s[:, k] == sk).all(), (A, C, B, mode_o, prune, k)
            assert_bits(util, want_u, "util")
            # read-only kernels on the updated table
            nu = t.node_util(0, U)
            cr = t.child_reach(0, reach)
            fs = t.final_strategy_node(0)
            for k in range(0, n, max(1, n // 97)):
                assert bits(nu[k])[0] == bits(orc.node_util(r[:, k], U[:, k]))[0]
                sig = orc.get_strategy(r[:, k])
                assert_bits(cr[:, k], (sig * reach[k]).astype(np.float32), "child reach")       # cfr.rs:585
                assert_bits(fs[:, k], orc.get_final_strategy(s[:, k]), "final strategy")        # infoset.rs:104-123
            t.destroy()


def test_rmplus_i32_vs_oracle():
    rng = np.random.Generator(np.random.PCG64(5))
    A, n = 3, 300
    R = rng.integers(-10**5, 10**5, size=(A, n)).astype(np.int32)
    S = rng.integers(0, 10**6, size=(A, n)).astype(np.int32)
    U = rng.uniform(-100, 100, size=(A, n)).astype(np.float32)
    reach = rng.uniform(0, 1, size=n).astype(np.float32)
    t = one_node_table(A, n)
    t.upload_node(0, R, S)
    util = t.update_node(0, U, reach, 100.0, rs.UPD_CLAMP_I64 | rs.UPD_RMPLUS)
    r, s = t.download_node(0)
    assert (r >= 0).all()
    for k in range(n):
        wu, rk, sk = orc.update_infoset_rmplus(R[:, k], S[:, k], U[:, k], reach[k], 100.0)
        assert bits(util[k])[0] == bits(wu)[0] and (r[:, k] == rk).all() and (s[:, k] == sk).all()


def test_null_reach_means_one_and_per_board_copies():
    rng = np.random.Generator(np.random.PCG64(6))
    A, C, B = 3, 37, 4
    t = one_node_table(A, C, rs.I32, B)
    for b in range(B):
        t.upload(0, b, rng.integers(-1000, 1000, size=(A, C)).astype(np.int32), rng.integers(0, 1000, size=(A, C)).astype(np.int32))
    R, S = t.download_node(0)
    for b in range(B):
        r, s = t.download(0, b)
        assert (r == R[:, b * C:(b + 1) * C]).all() and (s == S[:, b * C:(b + 1) * C]).all()
    U = rng.uniform(-35, 35, size=(A, B * C)).astype(np.float32)
    util = t.update_node(0, U, None, 100.0, rs.UPD_CLAMP_I64)
    r, s = t.download_node(0)
    for k in range(B * C):
        wu, rk, sk = orc.update_infoset(R[:, k], S[:, k], U[:, k], 1.0, 100.0, orc.UPD_CLAMP_I64)
        assert bits(util[k])[0] == bits(wu)[0] and (r[:, k] == rk).all() and (s[:, k] == sk).all()
    inf = t[0][2, 5]
    assert (inf.regrets == r[:, 2 * C + 5]).all() and (inf.strategy_sum == s[:, 2 * C + 5]).all()
    inf.set([1, 2, 3], [4, 5, 6])
    assert t[0][2, 5].regrets.tolist() == [1, 2, 3] and t[0][2, 5].strategy_sum.tolist() == [4, 5, 6]
    assert_bits(t[0][2, 5].get_final_strategy(), orc.get_final_strategy([4, 5, 6]))


def test_zero_init_and_fill_mirror():
    t = one_node_table(3, 130, rs.I32, 2)
    r, s = t.download_node(0)
    assert not r.any() and not s.any()                        # Infoset::init (infoset.rs:76-81)
    assert_bits(t.regret_match_node(0), np.full((3, 260), np.float32(1.0) / np.float32(3.0)))
    t.fill_random(77, (-10**6, 10**6), (0, 10**6))
    r, s = t.download_node(0)
    assert (r == rs.synth.table_node_values(t, 0, 77, -10**6, 10**6)).all()
    assert (s == rs.synth.table_node_values(t, 0, 77, 0, 10**6, ssum=True)).all()
    buf = t.lane_buffer(0, 1)
    L.check(L.load().rs_fill_uniform_f32(t._h, buf.ptr, t.pitch(0), 5, -1.0, 1.0))
    assert_bits(buf.download(np.float32, t.pitch(0)), rs.synth.uniform_f32(5, t.pitch(0), -1.0, 1.0))


def test_errors_instead_of_panics():
    t = one_node_table(3, 10)
    with pytest.raises(IndexError):
        t[1]
    with pytest.raises(IndexError):
        t[0][10].regrets
    with pytest.raises(rs.RsError) as e:
        L.check(L.load().rs_get_infoset(t._h, 0, 0, 10, None, None))
    assert e.value.code == L.ERR_OOB
    with pytest.raises(rs.RsError) as e:
        t.update_node(0, np.zeros((3, 10), np.float32), None, 100.0, 7)
    assert e.value.code == L.ERR_INVALID
    tf = one_node_table(3, 10, rs.F32)
    with pytest.raises(rs.RsError) as e:
        tf.update_node(0, np.zeros((3, 10), np.float32), None, 100.0, rs.UPD_PRUNE)
    assert e.value.code == L.ERR_UNSUPPORTED
    with pytest.raises(rs.RsError):
        rs.InfosetTable.create([(9, 10, 1, 0, 0)])           # more than RS_MAX_ACTIONS


# ---- rs_iterate / rs_train vs the oracle's per-lane recursion ---------------------------------------------------

def setup_pair(options_rs, options_orc, n_boards, C, seed, dtype=rs.I32, odtype=orc.T_I32, regret_scale=10**6):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_actions, tree = rs.build_game_tree(options_rs)
    table = rs.create_infosets(n_actions, tree, [C], n_boards, dtype)
    otree = orc.OracleTree(options_orc)
    otab = orc.OracleTable(otree, n_boards, C, odtype)
    for nd in tree.action_nodes():
        a, lanes = nd.n_children, table.lanes(nd.index)
        if dtype == rs.I32:
            R = rng.integers(-regret_scale, regret_scale, size=(a, lanes)).astype(np.int32)
            S = rng.integers(0, regret_scale, size=(a, lanes)).astype(np.int32)
            if regret_scale >= 10**6 and a > 0:                                        # prune + saturation lanes
                R[0, ::11] = -10_000_001
                R[a - 1, ::13] = 2_147_000_000
        else:
            R = rng.uniform(-1000, 1000, size=(a, lanes)).astype(np.float32)
            S = rng.uniform(0, 1000, size=(a, lanes)).astype(np.float32)
            if dtype == rs.F16:
                R, S = R.astype(np.float16).astype(np.float32), S.astype(np.float16).astype(np.float32)
        table.upload_node(nd.index, R, S)
        otab.set_node(nd.index, R, S)
    # one sign vector per round (lanes of that round), shared by its showdown / all-in terminals
    signs, leaves_g, leaves_o = {}, {}, {}
    for i, nd in enumerate(tree.nodes):
        if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
            parent = tree.nodes[nd.parent]
            r = parent.round_idx
            if r not in signs:
                sv = rng.integers(-1, 2, size=n_boards[r] * C).astype(np.float32)
                signs[r] = (sv, table.lane_buffer(parent.index, 1, sv))
            leaves_g[i] = (rs.LEAF_SIGN, signs[r][1])
            leaves_o[i] = (orc.LEAF_SIGN, signs[r][0])
    return tree, table, otree, otab, leaves_g, leaves_o


def compare_tables(tree, table, otab, exact=True):
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        if r.dtype == np.int32:
            assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index
        else:
            assert_bits(r, ro, "regrets node %d" % nd.index)
            assert_bits(s, so, "ssum node %d" % nd.index)


@pytest.mark.parametrize("C,B", [(5, 1), (250, 3), (1081, 1), (1000, 2)])
@pytest.mark.parametrize("mode", ["clamp", "wrap", "clamp+prune"])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("fuse", [1, 0])
def test_iterate_river_tree_vs_oracle(C, B, mode, graph, fuse):
    if graph and (C, B) not in ((250, 3), (1081, 1)):
        pytest.skip("graph replay covered on two shapes")
    tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [B], C, C * 7 + B)
    prune = "prune" in mode
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if mode == "wrap" else (100.0, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mg | (rs.UPD_PRUNE if prune else 0), chance_mode=rs.CHANCE_PASS,
                         use_graph=graph, fuse_subtrees=fuse)
    # fused: the whole river tree is ONE tree-specialised launch per traverser, pruned or not
    assert tr.n_launches(0) == (1 if fuse else (10 if prune else 8))
    osol = orc.OracleSolver(otree, otab, lo, scale=scale, mode=mo, prune=prune, chance_mode=orc.CHANCE_PASS)
    for it in range(3):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=4)
            assert_bits(got, want, "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)


@pytest.mark.parametrize("boards,chance", [([1, 2, 6], "enum"), ([1, 1, 1], "enum"), ([2, 2, 2], "pass"), ([1, 3, 3], "enum")])
@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("C", [9, 8])
def test_iterate_three_street_tree_vs_oracle(boards, chance, fuse, C):
    # C = 8 takes the 16-byte chance kernels (n_clusters % 4 == 0), C = 9 the scalar ones
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), boards, C, 42)
    cm_g, cm_o = (rs.CHANCE_ENUM, orc.CHANCE_ENUM) if chance == "enum" else (rs.CHANCE_PASS, orc.CHANCE_PASS)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=10000.0, mode=rs.UPD_WRAP_I32, chance_mode=cm_g, fuse_subtrees=fuse)
    osol = orc.OracleSolver(otree, otab, lo, scale=10000.0, mode=orc.UPD_WRAP_I32, chance_mode=cm_o)
    for it in range(2):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=4)
            assert_bits(got, want, "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)


@pytest.mark.parametrize("boards,chance", [([1, 2, 6], "enum"), ([1, 3, 3], "enum"), ([2, 2, 2], "pass")])
@pytest.mark.parametrize("fuse", [1, 0])
def test_iterate_three_street_tree_pruned_vs_oracle(boards, chance, fuse):
    """cfr() with prune = true over lanes (cfr.rs:379-386): since round 2 the generated kernels (river subtrees, round subtrees and their
    reach-down halves) have pruned forms, so a pruned sweep takes the same launch plan as an unpruned one.  Every 11th lane of action 0
    sits below the threshold."""
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), boards, 12, 43)
    cm_g, cm_o = (rs.CHANCE_ENUM, orc.CHANCE_ENUM) if chance == "enum" else (rs.CHANCE_PASS, orc.CHANCE_PASS)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, chance_mode=cm_g, fuse_subtrees=fuse)
    if fuse and chance == "enum":
        ref = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=cm_g, fuse_subtrees=fuse)
        assert tr.n_launches(0) == ref.n_launches(0) and tr.n_launches(1) == ref.n_launches(1)
        del ref
    osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, prune=True, chance_mode=cm_o)
    for it in range(3):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=4)
            assert_bits(got, want, "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)


@pytest.mark.parametrize("dtype", ["f32", "f16", "f32+rmplus", "i32+rmplus"])
@pytest.mark.parametrize("fuse", [1, 0])
def test_iterate_extension_dtypes_vs_oracle(dtype, fuse):
    rmplus = "rmplus" in dtype
    dt_g, dt_o = {"f32": (rs.F32, orc.T_F32), "f16": (rs.F16, orc.T_F16), "i32": (rs.I32, orc.T_I32)}[dtype.split("+")[0]]
    tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [2], 100, 9, dt_g, dt_o, 10**5)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=1.0 if dt_g != rs.I32 else 100.0,
                         mode=rs.UPD_CLAMP_I64 | (rs.UPD_RMPLUS if rmplus else 0), chance_mode=rs.CHANCE_PASS, fuse_subtrees=fuse)
    osol = orc.OracleSolver(otree, otab, lo, scale=1.0 if dt_g != rs.I32 else 100.0, mode=orc.UPD_CLAMP_I64, rmplus=rmplus,
                            chance_mode=orc.CHANCE_PASS)
    for it in range(3):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util")
    compare_tables(tree, table, otab)


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("variant", ["river", "river+prune", "river-f32", "three-street"])
def test_iterate_sampled_opponent_vs_oracle(fuse, graph, variant):
    """mccfr(): opponent nodes draw ONE action from sigma with rand's WeightedIndex (cfr.rs:467-476); the
    traverser update is cfr.rs:413-464 (clamp, scale 100).  Random bits = shared counter hash."""
    prune = "prune" in variant
    if variant == "three-street":
        tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), [3, 3, 3], 11, 71)
    elif variant == "river-f32":
        tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [2], 150, 72, rs.F32, orc.T_F32)
    else:
        tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [3], 333, 73)
    scale = 1.0 if variant == "river-f32" else 100.0
    tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=rs.UPD_CLAMP_I64 | (rs.UPD_PRUNE if prune else 0),
                         chance_mode=rs.CHANCE_PASS, use_graph=graph, fuse_subtrees=fuse, opp_mode=rs.OPP_SAMPLE, sample_seed=20261003)
    osol = orc.OracleSolver(otree, otab, lo, scale=scale, mode=orc.UPD_CLAMP_I64, prune=prune, chance_mode=orc.CHANCE_PASS,
                            opp_mode=orc.OPP_SAMPLE, base_seed=20261003)
    for it in range(3):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=4)
            assert_bits(got, want, "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)


def test_sampled_mode_needs_pass_through_chance():
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), [1, 2, 2], 4, 5)
    with pytest.raises(rs.RsError) as e:
        rs.MCCFRTrainer(tree, table, lg, chance_mode=rs.CHANCE_ENUM, opp_mode=rs.OPP_SAMPLE)
    assert e.value.code == L.ERR_UNSUPPORTED


# ---- deal batches: get-infoset addressing through cluster ids, batch-synchronous atomics -----------------

def setup_deals(options_rs, options_orc, sizes, n_deals, seed):
    """sizes[round_idx] = (clusters of player 0, clusters of player 1), different on purpose"""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_actions, tree = rs.build_game_tree(options_rs)
    table = rs.create_infosets(n_actions, tree, sizes, [1, 1, 1])
    otree = orc.OracleTree(options_orc)
    otab = orc.OracleDealTable(otree, sizes)
    for nd in tree.action_nodes():
        a, n = nd.n_children, sizes[nd.round_idx][nd.player]
        R = rng.integers(-10**6, 10**6, size=(a, n)).astype(np.int32)
        S = rng.integers(0, 10**6, size=(a, n)).astype(np.int32)
        if a > 0:
            R[0, ::5] = -10_000_001
            R[a - 1, ::7] = 2_147_000_000
        table.upload_node(nd.index, R, S)
        otab.set_node(nd.index, R, S)
    cidx = {(r, p): rng.integers(0, sizes[r][p], size=n_deals).astype(np.uint32) for r in range(len(sizes)) for p in (0, 1)}
    sign = rng.integers(-1, 2, size=n_deals).astype(np.float32)
    sbuf = rs.deal_buffer(table, n_deals, sign)
    lg = {i: (rs.LEAF_SIGN, sbuf) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    lo = {i: (orc.LEAF_SIGN, sign) for i in lg}
    return tree, table, otree, otab, lg, lo, cidx


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("variant", ["river-clamp", "river-wrap", "river-clamp+prune", "river-sampled", "three-street-sampled",
                                     "three-street-full", "river-sampled+prune-per-deal", "three-street-sampled+prune-per-deal"])
def test_deal_batches_vs_oracle(fuse, variant):
    """Many deals per info set (1000 deals on 13 / 17 clusters): collisions are the rule, results must still be exact.
    prune-per-deal: in train() `prune` is decided per deal (cfr.rs:213-221); a third of the deals carry the flag."""
    three = variant.startswith("three")
    sampled, prune, wrap = "sampled" in variant, "prune" in variant, "wrap" in variant
    n_deals = 1000 if not three else 300
    flags = None
    if "per-deal" in variant:
        flags = (np.random.Generator(np.random.PCG64(5)).integers(0, 3, n_deals) == 0).astype(np.uint8)
    if three:
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(),
                                                             [(7, 9), (11, 8), (13, 17)], n_deals, 31)
    else:
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [(13, 17)], n_deals, 32)
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if wrap else (100.0, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mg | (rs.UPD_PRUNE if prune else 0), fuse_subtrees=fuse, deals=cidx,
                         opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=777, use_graph=(variant == "river-clamp"), prune_deal=flags)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=scale, mode=mo, prune=prune, prune_deal=flags,
                                opp_mode=orc.OPP_SAMPLE if sampled else orc.OPP_FULL, base_seed=777)
    for it in range(3):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player)
            assert_bits(got, want, "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("sizes,lds_max", [([(5000, 4100)], None), ([(5000, 4100)], "65536"), ([(13000, 13000)], None)])
def test_deal_batches_large_cluster_counts(sizes, lds_max, monkeypatch):
    """5 000 clusters need a 121 KB LDS tile per traverser node (MI355X gives a workgroup 160 KB); capped at 64 KB, or with 13 000 clusters
    (312 KB), the kernels fall back to direct global atomics.  Same bits either way."""
    if lds_max:
        monkeypatch.setenv("RS_JIT_LDS_MAX", lds_max)
    n_deals = 6000
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), sizes, n_deals, 77)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=9)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=9)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("blocks", ["1", "2"])
def test_deal_batches_many_trips_per_workgroup(blocks, monkeypatch):
    """resident LDS tiles hold the sums of ALL trips of a workgroup and are flushed once at the end: force 1-2 workgroups over 7 000
    deals (4+ trips of 2 048 deals, the last one partial) and compare with the oracle"""
    monkeypatch.setenv("RS_JIT_MAX_BLOCKS", blocks)
    n_deals = 7000
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [(1081, 700)], n_deals, 78)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=3)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=3)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("blocks", [None, "2"])
@pytest.mark.parametrize("sparse", [True, "ordered", "scan-parent", "scan-parent-siblings", "siblings", "rows", "gathers", "rows+ordered"])
def test_sparse_subtree_sweeps_three_streets_many_deals(blocks, sparse, monkeypatch):
    """sampled three-street sweeps over 30 000 deals: every river subtree walks only the compacted list of its live deals (several trips per
    workgroup when the grid is capped); with the lists switched off (RS_JIT_NO_SPARSE) every lane is walked and masked; with RS_JIT_NO_ROUNDS the flop and
    turn rounds run as level kernels again.  The knobs are read when a solver is created.  Same bits, and equal to the oracle."""
    if blocks and sparse not in (True, "ordered", "rows", "rows+ordered"):
        pytest.skip("several trips per workgroup: run on the list walkers with tiles, rows and runs")
    if blocks:
        monkeypatch.setenv("RS_JIT_MAX_BLOCKS", blocks)
    if sparse not in ("rows", "rows+ordered", "gathers"):   # the list walkers with LDS tiles and work lists: what the engine itself only picks below 1 024 deals per batch
        monkeypatch.setenv("RS_JIT_ROWS", "0")
    if sparse not in ("ordered", "rows+ordered", "gathers"):
        monkeypatch.setenv("RS_JIT_ORDERED", "0")
    if sparse in (True, "ordered", "siblings"):   # the compaction scans the whole batch per root: the engine's choice up to 2 K deals per batch
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "1")
    if sparse in ("scan-parent", "scan-parent-siblings"):   # the compaction of a root's live deals walks its parent's lists (what batches beyond 2 K deals get) instead of the whole batch
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
        if "siblings" in sparse:   # what batches beyond 512 K deals get: one compaction job per parent (k_compact_siblings) instead of one per root (k_compact_live)
            monkeypatch.setenv("RS_JIT_NO_SIBLINGS", "0")
    elif sparse == "siblings":    # ... the whole batch scanned once per 16 roots
        monkeypatch.setenv("RS_JIT_NO_SIBLINGS", "0")
    elif sparse == "rows":        # what batches beyond 512 K deals get: the list walkers store delta rows, summed per round (rs_kernel_forms.delta_rows), sibling compaction
        monkeypatch.setenv("RS_JIT_ROWS", "1")
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
        monkeypatch.setenv("RS_JIT_NO_SIBLINGS", "0")
    elif sparse == "ordered":     # the batch walked in the order of the traverser's river cluster, river deltas summed along the runs (rs_kernel_forms.deal_order), tiles on flop / turn
        monkeypatch.setenv("RS_JIT_ORDERED", "1")
    elif sparse == "rows+ordered":   # what batches beyond 48 K deals get: delta rows on flop / turn lists, runs on the river, staged rows, parent-list compaction
        monkeypatch.setenv("RS_JIT_ROWS", "1")
        monkeypatch.setenv("RS_JIT_ORDERED", "1")
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
    elif sparse == "gathers":     # the same with the list walkers gathering a record per node instead of staging their deals' rows in LDS (the fallback for rows beyond 256 bytes)
        monkeypatch.setenv("RS_JIT_ROWS", "1")
        monkeypatch.setenv("RS_JIT_ORDERED", "1")
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
        monkeypatch.setenv("RS_JIT_NO_STAGE", "1")
    n_deals = 30000
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(7, 9), (11, 8), (13, 17)], n_deals, 91)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=12)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=12)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("seed", range(6))
def test_staged_rows_random_trees(seed, monkeypatch):
    """The forms big batches get (delta rows, ordered sweeps with runs, staged shadow rows) on RANDOM two- and three-round trees: node records of 8 / 16 / 32 / 64 bytes in one
    row, rows that go through LDS in one part and in two (rs_device.hpp stage_rows / stage_rows_two, cut where the emitter finds a record boundary), rows beyond 16 chunks (no
    staging: gathers), pruned deals.  Same bits as the oracle."""
    monkeypatch.setenv("RS_JIT_ROWS", "1")
    monkeypatch.setenv("RS_JIT_ORDERED", "1")
    monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    nb = int(rng.choice([4, 3]))
    rounds = 6 - nb
    pool = [0.33, 0.5, 0.75, 1.0, 2.0]
    bets = [sorted(rng.choice(pool, size=int(rng.integers(1, 4 if r else 3)), replace=False).tolist()) for r in range(rounds)]
    raises = [sorted(rng.choice([2.0, 3.0], size=int(rng.integers(1, 3)), replace=False).tolist()) for _ in range(rounds)]
    stacks, pot = (int(rng.integers(150, 900)), int(rng.integers(150, 900))), int(rng.integers(10, 80))
    if seed == 0:     # five-action nodes on the second round: 8-int and 16-int records beside the 4-int ones
        nb, rounds, bets, raises, stacks, pot = 4, 2, [[1.0], [0.33, 0.5, 1.0, 2.0]], [[3.0], [2.0, 3.0]], (400, 400), 40
    elif seed == 1:   # six-action nodes (32 / 64-byte records)
        nb, rounds, bets, raises, stacks, pot = 4, 2, [[0.5], [0.25, 0.5, 1.0, 2.0]], [[3.0], [2.0, 2.5, 3.0, 4.0]], (2000, 2000), 20
    og, oo = rs.Options(stacks, pot, nb, bets, raises), orc.make_options(stacks, pot, nb, bets, raises)
    _, probe = rs.build_game_tree(og)
    if any(nd.n_children == 0 for nd in probe.action_nodes()):
        pytest.skip("a node without valid actions makes mccfr panic (WeightedIndex::new(&[]).unwrap())")
    sizes = [(int(rng.integers(5, 60)), int(rng.integers(5, 60))) for _ in range(rounds)]
    n_deals = int(rng.integers(3000, 9000))
    prune = bool(seed % 2)
    flags = (rng.integers(0, 3, n_deals) == 0).astype(np.uint8) if prune else None
    tree, table, otree, otab, lg, lo, cidx = setup_deals(og, oo, sizes, n_deals, 7100 + seed)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64 | (rs.UPD_PRUNE if prune else 0), fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=seed,
                         prune_deal=flags)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, prune=prune, prune_deal=flags, opp_mode=orc.OPP_SAMPLE, base_seed=seed)
    tag = dict(nb=nb, bets=bets, raises=raises, stacks=stacks, pot=pot, sizes=sizes, n_deals=n_deals, prune=prune)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d %r" % (it, player, tag))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d: %r" % (nd.index, tag)


@pytest.mark.parametrize("rows", ["list-position", "whole-batch-scan", "many-ranges"])
def test_sparse_three_streets_cluster_ranges_on_every_round(rows, monkeypatch):
    """What batches beyond 256 K deals get, forced onto a small one: LDS capped so that EVERY round subtree (the first included) is cut into cluster ranges,
    the compaction of a root's live deals scans its parent's per-range lists, and -- "list-position" -- the parent's reach-down kernel writes the reach rows
    at its list position, the compaction stores every live deal's reach beside its list entry.  "deal-id" keeps rows indexed by deal (RS_JIT_NO_POSROWS),
    "whole-batch-scan" the small-batch compaction.  Same bits, equal to the oracle."""
    # "many-ranges": half of the LDS again (64-cluster ranges on the river), so that the launches of the river round carry several hundred (subtree, range) jobs each -- more than one
    # chunk of k_worklist's prefix scan -- most of them with a handful of deals or none
    monkeypatch.setenv("RS_JIT_LDS_MAX", "8256" if rows == "many-ranges" else "16384")
    monkeypatch.setenv("RS_JIT_ROWS", "0")      # cluster ranges belong to the list walkers with LDS tiles
    monkeypatch.setenv("RS_JIT_ORDERED", "0")
    monkeypatch.setenv("RS_JIT_SCAN_ALL", "1" if rows == "whole-batch-scan" else "0")
    n_deals = 30011
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(300, 280), (500, 450), (700, 650)], n_deals, 92)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=13)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, prune=True, opp_mode=orc.OPP_SAMPLE, base_seed=13)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


def test_deal_batches_reject_bad_inputs():
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [(5, 6)], 10, 3)
    with pytest.raises(rs.RsError):
        rs.MCCFRTrainer(tree, table, lg, deals={(0, 0): cidx[(0, 0)]})          # player 1 ids missing
    n, t2 = rs.build_game_tree(rs.default_flop())
    base = dict(scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=1)
    for dt in (rs.F32, rs.F16):   # float deal tables (binary16 since round 5): through the generated kernels, no pruning (cfr.rs:352 compares i32 regrets)
        tf = rs.create_infosets(n, t2, [(5, 6)], [1], dt)
        for kw in (dict(mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE), dict(fuse_subtrees=0)):
            with pytest.raises(rs.RsError) as e:
                rs.MCCFRTrainer(t2, tf, lg, deals=cidx, **dict(base, **kw))
            assert e.value.code == L.ERR_UNSUPPORTED


@pytest.mark.parametrize("fuse", [1, 0])
def test_wide_nodes_through_both_plans(fuse):
    """four bet sizes and two raise sizes: nodes with up to 6 actions (RS_MAX_ACTIONS = 8), tree kernels with A > 3"""
    opts_g = rs.Options((800, 800), 40, 5, [[0.25, 0.5, 1.0, 2.0]], [[2.0, 3.0]])
    opts_o = orc.make_options((800, 800), 40, 5, [[0.25, 0.5, 1.0, 2.0]], [[2.0, 3.0]])
    tree, table, otree, otab, lg, lo = setup_pair(opts_g, opts_o, [2], 37, 88)
    assert max(nd.n_children for nd in tree.action_nodes()) >= 5
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=fuse)
    osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player, threads=4), "root util")
    compare_tables(tree, table, otab)


@pytest.mark.parametrize("sampled", [True, False])
@pytest.mark.parametrize("fuse", [1, 0])
def test_wide_nodes_in_deal_batches(fuse, sampled):
    """nodes with 5-6 actions in deal mode: the AoS shadow's wide records (8 + 8 ints), LDS tiles of 2 * 6 rows, the sampled hash"""
    opts_g = rs.Options((800, 800), 40, 5, [[0.25, 0.5, 1.0, 2.0]], [[2.0, 3.0]])
    opts_o = orc.make_options((800, 800), 40, 5, [[0.25, 0.5, 1.0, 2.0]], [[2.0, 3.0]])
    n_deals = 5000
    tree, table, otree, otab, lg, lo, cidx = setup_deals(opts_g, opts_o, [(97, 61)], n_deals, 55)
    assert max(nd.n_children for nd in tree.action_nodes()) >= 5
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=fuse, deals=cidx,
                         opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=8)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64,
                                opp_mode=orc.OPP_SAMPLE if sampled else orc.OPP_FULL, base_seed=8)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("seed", range(32))
def test_randomised_differential(seed, monkeypatch):
    """random game options x engine modes, GPU vs oracle, bit for bit.  The form of the subtrees below ENUM chance nodes (rs_kernel_forms.lane_fan: the expand step inside
    the subtree kernel, or expand / reduce launches of their own; conftest's fan_loop fixture for the other lane tests) alternates with the seed."""
    lane_fan = [rs.FAN_EXPAND, rs.FAN_NONE][seed % 2]
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    nb = int(rng.choice([5, 5, 4, 3]))
    rounds = 6 - nb
    bets = [sorted(rng.choice([0.25, 0.33, 0.5, 0.75, 1.0, 1.5, 2.0], size=int(rng.integers(1, 4)), replace=False).tolist()) for _ in range(rounds)]
    raises = [sorted(rng.choice([2.0, 2.5, 3.0, 4.0], size=int(rng.integers(1, 3)), replace=False).tolist()) for _ in range(rounds)]
    stacks = (int(rng.integers(50, 1500)), int(rng.integers(50, 1500)))
    pot = int(rng.integers(4, 200))
    og, oo = rs.Options(stacks, pot, nb, bets, raises), orc.make_options(stacks, pot, nb, bets, raises)
    deals = bool(rng.integers(0, 3) == 0)
    sampled = bool(rng.integers(0, 2)) if (deals or rounds == 1 or rng.integers(0, 2)) else False
    fuse, graph = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
    prune = bool(rng.integers(0, 4) == 0)
    wrap = bool(rng.integers(0, 2)) and not prune
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if wrap else (100.0, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    mgf = mg | (rs.UPD_PRUNE if prune else 0)
    tag = dict(nb=nb, bets=bets, raises=raises, stacks=stacks, pot=pot, deals=deals, sampled=sampled, fuse=fuse, graph=graph, prune=prune, wrap=wrap)
    _, probe = rs.build_game_tree(og)
    if any(nd.n_children == 0 for nd in probe.action_nodes()):
        sampled = False    # a node without valid actions makes mccfr panic (WeightedIndex::new(&[]).unwrap()); cfr() is fine
    if deals:
        sizes = [(int(rng.integers(3, 40)), int(rng.integers(3, 40))) for _ in range(rounds)]
        n_deals = int(rng.integers(70, 600))
        tree, table, otree, otab, lg, lo, cidx = setup_deals(og, oo, sizes, n_deals, 5000 + seed)
        tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mgf, fuse_subtrees=fuse, deals=cidx, use_graph=graph, forms={"lane_fan": lane_fan},
                             opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=seed)
        osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=scale, mode=mo, prune=prune,
                                    opp_mode=orc.OPP_SAMPLE if sampled else orc.OPP_FULL, base_seed=seed)
    else:
        C = int(rng.integers(3, 70))
        if sampled or rng.integers(0, 2):
            b = int(rng.integers(1, 4))
            boards, cm_g, cm_o = [b] * rounds, rs.CHANCE_PASS, orc.CHANCE_PASS
        else:
            boards = [1]
            for _ in range(rounds - 1):
                boards.append(boards[-1] * int(rng.integers(1, 4)))
            cm_g, cm_o = rs.CHANCE_ENUM, orc.CHANCE_ENUM
        tree, table, otree, otab, lg, lo = setup_pair(og, oo, boards, C, 6000 + seed)
        tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mgf, chance_mode=cm_g, fuse_subtrees=fuse, use_graph=graph, forms={"lane_fan": lane_fan},
                             opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=seed)
        osol = orc.OracleSolver(otree, otab, lo, scale=scale, mode=mo, prune=prune, chance_mode=cm_o,
                                opp_mode=orc.OPP_SAMPLE if sampled else orc.OPP_FULL, base_seed=seed)
    for it in range(2):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player)
            assert_bits(got, want, "root util %r" % (tag,))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d: %r" % (nd.index, tag)


def test_action_node_without_valid_actions():
    """state.rs:125-157 can return no action: a short all-in raise that stays below the bet.  cfr(): value 0, no update;
    mccfr(): WeightedIndex::new(&[]).unwrap() panics -> rs_solver_create refuses the sampled mode."""
    og, oo = rs.Options((60, 1000), 100, 5, [[2.0]], [[3.0]]), orc.make_options((60, 1000), 100, 5, [[2.0]], [[3.0]])
    _, probe = rs.build_game_tree(og)
    assert any(nd.n_children == 0 for nd in probe.action_nodes())
    for fuse in (1, 0):
        tree, table, otree, otab, lg, lo = setup_pair(og, oo, [2], 9, 4)
        tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=fuse)
        osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util")
        compare_tables(tree, table, otab)
        with pytest.raises(rs.RsError):
            rs.MCCFRTrainer(tree, table, lg, chance_mode=rs.CHANCE_PASS, opp_mode=rs.OPP_SAMPLE)


def test_train_with_discount_schedule_vs_oracle():
    # cfr.rs:188-265 with a short interval so that several discount ticks happen
    tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [1], 64, 3)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, use_graph=True)
    osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    tr.train(11, discount_interval=3, discount_cap=9)
    osol.train(11, discount_interval=3, discount_cap=9)
    compare_tables(tree, table, otab)
    fs = table.final_strategy_all()
    for nd in tree.action_nodes():
        _, so = otab.get_node(nd.index)
        for k in range(0, so.shape[1], 7):
            assert_bits(fs[nd.index][:, k], orc.get_final_strategy(so[:, k]), "final strategy")


def test_leaf_util_buffers_per_traverser():
    """RS_LEAF_UTIL: utilities given verbatim, a different buffer per traverser"""
    rng = np.random.Generator(np.random.PCG64(77))
    C, B = 33, 2
    tree, table, otree, otab, lg, lo = setup_pair(rs.default_flop(), orc.options_default_river(), [B], C, 12)
    root = tree.nodes[tree.nodes[0].children[0]]
    lg0, lg1, lo0, lo1 = {}, {}, {}, {}
    for i in lg:
        u0 = rng.uniform(-500, 500, size=B * C).astype(np.float32)
        u1 = rng.uniform(-500, 500, size=B * C).astype(np.float32)
        lg0[i] = (rs.LEAF_UTIL, table.lane_buffer(root.index, 1, u0))
        lg1[i] = (rs.LEAF_UTIL, table.lane_buffer(root.index, 1, u1))
        lo0[i], lo1[i] = (orc.LEAF_UTIL, u0), (orc.LEAF_UTIL, u1)
    tr = rs.MCCFRTrainer(tree, table, lg0, leaves_p1=lg1, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS)
    os0 = orc.OracleSolver(otree, otab, lo0, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    os1 = orc.OracleSolver(otree, otab, lo1, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    for it in range(2):
        assert_bits(tr.iterate(0, True), os0.iterate(0), "p0")
        assert_bits(tr.iterate(1, True), os1.iterate(1), "p1")
    compare_tables(tree, table, otab)


def test_solver_rejects_mismatched_inputs():
    n, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n, tree, [(8, 9)], [1])          # different cluster counts per player
    sign = table.lane_buffer(0, 1)
    leaves = {i: (rs.LEAF_SIGN, sign) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    with pytest.raises(rs.RsError) as e:
        rs.MCCFRTrainer(tree, table, leaves)
    assert e.value.code == L.ERR_UNSUPPORTED
    table2 = rs.create_infosets(n, tree, [8], [1])
    with pytest.raises(rs.RsError) as e:
        rs.MCCFRTrainer(tree, table2, {})                        # showdown terminals without a leaf buffer
    assert e.value.code == L.ERR_INVALID


# ---- full BASELINE size: size-independent property (lanes are independent => any sampled board must match) ----------

@pytest.mark.parametrize("fuse", [1, 0])
def test_full_size_config2_sampled_boards_match_oracle(fuse):
    B, C, seed = 9216, 1000, 1235
    n, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n, tree, [C], [B])
    table.fill_random(seed, (-10**6, 10**6), (0, 10**6))
    root = tree.nodes[tree.nodes[0].children[0]]
    sign = table.lane_buffer(root.index, 1)
    L.check(L.load().rs_fill_uniform_f32(table._h, sign.ptr, table.pitch(root.index), seed + 17, -1.0, 1.0))
    leaves = {i: (rs.LEAF_SIGN, sign) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    tr = rs.MCCFRTrainer(tree, table, leaves, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=fuse)
    sample = [0, 4607, 9215]
    sign_host = rs.synth.uniform_f32(seed + 17, table.pitch(root.index), -1.0, 1.0)
    otree = orc.OracleTree(orc.options_default_river())
    otab = orc.OracleTable(otree, [len(sample)], C)
    for nd in tree.action_nodes():
        R = rs.synth.table_node_values(table, nd.index, seed, -10**6, 10**6)
        S = rs.synth.table_node_values(table, nd.index, seed, 0, 10**6, ssum=True)
        cols = np.concatenate([np.arange(b * C, (b + 1) * C) for b in sample])
        otab.set_node(nd.index, R[:, cols].astype(np.int32), S[:, cols].astype(np.int32))
    sgn = np.concatenate([sign_host[b * C:(b + 1) * C] for b in sample])
    lo = {d["id"]: (orc.LEAF_SIGN, sgn) for d in otree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    osol = orc.OracleSolver(otree, otab, lo, scale=100.0, mode=orc.UPD_CLAMP_I64, chance_mode=orc.CHANCE_PASS)
    for it in range(2):
        for player in (0, 1):
            got = tr.iterate(player, want_root_util=True)
            want = osol.iterate(player, threads=4)
            assert_bits(np.concatenate([got[b * C:(b + 1) * C] for b in sample]), want, "root util")
    for nd in tree.action_nodes():
        ro, so = otab.get_node(nd.index)
        for j, b in enumerate(sample):
            r, s = table.download(nd.index, b)
            assert (r == ro[:, j * C:(j + 1) * C]).all() and (s == so[:, j * C:(j + 1) * C]).all()
    # discount at full size: sampled boards again
    d = rs.discount_factor(300001)
    table.discount(d)
    otab.discount(d)
    for nd in tree.action_nodes():
        ro, so = otab.get_node(nd.index)
        r, s = table.download(nd.index, sample[1])
        assert (r == ro[:, C:2 * C]).all() and (s == so[:, C:2 * C]).all()


def _random_deals(rng, n):
    cards = np.zeros((9, n), dtype=np.uint8)
    for k in range(n):
        cards[:, k] = rng.permutation(52)[:9]
    return cards


def test_showdown_sign_vs_bruteforce_oracle():
    """rs_showdown_sign (bit-trick 7-card evaluator) vs the oracle's best-of-21 brute force, cfr.rs:324-333"""
    rng = np.random.Generator(np.random.PCG64(123))
    n = 20000
    cards = _random_deals(rng, n)
    # craft ties and near-ties: board plays (both players' holes irrelevant), shared straights / flushes
    R, S = "23456789TJQKA", "cdhs"
    c = lambda t: [4 * R.index(x[0]) + S.index(x[1]) for x in t.split()]
    crafted = [c("As Ks Qs Js Ts 2c 3d 4h 5c"), c("2c 7d 9h Js Kc Ac Ad Ah As"), c("5c 6d 7h 8s 9c Ad Kh Ac Kd"),
               c("2h 5h 9h Jh Kh Ah 3c Qh 4d"), c("9c 9d Kh Ks 2c Qc 3h Qd 4h"), c("Ac 2d 3h 4s 9c 5d Kh 5c Qh")]
    for i, h in enumerate(crafted):
        cards[:, i] = h
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n_actions, tree, [4], [1])
    buf, got = rs.showdown_sign(table, cards)
    want = orc.showdown_sign(cards)
    assert (got == want).all(), np.nonzero(got != want)[0][:10]
    assert set(np.unique(got).tolist()) == {-1.0, 0.0, 1.0} and got[0] == 0.0 and got[2] == 0.0


def test_showdown_golden_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "known_answers_n2_n3.json")))
    cards = np.array([d["board"] + d["p0"] + d["p1"] for d in fx["showdown"]], dtype=np.uint8).T
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n_actions, tree, [4], [1])
    _, got = rs.showdown_sign(table, cards)
    assert got.tolist() == [float(d["sign"]) for d in fx["showdown"]]


def test_cards_to_iteration_pipeline():
    """cards -> device showdown signs -> deal-batch sweep, against the oracle fed with its own brute-force signs"""
    rng = np.random.Generator(np.random.PCG64(9))
    n_deals = 500
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [(20, 20)], n_deals, 44)
    cards = _random_deals(rng, n_deals)
    sbuf, _ = rs.showdown_sign(table, cards)
    osign = orc.showdown_sign(cards)
    lg = {i: (rs.LEAF_SIGN, sbuf) for i in lg}
    lo = {i: (orc.LEAF_SIGN, osign) for i in lo}
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=5)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=5)
    for player in (0, 1):
        assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util")
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all()


def test_checkpoint_roundtrip(tmp_path, table_layout, monkeypatch):
    """RSTB files hold unpadded [A][lanes] rows whatever the in-memory block looks like: a tiled table (lanes % tile != 0 on the turn) must come back bit for bit,
    also when the file is loaded under a DIFFERENT tiling than it was saved under."""
    C = 50                                                               # lanes 50 / 100 / 200: turn and river nodes are tiled in the tiled64 run, 100 % 64 != 0
    for dtype, odt in ((rs.I32, orc.T_I32), (rs.F16, orc.T_F16)):
        tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), [1, 2, 4], C, 5, dtype, odt)
        n_tiled = sum(table.tile_lanes(nd.index) != table.pitch(nd.index) for nd in tree.action_nodes() if nd.n_children)
        assert (n_tiled > 0) == (table_layout == "tiled64"), "the tiled64 run must really tile some nodes"
        want = {nd.index: table.download_node(nd.index) for nd in tree.action_nodes()}
        for nd in tree.action_nodes():                                   # ... and the bits are the oracle's, not merely self-consistent
            ro, so = otab.get_node(nd.index)
            assert want[nd.index][0].tobytes() == ro.tobytes() and want[nd.index][1].tobytes() == so.tobytes()
        path = str(tmp_path / ("t%d.rstb" % dtype))
        table.save(path)
        loads = [rs.InfosetTable.load(path)]
        with monkeypatch.context() as m:                                 # the other layout: saved tiled -> loaded plain and the reverse
            if table_layout == "tiled64":
                m.setenv("RS_TABLE_TILE_LANES", "0")
            else:
                m.setenv("RS_TABLE_TILE_LANES", "64")
            loads.append(rs.InfosetTable.load(path))
            other_tiled = sum(loads[1].tile_lanes(nd.index) != loads[1].pitch(nd.index) for nd in tree.action_nodes() if nd.n_children)
            assert (other_tiled > 0) == (table_layout != "tiled64")
        for t2 in loads:
            assert t2.n_nodes == table.n_nodes and t2.dtype == table.dtype
            for nd in tree.action_nodes():
                a, b = want[nd.index], t2.download_node(nd.index)
                assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
                assert bytes(t2.node_desc(nd.index)) == bytes(table.node_desc(nd.index))
        # the file itself: header 16 B + 16 B per node, then regrets[A][lanes] | strategy_sum[A][lanes] per node, unpadded
        raw = open(path, "rb").read()
        off = 16 + 16 * table.n_nodes
        es = 4 if dtype == rs.I32 else 2
        for nd in tree.action_nodes():
            lanes = table.lanes(nd.index)
            n = nd.n_children * lanes * es
            if dtype == rs.I32:
                assert raw[off:off + n] == want[nd.index][0].astype(np.int32).tobytes(), "file rows of node %d are not [A][lanes]" % nd.index
                assert raw[off + n:off + 2 * n] == want[nd.index][1].astype(np.int32).tobytes()
            else:
                assert raw[off:off + n] == want[nd.index][0].astype(np.float16).tobytes()
            off += 2 * n
        assert off + 8 == len(raw)
        raw = bytearray(raw)
        raw[len(raw) // 2] ^= 0x40                                   # corrupt one byte -> checksum mismatch
        open(path, "wb").write(bytes(raw))
        with pytest.raises(rs.RsError):
            rs.InfosetTable.load(path)
        open(path, "wb").write(bytes(raw[:100]))                    # truncated
        with pytest.raises(rs.RsError):
            rs.InfosetTable.load(path)


# ---- sharded multi-round sweep (BASELINE configs[3]): W ranks emulated on one GPU ------------------------------------------

@pytest.mark.parametrize("world,fuse,dtype", [(2, 1, "i32"), (3, 1, "i32"), (2, 0, "i32"), (3, 0, "i32"), (2, 1, "f32"), (3, 1, "f32"), (3, 0, "f32"), (2, 1, "f16"), (3, 1, "f16"),
                                              (2, 0, "f16")])
def test_sharded_enum_sweep_equals_single_gpu(world, fuse, dtype):
    """Turn and river boards sharded over `world` ranks, flop replicated; the ranks exchange the turn-root utility rows at the
    sharded chance nodes (here: copied by the test between the ranks' exchange buffers; in production one ncclAllGather).
    Every rank must end up with the SAME flop table as the unsharded sweep and with its slice of the turn / river tables -- bit for bit for float tables too
    (f32, and binary16 = the sharded half of BASELINE configs[4]): the all-gather design never splits an f32 sum over ranks."""
    from rustsolver_amd.dist import shard_boards
    Cn, G = 8, [1, 5, 10]                       # global boards per round; turn = sharded round (5 boards over 2 or 3 ranks)
    fan_river = G[2] // G[1]
    dt_g, dt_o = {"i32": (rs.I32, orc.T_I32), "f32": (rs.F32, orc.T_F32), "f16": (rs.F16, orc.T_F16)}[dtype]
    # float tables: utilities of an ENUM chance node are sums over its deals; scale 2^-6 keeps binary16 regrets (|x| <= 1000 at the start) far from its range limit
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if dtype == "i32" else (2.0 ** -6, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), G, Cn, 99, dt_g, dt_o)
    full = {nd.index: table.download_node(nd.index) for nd in tree.action_nodes()}
    # global leaf signs per round, re-read from the full run's buffers
    signs = {}
    for i, nd in enumerate(tree.nodes):
        if i in lg:
            r = tree.nodes[nd.parent].round_idx
            if r not in signs:
                signs[r] = table.read_lane_buffer(lg[i][1], tree.nodes[nd.parent].index)[0].copy()
    ref = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mg, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=fuse)
    osol = orc.OracleSolver(otree, otab, lo, scale=scale, mode=mo, chance_mode=orc.CHANCE_ENUM)

    ranks = []
    for g in range(world):
        tlo, thi = shard_boards(G[1], g, world)
        boards = [1, thi - tlo, (thi - tlo) * fan_river]
        n_actions, tr_tree = rs.build_game_tree(rs.three_street_options())
        tb = rs.create_infosets(n_actions, tr_tree, [Cn], boards, dt_g)
        cols = {0: slice(0, Cn), 1: slice(tlo * Cn, thi * Cn), 2: slice(tlo * fan_river * Cn, thi * fan_river * Cn)}
        for nd in tr_tree.action_nodes():
            R, S = full[nd.index]
            tb.upload_node(nd.index, R[:, cols[nd.round_idx]], S[:, cols[nd.round_idx]])
        leaves, bufs = {}, {}
        for i, nd in enumerate(tr_tree.nodes):
            if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED:
                parent = tr_tree.nodes[nd.parent]
                r = parent.round_idx
                if r not in bufs:
                    bufs[r] = tb.lane_buffer(parent.index, 1, signs[r][cols[r]])
                leaves[i] = (rs.LEAF_SIGN, bufs[r])
        sv = rs.MCCFRTrainer(tr_tree, tb, leaves, scale=scale, mode=mg, chance_mode=rs.CHANCE_ENUM, fuse_subtrees=fuse,
                             shard=(world, g, 1, G[1]))
        ranks.append((tr_tree, tb, sv, cols))

    lib = L.load()
    for it in range(2):
        for player in (0, 1):
            want_g = ref.iterate(player, want_root_util=True)
            want_o = osol.iterate(player, threads=4)
            assert_bits(want_g, want_o, "unsharded GPU vs oracle")
            for (_, tb, sv, _) in ranks:
                sv.iterate_phase(player, 0)
            # the all-gather, done by hand: rank g's slot g goes to everybody
            slots = []
            for g, (_, tb, sv, _) in enumerate(ranks):
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
                assert_bits(got, want_g, "root util of a sharded rank")
    for (tr_tree, tb, sv, cols) in ranks:
        for nd in tr_tree.action_nodes():
            r, s2 = tb.download_node(nd.index)
            R, S = table.download_node(nd.index)
            assert r.tobytes() == np.ascontiguousarray(R[:, cols[nd.round_idx]]).tobytes() and s2.tobytes() == np.ascontiguousarray(S[:, cols[nd.round_idx]]).tobytes(), \
                "node %d" % nd.index
    compare_tables(tree, table, otab)


def test_sharded_solver_validates_its_slice():
    n, tree = rs.build_game_tree(rs.three_street_options())
    tb = rs.create_infosets(n, tree, [4], [1, 2, 4])
    sign = {r: tb.lane_buffer([nd.index for nd in tree.action_nodes() if nd.round_idx == r][0], 1) for r in range(3)}
    leaves = {i: (rs.LEAF_SIGN, sign[tree.nodes[nd.parent].round_idx]) for i, nd in enumerate(tree.nodes)
              if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    with pytest.raises(rs.RsError):      # rank 0 of 2 over 5 global boards must hold 3 boards, not 2
        rs.MCCFRTrainer(tree, tb, leaves, chance_mode=rs.CHANCE_ENUM, shard=(2, 0, 1, 5))
    with pytest.raises(rs.RsError):      # sharding needs the enumerating chance nodes
        rs.MCCFRTrainer(tree, tb, leaves, chance_mode=rs.CHANCE_PASS, shard=(2, 1, 1, 5))
    sv = rs.MCCFRTrainer(tree, tb, leaves, chance_mode=rs.CHANCE_ENUM, shard=(2, 1, 1, 5))
    with pytest.raises(rs.RsError):      # no communicator attached: rs_iterate refuses, phases must be driven by hand
        sv.iterate(0)


# ---- multi-GPU primitive on one rank ------------------------------------------------------------------------------------

def test_allreduce_replicated_single_rank_is_identity():
    import ctypes as C
    tree, table, otree, otab, lg, lo = setup_pair(rs.three_street_options(), orc.options_three_street(), [1, 2, 2], 6, 8)
    lib = L.load()
    ident = (C.c_char * L.COMM_ID_BYTES)()
    L.check(lib.rs_comm_unique_id(ident))
    comm = C.c_void_p()
    L.check(lib.rs_comm_create(table._h, ident, 0, 1, C.byref(comm)))
    tr = rs.MCCFRTrainer(tree, table, lg, scale=10000.0, mode=rs.UPD_WRAP_I32, chance_mode=rs.CHANCE_ENUM)
    osol = orc.OracleSolver(otree, otab, lo, scale=10000.0, mode=orc.UPD_WRAP_I32, chance_mode=orc.CHANCE_ENUM)
    # in-place all-gather with one rank: the buffer must come back unchanged
    probe = rs.DeviceBuffer.from_numpy(table, np.arange(256, dtype=np.float32))
    L.check(lib.rs_comm_allgather(comm, table._h, probe.ptr, 1024))
    table.sync()
    assert (probe.download(np.float32, 256) == np.arange(256, dtype=np.float32)).all()
    L.check(lib.rs_replicated_begin(table._h, 0b001))
    tr.iterate(0), tr.iterate(1)
    L.check(lib.rs_allreduce_replicated(table._h, comm, 0b001))
    table.sync()
    osol.iterate(0), osol.iterate(1)
    compare_tables(tree, table, otab)
    lib.rs_comm_destroy(comm)


@pytest.mark.parametrize("variant", ["river", "river+graph", "river+prune-per-deal", "three-street", "three-street+prune-per-deal", "three-street-wrap"])
@pytest.mark.parametrize("sizes", ["few-clusters", "many-clusters"])
def test_ordered_deal_sweeps_vs_oracle(variant, sizes, monkeypatch):
    """rs_kernel_forms.deal_order (forced here through its test override): the sweep of traverser p walks the batch sorted by p's cluster on the last round, the per-deal
    inputs travel as 32-byte records by rank, the sampler hashes the ORIGINAL deal id, and the last round's subtrees sum their deltas over the wave's runs of equal cluster
    (seg_add) instead of LDS tiles.  "few-clusters": thousands of deals per cluster, every wave is one run; "many-clusters": a handful per cluster, most waves hold more
    runs than kSegMax and fall back to per-lane atomics, the rest mix both.  Root utilities come back by deal id.  Same bits as the oracle, which never sorts."""
    monkeypatch.setenv("RS_JIT_ORDERED", "1")
    if sizes == "many-clusters" and variant in ("river", "river+graph"):
        pytest.skip("the fall-back to per-lane atomics runs on four of the six variants")
    three, prune = variant.startswith("three"), "prune" in variant
    n_deals = 20011 if three else 30005
    last = (13, 17) if sizes == "few-clusters" else (3001, 2500)
    flags = (np.random.Generator(np.random.PCG64(6)).integers(0, 3, n_deals) == 0).astype(np.uint8) if prune else None
    if three:
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(7, 9), (211, 190), last], n_deals, 41)
    else:
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [last], n_deals, 42)
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if "wrap" in variant else (100.0, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mg | (rs.UPD_PRUNE if prune else 0), fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=99,
                         use_graph="graph" in variant, prune_deal=flags)
    assert tr.ordered
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=scale, mode=mo, prune=prune, prune_deal=flags, opp_mode=orc.OPP_SAMPLE, base_seed=99)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("variant", ["three-street", "three-street+graph+prune-per-deal", "three-street-wrap", "three-street-big-river", "three-street-big-river-tiles",
                                     "three-street-big-river+prune-per-deal", "three-street-scan-parent"])
def test_delta_rows_deal_sweeps_vs_oracle(variant, monkeypatch):
    """rs_kernel_forms.delta_rows (forced through its test override onto a small batch): no delta tiles, no atomics inside the list walkers -- a visit stores its two delta
    vectors at the deal's list position ([2A][batch pitch] rows per traverser node), and k_row_sums adds every row up per cluster after the walks (a few hundred positions per
    workgroup here, so that rows are cut into many chunks); the first round's dense walk keeps its LDS tiles.  "big-river": the river's
    20 000 clusters exceed the summing pass's tile: its rows are added straight into the table once the river's walks are done (k_row_apply, rs_kernel_forms.direct_rows: what the
    lossless abstractions of solve_three_street get) -- "-tiles": direct rows off, that round keeps its tiles while the turn stores rows.  Same bits as the oracle."""
    monkeypatch.setenv("RS_JIT_ROWS", "1")
    monkeypatch.setenv("RS_JIT_ROWS_CHUNK", "1000")
    if variant.endswith("-tiles"):
        monkeypatch.setenv("RS_JIT_DIRECT_ROWS", "0")
    three, prune, full = True, "prune" in variant, False
    if "scan-parent" in variant:
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
    n_deals = (3001 if full else 20011) if three else 30005
    flags = (np.random.Generator(np.random.PCG64(6)).integers(0, 3, n_deals) == 0).astype(np.uint8) if prune else None
    if three:
        last = (20000, 17000) if "big-river" in variant else (301, 250)
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(7, 9), (211, 190), last], n_deals, 41)
    else:
        tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.default_flop(), orc.options_default_river(), [(1013, 977)], n_deals, 42)
    scale, mg, mo = (10000.0, rs.UPD_WRAP_I32, orc.UPD_WRAP_I32) if "wrap" in variant else (100.0, rs.UPD_CLAMP_I64, orc.UPD_CLAMP_I64)
    og, oo = (rs.OPP_FULL, orc.OPP_FULL) if full else (rs.OPP_SAMPLE, orc.OPP_SAMPLE)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=scale, mode=mg | (rs.UPD_PRUNE if prune else 0), fuse_subtrees=1, deals=cidx, opp_mode=og, sample_seed=99,
                         use_graph="graph" in variant, prune_deal=flags)
    assert tr.delta_rows
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=scale, mode=mo, prune=prune, prune_deal=flags, opp_mode=oo, base_seed=99)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


@pytest.mark.parametrize("variant", ["steps", "steps+graph", "steps-off", "train-loop", "train-loop-off", "host-loop", "host-loop-destroyed", "host-loop-read-inside", "host-phases-refused", "two-solvers"])
def test_kept_shadow_records_stay_in_step_with_the_table(variant, monkeypatch):
    """A round whose delta rows go straight into the table (20 000 / 17 000 river clusters: rs_kernel_forms.direct_rows) keeps its shadow records between sweeps: k_row_apply adds
    every delta to the record as well as to the table row, rs_discount sweeps the records too, and any other write to the table has them rebuilt before the next sweep
    (rs_solver.cpp setup_table_shadow).  "steps": sweeps with a discount, an upload and a single-infoset write between them; "train-loop": rs_train, inside which the records ARE
    the working copy (the table's river rows are written back when the loop ends) -- discount ticks every iteration; "two-solvers": two solvers on one table taking turns;
    "-off": rs_kernel_forms.kept_records = RS_FORM_OFF (those nodes have no shadow then: the walks gather the table's own rows).  Same bits as the oracle after every step."""
    monkeypatch.setenv("RS_JIT_ROWS", "1")
    monkeypatch.setenv("RS_JIT_ROWS_CHUNK", "1000")
    n_deals = 9001
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(7, 9), (211, 190), (20000, 17000)], n_deals, 41)
    def solver(seed):
        return (rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=seed, use_graph="graph" in variant,
                                forms={"kept_records": rs.FORM_OFF} if variant.endswith("-off") else None),
                orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=seed))
    def same_tables(what):
        for nd in tree.action_nodes():
            r, s_ = table.download_node(nd.index)
            ro, so = otab.get_node(nd.index)
            assert (r == ro).all() and (s_ == so).all(), "table differs at node %d after %s" % (nd.index, what)
    tr, osol = solver(99)
    assert tr.delta_rows
    river = [nd for nd in tree.action_nodes() if nd.round_idx == 2][:3]
    if variant.startswith("train-loop"):
        its, interval = 18, 4   # 16 trips or more: the records become the working copy (kKeptPrimaryMinTrips)
        tr.train(its, discount_interval=interval, discount_cap=10**9)      # rs_train: cfr.rs:207-262 with t counted per loop trip
        t, threshold = 0, interval
        while t < its:
            for player in (0, 1):
                osol.iterate(player)
            t += 1
            if t > threshold:
                otab.discount(orc.discount_factor(t, interval))
                threshold = t + interval
        same_tables("rs_train")
        for player in (0, 1):   # and the records are still good for a plain sweep afterwards
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util after the loop p=%d" % player)
        same_tables("a sweep after rs_train")
        return
    if variant == "host-loop-destroyed":   # a solver destroyed INSIDE its training loop (an exception in the host's loop, say): the records' rows go back to the table first
        tr.training_loop(True)
        for it in range(3):
            for player in (0, 1):
                tr.iterate(player)
                osol.iterate(player)
        tr.destroy()
        same_tables("a solver destroyed while its kept records were the working copy")
        tr2, osol2 = solver(7)   # and the table is good for the next solver
        for player in (0, 1):
            assert_bits(tr2.iterate(player, want_root_util=True), osol2.iterate(player), "root util of the next solver p=%d" % player)
        same_tables("the next solver's sweep")
        return
    if variant == "host-phases-refused":   # rs_iterate_phase promises "phase 0 leaves the table untouched": a solver whose river rows go straight into the table cannot keep it
        with pytest.raises(rs.RsError, match="direct_rows"):
            tr.iterate_phase(0, 0)
        tr_off = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=99,
                                 forms={"direct_rows": rs.FORM_OFF})
        before = {nd.index: table.download_node(nd.index) for nd in tree.action_nodes()}
        tr_off.iterate_phase(0, 0)
        for nd in tree.action_nodes():
            r, s_ = table.download_node(nd.index)
            assert (r == before[nd.index][0]).all() and (s_ == before[nd.index][1]).all(), "phase 0 wrote node %d" % nd.index
        tr_off.iterate_phase(0, 1)
        osol.iterate(0)
        same_tables("phases driven by hand, direct rows off")
        return
    if variant == "host-loop-read-inside":   # a download in the middle of the loop: the records' rows are written back first (and the loop goes on with table and records both up to date)
        tr.training_loop(True)
        for it in range(4):
            for player in (0, 1):
                tr.iterate(player)
                osol.iterate(player)
            if it == 1:
                same_tables("a download inside rs_solver_training_loop")
                nd = river[0]
                r, s_ = table.download_node(nd.index)
                r[:, 77] -= 4242
                table[nd.index][77].set(r[:, 77], s_[:, 77])      # ... and a write: nothing trained so far is lost
                otab.set_node(nd.index, r, s_)
        tr.training_loop(False)
        same_tables("the loop after a read and a write inside it")
        return
    if variant == "host-loop":   # rs_solver_training_loop around a loop the HOST writes: sweeps and discounts only in between, the table read after it
        tr.training_loop(True)
        for it in range(4):
            for player in (0, 1):
                tr.iterate(player)
                osol.iterate(player)
            if it % 2:
                table.discount(0.5 + 0.1 * it)
                otab.discount(np.float32(0.5 + 0.1 * it))
        tr.training_loop(False)
        same_tables("a host-written loop inside rs_solver_training_loop")
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util after the loop p=%d" % player)
        same_tables("a sweep after the loop")
        return
    if variant == "two-solvers":
        tr2, osol2 = solver(7)
        for it in range(2):
            for (a, b) in ((tr, osol), (tr2, osol2)):
                for player in (0, 1):
                    assert_bits(a.iterate(player, want_root_util=True), b.iterate(player), "root util it=%d p=%d" % (it, player))
        same_tables("two solvers taking turns")
        return
    rng = np.random.Generator(np.random.PCG64(5))
    for step in range(4):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util step=%d p=%d" % (step, player))
        if step == 0:     # rs_discount: the records take the same sweep
            table.discount(0.75)
            otab.discount(np.float32(0.75))
        elif step == 1:   # rs_table_upload_node: the records are rebuilt before the next sweep
            for nd in river:
                r, s_ = table.download_node(nd.index)
                r = (r + rng.integers(-50000, 50000, size=r.shape)).astype(np.int32)
                table.upload_node(nd.index, r, s_)
                otab.set_node(nd.index, r, s_)
        elif step == 2:   # rs_set_infoset
            nd = river[0]
            r, s_ = table.download_node(nd.index)
            r[:, 123] += 777
            table[nd.index][123].set(r[:, 123], s_[:, 123])
            otab.set_node(nd.index, r, s_)
    same_tables("sweeps, a discount, uploads")


@pytest.mark.parametrize("forms", ["tiles", "rows"])
def test_handoff_with_more_opponent_nodes_than_rows(forms, monkeypatch):
    """The reach-down kernel of a round subtree hands its draws at the opponent's nodes to the walk of the same subtree -- at most TEN nodes (3 bits of one packed word each); the
    opponent nodes beyond that are drawn again by the walk.  A flop subtree with three bet sizes and two raise sizes has 17 opponent nodes above next-round roots: both kinds in one
    pair of kernels, with LDS tiles and with delta rows, pruned deals included.  Same bits as the oracle."""
    if forms == "rows":
        monkeypatch.setenv("RS_JIT_ROWS", "1")
        monkeypatch.setenv("RS_JIT_SCAN_ALL", "0")
    else:
        monkeypatch.setenv("RS_JIT_ROWS", "0")
    bets, raises = ((0.25, 0.5, 1.0), (1.0,), (1.0,)), ((2.0, 3.0), (3.0,), (3.0,))
    n_deals = 6007
    flags = (np.random.Generator(np.random.PCG64(8)).integers(0, 3, n_deals) == 0).astype(np.uint8)
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.Options((200, 200), 40, 3, bets, raises), orc.make_options((200, 200), 40, 3, bets, raises), [(9, 7), (41, 37), (53, 59)],
                                                         n_deals, 77)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=5, prune_deal=flags)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, prune=True, prune_deal=flags, opp_mode=orc.OPP_SAMPLE, base_seed=5)
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index


def test_config1_plumbing_tree_through_rs_iterate():
    """BASELINE configs[0], "preflop-only 169-iso buckets, 2-action tree": the reference has no preflop round (state.rs:8, :59-64), so this is build-side plumbing --
    a tree adopted from plain node records (rs_tree_from_nodes) with ONE two-action node per player and 169 clusters (hand_indexer_s::init(1, [2]).size(0),
    gen_abstraction/ehs.rs:30), swept by rs_iterate and compared with the oracle's walk of the same records: player 0 {fold -> player 1 wins the pot, continue -> player 1's
    node}, player 1 {fold, call -> showdown}."""
    C_, pot = 169, 3
    nodes = [L.TreeNode() for _ in range(6)]

    def fill(i, kind, parent, children=(), **kw):
        nd = nodes[i]
        nd.kind, nd.parent, nd.n_children = kind, parent, len(children)
        for k, c in enumerate(children):
            nd.children[k] = c
        for key, v in kw.items():
            setattr(nd, key, v)
    fill(0, rs.NODE_PRIVATE_CHANCE, -1, (1,))
    fill(1, rs.NODE_ACTION, 0, (2, 3), index=0, player=0, round_idx=0)
    fill(2, rs.NODE_TERMINAL, 1, value=pot, ttype=rs.TERM_UNCONTESTED, last_to_act=0, round=0)
    fill(3, rs.NODE_ACTION, 1, (4, 5), index=1, player=1, round_idx=0)
    fill(4, rs.NODE_TERMINAL, 3, value=2 * pot, ttype=rs.TERM_UNCONTESTED, last_to_act=1, round=0)
    fill(5, rs.NODE_TERMINAL, 3, value=2 * pot, ttype=rs.TERM_SHOWDOWN, last_to_act=1, round=0)
    tree = rs.tree_from_nodes(nodes)
    assert tree.n_action_nodes == 2
    table = rs.create_infosets(2, tree, [C_], [1])
    otree = orc.OracleTree.from_nodes(nodes)
    otab = orc.OracleTable(otree, [1], C_)
    rng = np.random.Generator(np.random.PCG64(1234))   # regrets ~ U{-1000..1000}, reach = 1
    for idx in (0, 1):
        R = rng.integers(-1000, 1001, size=(2, C_)).astype(np.int32)
        S = rng.integers(0, 1001, size=(2, C_)).astype(np.int32)
        table.upload_node(idx, R, S)
        otab.set_node(idx, R, S)
    sign = rng.integers(-1, 2, size=C_).astype(np.float32)
    sbuf = table.lane_buffer(1, 1, sign)
    tr = rs.MCCFRTrainer(tree, table, {5: (rs.LEAF_SIGN, sbuf)}, scale=100.0, mode=rs.UPD_CLAMP_I64)
    osol = orc.OracleSolver(otree, otab, {5: (orc.LEAF_SIGN, sign)}, scale=100.0, mode=orc.UPD_CLAMP_I64)
    for it in range(3):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)
    for c in (0, 90, 168):   # get-infoset on the 169-bucket axis: `&self.infosets[an.index][cluster_idx]` and its two readers (infoset.rs:83-123)
        for idx in (0, 1):
            ro, so = otab.get_node(idx)
            assert_bits(table[idx][c].get_strategy(), orc.get_strategy(ro[:, c]), "get_strategy node %d cluster %d" % (idx, c))
            assert_bits(table[idx][c].get_final_strategy(), orc.get_final_strategy(so[:, c]), "get_final_strategy node %d cluster %d" % (idx, c))
    with pytest.raises(IndexError):
        table[0][169]


def test_c_examples_run_through_the_c_abi(tmp_path):
    """examples/*.c are plain C99 hosts of the C ABI: config1_main.c (BASELINE configs[0]: 169 buckets, a two-action tree adopted with rs_tree_from_nodes, three rs_iterate
    sweeps from a zero table) must print what the oracle computes for the same records; solver_main.c (the reference's `solver` binary, main.rs:29-36) must train and report
    an exploitability far below the uniform strategy's 77."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rustsolver_amd")
    exes = {}
    for name, extra in (("config1_main", []), ("solver_main", ["-D_POSIX_C_SOURCE=199309L"])):
        exes[name] = str(tmp_path / name)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror"] + extra + ["-I" + os.path.join(root, "include"),
                               os.path.join(root, "examples", name + ".c"), "-L" + libdir, "-lrustsolver_amd", "-Wl,-rpath," + libdir, "-o", exes[name]])
    r = subprocess.run([exes["config1_main"]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [int(x) for x in re.search(r"regrets (-?\d+) (-?\d+), strategy_sum (-?\d+) (-?\d+)", r.stdout).groups()]
    nodes = [L.TreeNode() for _ in range(6)]
    spec = [(rs.NODE_PRIVATE_CHANCE, -1, (1,), {}), (rs.NODE_ACTION, 0, (2, 3), dict(index=0, player=0)), (rs.NODE_TERMINAL, 1, (), dict(value=3, ttype=rs.TERM_UNCONTESTED, last_to_act=0)),
            (rs.NODE_ACTION, 1, (4, 5), dict(index=1, player=1)), (rs.NODE_TERMINAL, 3, (), dict(value=6, ttype=rs.TERM_UNCONTESTED, last_to_act=1)),
            (rs.NODE_TERMINAL, 3, (), dict(value=6, ttype=rs.TERM_SHOWDOWN, last_to_act=1))]
    for nd, (kind, parent, children, kw) in zip(nodes, spec):
        nd.kind, nd.parent, nd.n_children = kind, parent, len(children)
        for k, c in enumerate(children):
            nd.children[k] = c
        for key, v in kw.items():
            setattr(nd, key, v)
    otree = orc.OracleTree.from_nodes(nodes)
    otab = orc.OracleTable(otree, [1], 169)
    sign = np.array([0.0 if c % 3 == 0 else (-1.0 if c & 1 else 1.0) for c in range(169)], dtype=np.float32)
    osol = orc.OracleSolver(otree, otab, {5: (orc.LEAF_SIGN, sign)}, scale=100.0, mode=orc.UPD_CLAMP_I64)
    for it in range(3):
        for player in (0, 1):
            osol.iterate(player)
    ro, so = otab.get_node(0)
    assert got == [int(ro[0, 90]), int(ro[1, 90]), int(so[0, 90]), int(so[1, 90])], (got, r.stdout)
    r = subprocess.run([exes["solver_main"], "4000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "1081 / 1081 river clusters" in r.stdout
    assert float(re.search(r"exploitability (-?[\d.]+)", r.stdout).group(1)) < 20.0, r.stdout


@pytest.mark.parametrize("shadow", ["rule-mixed", "all"])
def test_deal_sweeps_with_and_without_table_shadows(shadow, monkeypatch):
    """sampled sweeps read a node through its AoS shadow (rebuilt per sweep) only where the batch is likely to read the record at all (rs_solver.cpp: n_deals * 8 >= clusters *
    actions * round subtrees); elsewhere the kernels gather the table's own rows (gather_node / gather_node2 with a null shadow).  3 000 deals against 30 / 300 / 140 clusters:
    every flop and turn node keeps its shadow, on the river the two-action nodes keep theirs (140 * 2 * 72 < 24 000) and the three-action nodes lose it -- both kinds inside
    ONE generated subtree.  "all" (rs_kernel_forms.shadow = RS_SHADOW_ALL) shadows everything.  An OPPONENT node's shadow record holds its strategy (matched once per sweep by
    k_build_shadow, sampled from as it comes); a node without a shadow has its regrets matched in the walk.  Same bits as the oracle every way, and the workspace figure shows the
    difference."""
    n_deals = 3000
    monkeypatch.setenv("RS_JIT_ROWS", "0")   # shadowed and unshadowed nodes side by side inside the tile kernels (with delta rows the walkers stage rows or gather all of a subtree)
    tree, table, otree, otab, lg, lo, cidx = setup_deals(rs.three_street_options(), orc.options_three_street(), [(30, 28), (300, 280), (140, 140)], n_deals, 93)
    tr = rs.MCCFRTrainer(tree, table, lg, scale=100.0, mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=14,
                         forms={"shadow": rs.SHADOW_ALL} if shadow == "all" else None)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=100.0, mode=orc.UPD_CLAMP_I64, prune=True, opp_mode=orc.OPP_SAMPLE, base_seed=14)
    test_deal_sweeps_with_and_without_table_shadows.workspace[shadow] = tr.workspace_bytes
    for it in range(2):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    for nd in tree.action_nodes():
        r, s = table.download_node(nd.index)
        ro, so = otab.get_node(nd.index)
        assert (r == ro).all() and (s == so).all(), "table differs at node %d" % nd.index
    ws = test_deal_sweeps_with_and_without_table_shadows.workspace
    if "all" in ws and "rule-mixed" in ws:   # rs_solver_workspace_bytes counts the shadow: shadowing every node costs the river's three-action records (16 + 32 bytes per cluster and traverser sweep)
        assert ws["all"] > ws["rule-mixed"]


test_deal_sweeps_with_and_without_table_shadows.workspace = {}


@pytest.mark.parametrize("variant", ["river-sampled", "river-full", "three-street-sampled", "three-street-full", "river-sampled-4-per-thread"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "f32+rmplus", "f16+rmplus"])
def test_f32_deal_batches_vs_oracle(variant, dtype, monkeypatch):
    """rs_solver_create_deals on an RS_F32 table (north_star: "f32 regrets"): deal sweeps whose sums are float.  f32 additions do not commute, so there are no atomics: every
    deal walks its subtrees densely, a traverser visit stores its two delta vectors per deal, and after the sweep every cell's deltas are added IN DEAL ORDER from 0.0 and
    then to the table -- exactly the oracle's sequential loop over the deals, hence bit-identical, with thousands of deals per info set.  Root utilities too.
    "f16" (round 5; BASELINE configs[4]: "fp16 regret/strategy tables with fp32 accumulators"): binary16 in memory, the same f32 deltas and f32 sums, ONE rounding per cell and
    sweep on the write-back; "+rmplus": a regret that does not end the sweep above 0 ends it at 0."""
    if dtype != "f32" and variant in ("river-full", "river-sampled-4-per-thread") and "rmplus" in dtype:
        pytest.skip("RM+ on two river variants and both three-street ones")
    three, sampled = variant.startswith("three"), "sampled" in variant
    half, rmplus = dtype.startswith("f16"), "rmplus" in dtype
    if variant.endswith("4-per-thread"):
        monkeypatch.setenv("RS_JIT_LANES", "4")
    n_deals = 2003 if three else 5001
    rng = np.random.Generator(np.random.PCG64(55))
    sizes = [(7, 9), (11, 8), (13, 17)] if three else [(13, 17)]
    n_actions, tree = rs.build_game_tree(rs.three_street_options() if three else rs.default_flop())
    table = rs.create_infosets(n_actions, tree, sizes, [1] * len(sizes), rs.F16 if half else rs.F32)
    otree = orc.OracleTree(orc.options_three_street() if three else orc.options_default_river())
    otab = orc.OracleDealTable(otree, sizes, orc.T_F16 if half else orc.T_F32)
    for nd in tree.action_nodes():
        a, n = nd.n_children, sizes[nd.round_idx][nd.player]
        R = rng.uniform(-1000, 1000, size=(a, n)).astype(np.float32)
        S = rng.uniform(0, 1000, size=(a, n)).astype(np.float32)
        if half:
            R, S = R.astype(np.float16).astype(np.float32), S.astype(np.float16).astype(np.float32)
        table.upload_node(nd.index, R, S)
        otab.set_node(nd.index, R, S)
    cidx = {(r, p): rng.integers(0, sizes[r][p], size=n_deals).astype(np.uint32) for r in range(len(sizes)) for p in (0, 1)}
    sign = rng.integers(-1, 2, size=n_deals).astype(np.float32)
    sbuf = rs.deal_buffer(table, n_deals, sign)
    lg = {i: (rs.LEAF_SIGN, sbuf) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
    lo = {i: (orc.LEAF_SIGN, sign) for i in lg}
    tr = rs.MCCFRTrainer(tree, table, lg, scale=0.25, mode=rs.UPD_CLAMP_I64 | (rs.UPD_RMPLUS if rmplus else 0), fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE if sampled else rs.OPP_FULL, sample_seed=5)
    osol = orc.OracleDealSolver(otree, otab, lo, cidx, n_deals, scale=0.25, mode=orc.UPD_CLAMP_I64, rmplus=rmplus, opp_mode=orc.OPP_SAMPLE if sampled else orc.OPP_FULL, base_seed=5)
    for it in range(3):
        for player in (0, 1):
            assert_bits(tr.iterate(player, want_root_util=True), osol.iterate(player), "root util it=%d p=%d" % (it, player))
    compare_tables(tree, table, otab)
    with pytest.raises(rs.RsError):   # pruning compares i32 regrets with the threshold (cfr.rs:352): not on float tables
        rs.MCCFRTrainer(tree, table, lg, scale=0.25, mode=rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, fuse_subtrees=1, deals=cidx, opp_mode=rs.OPP_SAMPLE, sample_seed=5)
