"""GPU (-m gpu): the readers of the average strategy through the C ABI -- rs_calc_br (MCCFRTrainer::calc_br as coded, cfr.rs:629-745) bit
for bit against the CPU oracle and the committed fixture, rs_best_response (the real thing) bit for bit (f64, fixed summation
order) against the oracle, and the domain properties on the device trainer: the average profile is zero-sum, a best response is worth at
least the average strategy, and the exploitability of the trained average strategy falls as training goes on."""
import json
import os

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import _lib as L
from rustsolver_amd import abstraction as ab
from test_best_response_cpu import BOARD, bits, river_game

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if rs.device_count() < 1:
        pytest.fail("no HIP device visible: GPU parity tests need a real MI355X (there is no CPU fallback)")


def fill_both(table, otab, tree, rng, sparse, dtype=L.I32):
    """the fill rule of tests/golden/make_calc_br_golden.py on the device table and the oracle table alike"""
    for nd in tree.action_nodes():
        a, n = otab.node_shape(nd.index)
        S = rng.integers(0, 1000, (a, n))
        S[rng.random((a, n)) < sparse] = 0
        R = rng.integers(-1000, 1000, (a, n))
        if dtype == L.I32:
            table.upload_node(nd.index, R.astype(np.int32), S.astype(np.int32))
        else:
            table.upload_node(nd.index, R.astype(np.float32), S.astype(np.float32))
        otab.set_node(nd.index, R, S)


def test_calc_br_fixture_through_the_abi(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "calc_br.json")))
    for case in fx["cases"]:
        river = case["tree"] == "river"
        n_actions, tree = rs.build_game_tree(rs.default_flop() if river else rs.three_street_options())
        rounds = 1 if river else 3
        table = rs.create_infosets(n_actions, tree, [case["n_clusters"]] * rounds, [1] * rounds)
        ot = orc.OracleTree(orc.options_default_river() if river else orc.options_three_street())
        otab = orc.OracleTable(ot, [1] * rounds, case["n_clusters"])
        fill_both(table, otab, tree, np.random.Generator(np.random.PCG64(case["seed"])), case["sparse"])
        got = table.calc_br(tree)
        assert bits(got) == case["br_bits"] == bits(otab.calc_br()), case


@pytest.mark.parametrize("dtype,odtype", [(L.I32, orc.T_I32), (L.F32, orc.T_F32), (L.F16, orc.T_F16)])
@pytest.mark.parametrize("river", [True, False])
def test_calc_br_equals_oracle(dtype, odtype, river):
    """every table element type, both trees, dense and sparse average strategies (zero-probability actions give NaN, as in the reference),
    a zero table (uniform strategies) and values beyond 2^24 (i32 -> f32 rounds)"""
    n_actions, tree = rs.build_game_tree(rs.default_flop() if river else rs.three_street_options())
    rounds = 1 if river else 3
    ot = orc.OracleTree(orc.options_default_river() if river else orc.options_three_street())
    for seed, sparse, n_clusters in [(1, 0.0, 70), (2, 0.5, 3), (3, 0.9, 1)]:
        table = rs.create_infosets(n_actions, tree, [n_clusters] * rounds, [1] * rounds, dtype=dtype)
        otab = orc.OracleTable(ot, [1] * rounds, n_clusters, odtype)
        assert bits(table.calc_br(tree)) == bits(otab.calc_br())   # zero table
        fill_both(table, otab, tree, np.random.Generator(np.random.PCG64(seed)), sparse, dtype)
        assert bits(table.calc_br(tree)) == bits(otab.calc_br())
    if dtype == L.I32:
        table = rs.create_infosets(n_actions, tree, [2] * rounds, [1] * rounds)
        otab = orc.OracleTable(ot, [1] * rounds, 2)
        rng = np.random.Generator(np.random.PCG64(9))
        for nd in tree.action_nodes():
            a, n = otab.node_shape(nd.index)
            S = rng.integers(0, 2**31 - 1, (a, n))
            table.upload_node(nd.index, np.zeros((a, n), np.int32), S.astype(np.int32))
            otab.set_node(nd.index, np.zeros((a, n)), S)
        assert bits(table.calc_br(tree)) == bits(otab.calc_br())


def device_and_oracle_game(rng, n0, n1, coarse, sparse=0.1):
    h, cid = river_game(rng, n0, n1, coarse)
    sizes = (int(cid[0].max()) + 1, int(cid[1].max()) + 1)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    table = rs.create_infosets(n_actions, tree, [sizes], [1])
    ot = orc.OracleTree(orc.options_default_river())
    otab = orc.OracleDealTable(ot, [sizes])
    fill_both(table, otab, tree, rng, sparse)
    return tree, table, otab, h, cid


@pytest.mark.parametrize("n0,n1,coarse", [(1081, 1081, 1), (700, 1081, 1), (300, 257, 4), (1, 64, 1), (65, 1, 1), (1081, 1081, 40)])
def test_best_response_equals_oracle_bit_for_bit(n0, n1, coarse):
    rng = np.random.Generator(np.random.PCG64(n0 + 3 * n1 + coarse))
    tree, table, otab, h, cid = device_and_oracle_game(rng, n0, n1, coarse)
    for mode in (L.BR_MAX, L.BR_AVERAGE):
        got = table.best_response(tree, BOARD, h[0], cid[0], h[1], cid[1], mode)
        want = otab.best_response(BOARD, h[0], cid[0], h[1], cid[1], mode)
        assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist(), (mode, got, want)
    ev = table.best_response(tree, BOARD, h[0], cid[0], h[1], cid[1], L.BR_AVERAGE)
    br = table.best_response(tree, BOARD, h[0], cid[0], h[1], cid[1], L.BR_MAX)
    assert abs(ev.sum()) < 1e-9 and (br >= ev - 1e-12).all()


def test_best_response_rejects_what_it_cannot_do():
    rng = np.random.Generator(np.random.PCG64(3))
    tree, table, otab, h, cid = device_and_oracle_game(rng, 30, 30, 1)
    bad = cid[0].copy()
    bad[7] = 30
    with pytest.raises(rs.RsError, match="outside the table"):
        table.best_response(tree, BOARD, h[0], bad, h[1], cid[1])
    with pytest.raises(rs.RsError, match="five distinct cards"):
        table.best_response(tree, [0, 0, 1, 2, 3], h[0], cid[0], h[1], cid[1])
    hb = h[0].copy()
    hb[0] = [BOARD[0], hb[0][1]]
    with pytest.raises(rs.RsError, match="board card"):
        table.best_response(tree, BOARD, hb, cid[0], h[1], cid[1])
    n_actions, tree3 = rs.build_game_tree(rs.three_street_options())
    table3 = rs.create_infosets(n_actions, tree3, [30, 30, 30], [1, 1, 1])
    with pytest.raises(rs.RsError, match="single-round"):
        table3.best_response(tree3, BOARD, h[0], cid[0], h[1], cid[1])
    # a combo twice in a range: the rank-order showdowns index the opponent's hands by their cards (at most 51 holders per card, one hand per card pair) -- rejected there,
    # accepted by the pair loop (the hand simply counts twice)
    hd = np.concatenate([h[1], h[1][:1]])
    cd = np.concatenate([cid[1], cid[1][:1]])
    with pytest.raises(rs.RsError, match="twice"):
        table.best_response(tree, BOARD, h[0], cid[0], hd, cd, L.BR_MAX | L.BR_SORTED)
    table.best_response(tree, BOARD, h[0], cid[0], hd, cd, L.BR_MAX)


def test_trainer_ticks_run_calc_br_and_exploitability_falls():
    """the reference's own configuration (default_flop board, random ranges, ISOMORPHIC river abstraction) on the device trainer:
    * with tick_br on, every discount tick stores calc_br of the table as it was BEFORE the discount (cfr.rs:244-247);
    * zero table: every strategy uniform -- exploitable; after training the average strategy is much less so, and keeps improving."""
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, 2, None)]
    n = 1 << 16
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=5, discount_interval=4 * n, discount_cap=10**12)
    tr.set_tick_br(True)
    with pytest.raises(rs.RsError, match="no discount tick"):
        tr.last_br()
    ev0, br0 = tr.best_response(L.BR_AVERAGE), tr.best_response(L.BR_MAX)
    assert abs(ev0.sum()) < 1e-9 and (br0 >= ev0 - 1e-12).all()
    e0 = tr.exploitability()
    assert e0 > 1.0
    tr.train(4)            # t = 4n: not yet beyond the threshold (cfr.rs:243 is a strict >)
    with pytest.raises(rs.RsError, match="no discount tick"):
        tr.last_br()
    before_tick = None
    tr.deal()
    for player in (0, 1):
        tr.iterate_phase(player, 0)
        tr.iterate_phase(player, 1)
    before_tick = tr.calc_br()
    tr.finish_batch()      # t = 5n > 4n: calc_br, then the discount
    pair, t = tr.last_br()
    assert t == 5 * n and bits(pair) == bits(before_tick)
    tr.train(60)
    e1 = tr.exploitability()
    tr.train(400)
    e2 = tr.exploitability()
    tr.status()
    ev = tr.best_response(L.BR_AVERAGE)
    assert abs(ev.sum()) < 1e-9
    assert e1 < 0.5 * e0 and e2 < e1, (e0, e1, e2)


# ---- best response over multi-round trees (rs_best_response_rounds; oracle: best_response.c orc_best_response_rounds) ------------------------------

def multi_round_device_game(rng, board0, n0, n1, bets, raises, n_clusters, sparse=0.15, tied=0, dtype=L.I32):
    K = 5 - len(board0)
    free = [c for c in range(52) if c not in board0]
    combos = np.array([(a, b) for i, a in enumerate(free) for b in free[i + 1:]], dtype=np.uint8)
    h = [combos[np.sort(rng.choice(len(combos), n, replace=False))] for n in (n0, n1)]
    D = 52 - len(board0)
    prefixes = [1, D, D * (D - 1)][: K + 1]
    cids = [[rng.integers(0, n_clusters[r], size=(prefixes[r], len(h[p]))).astype(np.uint32) for p in (0, 1)] for r in range(K + 1)]
    if tied:
        # last-round info sets that stay within `tied` run-outs each (what a lossless abstraction gives: the two orders of turn and river card, suit swaps): the run-outs
        # fall into small components and the level plan takes the round's own nodes by groups of run-outs (k_br_own_grouped_jobs)
        ro = orc.br_runouts(board0)
        pair = {}
        pid = np.array([pair.setdefault(tuple(sorted(int(c) for c in row[len(board0):])), len(pair)) for row in ro], dtype=np.uint32) // (tied // 2)
        assert len(ro) == prefixes[K] and n_clusters[K] >= (int(pid.max()) + 1) * max(n0, n1)
        cids[K] = [(pid[:, None] * len(h[p]) + np.arange(len(h[p]), dtype=np.uint32)[None, :]).astype(np.uint32) for p in (0, 1)]
        if K == 2:   # and turn info sets of one hand under one turn card (47 or 48 lanes, the first ones side by side): own nodes by columns (k_br_own_cols_jobs)
            assert n_clusters[1] >= D * max(n0, n1)
            cids[1] = [(np.arange(D, dtype=np.uint32)[:, None] * len(h[p]) + np.arange(len(h[p]), dtype=np.uint32)[None, :]).astype(np.uint32) for p in (0, 1)]
    n_actions, tree = rs.build_game_tree(rs.Options(n_board_cards=len(board0), bet_sizes=bets, raise_sizes=raises))
    sizes = [(n_clusters[r], n_clusters[r]) for r in range(K + 1)]
    table = rs.create_infosets(n_actions, tree, sizes, [1] * (K + 1), dtype=dtype)
    ot = orc.OracleTree(orc.make_options(n_board_cards=len(board0), bet_sizes=bets, raise_sizes=raises))
    otab = orc.OracleDealTable(ot, sizes, dtype={L.I32: orc.T_I32, L.F32: orc.T_F32, L.F16: orc.T_F16}[dtype])
    fill_both(table, otab, tree, rng, sparse, dtype)
    return tree, table, otab, h, cids


@pytest.mark.parametrize("board0,n0,n1,bets,raises,n_clusters", [
    ([9, 13, 51, 4], 60, 45, ((0.5, 1.0), (0.5, 1.0)), ((3.0,), (3.0,)), [7, 9]),          # turn start, 48 run-outs, the reference's sizes on both streets
    ([9, 13, 51, 4], 200, 150, ((0.5,), (1.0,)), ((), ()), [40, 300]),
    ([9, 13, 51], 24, 20, ((0.5,), (0.5,), (1.0,)), ((), (), ()), [4, 6, 8]),                # flop start, 2 352 ordered run-outs
    ([9, 13, 51, 4, 47], 90, 70, ((0.5, 1.0),), ((3.0,),), [11]),                          # the full board: one run-out, the single-round case through the same code
    ([9, 13, 51], 24, 20, ((0.5,), (0.5,), (1.0,)), ((), (), ()), [4, 6, -2]),               # river info sets of two lanes (the two orders of turn and river card): groups of run-outs
    ([9, 13, 51], 30, 33, ((0.5,), (0.5,), (0.5, 1.0)), ((), (), ()), [5, 7, -4]),           # ... of four lanes in four run-outs
])
def test_multi_round_best_response_equals_oracle_bit_for_bit(board0, n0, n1, bets, raises, n_clusters):
    rng = np.random.Generator(np.random.PCG64(n0 * 3 + n1 + len(board0)))
    tied = -n_clusters[-1] if n_clusters[-1] < 0 else 0
    if tied:
        n_clusters = [n_clusters[0], 49 * max(n0, n1), 1176 * 2 // tied * max(n0, n1)]
    tree, table, otab, h, cids = multi_round_device_game(rng, board0, n0, n1, bets, raises, n_clusters, tied=tied)
    assert len(orc.br_runouts(board0)) == {3: 2352, 4: 48, 5: 1}[len(board0)]
    import ctypes as C
    nb = L.load().rs_br_runouts(np.array(board0, dtype=np.uint8).ctypes.data_as(C.c_void_p), len(board0), None)
    dev_runouts = np.zeros((nb, 5), dtype=np.uint8)
    L.load().rs_br_runouts(np.array(board0, dtype=np.uint8).ctypes.data_as(C.c_void_p), len(board0), dev_runouts.ctypes.data_as(C.c_void_p))
    assert (dev_runouts == orc.br_runouts(board0)).all()
    vals = {}
    for mode in (L.BR_MAX, L.BR_AVERAGE):
        got = table.best_response_rounds(tree, board0, h[0], h[1], cids, mode)
        want = otab.best_response_rounds(board0, h[0], h[1], cids, mode)
        assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist(), (mode, got, want)
        vals[mode] = got
        # showdowns by rank order (RS_BR_SORTED): a summation order of its own, identical to the oracle's sorted mode bit for bit and equal to the pair loop within rounding
        got_s = table.best_response_rounds(tree, board0, h[0], h[1], cids, mode | L.BR_SORTED)
        want_s = otab.best_response_rounds(board0, h[0], h[1], cids, mode | orc.BR_SORTED)
        assert got_s.view(np.uint64).tolist() == want_s.view(np.uint64).tolist(), (mode, got_s, want_s)
        assert np.allclose(got_s, got, rtol=1e-11, atol=1e-12), (mode, got_s, got)
    assert abs(vals[L.BR_AVERAGE].sum()) < 1e-9
    # the calls above ran the level plan (every node of a tree depth in one launch, a buffer per tree edge); the depth-first walk (one launch per node, two buffers per depth:
    # what a card short of memory gets) must give the same bits
    import os
    os.environ["RS_BR_DEPTH_FIRST"] = "1"
    try:
        for mode in (L.BR_MAX, L.BR_AVERAGE, L.BR_MAX | L.BR_SORTED, L.BR_AVERAGE | L.BR_SORTED):
            dfs = table.best_response_rounds(tree, board0, h[0], h[1], cids, mode)
            want = otab.best_response_rounds(board0, h[0], h[1], cids, mode)
            assert dfs.view(np.uint64).tolist() == want.view(np.uint64).tolist(), (mode, dfs, want)
    finally:
        del os.environ["RS_BR_DEPTH_FIRST"]


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_multi_round_best_response_on_float_tables(dtype):
    """the level plan's kernels read the strategy sums through the table's type: binary32 and binary16 tables, river info sets of four lanes in four run-outs (own nodes and
    the opponent's reach by groups of run-outs), turn info sets by columns -- bit for bit the oracle's on its table of the same type"""
    rng = np.random.Generator(np.random.PCG64(77))
    board0, n0, n1 = [9, 13, 51], 26, 31
    n_clusters = [5, 49 * 31, 1176 * 2 // 4 * 31]
    tree, table, otab, h, cids = multi_round_device_game(rng, board0, n0, n1, ((0.5,), (0.5,), (1.0,)), ((), (), ()), n_clusters, tied=4, dtype=dtype)
    for mode in (L.BR_MAX | L.BR_SORTED, L.BR_AVERAGE | L.BR_SORTED, L.BR_MAX):
        got = table.best_response_rounds(tree, board0, h[0], h[1], cids, mode)
        want = otab.best_response_rounds(board0, h[0], h[1], cids, mode)
        assert got.view(np.uint64).tolist() == want.view(np.uint64).tolist(), (dtype, mode, got, want)


def test_sorted_showdowns_full_ranges_from_a_flop():
    """the case the rank-order showdowns exist for: both full 1 176-combo ranges from a flop (2 352 run-outs, the reference's three-street tree with one bet size): the
    exploitability of a random table through RS_BR_SORTED within a fraction of a second, zero-sum average profile, best response >= average -- and equal to the pair loop
    (3 s per call at this size) where the test can afford it: on the turn-start game below"""
    import time
    mask = ab.card_mask("7h8hQc")
    hands = ab.random_range(mask)
    assert len(hands) == 1176
    n_actions, tree = rs.build_game_tree(rs.Options(n_board_cards=3, bet_sizes=((1.0,),) * 3, raise_sizes=((),) * 3))
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, 1 << 16, seed=5, discount_interval=0)
    tr.train(8)
    tr.status()
    tr.best_response(L.BR_AVERAGE | L.BR_SORTED)          # the first call computes and caches the cluster ids of every (prefix, hand)
    t0 = time.perf_counter()
    ev = tr.best_response(L.BR_AVERAGE | L.BR_SORTED)
    br = tr.best_response(L.BR_MAX | L.BR_SORTED)
    dt = (time.perf_counter() - t0) / 2
    assert abs(ev.sum()) < 1e-9 and (br >= ev - 1e-9).all() and br.sum() / 2 > 0
    assert dt < 30.0, "one full-range best response took %.2f s (a sanity bound only: bench.py times it, solve_3s_full_br_s)" % dt
    # the river's own nodes went by groups of four run-outs (the lossless river abstraction: 4.3 lanes per info set, each in a run-out of its own) and the leaves through the
    # leaf loop: the depth-first walk -- a thread per info set, a launch per leaf -- gives the same bits
    assert tr.br_launches() > 0
    os.environ["RS_BR_DEPTH_FIRST"] = "1"
    try:
        tr.br_release()
        assert tr.best_response(L.BR_MAX | L.BR_SORTED).tobytes() == br.tobytes()
        assert tr.br_launches() == -1
    finally:
        del os.environ["RS_BR_DEPTH_FIRST"]
    tr.destroy()
    # turn start, full ranges (1 128 combos, 48 run-outs): sorted against the pair loop
    mask4 = ab.card_mask("7h8hQc2d")
    hands4 = ab.random_range(mask4)
    n4, tree4 = rs.build_game_tree(rs.Options(n_board_cards=4, bet_sizes=((0.5, 1.0), (0.5, 1.0)), raise_sizes=((3.0,), (3.0,))))
    abs4 = [ab.CardAbstraction.init([hands4, hands4], mask4, r, None) for r in (ab.TURN, ab.RIVER)]
    tr4 = rs.DealTrainer(tree4, abs4, [hands4, hands4], mask4, 1 << 16, seed=6, discount_interval=0)
    tr4.train(20)
    tr4.status()
    level = {}
    for mode in (L.BR_MAX, L.BR_AVERAGE):
        pair, srt = tr4.best_response(mode), tr4.best_response(mode | L.BR_SORTED)
        assert np.allclose(srt, pair, rtol=1e-10, atol=1e-12), (mode, pair, srt)
        level[mode] = srt
    # the level plan's leaf loop (one workgroup per run-out takes every leaf in turn: ranges of 512 combos and more) against the depth-first walk's one-leaf kernel: the same
    # sums in the same order, hence the same bits
    assert tr4.br_launches() > 0
    os.environ["RS_BR_DEPTH_FIRST"] = "1"
    try:
        tr4.br_release()
        for mode in (L.BR_MAX, L.BR_AVERAGE):
            assert tr4.best_response(mode | L.BR_SORTED).tobytes() == level[mode].tobytes()
        assert tr4.br_launches() == -1
    finally:
        del os.environ["RS_BR_DEPTH_FIRST"]
    tr4.destroy()


def test_three_street_trainer_is_solving_the_game():
    """a flop-start three-street game on the device trainer with lossless (ISOMORPHIC) abstractions on every street, small ranges: the exploitability of the average
    strategy -- best response over all 2 352 run-outs -- falls as training goes on (the curve the river game already had)"""
    mask = ab.card_mask("7h8hQc")
    rng = np.random.Generator(np.random.PCG64(12))
    hands = ab.random_range(mask)
    hands = hands[np.sort(rng.choice(len(hands), 70, replace=False))]
    n_actions, tree = rs.build_game_tree(rs.Options(n_board_cards=3, bet_sizes=((1.0,),) * 3, raise_sizes=((),) * 3))
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
    n = 1 << 16
    tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=3, discount_interval=0)
    ev0 = tr.best_response(L.BR_AVERAGE)
    assert abs(ev0.sum()) < 1e-9
    e0 = tr.exploitability()
    tr.train(40)
    e1 = tr.exploitability()
    tr.train(400)
    e2 = tr.exploitability()
    tr.status()
    ev = tr.best_response(L.BR_AVERAGE)
    assert abs(ev.sum()) < 1e-9
    assert e0 > 1.0 and e1 < 0.6 * e0 and e2 < e1, (e0, e1, e2)
    # the trainer keeps its best-response objects between calls (the game-only index and the walk's workspace): this small game takes the level plan -- one launch per tree
    # depth and kind -- its bytes are reported, can be given back, and the next call (which allocates them again) returns the same bits
    assert 0 < tr.br_launches() <= 200, tr.br_launches()
    held = tr.br_bytes()
    assert held > 0
    tr.br_release()
    assert 0 < tr.br_bytes() < held
    assert tr.exploitability() == e2 and tr.br_bytes() <= held   # the object of the showdown mode just asked for has its workspace again; the other mode's stays released
