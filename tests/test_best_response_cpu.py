"""CPU: the two readers of the average strategy in the oracle (oracle/best_response.c) -- calc_br as coded (cfr.rs:629-745) and the real
best response -- against hand-derived answers, the independent Python restatement (oracle/np_restate.py), the committed fixture
(tests/golden/calc_br.json) and the properties a best response must have.  PARITY UNPINNED: the reference has no test for calc_br."""
import json
import os

import numpy as np
import pytest

from oracle import np_restate as npr
from oracle import orc

BOARD = [4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3, 4 * 1 + 0, 4 * 11 + 3]   # 4d 5d As 3c Ks, options.rs:55


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32).tolist()


def fill(table, tree, rng, sparse=0.0, lo=0, hi=1000):
    """random strategy sums; `sparse` = share of cells forced to 0 (zero-probability actions make the as-coded walk produce NaN)"""
    fs = {}
    for d in tree.as_dicts():
        if d["kind"] != orc.ACTION:
            continue
        a, n = table.node_shape(d["index"])
        S = rng.integers(lo, hi, (a, n))
        S[rng.random((a, n)) < sparse] = 0
        table.set_node(d["index"], rng.integers(-1000, 1000, (a, n)), S)
        fs[d["index"]] = npr.get_strategy(S.astype(np.int32))
    return fs


def test_calc_br_hand_derived():
    """no bet sizes: P0 check, P1 check, showdown of the 35 pot.  op stays 1.0, a SHOWDOWN pays +pot to BOTH players (cfr.rs:726):
    (1 * 35) * (1 / 1) = 35 for each.  With bets, a zero table: every strategy is uniform and both players' maxima are the largest
    showdown pot (1035 after Bet1.0-Raise3-Raise3-Call) up to the rounding of op * pot * (1 / op)."""
    ot = orc.OracleTree(orc.make_options(bet_sizes=((),), raise_sizes=((),)))
    assert ot.n_action_nodes == 2
    tb = orc.OracleTable(ot, [1], 3)
    assert tb.calc_br().tolist() == [35.0, 35.0]
    ot = orc.OracleTree(orc.options_default_river())
    tb = orc.OracleTable(ot, [1], 3)
    got = tb.calc_br()
    assert np.allclose(got, [1035.0, 1035.0], rtol=1e-6)


@pytest.mark.parametrize("tree_name", ["river", "three_street"])
@pytest.mark.parametrize("sparse", [0.0, 0.4, 1.0])
def test_calc_br_c_oracle_equals_python_restatement(tree_name, sparse):
    if tree_name == "river":
        ot, (nodes, _) = orc.OracleTree(orc.options_default_river()), npr.build_tree()
        n_boards = [1]
    else:
        ot = orc.OracleTree(orc.options_three_street())
        nodes, _ = npr.build_tree(n_board_cards=3, bet_sizes=((0.5, 1.0),) * 3, raise_sizes=((3.0,),) * 3)
        n_boards = [1, 1, 1]
    assert len(nodes) == ot.n_nodes
    for seed in range(4):
        rng = np.random.Generator(np.random.PCG64(900 + seed))
        tb = orc.OracleTable(ot, n_boards, 4)
        fs = fill(tb, ot, rng, sparse=sparse)
        want = npr.calc_br(nodes, {k: v[:, 0] for k, v in fs.items()})
        assert bits(tb.calc_br()) == bits(want)


def test_calc_br_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "calc_br.json")))
    for case in fx["cases"]:
        ot = orc.OracleTree(orc.options_default_river() if case["tree"] == "river" else orc.options_three_street())
        tb = orc.OracleTable(ot, [1, 1, 1][: 1 if case["tree"] == "river" else 3], case["n_clusters"])
        fill(tb, ot, np.random.Generator(np.random.PCG64(case["seed"])), sparse=case["sparse"])
        assert bits(tb.calc_br()) == case["br_bits"], case


def river_game(rng, n0, n1, coarse):
    """two ranges on the default board, cluster ids either one per hand or `coarse` hands per cluster"""
    free = [c for c in range(52) if c not in BOARD]
    combos = np.array([(a, b) for i, a in enumerate(free) for b in free[i + 1:]], dtype=np.uint8)
    h = [combos[np.sort(rng.choice(len(combos), n, replace=False))] for n in (n0, n1)]
    cid = [np.arange(len(x), dtype=np.uint32) // coarse for x in h]
    return h, cid


def oracle_game(rng, n0, n1, coarse, sparse=0.1):
    h, cid = river_game(rng, n0, n1, coarse)
    ot = orc.OracleTree(orc.options_default_river())
    tb = orc.OracleDealTable(ot, [(int(cid[0].max()) + 1, int(cid[1].max()) + 1)])
    fs = fill(tb, ot, rng, sparse=sparse)
    return ot, tb, fs, h, cid


@pytest.mark.parametrize("n0,n1,coarse", [(40, 40, 1), (55, 30, 1), (60, 60, 7), (1, 50, 1), (1081, 300, 3)])
def test_best_response_c_oracle_equals_numpy_restatement(n0, n1, coarse):
    rng = np.random.Generator(np.random.PCG64(n0 * 7 + n1))
    ot, tb, fs, h, cid = oracle_game(rng, n0, n1, coarse)
    nodes, _ = npr.build_tree()
    masks = [(np.uint64(1) << x[:, 0].astype(np.uint64)) | (np.uint64(1) << x[:, 1].astype(np.uint64)) for x in h]
    scores = [np.array([orc.evaluate7(list(x) + BOARD) for x in hp]) for hp in h]
    for mode, name in ((0, "max"), (1, "avg")):
        got = tb.best_response(BOARD, h[0], cid[0], h[1], cid[1], mode)
        want = npr.best_response(nodes, lambda i: fs[i], masks, scores, cid, name)
        assert np.allclose(got, want, rtol=1e-11, atol=1e-12), (mode, got, want)


@pytest.mark.parametrize("coarse", [1, 5])
def test_best_response_properties(coarse):
    """zero-sum: the average profile's values cancel; a best response is worth at least the average strategy; exploitability >= 0"""
    rng = np.random.Generator(np.random.PCG64(77 + coarse))
    ot, tb, fs, h, cid = oracle_game(rng, 120, 90, coarse)
    ev = tb.best_response(BOARD, h[0], cid[0], h[1], cid[1], 1)
    br = tb.best_response(BOARD, h[0], cid[0], h[1], cid[1], 0)
    assert abs(ev.sum()) < 1e-9
    assert (br >= ev - 1e-12).all()
    assert br.sum() / 2 > 0


def test_best_response_folding_everything_loses_the_pot():
    """hand-derived: player 0's average strategy is all-in on action 0 (Check) and, facing a bet, Fold (action 1 of Call / Fold / Raise):
    player 1's best response is worth at least the 35 pot of an UNCONTESTED leaf per deal -- betting wins it outright"""
    rng = np.random.Generator(np.random.PCG64(5))
    h, cid = river_game(rng, 80, 80, 1)
    ot = orc.OracleTree(orc.options_default_river())
    tb = orc.OracleDealTable(ot, [(80, 80)])
    for d in ot.as_dicts():
        if d["kind"] != orc.ACTION:
            continue
        a, n = tb.node_shape(d["index"])
        S = np.zeros((a, n), dtype=np.int64)
        kinds = [k for k, _ in d["actions"]]
        S[kinds.index(orc.ACT_FOLD) if orc.ACT_FOLD in kinds else 0] = 100
        tb.set_node(d["index"], np.zeros((a, n)), S)
    br = tb.best_response(BOARD, h[0], cid[0], h[1], cid[1], 0)
    assert br[1] >= 35.0 - 1e-9


# ---- multi-round best response (oracle/best_response.c orc_best_response_rounds) ------------------------------------------------------------

def multi_round_game(rng, board0, n0, n1, bets, raises, n_clusters):
    """a turn- or flop-start game: ranges avoiding the initial board, a tree with one betting round per street left, random cluster ids per (round, board prefix, hand)
    -- imperfect recall on purpose -- and a random table"""
    K = 5 - len(board0)
    free = [c for c in range(52) if c not in board0]
    combos = np.array([(a, b) for i, a in enumerate(free) for b in free[i + 1:]], dtype=np.uint8)
    h = [combos[np.sort(rng.choice(len(combos), n, replace=False))] for n in (n0, n1)]
    ot = orc.OracleTree(orc.make_options(n_board_cards=len(board0), bet_sizes=bets, raise_sizes=raises))
    runouts = orc.br_runouts(board0)
    D = 52 - len(board0)
    prefixes = [1, D, D * (D - 1)][: K + 1]
    cids = [[rng.integers(0, n_clusters[r], size=(prefixes[r], len(h[p]))).astype(np.uint32) for p in (0, 1)] for r in range(K + 1)]
    tb = orc.OracleDealTable(ot, [(n_clusters[r], n_clusters[r]) for r in range(K + 1)])
    fs = fill(tb, ot, rng, sparse=0.15)
    return ot, tb, fs, h, cids, runouts


def brute_force_average_value(ot, fs, board0, h, cids, runouts):
    """EV of player 0 when both play their average strategies: an explicit sum over every deal generate_hand can draw (cfr.rs:100-143), one scalar tree walk
    per deal -- written independently of the vector walk in best_response.c"""
    nodes = ot.as_dicts()
    K = 5 - len(board0)
    D = 52 - len(board0)
    per_prefix = [int(np.prod([D - i for i in range(r, K)])) for r in range(K + 1)]
    pb = 1.0 / len(runouts)

    def walk(i, b, hi, s0, s1):
        d = nodes[i]
        if d["kind"] == orc.TERMINAL:
            pot = float(np.float32(d["value"]))
            if d["ttype"] == orc.UNCONTESTED:
                return -pot if d["last_to_act"] == 0 else pot
            return pot if s0 > s1 else (-pot if s0 < s1 else 0.0)
        if d["kind"] != orc.ACTION:
            return walk(d["children"][0], b, hi, s0, s1)
        r, p = d["round_idx"], d["player"]
        k = cids[r][p][b // per_prefix[r], hi[p]]
        sig = fs[d["index"]][:, k]
        return sum(float(sig[a]) * walk(c, b, hi, s0, s1) for a, c in enumerate(d["children"]) if sig[a] != 0)

    total = 0.0
    for b, cards in enumerate(runouts):
        new = set(int(c) for c in cards[len(board0):])
        ok0 = [i for i, x in enumerate(h[0]) if not (set(map(int, x)) & new)]
        for i0 in ok0:
            used = new | set(map(int, h[0][i0]))
            ok1 = [j for j, y in enumerate(h[1]) if not (set(map(int, y)) & used)]
            if not ok1:
                continue
            w = pb / (len(ok0) * len(ok1))
            s0 = orc.evaluate7(list(h[0][i0]) + list(cards))
            for i1 in ok1:
                s1 = orc.evaluate7(list(h[1][i1]) + list(cards))
                total += w * walk(0, b, (i0, i1), s0, s1)
    return total


def test_multi_round_oracle_reduces_to_the_single_round_one():
    rng = np.random.Generator(np.random.PCG64(31))
    ot, tb, fs, h, cid = oracle_game(rng, 70, 50, 3)
    for mode in (0, 1):
        one = tb.best_response(BOARD, h[0], cid[0], h[1], cid[1], mode)
        many = tb.best_response_rounds(BOARD, h[0], h[1], [[cid[0][None, :], cid[1][None, :]]], mode)
        assert one.view(np.uint64).tolist() == many.view(np.uint64).tolist()


def test_multi_round_average_value_equals_a_deal_by_deal_enumeration():
    """turn start (48 run-outs), two betting rounds, imperfect-recall clusters: the vector walk's average-profile value = the explicit sum over all deals"""
    rng = np.random.Generator(np.random.PCG64(32))
    board0 = [4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3, 4 * 1 + 0]
    ot, tb, fs, h, cids, runouts = multi_round_game(rng, board0, 9, 7, ((0.5,), (1.0,)), ((), ()), [5, 4])
    assert len(runouts) == 48 and ot.n_action_nodes > 8
    got = tb.best_response_rounds(board0, h[0], h[1], cids, 1)
    want = brute_force_average_value(ot, fs, board0, h, cids, runouts)
    assert abs(got[0] - want) < 1e-9 * max(1.0, abs(want)) and abs(got[0] + got[1]) < 1e-9


@pytest.mark.parametrize("board0,n_clusters", [([4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3], [3, 4, 5]), ([4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3, 4 * 1 + 0], [6, 2])])
def test_multi_round_best_response_properties(board0, n_clusters):
    """flop start (2 352 ordered run-outs) and turn start: zero-sum average profile, best response >= average, and one cluster per lane (perfect information about
    one's own lane) can only do better than coarse clusters"""
    rng = np.random.Generator(np.random.PCG64(33 + len(board0)))
    K = 5 - len(board0)
    ot, tb, fs, h, cids, runouts = multi_round_game(rng, board0, 10, 8, ((0.5,),) * (K + 1), ((),) * (K + 1), n_clusters)
    ev = tb.best_response_rounds(board0, h[0], h[1], cids, 1)
    br = tb.best_response_rounds(board0, h[0], h[1], cids, 0)
    assert abs(ev.sum()) < 1e-9
    assert (br >= ev - 1e-9).all()
    assert br.sum() / 2 > 0


@pytest.mark.parametrize("board0,n_clusters,n0,n1", [([4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3], [3, 4, 5], 40, 33), ([4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3, 4 * 1 + 0], [6, 2], 90, 120),
                                                      ([4 * 2 + 1, 4 * 3 + 1, 4 * 12 + 3, 4 * 1 + 0, 4 * 7 + 2], [9], 300, 280)])
def test_sorted_showdowns_equal_the_pair_loop(board0, n_clusters, n0, n1):
    """RS_BR_SORTED restated (orc_best_response_rounds, mode | 0x100): the opponent's hands of a run-out sorted by (score, index), a leaf = a difference of prefix sums of
    its reach corrected for the holders of the traverser's two cards.  A different summation order from the pair loop of cfr.rs:323-347, hence pinned to it within f64
    rounding -- on ranges with plenty of shared cards, equal scores (ties), hands blocked by the run-out and the same hand in both ranges -- and, through it, to the deal-by-deal
    enumeration the pair loop is pinned by."""
    rng = np.random.Generator(np.random.PCG64(77 + len(board0)))
    K = 5 - len(board0)
    ot, tb, fs, h, cids, runouts = multi_round_game(rng, board0, n0, n1, ((0.5,),) * (K + 1), ((),) * (K + 1), n_clusters)
    if n0 >= 90:
        assert len(set(map(tuple, h[0])) & set(map(tuple, h[1]))) > 0      # the same hand on both sides exists (uncontested leaves add it back)
    for mode in (0, 1):
        pair = tb.best_response_rounds(board0, h[0], h[1], cids, mode)
        srt = tb.best_response_rounds(board0, h[0], h[1], cids, mode | orc.BR_SORTED)
        assert np.allclose(srt, pair, rtol=1e-11, atol=1e-12), (mode, pair, srt)
