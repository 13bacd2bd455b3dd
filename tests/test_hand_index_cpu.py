"""CPU (-m "not gpu"): the canonical hand index in front of get-infoset addressing.

  * the oracle (oracle/hand_index.c) against the reference's own known answers for this boundary -- 1 286 792 flop hands
    (out.txt:1), the 12 888 turn clusters and the equalities of test_init_iso_turn (card_abstraction.rs:307-330), 1 081 river
    clusters on options::default_flop()'s board -- and against the published index-set sizes;
  * the oracle's index partition against an independent brute-force canonical form (minimum over 24 suit relabellings);
  * the product's HOST side of the same arithmetic (rs_hand_indexer / rs_card_abs: no GPU involved) against the oracle, bit for bit.
The GPU kernels of the same code are covered by tests/test_gpu_cards.py."""
import json
import os

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import abstraction as ab

SHAPES = [[2], [2, 3], [2, 4], [2, 5], [2, 3, 1], [2, 3, 1, 1], [1, 1, 1, 1, 1, 1, 1], [5, 2], [3, 3, 1], [7]]


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "hand_index.json")))


def random_hands(rng, n, n_cards):
    return np.stack([rng.permutation(52)[:n_cards] for _ in range(n)]).astype(np.uint8)


# ---- oracle vs the reference's numbers -----------------------------------------------------------------------------------------
def test_oracle_sizes_match_reference_and_paper(fx):
    for key, sizes in fx["published"]["sizes"].items():
        cpr = [int(x) for x in key.split(",")]
        ix = orc.HandIndexer(cpr)
        assert [ix.size(r) for r in range(len(cpr))] == sizes, key
    ref = fx["reference"]["flop_size"]
    assert orc.HandIndexer(ref["cards_per_round"]).size(1) == ref["size"]   # out.txt:1


def test_oracle_reproduces_test_init_iso_turn(fx):
    """card_abstraction.rs:307-330 -- the reference's only runnable test at this boundary"""
    t = fx["reference"]["iso_turn"]
    ix = orc.HandIndexer([2, 4])   # BettingRound::Turn => init(2, [2, 4]) (card_abstraction.rs:89)
    rng_hands = ab.random_range(t["flop_mask"])
    keys = ix.generate_map(rng_hands, t["flop_mask"], 4)
    assert len(keys) == t["size"][0] == t["size"][1]
    for a, b in t["equal"]:
        assert ix.get_index(a[:6]) == ix.get_index(b[:6])
    for a, b in t["not_equal"]:
        assert ix.get_index(a[:6]) != ix.get_index(b[:6])


def test_oracle_default_river_abstraction_has_1081_clusters(fx):
    d = fx["reference"]["default_river"]
    mask = ab.card_mask(d["board"])
    assert bin(mask).count("1") == 5
    keys = orc.HandIndexer([2, 5]).generate_map(ab.random_range(mask), mask, 5)
    assert len(keys) == d["size"][0] and len(set(keys.tolist())) == len(keys)


def test_oracle_hand_derived_and_restated_fixtures(fx):
    pre = orc.HandIndexer([2])
    for e in fx["hand_derived"]["preflop"]:
        assert pre.get_index(e["cards"]) == e["index"], e["name"]
    for key, cases in fx["restated"].items():
        if key == "generate_hand":
            continue
        cpr = [int(x) for x in key.split(",")]
        ix = orc.HandIndexer(cpr)
        for c in cases:
            assert [ix.get_index(c["cards"], r) for r in range(len(cpr))] == c["index"]


@pytest.mark.parametrize("cpr", [[2], [2, 3], [2, 4], [2, 5], [2, 3, 1, 1]])
def test_oracle_partition_equals_bruteforce_canonical_form(cpr):
    """two hands share an index exactly when some suit relabelling maps one onto the other (checked by brute force)"""
    rng = np.random.Generator(np.random.PCG64(5))
    ix = orc.HandIndexer(cpr)
    r = len(cpr) - 1
    nc = sum(cpr)
    base = random_hands(rng, 1500, nc)
    perms = [rng.permutation(4) for _ in range(3)]
    hands = [base]
    for p in perms:   # relabelled copies (must collide) ...
        hands.append(((base >> 2) << 2 | p[base & 3]).astype(np.uint8))
    shuffled = base.copy()   # ... and cards reordered inside each round (must collide too)
    at = 0
    for k in cpr:
        shuffled[:, at:at + k] = shuffled[:, at:at + k][:, ::-1]
        at += k
    hands.append(shuffled)
    near = base.copy()       # ... and near misses: one card's rank moved (must NOT collide unless truly isomorphic)
    near[:, -1] = (near[:, -1] + 4) % 52
    ok = np.array([len(set(h.tolist())) == nc for h in near])
    hands.append(near[ok])
    allh = np.concatenate(hands)
    idx = ix.get_index(allh, r)
    canon = [ix.canon(h, r) for h in allh]
    by_idx, by_canon = {}, {}
    for i, (a, c) in enumerate(zip(idx.tolist(), canon)):
        assert by_idx.setdefault(a, c) == c, "one index, two canonical forms: hands %s" % allh[i]
        assert by_canon.setdefault(c, a) == a, "one canonical form, two indices: hands %s" % allh[i]
    assert idx.max() < ix.size(r)
    assert (idx[: len(base)] == idx[len(base): 2 * len(base)]).all()


def test_oracle_unindex_is_a_right_inverse_over_the_whole_flop():
    """every index below size(round) decodes to a hand that encodes back to it: the index is a bijection onto 0..size-1"""
    ix = orc.HandIndexer([2, 3])
    assert ix.size(0) == 169
    for i in range(169):
        h = ix.get_hand(0, i)
        assert ix.get_index(h, 0) == i and len(set(h.tolist())) == 2
    rng = np.random.Generator(np.random.PCG64(6))
    for i in np.concatenate([np.arange(2000), rng.integers(0, ix.size(1), 4000), [ix.size(1) - 1]]):
        h = ix.get_hand(1, int(i))
        assert len(set(h.tolist())) == 5 and ix.get_index(h, 1) == int(i)
    with pytest.raises(IndexError):
        ix.get_hand(1, ix.size(1))


# ---- product host side vs the oracle -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cpr", SHAPES)
def test_host_indexer_matches_oracle_bit_for_bit(cpr):
    rng = np.random.Generator(np.random.PCG64(sum(cpr) * 31 + len(cpr)))
    oix, pix = orc.HandIndexer(cpr), ab.HandIndexer(cpr)
    hands = random_hands(rng, 3000, sum(cpr))
    for r in range(len(cpr)):
        assert pix.size(r) == oix.size(r)
        got = pix.get_index(hands, r)
        want = oix.get_index(hands[:, : pix.n_cards(r)].copy(), r)
        assert (got == want).all(), (cpr, r, np.nonzero(got != want)[0][:5])
        back = pix.get_hand(r, got[:300])
        assert (pix.get_index(back, r) == got[:300]).all()
        for i in range(0, 300, 7):   # same representative as the oracle's get_hand
            assert (back[i] == oix.get_hand(r, got[i])).all()


def test_host_indexer_whole_flop_is_a_bijection():
    pix = ab.HandIndexer([2, 3])
    n = pix.size(1)
    idx = np.arange(n, dtype=np.uint64)
    hands = pix.get_hand(1, idx)
    assert (pix.get_index(hands, 1) == idx).all()
    assert ((hands[:, :, None] == hands[:, None, :]).sum(axis=(1, 2)) == 5).all()   # five distinct cards in every hand
    oix = orc.HandIndexer([2, 3])
    for i in range(0, n, 9973):
        assert (oix.get_hand(1, i) == hands[i]).all()


def test_host_fixtures(fx):
    pre = ab.HandIndexer([2])
    for e in fx["hand_derived"]["preflop"]:
        assert pre.get_index(e["cards"]) == e["index"], e["name"]
    for key, cases in fx["restated"].items():
        if key == "generate_hand":
            continue
        cpr = [int(x) for x in key.split(",")]
        ix = ab.HandIndexer(cpr)
        cards = np.array([c["cards"] for c in cases], dtype=np.uint8)
        for r in range(len(cpr)):
            assert ix.get_index(cards, r).tolist() == [c["index"][r] for c in cases]


def test_host_indexer_errors_instead_of_panics():
    with pytest.raises(rs.RsError):
        ab.HandIndexer([])
    with pytest.raises(rs.RsError):
        ab.HandIndexer([2, 0])
    with pytest.raises(rs.RsError):
        ab.HandIndexer([7, 7, 7])        # more than 16 cards
    ix = ab.HandIndexer([2, 3])
    with pytest.raises(rs.RsError):
        ix.get_index([[0, 0, 1, 2, 3]])  # a card twice
    with pytest.raises(rs.RsError):
        ix.get_index([[0, 52, 1, 2, 3]])  # not a card
    with pytest.raises(IndexError):
        ix.get_hand(1, [ix.size(1)])
    with pytest.raises(ValueError):
        ix.get_index([[0, 1, 2]])        # too few cards for the round


# ---- card abstraction: generate_maps + get_cluster ---------------------------------------------------------------------------------
def test_card_abs_reproduces_test_init_iso_turn(fx):
    """the reference's test, through the product's ABI (ISOMORPHIC::init + get_cluster, card_abstraction.rs:307-330)"""
    t = fx["reference"]["iso_turn"]
    rng_hands = ab.random_range(t["flop_mask"])          # HandRange "random" + remove_invalid_combos (:311-312)
    card_abs = ab.CardAbstraction.init([rng_hands, rng_hands], t["flop_mask"], ab.TURN)
    assert [card_abs.get_size(0), card_abs.get_size(1)] == t["size"]
    for a, b in t["equal"]:
        for player in (0, 1):
            assert card_abs.get_cluster(a, player) == card_abs.get_cluster(b, player)
    for a, b in t["not_equal"]:
        assert card_abs.get_cluster(a, 1) != card_abs.get_cluster(b, 1)


def test_card_abs_default_river_is_1081_and_matches_oracle_order(fx):
    d = fx["reference"]["default_river"]
    mask = ab.card_mask(d["board"])
    hands = ab.random_range(mask)
    card_abs = ab.CardAbstraction.init([hands, hands[::-1]], mask, ab.RIVER)
    assert [card_abs.get_size(0), card_abs.get_size(1)] == d["size"]
    oix = orc.HandIndexer([2, 5])
    assert (card_abs.keys(0) == oix.generate_map(hands, mask, 5)).all()
    assert (card_abs.keys(1) == oix.generate_map(hands[::-1], mask, 5)).all()
    board = [c for c in range(52) if mask >> c & 1]
    cards = np.array([[a, b] + board for a, b in hands], dtype=np.uint8)
    assert (card_abs.get_cluster(cards, 0) == np.arange(len(hands))).all()          # first-appearance order = range order
    assert (card_abs.get_cluster(cards, 1) == np.arange(len(hands))[::-1]).all()
    with pytest.raises(KeyError):                                                    # a hand outside the map: Rust unwrap()s a None
        other = [c for c in range(52) if not mask >> c & 1][:3]
        card_abs.get_cluster([other[0], other[1]] + board[:4] + [other[2]], 0)


@pytest.mark.parametrize("round_,n_board", [(ab.FLOP, 3), (ab.TURN, 3), (ab.RIVER, 3), (ab.RIVER, 4), (ab.TURN, 4)])
def test_card_abs_with_bucket_file_matches_oracle(round_, n_board, tmp_path):
    """EMD / OCHS shape: index -> bucket file -> dense id (card_abstraction.rs:245-251), small ranges, 0..2 missing board cards"""
    rng = np.random.Generator(np.random.PCG64(40 + round_ * 8 + n_board))
    deck = rng.permutation(52)
    mask = sum(1 << int(c) for c in deck[:n_board])
    if 3 + round_ - n_board > 2:
        with pytest.raises(rs.RsError):   # "invalid number of board cards" (card_abstraction.rs:171)
            ab.CardAbstraction.init([ab.random_range(mask)[:5]] * 2, mask, round_)
        return
    allh = ab.random_range(mask)
    h0 = allh[rng.permutation(len(allh))[:40]]
    h1 = allh[rng.permutation(len(allh))[:25]]
    oix = orc.HandIndexer([2, 3 + round_])
    arr = rng.integers(0, 50, size=oix.size(1), dtype=np.uint32) if round_ < ab.RIVER else None
    if arr is None:   # the river indexer has 123 M entries: emulate a bucket file sparsely through the keys instead
        card_abs = ab.CardAbstraction.init([h0, h1], mask, round_)
        assert (card_abs.keys(0) == oix.generate_map(h0, mask, 3 + round_)).all()
        assert (card_abs.keys(1) == oix.generate_map(h1, mask, 3 + round_)).all()
        return
    path = str(tmp_path / ("round_%d_emd.dat" % (round_ + 1)))
    ab.write_cluster_file(path, arr)
    card_abs = ab.CardAbstraction.init([h0, h1], mask, round_, ab.read_cluster_file(path))
    assert card_abs.index_size() == oix.size(1) == len(arr)
    for p, h in ((0, h0), (1, h1)):
        keys = oix.generate_map(h, mask, 3 + round_, arr)
        assert (card_abs.keys(p) == keys).all() and card_abs.get_size(p) == len(keys) <= 50
        board = [int(c) for c in range(52) if mask >> c & 1]
        free = [c for c in range(52) if not (mask >> c & 1) and c not in h[0]]
        cards = np.array(list(h[0]) + board + free[: 3 + round_ - n_board], dtype=np.uint8)
        bucket = int(arr[oix.get_index(cards)])
        assert card_abs.get_cluster(cards, p) == keys.tolist().index(bucket)
    with pytest.raises(IndexError):   # a bucket file shorter than the indexer: Rust panics with index out of bounds
        ab.CardAbstraction.init([h0, h1], mask, round_, arr[:1000])


def test_card_abs_rejects_bad_ranges():
    with pytest.raises(rs.RsError):
        ab.CardAbstraction.init([[(0, 0)], [(1, 2)]], 0b111000, ab.FLOP)     # a card twice
    with pytest.raises(rs.RsError):
        ab.CardAbstraction.init([[(3, 9)], [(1, 2)]], 0b111000, ab.FLOP)     # combo on the board
    with pytest.raises(rs.RsError):
        ab.CardAbstraction.init([[(8, 9)], [(1, 2)]], 0b1111110000, ab.RIVER)  # six board cards


# ---- deal sampler (oracle level; the GPU kernel is compared with it in test_gpu_cards.py) ----------------------------------------
def test_oracle_generate_hand_fixture_and_properties(fx):
    for case in fx["restated"]["generate_hand"]:
        h = ab.random_range(case["board_mask"])
        got = orc.generate_hands(case["seed"], case["first_deal"], case["board_mask"], h, h, len(case["cards9"]))
        assert got.T.tolist() == case["cards9"]
    mask = 0b1011 << 20
    h0 = ab.random_range(mask)[:30]
    h1 = ab.random_range(mask)[[100, 400, 900]]
    deals = orc.generate_hands(9, 0, mask, h0, h1, 4000).T
    board = [c for c in range(52) if mask >> c & 1]
    assert (deals[:, :3] == board).all()                                  # cfr.rs:110-113: given cards first, ascending
    assert all(len(set(d.tolist())) == 9 for d in deals)                  # nothing dealt twice
    s0 = {tuple(x) for x in h0.tolist()}
    s1 = {tuple(x) for x in h1.tolist()}
    assert all(tuple(d[5:7]) in s0 and tuple(d[7:9]) in s1 for d in deals.tolist())
    counts = np.bincount(deals[:, 3:5].reshape(-1), minlength=52)         # free board cards roughly uniform over the 49 left
    free = [c for c in range(52) if c not in board]
    assert counts[board].sum() == 0 and counts[free].min() > 60


def test_oracle_generate_hand_gives_up_where_the_reference_would_spin():
    """cfr.rs:127-137 loops until a combo fits; a range that cannot fit never ends there, here it is an error"""
    mask = 0b111
    with pytest.raises(RuntimeError):
        orc.generate_hands(3, 0, mask, [(10, 11)], [(11, 12)], 1)       # the two ranges always collide


# ---- the integrator's self-check: rs_hand_index_verify (card_abstraction.rs:204-209, :227-229) ---------------------------------------------
def test_hand_index_verify_reports_the_first_disagreeing_hand():
    """The order of hand indices is not pinned by any reference fixture (rust_poker is not vendored), so the library ships the check an integrator runs against
    hand_indexer_s::get_index before loading bucket files from the other side.  Here the `other side` is the oracle's indexer (agreement expected), then a copy
    with two entries tampered with (the first must be reported, with this library's own index for it) and malformed input (errors, not reports)."""
    rng = np.random.Generator(np.random.PCG64(20))
    for cpr in ([2, 3], [2, 4], [2, 5]):
        ix, oix = ab.HandIndexer(cpr), orc.HandIndexer(cpr)
        hands = random_hands(rng, 1000, sum(cpr))
        theirs = np.array([oix.get_index(h) for h in hands], dtype=np.uint64)
        assert ix.verify(hands, theirs) == (1000, None)
        assert ix.verify(hands[:, :2], np.array([oix.get_index(h[:2], 0) for h in hands], dtype=np.uint64), round_=0) == (1000, None)
        bad = theirs.copy()
        bad[417] += 1
        bad[800] ^= 5
        first, got = ix.verify(hands, bad)
        assert (first, got) == (417, int(theirs[417]))
        assert "hand 417" in rs._lib.load().rs_last_error().decode()
        assert ix.verify(hands[:0], theirs[:0]) == (0, None)                      # empty input: nothing to disagree on
        dup = hands.copy()
        dup[3, 1] = dup[3, 0]                                                     # a card twice: an error, not a report
        with pytest.raises(Exception):
            ix.verify(dup, theirs)
        with pytest.raises(ValueError):
            ix.verify(hands, theirs[:-1])
