"""GPU (-m gpu): the card kernels in front of the update path, through the C ABI, bit-exact against the CPU oracle:
canonical hand index, get_cluster for deal batches, the deal sampler, and the whole MCCFRTrainer loop on the device
(sample -> index -> bucket -> dense id -> showdown -> sampled mccfr sweep -> discount)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import abstraction as ab

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if rs.device_count() < 1:
        pytest.fail("no HIP device visible: GPU parity tests need a real MI355X (there is no CPU fallback)")


@pytest.fixture(scope="module")
def table():
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    return rs.create_infosets(n_actions, tree, [4], [1])


def random_hands(rng, n, n_cards):
    return np.stack([rng.permutation(52)[:n_cards] for _ in range(n)]).astype(np.uint8)


@pytest.mark.parametrize("cpr", [[2], [2, 3], [2, 4], [2, 5], [2, 3, 1, 1], [1, 1, 1, 1, 1, 1, 1], [5, 2], [7]])
def test_device_hand_index_equals_host_and_oracle(table, cpr):
    rng = np.random.Generator(np.random.PCG64(sum(cpr) + 100 * len(cpr)))
    pix, oix = ab.HandIndexer(cpr), orc.HandIndexer(cpr)
    n = 20000 if len(cpr) < 4 else 5000
    hands = random_hands(rng, n, sum(cpr))
    for r in range(len(cpr)):
        got = pix.get_index_device(table, hands, r)
        assert (got == pix.get_index(hands, r)).all()
        assert (got[:1500] == oix.get_index(hands[:1500, : pix.n_cards(r)].copy(), r)).all()
        assert got.max() < pix.size(r)


def test_device_hand_index_fixture_and_ragged_sizes(table, golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "hand_index.json")))
    for key, cases in fx["restated"].items():
        if key == "generate_hand":
            continue
        cpr = [int(x) for x in key.split(",")]
        ix = ab.HandIndexer(cpr)
        cards = np.array([c["cards"] for c in cases], dtype=np.uint8)
        for r in range(len(cpr)):
            assert ix.get_index_device(table, cards, r).tolist() == [c["index"][r] for c in cases]
    ix = ab.HandIndexer([2, 5])
    rng = np.random.Generator(np.random.PCG64(8))
    for n in (0, 1, 63, 64, 65, 257):   # empty, below / at / above the 64-lane pitch
        hands = random_hands(rng, n, 7) if n else np.zeros((0, 7), dtype=np.uint8)
        assert (ix.get_index_device(table, hands) == (ix.get_index(hands) if n else np.zeros(0, np.uint64))).all()


def whole_flop_on_device(table):
    ix = ab.HandIndexer([2, 3])
    n = ix.size(1)
    idx = np.arange(n, dtype=np.uint64)
    return ix, idx, ix.get_hand(1, idx)


def test_device_hand_index_whole_flop_is_the_identity(table):
    """size-independent property at full size: unindex (host) then index (GPU) over all 1 286 792 flop classes is the identity"""
    ix, idx, hands = whole_flop_on_device(table)
    assert (ix.get_index_device(table, hands, 1) == idx).all()


def deals_from(rng, n, mask, h0, h1):
    """valid deals (uint8 [9][n]) drawn with numpy: board completes `mask`, one combo per range, nothing dealt twice"""
    board = [c for c in range(52) if mask >> c & 1]
    out = np.zeros((9, n), dtype=np.uint8)
    k = 0
    while k < n:
        a, b = h0[rng.integers(len(h0))], h1[rng.integers(len(h1))]
        used = set(board) | set(a.tolist()) | set(b.tolist())
        if len(used) != len(board) + 4:
            continue
        free = [c for c in range(52) if c not in used]
        extra = rng.permutation(free)[: 5 - len(board)].tolist()
        out[:, k] = board + extra + a.tolist() + b.tolist()
        k += 1
    return out


@pytest.mark.parametrize("round_,n_board,bucketed", [(ab.RIVER, 5, False), (ab.TURN, 3, False), (ab.RIVER, 3, False), (ab.FLOP, 3, True),
                                                    (ab.TURN, 3, True), (ab.RIVER, 3, True), (ab.RIVER, 4, False), (ab.TURN, 4, True)])
def test_device_get_cluster_equals_host(table, round_, n_board, bucketed):
    rng = np.random.Generator(np.random.PCG64(round_ * 10 + n_board + 50 * bucketed))
    mask = sum(1 << int(c) for c in rng.permutation(52)[:n_board])
    allh = ab.random_range(mask)
    h0 = allh if n_board == 5 else allh[rng.permutation(len(allh))[:60]]
    h1 = allh[::-1] if n_board == 5 else allh[rng.permutation(len(allh))[:45]]
    arr = None
    if bucketed:
        arr = rng.integers(0, 200, size=ab.HandIndexer([2, 3 + round_]).size(1), dtype=np.uint32)
    card_abs = ab.CardAbstraction.init([h0, h1], mask, round_, arr)
    n = 3000
    deals = deals_from(rng, n, mask, h0, h1)
    c0, c1 = card_abs.get_clusters_device(table, deals)
    nb = 3 + round_
    hand0 = np.concatenate([deals[5:7], deals[:nb]]).T   # the acting player's hole cards, then the board (cfr.rs:357-365)
    hand1 = np.concatenate([deals[7:9], deals[:nb]]).T
    assert (c0 == card_abs.get_cluster(hand0, 0)).all() and (c1 == card_abs.get_cluster(hand1, 1)).all()
    assert c0.max() < card_abs.get_size(0) and c1.max() < card_abs.get_size(1)
    # and against the oracle's own chain: index -> bucket -> position among the first-appearance keys
    oix = orc.HandIndexer([2, 3 + round_])
    keys0 = oix.generate_map(h0, mask, nb, arr).tolist()
    pos = {k: i for i, k in enumerate(keys0)}
    for i in range(0, n, 37):
        b = oix.get_index(hand0[i])
        b = int(arr[b]) if arr is not None else b
        assert pos[b] == c0[i]


def test_device_get_cluster_on_boards_the_tables_do_not_cover(table):
    """One or two board cards beyond the initial board: the device reads get_cluster from a table over (hole pair, those cards), filled once through the index path
    (k_cluster_xlut).  A deal whose board does NOT start with the initial board (any order) is none of the table's: it takes the index path itself.  Bucket files with few
    buckets, so that every bucket has a cluster id whatever the board: both kinds of deals in one batch must equal the host's get_cluster."""
    rng = np.random.Generator(np.random.PCG64(77))
    for round_ in (ab.TURN, ab.RIVER):
        mask = ab.card_mask("2c9dKh")
        other = ab.card_mask("2c9dQs")
        allh = [h for h in ab.random_range(mask).tolist() if not ((1 << h[0] | 1 << h[1]) & other)]
        hands = np.array(allh, dtype=np.uint8)[rng.permutation(len(allh))[:80]]
        arr = rng.integers(0, 19, size=ab.HandIndexer([2, 3 + round_]).size(1), dtype=np.uint32)
        card_abs = ab.CardAbstraction.init([hands, hands], mask, round_, arr)
        home, away = deals_from(rng, 1500, mask, hands, hands), deals_from(rng, 1500, other, hands, hands)
        home[:3] = home[:3][rng.permutation(3)]              # the initial board in another order is still the initial board
        deals = np.concatenate([home, away], axis=1)[:, rng.permutation(3000)]
        c0, c1 = card_abs.get_clusters_device(table, deals)
        nb = 3 + round_
        assert (c0 == card_abs.get_cluster(np.concatenate([deals[5:7], deals[:nb]]).T, 0)).all()
        assert (c1 == card_abs.get_cluster(np.concatenate([deals[7:9], deals[:nb]]).T, 1)).all()


def test_device_get_cluster_reports_hands_outside_the_map(table):
    """Rust: `*self.cluster_map[player].get(&hand_index).unwrap()` panics (card_abstraction.rs:208); here an error code"""
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    card_abs = ab.CardAbstraction.init([hands[:10], hands[:10]], mask, ab.RIVER)
    rng = np.random.Generator(np.random.PCG64(3))
    deals = deals_from(rng, 200, mask, hands[500:600], hands[:10])   # player 0 holds hands the abstraction never saw
    with pytest.raises(KeyError):
        card_abs.get_clusters_device(table, deals)
    ok = deals_from(rng, 200, mask, hands[:5], hands[5:10])
    c0, c1 = card_abs.get_clusters_device(table, ok)                 # the error word was cleared: the next batch is fine
    assert c0.max() < 10 and c1.max() < 10


@pytest.mark.parametrize("n_board", [3, 4, 5])
def test_device_deal_sampler_equals_oracle(table, n_board):
    rng = np.random.Generator(np.random.PCG64(70 + n_board))
    mask = sum(1 << int(c) for c in rng.permutation(52)[:n_board])
    allh = ab.random_range(mask)
    h0, h1 = allh[rng.permutation(len(allh))[:300]], allh[rng.permutation(len(allh))[:7]]
    n = 5000
    got = ab.sample_deals(table, 1234, 1000, mask, [h0, h1], n)
    want = orc.generate_hands(1234, 1000, mask, h0, h1, n)
    assert (got == want).all()
    assert (ab.sample_deals(table, 1234, 1500, mask, [h0, h1], 100) == want[:, 500:600]).all()   # deal numbers, not batch positions
    assert all(len(set(col.tolist())) == 9 for col in got.T[:500])


def test_device_deal_sampler_gives_up_where_the_reference_would_spin(table):
    with pytest.raises(RuntimeError):
        ab.sample_deals(table, 3, 0, 0b111, [[(10, 11)], [(11, 12)]], 64)
    with pytest.raises(rs.RsError):
        ab.sample_deals(table, 3, 0, 0b11, [[(10, 11)], [(12, 13)]], 64)      # invalid board mask (options.rs:41)


def load_trainer_pair(options_rs, options_orc, board, ranges, rounds, n_deals, seed, interval, cap, bucket_files=None, fuse=None, prune_threshold=None, odtype=None, scale=100.0,
                      **trainer_kw):
    mask = ab.card_mask(board) if isinstance(board, str) else board
    n_actions, tree = rs.build_game_tree(options_rs)
    first = bin(mask).count("1") - 3
    card_abs = [ab.CardAbstraction.init(ranges, mask, first + r, None if bucket_files is None else bucket_files[r]) for r in range(rounds)]
    tr = rs.DealTrainer(tree, card_abs, ranges, mask, n_deals, seed=seed, discount_interval=interval, discount_cap=cap, fuse_subtrees=fuse,
                        prune_threshold=prune_threshold, scale=scale, **trainer_kw)
    sizes = [(a.get_size(0), a.get_size(1)) for a in card_abs]
    otree = orc.OracleTree(options_orc)
    otab = orc.OracleDealTable(otree, sizes) if odtype is None else orc.OracleDealTable(otree, sizes, odtype)
    cidx = {(r, p): np.zeros(n_deals, dtype=np.uint32) for r in range(rounds) for p in (0, 1)}
    sign = np.zeros(n_deals, dtype=np.float32)
    leaves = {d["id"]: (orc.LEAF_SIGN, sign) for d in otree.as_dicts() if d["kind"] == orc.TERMINAL and d["ttype"] != orc.UNCONTESTED}
    prune = np.zeros(n_deals, dtype=np.uint8)   # all zero = every deal unpruned, whatever ctx.prune says
    osol = orc.OracleDealSolver(otree, otab, leaves, cidx, n_deals, scale=scale, mode=orc.UPD_CLAMP_I64, opp_mode=orc.OPP_SAMPLE, base_seed=seed,
                                prune=True, prune_deal=prune)
    return dict(prune=prune, prune_threshold=prune_threshold, tr=tr, tree=tree, card_abs=card_abs, mask=mask, first=first, otab=otab, osol=osol, cidx=cidx, sign=sign, ranges=ranges,
                rounds=rounds, n_deals=n_deals, seed=seed, interval=interval, cap=cap, bucket_files=bucket_files, t=0, threshold=interval,
                batches=0)


def oracle_batch(ctx):
    """one batch of MCCFRTrainer::train on the CPU: orc_generate_hand, oracle index -> bucket -> first-appearance id, brute-force
    showdown, orc_iterate_deals for both players, then the discount check of cfr.rs:240-262"""
    n = ctx["n_deals"]
    cards = orc.generate_hands(ctx["seed"], ctx["batches"] * n, ctx["mask"], ctx["ranges"][0], ctx["ranges"][1], n)
    for r in range(ctx["rounds"]):
        nb = 3 + ctx["first"] + r
        oix = orc.HandIndexer([2, nb])
        arr = None if ctx["bucket_files"] is None else ctx["bucket_files"][r]
        for p in (0, 1):
            if (r, p) not in ctx.setdefault("pos", {}):   # first-appearance ids of the abstraction: a function of ranges, board and bucket file, not of the batch
                ctx["pos"][(r, p)] = {k: i for i, k in enumerate(oix.generate_map(ctx["ranges"][p], ctx["mask"], nb, arr).tolist())}
            pos = ctx["pos"][(r, p)]
            hands = np.concatenate([cards[5 + 2 * p: 7 + 2 * p], cards[:nb]]).T.copy()
            idx = oix.get_index(hands)
            buckets = idx if arr is None else arr[idx.astype(np.int64)]
            ctx["cidx"][(r, p)][:] = [pos[int(b)] for b in buckets]
    ctx["sign"][:] = orc.showdown_sign(cards)
    if ctx["prune_threshold"] is not None:   # cfr.rs:213-221: q per deal, prune = t > PRUNE_THRESHOLD && q > 0.05
        ctx["prune"][:] = orc.deal_prune_flags(ctx["seed"], ctx["batches"] * n, ctx["prune_threshold"], n)
    for player in (0, 1):
        ctx["osol"].iterate(player)
    ctx["batches"] += 1
    ctx["t"] += n
    if ctx["interval"] and ctx["t"] <= ctx["cap"] and ctx["t"] > ctx["threshold"]:
        orc.lib().orc_discount_table(C.byref(ctx["otab"].tb), orc.discount_factor(ctx["t"], ctx["interval"]))
        ctx["threshold"] = ctx["t"] + ctx["interval"]
    return cards


def compare_trainer_tables(ctx):
    for nd in ctx["tree"].action_nodes():
        r, s = ctx["tr"].infosets.download_node(nd.index)
        ro, so = ctx["otab"].get_node(nd.index)
        assert r.tobytes() == np.ascontiguousarray(ro).tobytes() and s.tobytes() == np.ascontiguousarray(so).tobytes(), "table differs from the oracle at node %d" % nd.index


@pytest.mark.parametrize("fuse", [1, 0])
def test_deal_trainer_reference_as_coded(fuse):
    """options::default_flop(): board 4d5dAs3cKs, random ranges, ISOMORPHIC river abstraction (cfr.rs:159-184), discount ticks on"""
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    ctx = load_trainer_pair(rs.default_flop(), orc.options_default_river(), mask, [hands, hands], 1, 3000, seed=11, interval=4000, cap=13000,
                            fuse=fuse)
    assert [ctx["card_abs"][0].get_size(p) for p in (0, 1)] == [1081, 1081]
    for b in range(5):
        ctx["tr"].train(1)
        cards = oracle_batch(ctx)
        assert (ctx["tr"].cards() == cards).all()
        for p in (0, 1):
            assert (ctx["tr"].clusters(0, p) == ctx["cidx"][(0, p)]).all()
        assert (ctx["tr"].signs() == ctx["sign"]).all()
    ctx["tr"].status()
    assert ctx["tr"].iterations == 15000 == ctx["t"]
    compare_trainer_tables(ctx)


@pytest.mark.parametrize("streets", [1, 3])
def test_deal_trainer_prune_schedule(streets):
    """train()'s prune flag (cfr.rs:213-221) with PRUNE_THRESHOLD moved into reach: deals numbered beyond it whose q > 0.05 are traversed
    with prune = true (explored[] of cfr.rs:379-386, updates of :419-441).  The table starts with regrets on both sides of -10 000 000 so
    that pruning bites; batch 0 carries no flag (its deals are numbered below the threshold), batch 1's tail and batches 2-3 do.  Cards, flags and
    tables equal the oracle's."""
    if streets == 1:
        mask = ab.card_mask("4d5dAs3cKs")
        hands = ab.random_range(mask)
        ctx = load_trainer_pair(rs.default_flop(), orc.options_default_river(), mask, [hands, hands], 1, 3000, seed=21, interval=7000, cap=10**9,
                                prune_threshold=4500)
    else:
        mask = ab.card_mask("7h8hQc")
        hands = ab.random_range(mask)[:300]
        rng = np.random.Generator(np.random.PCG64(12))
        files = [rng.integers(0, 40, size=1286792, dtype=np.uint32), rng.integers(0, 30, size=13960050, dtype=np.uint32), None]
        ctx = load_trainer_pair(rs.three_street_options(), orc.options_three_street(), mask, [hands, hands], 3, 1500, seed=22, interval=4000,
                                cap=10**9, bucket_files=files, prune_threshold=2000)
    rng = np.random.Generator(np.random.PCG64(99))
    for nd in ctx["tree"].action_nodes():
        a, n = ctx["otab"].node_shape(nd.index)
        R = rng.integers(-10**6, 10**6, size=(a, n)).astype(np.int32)
        R[rng.random((a, n)) < 0.3] = -10_000_001
        R[rng.random((a, n)) < 0.05] = -10_000_000   # the threshold itself is NOT explored (cfr.rs:380 is a strict >)
        S = rng.integers(0, 10**5, size=(a, n)).astype(np.int32)
        ctx["tr"].infosets.upload_node(nd.index, R, S)
        ctx["otab"].set_node(nd.index, R, S)
    n = ctx["n_deals"]
    seen = 0
    for b in range(4):
        ctx["tr"].train(1)
        cards = oracle_batch(ctx)
        assert (ctx["tr"].cards() == cards).all()
        got = ctx["tr"].prune_flags()
        assert (got == ctx["prune"]).all()
        seen += int(got.sum())
        assert int(got.sum()) == 0 if (b + 1) * n - 1 <= ctx["prune_threshold"] else got.sum() > 0
    ctx["tr"].status()
    assert seen > n
    compare_trainer_tables(ctx)


@pytest.mark.parametrize("flow", ["one-call", "batch-by-batch", "batch-by-batch+graph", "resweep", "host-loop", "unordered"])
def test_deal_trainer_deals_and_sorts_ahead(flow, monkeypatch):
    """rs_deal_trainer_params.prefetch (the trainer's own choice beyond 256 K deals per batch, forced here): the NEXT batch is dealt on a second stream beside the sweeps, and
    -- round 5 -- where the sweeps are ordered their 32-byte records are sorted there too, traverser p's as soon as sweep p has let go of its buffer; the live arrays are then
    only filled for the accessors.  Nothing observable may move: cards, cluster ids, signs, prune flags after every call are the LIVE batch's, tables equal the oracle's.
    "resweep": a sweep of the live batch asked of the trainer's solver directly, after train() has already sorted the records of the batch dealt ahead (the solver asks the
    trainer before every sweep); "host-loop": rs_deal_trainer_deal / rs_iterate_phase / rs_deal_trainer_finish_batch driven from outside; "unordered": the same with sweeps
    that sort nothing (the round-4 hand-over)."""
    monkeypatch.setenv("RS_JIT_ORDERED", "0" if flow == "unordered" else "1")
    monkeypatch.setenv("RS_JIT_ROWS", "1")
    rng = np.random.Generator(np.random.PCG64(78))
    mask = ab.card_mask("7h8hQc")
    allh = ab.random_range(mask)
    ranges = [allh[rng.permutation(len(allh))[:60]], allh[rng.permutation(len(allh))[:45]]]
    files = [rng.integers(0, 37, size=1286792, dtype=np.uint32), rng.integers(0, 61, size=13960050, dtype=np.uint32), rng.integers(0, 150, size=123156254, dtype=np.uint32)]
    ctx = load_trainer_pair(rs.three_street_options(), orc.options_three_street(), mask, ranges, 3, 1800, seed=6, interval=3000, cap=10**9, bucket_files=files,
                            prune_threshold=4000, prefetch=True, use_graph="graph" in flow)
    tr = ctx["tr"]

    def live_batch_is(cards):
        assert (tr.cards() == cards).all()
        for r in range(3):
            for p in (0, 1):
                assert (tr.clusters(r, p) == ctx["cidx"][(r, p)]).all(), "cluster ids of round %d, player %d" % (r, p)
        assert (tr.signs() == ctx["sign"]).all() and (tr.prune_flags() == ctx["prune"]).all()

    if flow == "one-call":
        tr.train(5)
        for b in range(5):
            cards = oracle_batch(ctx)
        live_batch_is(cards)
    elif flow == "host-loop":
        for b in range(4):
            tr.deal()
            for player in (0, 1):
                tr.iterate_phase(player, 0)
                tr.iterate_phase(player, 1)
            tr.finish_batch()
            live_batch_is(oracle_batch(ctx))
    else:
        for b in range(4):
            tr.train(1 if b != 2 else 2)
            cards = oracle_batch(ctx)
            if b == 2:
                cards = oracle_batch(ctx)
            live_batch_is(cards)
            if flow == "resweep" and b in (1, 2):   # the live batch once more, through the solver (the oracle: the same deals, the next two sweep seeds)
                for player in (0, 1):
                    tr.iterate_phase(player, 0)
                    tr.iterate_phase(player, 1)
                    ctx["osol"].iterate(player)
                live_batch_is(cards)
    tr.status()
    compare_trainer_tables(ctx)


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_deal_trainer_on_float_tables(dtype):
    """rs_deal_trainer_params.table_dtype (round 5; BASELINE configs[4], "fp16 regret/strategy tables with fp32 accumulators", on the algorithm the reference runs): the whole
    device trainer -- dealing, cluster ids, showdowns, the sampled sweeps, the reference's discount ticks -- on a binary16 (or f32) table: f32 per-deal deltas summed in deal
    order, one rounding per cell and sweep.  Same bits as the oracle chain, batch after batch; pruning is refused."""
    rng = np.random.Generator(np.random.PCG64(79))
    mask = ab.card_mask("7h8hQc")
    allh = ab.random_range(mask)
    ranges = [allh[rng.permutation(len(allh))[:40]], allh[rng.permutation(len(allh))[:55]]]
    files = [rng.integers(0, 37, size=1286792, dtype=np.uint32), rng.integers(0, 61, size=13960050, dtype=np.uint32), rng.integers(0, 90, size=123156254, dtype=np.uint32)]
    dt_g, dt_o = (rs.F16, orc.T_F16) if dtype == "f16" else (rs.F32, orc.T_F32)
    ctx = load_trainer_pair(rs.three_street_options(), orc.options_three_street(), mask, ranges, 3, 1200, seed=8, interval=2000, cap=10**9, bucket_files=files, odtype=dt_o,
                            scale=0.5, dtype=dt_g)
    assert ctx["tr"].infosets.download_node(0)[0].dtype == np.float32
    for b in range(4):
        ctx["tr"].train(1 if b else 2)
        cards = oracle_batch(ctx)
        if not b:
            cards = oracle_batch(ctx)
        assert (ctx["tr"].cards() == cards).all()
    ctx["tr"].status()
    compare_trainer_tables(ctx)
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    with pytest.raises(rs.RsError):
        rs.DealTrainer(tree, ctx["card_abs"], ranges, mask, 64, dtype=dt_g, prune_threshold=10**7)


@pytest.mark.parametrize("parts", [True, False])
def test_deal_trainer_three_streets_from_a_flop_with_bucket_files(parts, monkeypatch):
    """flop start (3 board cards), three rounds: EMD-style bucket files on flop and turn, ISOMORPHIC river; narrow ranges so that many
    deals share an info set; all five batches in one train() call"""
    if not parts:   # the ISOMORPHIC river abstraction has ~40-55 K clusters: LDS tiles over ~40-54 cluster ranges with the rows off; with them (the engine's choice for any
                    # batch that has list walkers) its delta rows go straight into the table
        monkeypatch.setenv("RS_JIT_ROWS", "1")
    else:
        monkeypatch.setenv("RS_JIT_ROWS", "0")
    rng = np.random.Generator(np.random.PCG64(77))
    mask = ab.card_mask("7h8hQc")
    allh = ab.random_range(mask)
    ranges = [allh[rng.permutation(len(allh))[:40]], allh[rng.permutation(len(allh))[:55]]]
    files = [rng.integers(0, 37, size=1286792, dtype=np.uint32), rng.integers(0, 61, size=13960050, dtype=np.uint32), None]
    opts_rs, opts_orc = rs.three_street_options(), orc.options_three_street()
    ctx = load_trainer_pair(opts_rs, opts_orc, mask, ranges, 3, 1500, seed=5, interval=2500, cap=10**9, bucket_files=files)
    assert ctx["card_abs"][0].get_size(0) <= 37 and ctx["card_abs"][1].get_size(1) <= 61
    ctx["tr"].train(5)
    for b in range(5):
        cards = oracle_batch(ctx)
    assert (ctx["tr"].cards() == cards).all()
    ctx["tr"].status()
    compare_trainer_tables(ctx)


@pytest.mark.parametrize("world,streets,n", [(2, 1, 700), (3, 1, 700), (2, 3, 700), (2, 3, 600_000), (2, 1, 1 << 21)])
def test_data_parallel_ranks_equal_one_gpu_with_the_union_batch(world, streets, n):
    """`world` ranks x n deals on replicated tables, their i32 deltas summed between sweep and apply (what rs_comm_allreduce_deltas does
    over xGMI; here the test adds them on the host), equal ONE trainer with world*n deals per batch, bit for bit: cards, tables, discount
    ticks.  Ranks are emulated on one GPU, one trainer per rank.  streets = 3: a flop-start tree, i.e. round subtrees, live-deal lists and the
    rank's lane base in the sampling hash together.  n = 2 M: the property at bench.py's size (4 M deals per union batch, the whole range,
    four deals per thread, staged dealing on the second stream).  streets = 3 with 600 000 deals per rank: the list walkers store delta rows (batches beyond 512 K deals), whose
    summing launches belong to phase 0 -- the delta tables must be complete when the ranks exchange them."""
    if streets == 1:
        mask = ab.card_mask("4d5dAs3cKs")
        hands = ab.random_range(mask)[:: (3 if n < 10000 else 1)]
        n_actions, tree = rs.build_game_tree(rs.default_flop())
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)]
    else:
        rng = np.random.Generator(np.random.PCG64(4))
        mask = ab.card_mask("2c9dKh")
        allh = ab.random_range(mask)
        hands = allh[rng.permutation(len(allh))[:35]]
        n_actions, tree = rs.build_game_tree(rs.three_street_options())
        files = [rng.integers(0, 23, size=1286792, dtype=np.uint32), rng.integers(0, 41, size=13960050, dtype=np.uint32), None]
        card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
    kw = dict(seed=21, discount_interval=2 * world * n - 100, discount_cap=10**9)
    # ranks driven phase by phase from outside exchange the delta TABLES: every delta has to pass through them (rs_iterate_phase refuses a solver whose big rounds add their rows
    # straight into the table; the library's own data-parallel path, rs_iterate under a communicator, exchanges those rows as items: tests/test_gpu_multiproc.py)
    ranks = [rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, world=world, rank=r, forms={"direct_rows": rs.FORM_OFF}, **kw) for r in range(world)]
    single = rs.DealTrainer(tree, card_abs, [hands, hands], mask, world * n, **kw)
    for batch in range(4 if streets == 1 else 2):
        for tr in ranks:
            tr.deal()
        for player in (0, 1):
            for tr in ranks:
                tr.iterate_phase(player, 0)
            parts = [tr.deltas() for tr in ranks]
            dreg = np.sum([p[0].view(np.uint32) for p in parts], axis=0, dtype=np.uint32).view(np.int32)    # wrapping, like ncclInt32 sum
            dssm = np.sum([p[1].view(np.uint32) for p in parts], axis=0, dtype=np.uint32).view(np.int32)
            for tr in ranks:
                tr.set_deltas(dreg, dssm)
                tr.iterate_phase(player, 1)
        for tr in ranks:
            tr.finish_batch()
        single.train(1)
        assert (np.concatenate([tr.cards() for tr in ranks], axis=1) == single.cards()).all()
        for nd in tree.action_nodes():
            want = single.infosets.download_node(nd.index)
            for tr in ranks:
                got = tr.infosets.download_node(nd.index)
                assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), (batch, nd.index)
    assert all(tr.iterations == single.iterations for tr in ranks)
    for tr in ranks + [single]:
        tr.status()


def test_data_parallel_trainer_over_rccl_with_one_rank():
    """the RCCL path itself (communicator, ncclAllReduce of the two delta arrays) with a 1-rank communicator: must equal the plain trainer"""
    import ctypes as C2
    from rustsolver_amd import _lib as L
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)[::5]
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    a = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, 900, seed=4, world=1, rank=0)
    b = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, 900, seed=4)
    ident = (C2.c_char * L.COMM_ID_BYTES)()
    rc = L.load().rs_comm_unique_id(ident)
    if rc == L.ERR_COMM:
        pytest.skip("librccl.so cannot be loaded here")
    L.check(rc)
    comm = C2.c_void_p()
    L.check(L.load().rs_comm_create(a.infosets._h, ident, 0, 1, C2.byref(comm)))
    a.attach_comm(comm)
    a.train(3)
    b.train(3)
    for nd in tree.action_nodes():
        ga, gb = a.infosets.download_node(nd.index), b.infosets.download_node(nd.index)
        assert (ga[0] == gb[0]).all() and (ga[1] == gb[1]).all()
    a.attach_comm(None)
    L.load().rs_comm_destroy(comm)


@pytest.mark.parametrize("n_deals", [1, 3, 65, 257])
def test_deal_trainer_ragged_batches(n_deals):
    """batches that are not multiples of the 4-deal vector or the 64-lane pitch, ranges of two combos: against the oracle chain"""
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    ranges = [hands[[10, 700]], hands[[333, 20]]]
    ctx = load_trainer_pair(rs.default_flop(), orc.options_default_river(), mask, ranges, 1, n_deals, seed=3, interval=0, cap=0)
    for b in range(3):
        ctx["tr"].train(1)
        cards = oracle_batch(ctx)
        assert (ctx["tr"].cards() == cards).all()
    ctx["tr"].status()
    compare_trainer_tables(ctx)


def test_deal_trainer_rejects_bad_inputs():
    mask = ab.card_mask("4d5dAs3cKs")
    hands = ab.random_range(mask)
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    river = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
    turn = ab.CardAbstraction.init([hands[:5], hands[:5]], ab.card_mask("4d5dAs3c"), ab.TURN)
    with pytest.raises(rs.RsError):
        rs.DealTrainer(tree, [turn], [hands, hands], mask, 64)            # abstraction of the wrong street
    with pytest.raises(rs.RsError):
        rs.DealTrainer(tree, [river], [hands, hands], mask, 0)            # empty batch
    with pytest.raises(rs.RsError):
        rs.DealTrainer(tree, [river], [hands, hands], 0b11, 64)           # invalid board mask
    with pytest.raises(rs.RsError):
        rs.DealTrainer(tree, [river], [hands, ab.random_range(0)], mask, 64)   # a combo on the board
    tr = rs.DealTrainer(tree, [river], [hands[:1], hands[:1]], mask, 64)   # both players can only hold the same combo
    tr.train(1)
    with pytest.raises(rs.RsError):
        tr.status()


def test_kept_records_at_size_equal_the_table_only_training():
    """solve_three_street's game at its own size (flop start, lossless abstractions: 180 234 river clusters with 200-combo ranges, 2 GB of table, 65 536 deals per batch): forty
    batches with the reference's discount ticks (one every 1.5 batches) by two trainers, one with kept shadow records (the working copy inside rs_deal_trainer_train; written
    back to the table when the call returns), one without (rs_kernel_forms.kept_records = RS_FORM_OFF: the walks gather the table's rows).  Same table checksums after
    every call -- also after a short call (fewer than 16 batches: table and records both written) and after a discount from outside."""
    mask = ab.card_mask("7h8hQc")
    rng = np.random.Generator(np.random.PCG64(2))
    hands = ab.random_range(mask)
    hands = hands[np.sort(rng.choice(len(hands), 200, replace=False))]
    n_actions, tree = rs.build_game_tree(rs.three_street_options())
    card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
    n = 1 << 16
    kept = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=1, use_graph=True)
    plain = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=1, use_graph=True, forms={"kept_records": rs.FORM_OFF})
    for step, batches in enumerate((24, 3, 16)):
        for tr in (kept, plain):
            tr.train(batches)
            tr.status()
        assert kept.infosets.checksum() == plain.infosets.checksum(), "after call %d (%d batches)" % (step, batches)
        if step == 1:   # a write from outside the loop: the records follow (rs_discount) or are rebuilt (anything else)
            for tr in (kept, plain):
                tr.infosets.discount(0.9)
    assert kept.infosets.checksum() == plain.infosets.checksum()
    kept.destroy()
    plain.destroy()
