"""Second, independent restatement of the reference semantics (numpy, vectorised over lanes).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/rs_oracle.h).  It exists so that the C
oracle (rs_oracle.c) is not the only reading of the Rust source: tests compare the two on random
inputs, and tests/golden/make_golden.py uses this file to generate the committed fixtures.

Written array-at-a-time on purpose (the C oracle is scalar and recursive) so that a shared
misreading is less likely.  Reference lines are cited per function.
"""
import numpy as np

F32 = np.float32
I32_MAX, I32_MIN = 2**31 - 1, -(2**31)
PRUNE_THRESHOLD = -10_000_000  # cfr.rs:352


def rust_f32_as_i64(x):
    """Rust `f32 as i64`: trunc toward zero, saturating, NaN -> 0."""
    x = np.asarray(x, dtype=np.float32)
    out = np.zeros(x.shape, dtype=np.int64)
    ok = np.isfinite(x) & (np.abs(x) < F32(2.0**63))
    out[ok] = np.trunc(x[ok].astype(np.float64)).astype(np.int64)
    out[(x >= F32(2.0**63))] = np.iinfo(np.int64).max
    out[(x <= F32(-(2.0**63)))] = np.iinfo(np.int64).min
    return out


def rust_f32_as_i32(x):
    """Rust `f32 as i32`."""
    return np.clip(rust_f32_as_i64(x), I32_MIN, I32_MAX).astype(np.int32)


def get_strategy(R):
    """infoset.rs:83-102 over an [A, n] integer array -> [A, n] f32."""
    R = np.asarray(R)
    A, n = R.shape
    Rf = R.astype(np.float32)  # `as f32` (RNE)
    pos = R > 0
    norm = np.zeros(n, dtype=np.float32)
    for i in range(A):  # index order, f32 accumulation
        norm = np.where(pos[i], norm + Rf[i], norm).astype(np.float32)
    has = norm > 0
    safe = np.where(has, norm, F32(1.0)).astype(np.float32)
    sig = np.zeros((A, n), dtype=np.float32)
    for i in range(A):
        sig[i] = np.where(has, np.where(pos[i], Rf[i] / safe, F32(0.0)), F32(1.0) / F32(A))
    return sig


def get_strategy_f32(R):
    """Extension: the same formula on float regrets."""
    R = np.asarray(R, dtype=np.float32)
    A, n = R.shape
    pos = R > 0
    norm = np.zeros(n, dtype=np.float32)
    for i in range(A):
        norm = np.where(pos[i], norm + R[i], norm).astype(np.float32)
    has = norm > 0
    safe = np.where(has, norm, F32(1.0)).astype(np.float32)
    sig = np.zeros((A, n), dtype=np.float32)
    for i in range(A):
        sig[i] = np.where(has, np.where(pos[i], R[i] / safe, F32(0.0)), F32(1.0) / F32(A))
    return sig


def node_util(sig, U, explored=None):
    """cfr.rs:384/:391/:588: util += utils[i] * strategy[i], sequential, no FMA."""
    A, n = sig.shape
    util = np.zeros(n, dtype=np.float32)
    for i in range(A):
        term = (U[i].astype(np.float32) * sig[i]).astype(np.float32)
        nxt = (util + term).astype(np.float32)
        util = nxt if explored is None else np.where(explored[i], nxt, util)
    return util


def update(R, S, U, reach, scale, mode="clamp", prune=False):
    """One traverser visit for n lanes.  mode 'clamp' = cfr.rs:413-464, 'wrap' = cfr.rs:612-621.
    R, S: [A, n] int32; U: [A, n] f32; reach: [n] f32.  Returns (util, R', S')."""
    R = np.asarray(R, dtype=np.int32)
    S = np.asarray(S, dtype=np.int32)
    U = np.asarray(U, dtype=np.float32)
    reach = np.broadcast_to(np.asarray(reach, dtype=np.float32), R.shape[1:]).astype(np.float32)
    A, n = R.shape
    sig = get_strategy(R)
    explored = (R > PRUNE_THRESHOLD) if prune else np.ones((A, n), dtype=bool)
    util = node_util(sig, U, explored)
    k = (F32(scale) * reach).astype(np.float32)  # (scale * cfr_reach) first
    with np.errstate(invalid="ignore", over="ignore"):
        dR = (k * (U - util).astype(np.float32)).astype(np.float32)
        dS = (k * sig).astype(np.float32)
    if mode == "clamp":
        Rn = np.clip(R.astype(np.int64) + rust_f32_as_i64(dR), I32_MIN, I32_MAX).astype(np.int32)
        Sn = np.clip(S.astype(np.int64) + rust_f32_as_i64(dS), I32_MIN, I32_MAX).astype(np.int32)
    elif mode == "wrap":
        Rn = (R.astype(np.uint32) + rust_f32_as_i32(dR).astype(np.uint32)).astype(np.uint32).view(np.int32)
        Sn = (S.astype(np.uint32) + rust_f32_as_i32(dS).astype(np.uint32)).astype(np.uint32).view(np.int32)
    elif mode == "rmplus":
        Rn = np.clip(R.astype(np.int64) + rust_f32_as_i64(dR), 0, I32_MAX).astype(np.int32)
        Sn = np.clip(S.astype(np.int64) + rust_f32_as_i64(dS), I32_MIN, I32_MAX).astype(np.int32)
    else:
        raise ValueError(mode)
    Rn = np.where(explored, Rn, R)
    Sn = np.where(explored, Sn, S)
    return util, Rn, Sn


def round_f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def update_f32(R, S, U, reach, scale, rmplus=False, f16=False):
    """Extension modes: float tables (optionally binary16 storage), f32 accumulate."""
    R = np.asarray(R, dtype=np.float32)
    S = np.asarray(S, dtype=np.float32)
    U = np.asarray(U, dtype=np.float32)
    reach = np.broadcast_to(np.asarray(reach, dtype=np.float32), R.shape[1:]).astype(np.float32)
    sig = get_strategy_f32(R)
    util = node_util(sig, U)
    k = (F32(scale) * reach).astype(np.float32)
    Rn = (R + (k * (U - util).astype(np.float32)).astype(np.float32)).astype(np.float32)
    Sn = (S + (k * sig).astype(np.float32)).astype(np.float32)
    if rmplus:
        Rn = np.where(Rn > 0, Rn, F32(0.0)).astype(np.float32)
    if f16:
        with np.errstate(over="ignore"):
            Rn, Sn = round_f16(Rn), round_f16(Sn)
    return util, Rn, Sn


def discount_factor(tc, interval=100_000):
    """cfr.rs:248-249: p = (tc / DISCOUNT_INTERVAL) as f32 (integer divide first); d = p/(p+1)."""
    p = F32(tc // interval)
    return F32(p / F32(p + F32(1.0)))


def discount(X, d):
    """cfr.rs:256-257: ((x as f32) * d) as i32."""
    X = np.asarray(X, dtype=np.int32)
    return rust_f32_as_i32((X.astype(np.float32) * F32(d)).astype(np.float32))


# ---------------------------------------------------------------------------------------------------
# public tree (tree_builder.rs + state.rs), written as plain dict/list Python
# ---------------------------------------------------------------------------------------------------
ALLIN_THRESHOLD, MAX_RAISES = 0.67, 2  # constants.rs:2,5


def build_tree(stacks=(500, 500), pot=35, n_board_cards=5, bet_sizes=((0.5, 1.0),), raise_sizes=((3.0,),)):
    """Returns (nodes, n_action_nodes); node = dict(kind, children, ...), ids in creation order."""
    nodes = []
    counter = [0]
    first_round = {3: 0, 4: 1, 5: 2}[n_board_cards]  # state.rs:60-65

    def new(parent, **kw):
        nodes.append(dict(parent=parent, children=[], **kw))
        return len(nodes) - 1

    def valid_actions(st, ridx):  # state.rs:125-157
        cur, oth = st["p"][st["cur"]], st["p"][1 - st["cur"]]
        acts = []
        if oth["wager"] == 0:
            acts.append(("check", 0.0))
        if oth["wager"] > cur["wager"]:
            acts.append(("call", 0.0))
            acts.append(("fold", 0.0))
        if oth["wager"] == 0:
            for b in bet_sizes[ridx]:
                acts.append(("bet", b))
                if b * float(st["pot"]) > ALLIN_THRESHOLD * float(cur["stack"]):
                    break
        allin = any(p["stack"] == 0 for p in st["p"])
        if st["raises"] < MAX_RAISES and not allin and oth["wager"] > cur["wager"]:
            for r in raise_sizes[ridx]:
                acts.append(("raise", r))
                if r * float(oth["wager"]) > ALLIN_THRESHOLD * float(cur["stack"]):
                    break
        return acts

    def clone(st):
        return dict(p=[dict(q) for q in st["p"]], pot=st["pot"], raises=st["raises"], cur=st["cur"],
                    round=st["round"], settled=st["settled"])

    def apply(st, kind, amt):  # state.rs:158-212
        n = clone(st)
        cur, oth = n["p"][n["cur"]], n["p"][1 - n["cur"]]
        if kind == "bet":
            chips = int(n["pot"] * amt)
            if chips > int(cur["stack"] * ALLIN_THRESHOLD):
                chips = cur["stack"]
            cur["stack"] -= chips
            cur["wager"] = chips
            n["pot"] += chips
            n["cur"] = 1 - n["cur"]
        elif kind == "raise":
            chips = int(oth["wager"] * amt)
            if chips > int(cur["stack"] * ALLIN_THRESHOLD):
                chips = cur["stack"]
            cur["stack"] -= chips
            cur["wager"] += chips
            n["raises"] += 1
            n["pot"] += chips
            n["cur"] = 1 - n["cur"]
        elif kind == "call":
            diff = oth["wager"] - cur["wager"]
            if cur["stack"] >= diff:
                n["pot"] += diff
                cur["stack"] -= diff
            else:
                n["pot"] += cur["stack"]
                cur["stack"] = 0
            n["settled"] = True
        elif kind == "check":
            if n["cur"] == 1:
                n["settled"] = True
            n["cur"] = 1 - n["cur"]
        elif kind == "fold":
            cur["folded"] = True
            n["pot"] -= oth["wager"] - cur["wager"]
            n["settled"] = True
        return n

    def action_nodes(parent, ridx, st):  # tree_builder.rs:67-90
        nid = new(parent, kind="action", player=st["cur"], index=counter[0], round_idx=ridx, actions=[])
        counter[0] += 1
        for kind, amt in valid_actions(st, ridx):
            nxt = apply(st, kind, amt)
            folded = any(p["folded"] for p in nxt["p"])
            allin = any(p["stack"] == 0 for p in nxt["p"])
            if nxt["settled"]:
                if nxt["round"] == 2 or allin or folded:  # is_terminal, state.rs:95-99
                    ttype = "SHOWDOWN"
                    if allin and nxt["round"] != 2:
                        ttype = "ALLIN"
                    if folded:
                        ttype = "UNCONTESTED"
                    child = new(nid, kind="terminal", value=nxt["pot"], ttype=ttype, last_to_act=nxt["cur"],
                                round=nxt["round"])
                else:
                    street = clone(nxt)  # to_next_street, state.rs:108-124
                    street["settled"] = False
                    street["cur"] = 0
                    for p in street["p"]:
                        p["wager"] = 0
                    street["round"] += 1
                    child = new(nid, kind="public_chance", round=street["round"])
                    gc = action_nodes(child, ridx + 1, street)
                    nodes[child]["children"].append(gc)
            else:
                child = action_nodes(nid, ridx, nxt)
            nodes[nid]["children"].append(child)
            nodes[nid]["actions"].append([kind, amt])
        return nid

    st0 = dict(p=[dict(stack=stacks[0], wager=0, folded=False), dict(stack=stacks[1], wager=0, folded=False)],
               pot=pot, raises=0, cur=0, round=first_round, settled=False)
    root = new(-1, kind="private_chance")
    nodes[root]["children"].append(action_nodes(root, 0, st0))
    return nodes, counter[0]


# ---------------------------------------------------------------------------------------------------
# opponent sampling (cfr.rs:467-476): rand 0.7 WeightedIndex over supplied bits, second opinion
# ---------------------------------------------------------------------------------------------------
def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def sample_bits(seed, node_index, lanes):
    """the 32-bit counter hash of the opponent sampler (rs_device.hpp sample_bits): node and lane words xor-ed, the upper seed half added to the lane before its multiply,
    the lower one xor-ed in, lowbias32 finisher"""
    lanes = np.asarray(lanes, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    seed = np.uint64(seed)
    with np.errstate(over="ignore"):
        s_lo, s_hi = seed & M, seed >> np.uint64(32)
        n_mix = (np.uint64(node_index + 1) * np.uint64(0xC2B2AE35)) & M
        l_mix = (((((lanes & M) + s_hi) & M) * np.uint64(0x9E3779B9)) & M) ^ (((lanes >> np.uint64(32)) * np.uint64(0x27D4EB2F)) & M)
        x = (s_lo ^ n_mix ^ l_mix) & M
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(0x7FEB352D)) & M
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(0x846CA68B)) & M
        x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def sweep_seed(base_seed, call_index):
    with np.errstate(over="ignore"):
        return int(splitmix64(np.uint64(base_seed) + np.uint64(call_index) * np.uint64(0x632BE59BD9B4E019)))


def weighted_index(W, bits):
    """W: [A, n] f32 weights, bits: [n] uint32 -> [n] sampled indices (rand 0.7.3 WeightedIndex + UniformFloat<f32>)"""
    W = np.asarray(W, dtype=np.float32)
    A, n = W.shape
    total = W[0].copy()
    cum = []
    for i in range(1, A):
        cum.append(total.copy())
        total = (total + W[i]).astype(np.float32)
    u01 = ((np.asarray(bits, dtype=np.uint32) >> np.uint32(9)).astype(np.float32) * F32(2.0 ** -23)).astype(np.float32)
    chosen = ((u01 * total).astype(np.float32) + F32(0.0)).astype(np.float32)
    idx = np.zeros(n, dtype=np.int64)
    for i, c in enumerate(cum):
        idx = np.where(c <= chosen, i + 1, idx)
    return idx


# ---------------------------------------------------------------------------------------------------
# calc_br as coded (cfr.rs:629-745) and the real best response, second opinions
# ---------------------------------------------------------------------------------------------------
def calc_br(nodes, final_strategy_bucket0):
    """MCCFRTrainer::calc_br with its length-1 `op` vectors.  final_strategy_bucket0[index] = get_final_strategy() of bucket 0 of
    the action node with that index (f32 array).  Written with explicit [player][hand] lists like the Rust."""
    def terminal(n, op):  # cfr.rs:695-745
        res = [[F32(0.0)], [F32(0.0)]]
        money = F32(n["value"])
        for p in range(2):
            opp = 1 - p
            ges = F32(0.0)
            for g in range(len(op[0])):
                if n["ttype"] == "UNCONTESTED":
                    pay = F32(F32(op[opp][g] * (F32(-1.0) if p == n["last_to_act"] else F32(1.0))) * money)
                else:
                    pay = F32(op[opp][g] * money)
                res[p][0] = F32(res[p][0] + pay)
                ges = F32(ges + op[opp][g])
            with np.errstate(divide="ignore", invalid="ignore"):
                res[p][0] = F32(res[p][0] * F32(F32(1.0) / ges))
        return res

    def walk(i, op):
        n = nodes[i]
        if n["kind"] == "terminal":
            return terminal(n, op)
        if n["kind"] != "action":
            return walk(n["children"][0], op)  # cfr.rs:646-651
        pl, opp = n["player"], 1 - n["player"]
        probs = [final_strategy_bucket0[n["index"]]]  # only probabilites[h] for h < len(op[player]) == 1 is read (cfr.rs:677-679)
        pay = []
        for a, ch in enumerate(n["children"]):
            newop = [list(op[0]), list(op[1])]
            for h in range(len(newop[pl])):
                newop[pl][h] = F32(newop[pl][h] * F32(probs[h][a]))
            pay.append(walk(ch, newop))
        max_val, max_index = pay[0][pl][0], 0
        for a in range(1, len(pay)):
            if max_val < pay[a][pl][0]:  # cfr.rs:686
                max_val, max_index = pay[a][pl][0], a
        res = [[F32(0.0)], [F32(0.0)]]
        res[pl][0] = max_val
        res[opp][0] = pay[max_index][opp][0]
        return res

    r = walk(0, [[F32(1.0)], [F32(1.0)]])
    return np.array([r[0][0], r[1][0]], dtype=np.float32)


def best_response(nodes, sigma_bar, masks, scores, cids, mode="max"):
    """Value per deal of each player against the other's average strategy (matrix form, f64; sums in numpy's order, so equal to the C
    oracle up to rounding).  sigma_bar(index) -> f32 [A][n_clusters of the acting player]; masks[p] u64 [n_p]; scores[p] [n_p];
    cids[p] [n_p].  Deal probability as generate_hand draws on a full board (cfr.rs:124-137): uniform h0, then h1 uniform among the
    combos of range 1 that avoid h0."""
    n = [len(masks[0]), len(masks[1])]
    compat = (masks[0][:, None] & masks[1][None, :]) == 0  # [n0][n1]
    cnt = compat.sum(axis=1).astype(np.float64)
    w0 = np.where(cnt > 0, 1.0 / (n[0] * np.maximum(cnt, 1.0)), 0.0)
    prob = compat * w0[:, None]  # P(h0, h1)
    s0, s1 = scores[0].astype(np.int64)[:, None], scores[1].astype(np.int64)[None, :]
    cmp01 = np.sign(s0 - s1).astype(np.float64)  # +1: player 0's hand wins
    out = np.zeros(2)
    for p in range(2):
        P = prob if p == 0 else prob.T           # [n_p][n_o]
        sgn = cmp01 if p == 0 else -cmp01.T

        def walk(i, q):  # q: opponent reach [n_o]; returns values [n_p]
            nd = nodes[i]
            if nd["kind"] == "terminal":
                pot = float(np.float32(nd["value"]))
                if nd["ttype"] == "UNCONTESTED":
                    return (P * q[None, :]).sum(axis=1) * (-pot if p == nd["last_to_act"] else pot)
                return (P * sgn * q[None, :]).sum(axis=1) * pot
            if nd["kind"] != "action":
                return walk(nd["children"][0], q)
            sig = sigma_bar(nd["index"]).astype(np.float64)  # [A][C]
            if nd["player"] == p:
                vch = np.stack([walk(ch, q) for ch in nd["children"]])  # [A][n_p]
                if mode == "max":
                    C = sig.shape[1]
                    per = np.zeros((len(vch), C))
                    for a in range(len(vch)):
                        np.add.at(per[a], cids[p], vch[a])
                    best = per.argmax(axis=0)  # first maximum
                    return vch[best[cids[p]], np.arange(n[p])]
                return (sig[:, cids[p]] * vch).sum(axis=0)
            o = 1 - p
            return sum(walk(ch, q * sig[a, cids[o]]) for a, ch in enumerate(nd["children"]))

        out[p] = walk(0, np.ones(n[1 - p])).sum()
    return out
