#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/.

The reference is Rust and cannot be built or imported here, so these
vectors do NOT come from running the reference.  They come from:
  * KNOWN_ANSWERS / DISCOUNT_ANSWERS below: worked out by hand from the Rust semantics of
    infoset.rs:83-102, cfr.rs:413-464 and cfr.rs:248-258 (the table in SURVEY.md section 8(c)),
    typed in here as literals;
  * oracle/np_restate.py (the independent numpy restatement), which must reproduce every literal
    before anything is written, and which generates the seeded random cases and the tree fixtures.

Run:  python tests/golden/make_golden.py      (rewrites the .json files next to this script)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import np_restate as npr  # noqa: E402


def bits(x):
    return [int(v) for v in np.asarray(x, dtype=np.float32).reshape(-1).view(np.uint32)]


def f32_from_bits(b):
    return np.array(b, dtype=np.uint32).view(np.float32)


# --- hand-derived literals; strategy / util given as f32 bit patterns -------
THIRD = 0x3EAAAAAB
KNOWN_ANSWERS = [
    dict(regrets=[0, 0, 0], ssum=[0, 0, 0], utils=[35.0, -35.0, 69.0], reach=1.0,
         strategy_bits=[THIRD, THIRD, THIRD], util_bits=bits(23.0)[0],
         new_regrets=[1200, -5800, 4600], new_ssum=[33, 33, 33]),
    dict(regrets=[300, -100, 100], ssum=[10, 20, 30], utils=[35.0, -52.0, 105.0], reach=0.5,
         strategy_bits=bits([0.75, 0.0, 0.25]), util_bits=bits(52.5)[0],
         new_regrets=[-575, -5325, 2725], new_ssum=[47, 20, 42]),
    dict(regrets=[-5, -7], ssum=[0, 0], utils=[137.0, -69.0], reach=0.25,
         strategy_bits=bits([0.5, 0.5]), util_bits=bits(34.0)[0],
         new_regrets=[2570, -2582], new_ssum=[12, 12]),
    dict(regrets=[1, 2, 3], ssum=[5, 5, 5], utils=[1.0, 2.0, 3.0], reach=float(np.float32(0.3333333)),
         strategy_bits=[0x3E2AAAAB, 0x3EAAAAAB, bits(0.5)[0]], util_bits=0x40155556,
         new_regrets=[-43, -9, 25], new_ssum=[10, 16, 21]),
    dict(regrets=[2147483000, -2147483000, 7], ssum=[2147483600, 0, 1], utils=[1035.0, -1035.0, 0.0], reach=1.0,
         strategy_bits=[bits(1.0)[0], 0, 0x31600004], util_bits=bits(1035.0)[0],
         new_regrets=[2147483000, -2147483648, -103493], new_ssum=[2147483647, 0, 1]),
    dict(regrets=[16777217, 33554433, -1], ssum=[0, 0, 0], utils=[float(np.float32(0.1)), float(np.float32(0.2)),
                                                                   float(np.float32(0.3))],
         reach=float(np.float32(0.7)),
         strategy_bits=[0x3EAAAAAB, 0x3F2AAAAB, 0], util_bits=0x3E2AAAAB,
         new_regrets=[16777213, 33554435, 8], new_ssum=[23, 46, 0]),
]
DISCOUNT_IN = [1000, -1000, 7, -7, 2147483647, -2147483648, 16777217, 3]
DISCOUNT_ANSWERS = [
    dict(tc=100001, d_bits=bits(0.5)[0], out=[500, -500, 3, -3, 1073741824, -1073741824, 8388608, 1]),
    dict(tc=250000, d_bits=0x3F2AAAAB, out=[666, -666, 4, -4, 1431655808, -1431655808, 11184811, 2]),
    dict(tc=1999999, d_bits=0x3F733333, out=[950, -950, 6, -6, 2040109440, -2040109440, 15938355, 2]),
]


def check_literals():
    for ka in KNOWN_ANSWERS:
        R = np.array(ka["regrets"], dtype=np.int32)[:, None]
        S = np.array(ka["ssum"], dtype=np.int32)[:, None]
        U = np.array(ka["utils"], dtype=np.float32)[:, None]
        sig = npr.get_strategy(R)
        assert bits(sig) == ka["strategy_bits"], (ka, bits(sig))
        util, Rn, Sn = npr.update(R, S, U, np.float32(ka["reach"]), 100.0, "clamp")
        assert bits(util)[0] == ka["util_bits"], (ka, bits(util))
        assert Rn[:, 0].tolist() == ka["new_regrets"], (ka, Rn[:, 0])
        assert Sn[:, 0].tolist() == ka["new_ssum"], (ka, Sn[:, 0])
    for da in DISCOUNT_ANSWERS:
        d = npr.discount_factor(da["tc"])
        assert bits(d)[0] == da["d_bits"], (da, bits(d))
        assert npr.discount(DISCOUNT_IN, d).tolist() == da["out"], (da, npr.discount(DISCOUNT_IN, d))


def random_cases():
    cases = []
    rng = np.random.Generator(np.random.PCG64(20261003))
    n = 48
    for A in (2, 3, 4, 5):
        for mode, scale in (("clamp", 100.0), ("wrap", 10000.0)):
            for prune in (False, True):
                R = rng.integers(-10**6, 10**6, size=(A, n)).astype(np.int32)
                S = rng.integers(0, 10**6, size=(A, n)).astype(np.int32)
                U = rng.uniform(-1035, 1035, size=(A, n)).astype(np.float32)
                reach = rng.uniform(0, 1, size=n).astype(np.float32)
                # saturation / prune / degenerate lanes
                R[:, 0] = 0
                R[:, 1] = -5
                R[0, 2], R[1, 2] = 2147483000, -2147483000
                S[0, 2] = 2147483600
                R[:, 3] = rng.integers(-2147483648, 2147483647, size=A)
                R[0, 4] = -10_000_000       # exactly at the prune threshold (not explored: needs > threshold)
                R[1, 5] = -10_000_001
                R[A - 1, 6] = -2_000_000_000
                reach[7] = 0.0
                U[0, 8] = np.float32(3.0e9)  # delta beyond i32 but inside i64
                reach[8] = 1.0
                util, Rn, Sn = npr.update(R, S, U, reach, scale, mode, prune)
                cases.append(dict(A=A, n=n, mode=mode, scale=scale, prune=prune,
                                  regrets=R.reshape(-1).tolist(), ssum=S.reshape(-1).tolist(),
                                  utils_bits=bits(U), reach_bits=bits(reach),
                                  strategy_bits=bits(npr.get_strategy(R)), util_bits=bits(util),
                                  new_regrets=Rn.reshape(-1).tolist(), new_ssum=Sn.reshape(-1).tolist()))
    # discount sweeps
    dcases = []
    for tc in (100001, 200002, 777777, 1999999, 19999999):
        X = rng.integers(-2147483648, 2147483647, size=64).astype(np.int32)
        d = npr.discount_factor(tc)
        dcases.append(dict(tc=tc, d_bits=bits(d)[0], x=X.tolist(), out=npr.discount(X, d).tolist()))
    # extension modes
    ecases = []
    for A in (2, 3):
        for rmplus in (False, True):
            for f16 in (False, True):
                R = rng.uniform(-1000, 1000, size=(A, n)).astype(np.float32)
                S = rng.uniform(0, 1000, size=(A, n)).astype(np.float32)
                if f16:
                    R, S = npr.round_f16(R), npr.round_f16(S)
                U = rng.uniform(-35, 35, size=(A, n)).astype(np.float32)
                reach = rng.uniform(0, 1, size=n).astype(np.float32)
                util, Rn, Sn = npr.update_f32(R, S, U, reach, 1.0, rmplus, f16)
                ecases.append(dict(A=A, n=n, rmplus=rmplus, f16=f16, scale=1.0, regrets_bits=bits(R), ssum_bits=bits(S),
                                   utils_bits=bits(U), reach_bits=bits(reach), util_bits=bits(util),
                                   new_regrets_bits=bits(Rn), new_ssum_bits=bits(Sn)))
    return cases, dcases, ecases


def tree_fixtures():
    river, n_river = npr.build_tree()
    three, n_three = npr.build_tree(n_board_cards=3, bet_sizes=((0.5, 1.0),) * 3, raise_sizes=((3.0,),) * 3)
    summary = dict(
        n_nodes=len(three), n_action_nodes=n_three,
        action_nodes=[[nd["index"], nd["player"], nd["round_idx"], len(nd["children"])]
                      for nd in three if nd["kind"] == "action"],
        terminals=[[i, nd["ttype"], nd["value"], nd["last_to_act"], nd["round"]]
                   for i, nd in enumerate(three) if nd["kind"] == "terminal"],
    )
    return dict(n_action_nodes=n_river, nodes=river), summary


def main():
    check_literals()
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(dict(scale=100.0, mode="clamp", update=KNOWN_ANSWERS, discount_in=DISCOUNT_IN,
                       discount=DISCOUNT_ANSWERS), f, indent=1)
    cases, dcases, ecases = random_cases()
    with open(os.path.join(HERE, "random_cases.json"), "w") as f:
        json.dump(dict(update=cases, discount=dcases, extension=ecases), f)
    river, summary = tree_fixtures()
    with open(os.path.join(HERE, "tree_river.json"), "w") as f:
        json.dump(river, f, indent=1)
    with open(os.path.join(HERE, "tree_three_street.json"), "w") as f:
        json.dump(summary, f)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
