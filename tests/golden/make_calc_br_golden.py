"""Generates tests/golden/calc_br.json from the independent Python restatement of MCCFRTrainer::calc_br (oracle/np_restate.py, following
cfr.rs:629-745).  The reference cannot be built here and holds no test for calc_br: these are RESTATED vectors (parity unpinned), kept so
that the C oracle and the GPU product are checked against numbers neither of them produced.
    python tests/golden/make_calc_br_golden.py
Table fill rule (the tests repeat it): for every action node in index order, rng = PCG64(seed): strategy_sum = rng.integers(0, 1000, (A, n)),
strategy_sum[rng.random((A, n)) < sparse] = 0, then regrets = rng.integers(-1000, 1000, (A, n))."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import np_restate as npr  # noqa: E402


def main():
    cases = []
    for tree, seed, n_clusters, sparse in [("river", 1, 3, 0.0), ("river", 2, 5, 0.3), ("river", 3, 2, 0.7), ("river", 4, 1, 1.0),
                                           ("three_street", 5, 2, 0.0), ("three_street", 6, 3, 0.5)]:
        if tree == "river":
            nodes, _ = npr.build_tree()
        else:
            nodes, _ = npr.build_tree(n_board_cards=3, bet_sizes=((0.5, 1.0),) * 3, raise_sizes=((3.0,),) * 3)
        rng = np.random.Generator(np.random.PCG64(seed))
        fs = {}
        for nd in sorted((n for n in nodes if n["kind"] == "action"), key=lambda n: n["index"]):
            a = len(nd["children"])
            S = rng.integers(0, 1000, (a, n_clusters))
            S[rng.random((a, n_clusters)) < sparse] = 0
            rng.integers(-1000, 1000, (a, n_clusters))   # the regrets the tests draw next (calc_br never reads them)
            fs[nd["index"]] = npr.get_strategy(S.astype(np.int32))[:, 0]
        br = npr.calc_br(nodes, fs)
        cases.append(dict(tree=tree, seed=seed, n_clusters=n_clusters, sparse=sparse, br_bits=br.view(np.uint32).tolist(),
                          br=[None if np.isnan(x) else float(x) for x in br]))
    out = os.path.join(os.path.dirname(__file__), "calc_br.json")
    json.dump(dict(source="oracle/np_restate.py calc_br (restated from cfr.rs:629-745; parity unpinned)", cases=cases), open(out, "w"), indent=1)
    print(out, [c["br"] for c in cases])


if __name__ == "__main__":
    main()
