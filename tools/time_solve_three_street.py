#!/usr/bin/env python3
"""Times the training loop of bench.py's `solve_three_street` leg on its own: flop start 7h8hQc, the three-street tree, lossless (ISOMORPHIC) abstractions on
every street (190 / 8 213 / 180 234 clusters with 200-combo ranges: far too many for LDS delta tiles on the river), 65 536 deals per batch.

    BATCHES=100 GRAPH=1 python tools/time_solve_three_street.py        (DISCOUNT=0: no discount ticks; NO_KEPT=1: rs_kernel_forms.kept_records off)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustsolver_amd as rs
from rustsolver_amd import abstraction as ab

mask = ab.card_mask("7h8hQc")
rng = np.random.Generator(np.random.PCG64(2))
hands = ab.random_range(mask)
hands = hands[np.sort(rng.choice(len(hands), int(os.environ.get("COMBOS", "200")), replace=False))]
n_actions, tree = rs.build_game_tree(rs.three_street_options())
card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
n = int(os.environ.get("N", str(1 << 16)))
tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=1, use_graph=bool(int(os.environ.get("GRAPH", "1"))), discount_interval=int(os.environ.get("DISCOUNT", "100000")),   # DISCOUNT=0: no discount ticks (what training looks like after cfr.rs:194's 20 M iterations)
                   
                    forms={"shadow": rs.SHADOW_ALL} if os.environ.get("SHADOW_ALL") else ({"kept_records": rs.FORM_OFF} if os.environ.get("NO_KEPT") else None))
# SHADOW_ALL=1: every node read through an AoS shadow rebuilt per sweep (2 GB here); NO_KEPT=1: the river nodes without kept records (their walks gather the table's rows)
tr.train(3); tr.status()
K = int(os.environ.get("BATCHES", "100"))
t0 = time.perf_counter(); tr.train(K); tr.status(); dt = (time.perf_counter() - t0) / K
print("solve_three_street training: clusters %s, table %.2f GB, %d deals per batch: %.3f ms per batch = %.3g deal-iterations/s"
      % ("/".join(str(a.get_size(0)) for a in card_abs), tr.infosets.nbytes / 1e9, n, dt * 1e3, n / dt))
