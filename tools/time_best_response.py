#!/usr/bin/env python3
"""Times DealTrainer.exploitability() (rs_best_response_rounds with rank-order showdowns) from a flop on the three-street tree with lossless abstractions.
    python tools/time_best_response.py                 # the full 1 176-combo ranges: 2.8 M (run-out, hand) lanes
    HANDS=200 python tools/time_best_response.py       # 200 combos per range
    RS_BR_DEPTH_FIRST=1 ...                            # one launch per tree node instead of one per tree depth and kind
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustsolver_amd as rs
from rustsolver_amd import abstraction as ab

mask = ab.card_mask("7h8hQc")
hands = ab.random_range(mask)
if os.environ.get("HANDS"):
    hands = hands[np.random.Generator(np.random.PCG64(1)).permutation(len(hands))[: int(os.environ["HANDS"])]]
n_actions, tree = rs.build_game_tree(rs.three_street_options())
card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, 1 << 16, seed=1)
print("hands", len(hands), "clusters", [a.get_size(0) for a in card_abs], "table GB %.2f" % (tr.infosets.nbytes / 1e9))
for k in range(4):
    t0 = time.perf_counter()
    e = tr.exploitability()
    print("exploitability %.6f  %.3f s  launches %d  held %.2f GB" % (e, time.perf_counter() - t0, tr.br_launches(), tr.br_bytes() / 1e9))
