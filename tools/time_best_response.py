import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import rustsolver_amd as rs
from rustsolver_amd import _lib as L
from rustsolver_amd import abstraction as ab
mask = ab.card_mask("7h8hQc")
hands = ab.random_range(mask)
print("hands", len(hands))
n_actions, tree = rs.build_game_tree(rs.three_street_options())
t0 = time.perf_counter()
card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, None) for r in range(3)]
print("abstractions %.1f s" % (time.perf_counter() - t0), [a.get_size(0) for a in card_abs])
tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, 1 << 16, seed=1)
print("table GB", tr.infosets.nbytes / 1e9)
for k in range(2):
    t0 = time.perf_counter(); e = tr.exploitability(); print("exploitability", e, "%.2f s" % (time.perf_counter() - t0))
