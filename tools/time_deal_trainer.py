#!/usr/bin/env python3
"""Times the device trainer on the reference's as-coded game (options::default_flop(): board 4d5dAs3cKs, random ranges, ISOMORPHIC river).
Knobs are read when a solver is created (RS_JIT_LANES, RS_JIT_MAX_BLOCKS, ...): run one setting per process.

    TAG=default python tools/time_deal_trainer.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustsolver_amd as rs
# PRUNE=none: never prune; PRUNE=<t>: train()'s schedule with PRUNE_THRESHOLD = t (cfr.rs:190; default 10 000 000): batches beyond it run the `_prune` kernel forms
PRUNE = None if os.environ.get("PRUNE", "10000000") == "none" else int(os.environ.get("PRUNE", "10000000"))
from rustsolver_amd import abstraction as ab
mask = ab.card_mask("4d5dAs3cKs"); hands = ab.random_range(mask)
n_actions, tree = rs.build_game_tree(rs.default_flop())
card_abs = ab.CardAbstraction.init([hands, hands], mask, ab.RIVER)
tr = rs.DealTrainer(tree, [card_abs], [hands, hands], mask, int(os.environ.get("N", str(1 << 22))), seed=7, discount_interval=0, use_graph=bool(int(os.environ.get("GRAPH", "0"))), prune_threshold=PRUNE)
tr.infosets.fill_random(4321, (-10**6, 10**6), (0, 10**6))
K = int(os.environ.get("BATCHES", "20"))
tr.train(5); tr.infosets.sync()
best = 1e9
for _ in range(int(os.environ.get("REPS", "5"))):
    t0 = time.perf_counter(); tr.train(K); tr.infosets.sync(); best = min(best, (time.perf_counter() - t0) / K * 1e3)
print(os.environ.get("TAG", ""), "best %.3f ms/batch" % best)
