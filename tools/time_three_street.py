#!/usr/bin/env python3
"""Times the device trainer (rs_deal_trainer) on a flop-start three-street tree with K-bucket files on every street: the reference's
"real" configuration (options.rs:68-77 commented vectors), sampled mccfr.  DESIGN.md section 8 quotes its output.

    python tools/time_three_street.py            # K=5000 N=1048576 by default (environment variables K, N)
    RS_JIT_ROWS=0 RS_JIT_ORDERED=0 / RS_JIT_NO_STAGE=1 python tools/time_three_street.py   # the earlier forms, for comparison (one setting per process)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustsolver_amd as rs
# PRUNE=none: never prune; PRUNE=<t>: train()'s schedule with PRUNE_THRESHOLD = t (cfr.rs:190; default 10 000 000): batches beyond it run the `_prune` kernel forms
PRUNE = None if os.environ.get("PRUNE", "10000000") == "none" else int(os.environ.get("PRUNE", "10000000"))
from rustsolver_amd import abstraction as ab
rng = np.random.Generator(np.random.PCG64(1))
mask = ab.card_mask("7h8hQc")
hands = ab.random_range(mask)
K = int(os.environ.get("K", "5000"))
files = [rng.integers(0, K, size=1286792, dtype=np.uint32), rng.integers(0, K, size=13960050, dtype=np.uint32), rng.integers(0, K, size=123156254, dtype=np.uint32)]
t0 = time.perf_counter()
n_actions, tree = rs.build_game_tree(rs.three_street_options())
card_abs = [ab.CardAbstraction.init([hands, hands], mask, r, files[r]) for r in range(3)]
print("abstractions: %.1f s, sizes" % (time.perf_counter() - t0), [a.get_size(0) for a in card_abs], "action nodes", n_actions)
n = int(os.environ.get("N", str(1 << 20)))
t0 = time.perf_counter()
tr = rs.DealTrainer(tree, card_abs, [hands, hands], mask, n, seed=7, discount_interval=0, prune_threshold=PRUNE, use_graph=bool(int(os.environ.get("GRAPH", "0"))), prefetch=(None if os.environ.get("PREFETCH") is None else bool(int(os.environ["PREFETCH"]))),
                    forms={"shadow": rs.SHADOW_ALL} if os.environ.get("SHADOW_ALL") else ({"kept_records": rs.FORM_OFF} if os.environ.get("NO_KEPT") else None))
# SHADOW_ALL=1: rs_kernel_forms.shadow = RS_SHADOW_ALL instead of the rule; NO_KEPT=1: rs_kernel_forms.kept_records = RS_FORM_OFF
print("trainer create: %.1f s, table %.1f MB" % (time.perf_counter() - t0, tr.infosets.nbytes / 1e6 if not callable(tr.infosets.nbytes) else tr.infosets.nbytes() / 1e6))
tr.train(2); tr.status()
KB = int(os.environ.get("BATCHES", "5"))
t0 = time.perf_counter(); tr.train(KB); tr.infosets.sync(); dt = (time.perf_counter() - t0) / KB
print("walks of the last batch per round (deal, round subtree): traverser 0", tr.walk_counts(0), "traverser 1", tr.walk_counts(1))
print("three-street, %d clusters, %d deals per batch: %.2f ms per batch = %.3g deal-iterations/s" % (K, n, dt * 1e3, n / dt))
