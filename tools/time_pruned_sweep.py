#!/usr/bin/env python3
"""Times BASELINE configs[1] (river tree, 9 216 boards x 1 000 clusters, i32) with and without RS_UPD_PRUNE (cfr() with prune = true, cfr.rs:379-386),
through the generated subtree kernels and through the level plan.  The synthetic regrets stay above the threshold: this prices the pruned forms of the
kernels (one more read of every traverser node's regret rows, the explored masks), not skipped work.

    python tools/time_pruned_sweep.py
"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import rustsolver_amd as rs

for fuse in (1, 0):
    for mode in ("clamp", "clamp+prune"):
        tr = bench.make_trainer(rs, 9216, 1000, mode, 0, 0, 1235, fuse)
        table = tr.infosets
        bench.run_steps(tr, 5)
        table.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            bench.run_steps(tr, 20)
            table.sync()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        print("fuse=%d %-12s %.3f ms/iteration, %d launches, checksum %016x %016x" % (fuse, mode, best, tr.n_launches(0) + tr.n_launches(1), *table.checksum()), flush=True)
        del tr, table
