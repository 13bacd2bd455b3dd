#!/usr/bin/env python3
"""Times BASELINE configs[2] (706-node flop+turn+river tree, 5 000 clusters, boards 1/49/2 352, i32, ENUM chance) under the knobs of the environment --
one setting per process: RS_TABLE_TILE_LANES, DTYPE, BOARDS ...

    TAG=untiled RS_TABLE_TILE_LANES=0 python tools/time_config3.py
BOARDS=1,24,1152 emulates the per-rank share of a 2-GPU run (49 turn boards over 2 ranks).
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

boards = [int(x) for x in os.environ.get("BOARDS", "1,49,2352").split(",")]
C = int(os.environ.get("C", "5000"))
t0 = time.perf_counter()
tr = bench.make_trainer(rs, boards, C, os.environ.get("MODE", "clamp"), 0, 0, 1236, 1, "three-street", os.environ.get("DTYPE", "i32"), "full", None)
create = time.perf_counter() - t0
table = tr.infosets
bench.run_steps(tr, 2)
table.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    bench.run_steps(tr, 3)
    table.sync()
    best = min(best, (time.perf_counter() - t0) / 3 * 1e3)
print("%s boards %s: best %.2f ms/iteration, %d launches, workspace %.2f GB, create %.1f s, checksum %016x %016x"
      % (os.environ.get("TAG", ""), boards, best, tr.n_launches(0) + tr.n_launches(1), tr.workspace_bytes / 1e9, create, *table.checksum()))
