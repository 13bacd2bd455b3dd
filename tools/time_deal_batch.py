#!/usr/bin/env python3
"""Times bench.py's deal-batch leg (river tree, given cluster ids, sampled opponents) at a chosen cluster count -- e.g. 1 081, where the delta tiles of the seven
traverser nodes miss one workgroup's LDS by 1 %.  One knob setting per process.

    C=1081 N=4194304 python tools/time_deal_batch.py
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import rustsolver_amd as rs

C, N = int(os.environ.get("C", "1081")), int(os.environ.get("N", str(1 << 22)))
out = bench.deal_batch_leg(rs, 0, N, C, False, 0.0)
print("%s %d clusters, %d deals per batch: %.3f ms per batch = %.3g deal-iterations/s" % (os.environ.get("TAG", ""), C, N, out["ms_per_batch"], out["value"]))
