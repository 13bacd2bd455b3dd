#!/usr/bin/env python3
"""In-process, interleaved A/B of tree-kernel code-generation variants (cdna guide rule 24: perf deltas come
from interleaved rounds in ONE process).  Variants are selected through the RS_JIT_* environment variables that
rs_jit.cpp reads at solver creation; all trainers share one table, so every variant streams the same bytes.

    python tools/ab_tree_kernel.py "" "RS_TABLE_TILE_LANES=0" "FUSE=0" --rounds 7
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustsolver_amd as rs  # noqa: E402
from rustsolver_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", help='each: "KEY=VAL KEY=VAL" (empty string = defaults); FUSE=0 selects the level plan')
ap.add_argument("--boards", type=int, default=9216)
ap.add_argument("--clusters", type=int, default=1000)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()

n_actions, tree = rs.build_game_tree(rs.default_flop())
root = tree.nodes[tree.nodes[0].children[0]]


def make_table():
    tb = rs.create_infosets(n_actions, tree, [a.clusters], [a.boards])
    tb.fill_random(1235, (-10**6, 10**6), (0, 10**6))
    sg = tb.lane_buffer(root.index, 1)
    L.check(L.load().rs_fill_uniform_f32(tb._h, sg.ptr, tb.pitch(root.index), 99, -1.0, 1.0))
    return tb, sg, {i: (rs.LEAF_SIGN, sg) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}


table, sign, leaves = make_table()
own = []   # variants with RS_TABLE_PITCH_SKEW get a table of their own (the knob is read once per PROCESS: only one skew value per run)
trainers = []
for v in a.variants:
    env = dict(kv.split("=") for kv in v.split()) if v.strip() else {}
    for k in list(os.environ):
        if (k.startswith("RS_JIT_") and k != "RS_JIT_CACHE") or k.startswith("RS_TABLE_"):
            del os.environ[k]
    fuse = int(env.pop("FUSE", "1"))
    os.environ.update(env)
    tb, lv = table, leaves
    if any(k.startswith("RS_TABLE_") for k in env):   # a layout knob (read when a table is created): this variant gets a table of its own
        tb, sg, lv = make_table()
        own.append((tb, sg))
    trainers.append(rs.MCCFRTrainer(tree, tb, lv, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS, fuse_subtrees=fuse))
lib = L.load()


def run(tr, k):
    for _ in range(k):
        L.check(lib.rs_iterate(tr._h, 0, None))
        L.check(lib.rs_iterate(tr._h, 1, None))
    tr.infosets.sync()


for tr in trainers:
    run(tr, 3)
times = [[] for _ in trainers]
for r in range(a.rounds):
    for i, tr in enumerate(trainers):
        t0 = time.perf_counter()
        run(tr, a.steps)
        times[i].append((time.perf_counter() - t0) / a.steps * 1e3)
for v, ts in zip(a.variants, times):
    print("%-40s median %.3f ms  min %.3f  max %.3f   (%s)" % (v or "<default>", statistics.median(ts), min(ts), max(ts),
                                                            " ".join("%.3f" % t for t in ts)))
import ctypes as C  # noqa: E402
g = C.c_double()
L.check(lib.rs_stream_probe(table._h, 2 << 30, 10, C.byref(g)))   # which kind of card was this?
lanes = a.boards * a.clusters
print("this card: plain copy %.0f GB/s; default variant moves %.0f GB/s (2 x 3.576 GB per iteration at 9216 x 1000 lanes)"
      % (g.value, 2 * 3.576e9 * (lanes / 9.216e6) / (statistics.median(times[0]) * 1e-3) / 1e9))
