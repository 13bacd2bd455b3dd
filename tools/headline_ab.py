#!/usr/bin/env python3
"""One leg of the same-card headline A/B (task 1): times the config-2 river sweep (14 action nodes, 1 000 clusters,
9 216 boards, clamp update) of the package found under --root -- this repository's HEAD or an older tree unpacked in a side directory
(git archive <rev> | tar -x -C _ab/<rev>; python -m rustsolver_amd.build there) -- and the card's plain-copy rate right after, in ONE
process, and prints one JSON line.  tools/headline_ab.sh interleaves the legs on one card.

    python tools/headline_ab.py --root . --saturating 100
    python tools/headline_ab.py --root _ab/r01
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--root", default=".")
ap.add_argument("--label", default=None)
ap.add_argument("--boards", type=int, default=9216)
ap.add_argument("--clusters", type=int, default=1000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--saturating", type=int, default=0, help="one regret cell in N beyond +-2.1e9 (trees that have rs_table_plant_saturating)")
ap.add_argument("--seed", type=int, default=1235)
a = ap.parse_args()

root = os.path.abspath(a.root)
sys.path.insert(0, root)
import rustsolver_amd as rs  # noqa: E402
from rustsolver_amd import _lib as L  # noqa: E402

assert os.path.abspath(rs.__file__).startswith(root), (rs.__file__, root)
lib = L.load()
n_actions, tree = rs.build_game_tree(rs.default_flop())
tb = rs.create_infosets(n_actions, tree, [a.clusters], [a.boards])
tb.fill_random(a.seed, (-10**6, 10**6), (0, 10**6))
if a.saturating:
    L.check(lib.rs_table_plant_saturating(tb._h, a.seed, a.saturating))
rootnode = tree.nodes[tree.nodes[0].children[0]]
sg = tb.lane_buffer(rootnode.index, 1)
L.check(lib.rs_fill_uniform_f32(tb._h, sg.ptr, tb.pitch(rootnode.index), a.seed + 17 + rootnode.round_idx, -1.0, 1.0))
leaves = {i: (rs.LEAF_SIGN, sg) for i, nd in enumerate(tree.nodes) if nd.kind == rs.NODE_TERMINAL and nd.ttype != rs.TERM_UNCONTESTED}
tr = rs.MCCFRTrainer(tree, tb, leaves, scale=100.0, mode=rs.UPD_CLAMP_I64, chance_mode=rs.CHANCE_PASS)
tb.sync()


def run(k):
    for _ in range(k):
        L.check(lib.rs_iterate(tr._h, 0, None))
        L.check(lib.rs_iterate(tr._h, 1, None))


run(a.warmup)
tb.sync()
walls = []
for _ in range(3):          # three timed passes of K steps each: the spread inside one process
    t0 = time.perf_counter()
    run(a.steps)
    tb.sync()
    walls.append((time.perf_counter() - t0) / a.steps * 1e3)
tb.profile_reset()
tb.profile_enable(True)
run(a.steps)
tb.sync()
prof = tb.profile_read()
tb.profile_enable(False)
k = prof["tree"] if prof["tree"]["launches"] else prof["update"]
avg_ms = k["ms"] / k["launches"]
algo = k["algo_bytes"] / k["launches"]
g = C.c_double()
L.check(lib.rs_stream_probe(tb._h, 2 << 30, 10, C.byref(g)))
out = {"label": a.label or a.root, "saturating": a.saturating, "ms_per_step": round(statistics.median(walls), 4),
       "ms_per_step_passes": [round(w, 4) for w in walls], "avg_launch_ms": round(avg_ms, 4), "launches": k["launches"],
       "algo_bytes_per_launch": algo, "algo_GBps": round(algo / (avg_ms * 1e-3) / 1e9, 1), "copy_GBps": round(g.value, 1),
       "over_copy": round(algo / (avg_ms * 1e-3) / 1e9 / g.value, 4), "value": round(a.boards / (statistics.median(walls) * 1e-3)),
       "tiled": int(tb.tile_lanes(rootnode.index)) if hasattr(tb, "tile_lanes") else None,
       "env": {k2: v for k2, v in os.environ.items() if k2.startswith("RS_") and k2 != "RS_JIT_CACHE"}}
print(json.dumps(out))
