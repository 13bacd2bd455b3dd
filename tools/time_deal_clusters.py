#!/usr/bin/env python3
"""Times rs_card_abs_clusters_device (k_deal_clusters) alone: the river game's fixed board (get_cluster tabulated by hole cards) and a flop-start river abstraction (hand index +
hash probe per deal), N deals.   N=4194304 python tools/time_deal_clusters.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustsolver_amd as rs
from rustsolver_amd import _lib as L
from rustsolver_amd import abstraction as ab
from rustsolver_amd.solver import DeviceBuffer, deal_pitch

n = int(os.environ.get("N", str(1 << 22)))
lib = L.load()
for name, board, rnd in (("fixed 5-card board (table by hole cards)", "4d5dAs3cKs", ab.RIVER), ("flop start, river index", "7h8hQc", ab.RIVER), ("flop start, flop index", "7h8hQc", ab.FLOP)):
    mask = ab.card_mask(board)
    hands = ab.random_range(mask)
    _, tree = rs.build_game_tree(rs.default_flop())
    ca = ab.CardAbstraction.init([hands, hands], mask, rnd)
    table = rs.create_infosets(tree.n_action_nodes, tree, [8] * 3, [1] * 3)
    cards = ab.sample_deals(table, 7, 0, mask, [hands, hands], n)
    pitch = deal_pitch(n)
    host = np.zeros((9, pitch), dtype=np.uint8)
    host[:, :n] = cards
    dc = DeviceBuffer.from_numpy(table, host)
    d0, d1 = DeviceBuffer(table, pitch * 4), DeviceBuffer(table, pitch * 4)
    for rep in range(3):
        L.check(lib.rs_card_abs_clusters_device(ca._h, table._h, dc.ptr, n, d0.ptr, d1.ptr))
    table.sync()
    t0 = time.perf_counter()
    K = 20
    for rep in range(K):
        L.check(lib.rs_card_abs_clusters_device(ca._h, table._h, dc.ptr, n, d0.ptr, d1.ptr))
    table.sync()
    print("%-45s %d deals: %.1f us per call" % (name, n, (time.perf_counter() - t0) / K * 1e6))
