"""Writes tests/golden/hand_index.json.

Three kinds of entries, kept apart on purpose:
  reference   -- numbers typed from the reference's own files (file:line given); they pin the PARTITION into classes
  hand_derived -- preflop indices worked out by hand from the published algorithm (comments show the arithmetic)
  restated    -- seeded random hands with the index the CPU oracle (oracle/hand_index.c) gives them; they pin nothing about
                 the reference, they only freeze this repository's index ORDER so that oracle and GPU cannot drift together

    python tests/golden/make_hand_index_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402


def main():
    out = {
        "reference": {
            "flop_size": {"cards_per_round": [2, 3], "size": 1286792, "where": "out.txt:1 'Generating 1286792 histograms for round 1'"},
            "iso_turn": {"where": "card_abstraction.rs:307-330 test_init_iso_turn", "flop_mask": 7, "round": 1, "range": "random",
                         "size": [12888, 12888],
                         "equal": [[[51, 5, 0, 1, 2, 3, 4], [50, 5, 0, 1, 2, 3, 4]]],
                         "not_equal": [[[6, 5, 0, 1, 2, 3, 4], [50, 5, 0, 1, 2, 3, 4]]]},
            "default_river": {"where": "options.rs:57 board 4d5dAs3cKs, cfr.rs:171 ISOMORPHIC river; C(47,2) hands, no two suits of the "
                                       "board alike, so every hand is its own class", "board": "4d5dAs3cKs", "size": [1081, 1081]},
            "aa_cards": {"where": "gen_abstraction/ehs.rs:109 'cards = vec![48u8, 49]' in test_get_ehs_aa", "cards": [48, 49]},
        },
        "published": {"where": "Waugh 2013, table of index-set sizes",
                      "sizes": {"2": [169], "2,3": [169, 1286792], "2,4": [169, 13960050], "2,5": [169, 123156254],
                                "2,3,1": [169, 1286792, 55190538], "2,3,1,1": [169, 1286792, 55190538, 2428287420]}},
        "hand_derived": {
            "why": "preflop: configuration (1,1,0,0) comes first (smaller count word) with C(13+1,2) = 91 classes, index = lo + C(hi+1,2) "
                   "over the two ranks lo <= hi; configuration (2,0,0,0) follows at offset 91 with the colex rank C(r1,1) + C(r2,2)",
            "preflop": [
                {"cards": [0, 1], "name": "22", "index": 0},
                {"cards": [4, 1], "name": "32o", "index": 1},        # 0 + C(2,2)
                {"cards": [48, 49], "name": "AA", "index": 90},      # 12 + C(13,2)
                {"cards": [48, 45], "name": "AKo", "index": 89},     # 11 + C(13,2)
                {"cards": [0, 4], "name": "32s", "index": 91},       # 91 + C(0,1) + C(1,2)
                {"cards": [48, 44], "name": "AKs", "index": 168},    # 91 + 11 + C(12,2)
                {"cards": [50, 2], "name": "A2s (suit 2)", "index": 157},   # 91 + 0 + C(12,2)
            ]},
        "restated": {},
    }
    rng = np.random.Generator(np.random.PCG64(20261003))
    for cpr in ([2, 3], [2, 4], [2, 5], [2, 3, 1, 1]):
        ix = orc.HandIndexer(cpr)
        hands = np.stack([rng.permutation(52)[: sum(cpr)] for _ in range(64)]).astype(np.uint8)
        out["restated"][",".join(map(str, cpr))] = [
            {"cards": h.tolist(), "index": [int(ix.get_index(h, r)) for r in range(len(cpr))]} for h in hands]
    # deal sampler: first deals of two seeds on the reference's default board and on a flop
    from_mask = lambda m: np.array([(a, b) for a in range(52) for b in range(a) if not ((m >> a) & 1 or (m >> b) & 1)], dtype=np.uint8)
    out["restated"]["generate_hand"] = []
    for seed, mask in ((1, 0b111), (2, (1 << 9) | (1 << 13) | (1 << 48) | (1 << 7) | (1 << 44))):
        h = from_mask(mask)
        out["restated"]["generate_hand"].append({"seed": seed, "board_mask": mask, "range": "random", "first_deal": 5,
                                                 "cards9": orc.generate_hands(seed, 5, mask, h, h, 8).T.tolist()})
    with open(os.path.join(HERE, "hand_index.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
