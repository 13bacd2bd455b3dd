"""Writes tests/golden/kmeans_emd.json: the histograms and expected distances of the reference's own EMD known-answer tests
(gen_abstraction/emd.rs:122-180: test_same, test_emd_66_jt, test_emd_27_aa, tolerance ERROR = 0.01, emd.rs:120), read from the
reference file as DATA (numbers only), plus seeded random cases with the value the CPU oracle gives them (those freeze this
repository's arithmetic; they say nothing about the reference).

    python tests/golden/make_kmeans_emd_golden.py          # needs /root/reference (not present on the GPU box; the JSON is committed)
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402


def vec_literals(text):
    return [[float(x) for x in re.findall(r"[-+]?\d*\.\d+|\d+", body)] for body in re.findall(r"vec!\[(.*?)\]", text, flags=re.S)]


def main():
    src = open("/root/reference/src/gen_abstraction/emd.rs").read()
    tests = src[src.index("mod tests"):]
    vecs = vec_literals(tests)
    values = [float(x) for x in re.findall(r"let actual_value = ([0-9.]+);", tests)]
    assert len(vecs) == 5 and len(values) == 2 and [len(v) for v in vecs] == [8, 30, 30, 30, 30]
    out = {"reference": {"where": "gen_abstraction/emd.rs:122-180", "tolerance": 0.01, "cases": [
        {"name": "test_same", "p": vecs[0], "q": vecs[0], "emd": 0.0, "exact": True},
        {"name": "test_emd_66_jt", "p": vecs[1], "q": vecs[2], "emd": values[0], "exact": False},
        {"name": "test_emd_27_aa", "p": vecs[3], "q": vecs[4], "emd": values[1], "exact": False}]}, "restated": []}
    rng = np.random.Generator(np.random.PCG64(31))
    for n_bins in (1, 2, 8, 20, 30, 35, 64):
        for kind in ("dense", "sparse", "peaked", "zero"):
            p = rng.random(n_bins).astype(np.float32)
            q = rng.random(n_bins).astype(np.float32)
            if kind == "sparse":
                p[rng.random(n_bins) < 0.6] = 0
                q[rng.random(n_bins) < 0.6] = 0
            elif kind == "peaked":
                p = (p ** 8).astype(np.float32)
                q = np.roll(q ** 8, n_bins // 2).astype(np.float32)
            elif kind == "zero":
                q[:] = 0
            out["restated"].append({"p_bits": p.view(np.uint32).tolist(), "q_bits": q.view(np.uint32).tolist(),
                                    "emd_bits": int(np.float32(orc.emd_1d(p, q)).view(np.uint32)),
                                    "l2_bits": int(np.float32(orc.l2_dist(p, q)).view(np.uint32))})
    with open(os.path.join(HERE, "kmeans_emd.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in out["reference"]["cases"]:
        print(c["name"], orc.emd_1d(c["p"], c["q"]), "want", c["emd"])


if __name__ == "__main__":
    main()
