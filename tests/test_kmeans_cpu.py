"""CPU (-m "not gpu"): the abstraction generator's distance functions.

  * the oracle (oracle/kmeans_emd.c, a literal restatement of emd.rs:53-113 / kmeans.rs:622-630) against the reference's OWN known answers
    (gen_abstraction/emd.rs:122-180, tolerance ERROR = 0.01 from emd.rs:120);
  * the product's host-side distance (rs_histogram_distance: the bitmask formulation the GPU kernel uses) against the oracle, bit for bit."""
import json
import os

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import abstraction as ab


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "kmeans_emd.json")))


def f32bits(x):
    return int(np.float32(x).view(np.uint32))


def test_oracle_meets_the_reference_known_answers(fx):
    tol = fx["reference"]["tolerance"]
    for c in fx["reference"]["cases"]:
        got = float(orc.emd_1d(c["p"], c["q"]))
        assert abs(got - c["emd"]) < tol, (c["name"], got, c["emd"])   # assert!(emd < actual + ERROR && emd > actual - ERROR)


def test_emd_of_a_histogram_with_itself_is_three_ulps_not_zero(fx):
    """emd.rs:136 asserts emd(h, h) == 0.0 inside a #[bench].  Evaluated in f32 as written, w = sum of the normalised bins = 1 - 2^-24 for that
    histogram, u = 3, so the function returns (1 - w) * u = 3 * 2^-24: far inside ERROR = 0.01, not equal to 0.0 (numpy float32 agrees)."""
    c = fx["reference"]["cases"][0]
    p = np.array(c["p"], dtype=np.float32)
    s = np.float32(0)
    for x in p:
        s = np.float32(s + x)
    w = np.float32(0)
    for x in p:
        w = np.float32(w + np.float32(x / s))
    assert w == np.float32(1) - np.float32(2.0 ** -24)
    assert orc.emd_1d(p, p) == np.float32(3 * 2.0 ** -24)


def test_oracle_and_host_restated_fixture(fx):
    for c in fx["restated"]:
        p = np.array(c["p_bits"], dtype=np.uint32).view(np.float32)
        q = np.array(c["q_bits"], dtype=np.uint32).view(np.float32)
        assert f32bits(orc.emd_1d(p, q)) == c["emd_bits"] and f32bits(orc.l2_dist(p, q)) == c["l2_bits"]
        assert f32bits(ab.histogram_distance(p, q)) == c["emd_bits"]
        assert f32bits(ab.histogram_distance(p, q, ab.DIST_L2)) == c["l2_bits"]
    for c in fx["reference"]["cases"]:
        assert f32bits(ab.histogram_distance(c["p"], c["q"])) == f32bits(orc.emd_1d(c["p"], c["q"]))


def test_host_distance_equals_oracle_on_fuzzed_histograms():
    """the bitmask formulation (at most n_bins transfers) against the literal 2(u-1) x n_bins probe loop"""
    rng = np.random.Generator(np.random.PCG64(17))
    for _ in range(6000):
        n = int(rng.integers(1, 65))
        p = (rng.random(n) ** rng.integers(1, 7)).astype(np.float32)
        q = (rng.random(n) ** rng.integers(1, 7)).astype(np.float32)
        if rng.random() < 0.3:
            p[rng.random(n) < 0.5] = 0
        if rng.random() < 0.3:
            q[rng.random(n) < 0.5] = 0
        if rng.random() < 0.1:
            q = np.roll(p, int(rng.integers(0, n)))       # same mass, shifted: pure cross-bin work
        assert f32bits(ab.histogram_distance(p, q)) == f32bits(orc.emd_1d(p, q)), (n, p, q)
        assert f32bits(ab.histogram_distance(p, q, ab.DIST_L2)) == f32bits(orc.l2_dist(p, q))


def test_emd_edge_cases():
    z = np.zeros(8, dtype=np.float32)
    h = np.arange(8, dtype=np.float32)
    for f in (orc.emd_1d, ab.histogram_distance):
        assert f(z, h) == 0.0 and f(h, z) == 0.0 and f(z, z) == 0.0          # emd.rs:59-61
        assert f(np.float32([1]), np.float32([5])) == 0.0                      # one bin: all mass matches
    a = np.float32([1, 0, 0, 0]); b = np.float32([0, 0, 0, 1])
    assert orc.emd_1d(a, b) == ab.histogram_distance(a, b) == np.float32(3.0)  # u = 4: offset 3 is reachable, cost 1 * 3
    with pytest.raises(rs.RsError):
        ab.histogram_distance(np.zeros(65, np.float32), np.zeros(65, np.float32))
    with pytest.raises(ValueError):
        ab.histogram_distance(np.zeros(4, np.float32), np.zeros(5, np.float32))


def test_oracle_predict_takes_the_first_strict_minimum():
    """kmeans.rs:194-202: `if variance[k] < min_variance` -- ties keep the earlier center"""
    data = np.float32([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0]])
    centers = np.float32([[0, 1, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0]])
    cl, md = orc.kmeans_predict(data, centers, orc.DIST_EMD)
    assert cl.tolist() == [1, 3, 0] and md.tolist() == [0.0, 1.0, 0.0]
    cl2, _ = orc.kmeans_predict(data, centers, orc.DIST_L2, threads=3)
    assert cl2.tolist() == [1, 0, 0]      # [0,0,0,1] and the empty histogram are equidistant from every center: the first one wins


# ---- the oracle's training loops (oracle/kmeans_fit.c): what they are pinned by, since the reference holds no test for them ---------------------

def _hist(rng, n, n_bins):
    centres = rng.random(n)[:, None] * n_bins
    width = (0.5 + 6 * rng.random(n))[:, None]
    x = np.exp(-0.5 * ((np.arange(n_bins)[None, :] - centres) / width) ** 2)
    return (np.floor(x / x.sum(axis=1, keepdims=True) * 250) / 250.0).astype(np.float32)


def test_oracle_hamerly_assignment_is_the_brute_force_assignment():
    """reassign_clusters only SKIPS distance evaluations (kmeans.rs:287-334): after every round of fit_regular the clusters must be Kmeans::predict's arg-min against
    the centers that round's reassign saw.  Exact for l2_dist (a metric).  emd_1d is a heuristic WITHOUT a triangle inequality, so with it the reference's own
    bounds skip evaluations they should not (about one datum in ten changes cluster here): for EMD the check is only that most assignments agree -- the
    disagreement is the reference's algorithm, reproduced, not an error of the restatement (the GPU path must match the oracle bit for bit either way)."""
    rng = np.random.Generator(np.random.PCG64(5))
    data = _hist(rng, 4000, 20)
    centers = data[rng.choice(len(data), size=25, replace=False)].copy()
    for kind, floor in ((orc.DIST_L2, 1.0), (orc.DIST_EMD, 0.8)):
        prev = centers
        for it in range(1, 6 if kind == orc.DIST_L2 else 3):   # EMD: a cluster that runs empty becomes the zero histogram, at emd distance 0 from everything
            cl, cent, bounds, inertia = orc.kmeans_fit_regular(data, centers, kind, it)
            want, _ = orc.kmeans_predict(data, prev, kind)
            assert (cl == want).mean() >= floor, (kind, it, (cl == want).mean())
            assert (bounds[:, 1] >= 0).all()
            prev = cent


def test_oracle_fit_regular_first_round_by_hand():
    """one round from scratch is plain k-means: every bound starts at (0, f32::MAX), so every datum scans all centers; means are sums in data order / counts,
    bins that sum to 0 stay 0 (kmeans.rs:537), and the centers come back as those means"""
    rng = np.random.Generator(np.random.PCG64(6))
    data = _hist(rng, 500, 8)
    centers = data[:5].copy()
    cl, cent, bounds, inertia = orc.kmeans_fit_regular(data, centers, orc.DIST_L2, 1)
    want_cl, want_d = orc.kmeans_predict(data, centers, orc.DIST_L2)
    assert (cl == want_cl).all()
    for j in range(5):
        acc = np.zeros(8, dtype=np.float32)
        cnt = np.float32(0)
        for i in np.nonzero(cl == j)[0]:                      # ascending data order, f32
            acc = (acc + data[i]).astype(np.float32)
            cnt = np.float32(cnt + np.float32(1))
        mean = np.where(acc > 0, (acc / cnt).astype(np.float32), acc)
        assert mean.view(np.uint32).tolist() == cent[j].view(np.uint32).tolist()
    mv = np.array([orc.l2_dist(cent[j], centers[j]) for j in range(5)], dtype=np.float32)
    assert (bounds[:, 1].view(np.uint32) == (want_d + mv[cl]).astype(np.float32).view(np.uint32)).all()   # upper = distance + own center's movement


def test_oracle_init_s_is_in_out_as_coded():
    """kmeans.rs:518 creates s once; init_s (:267-285) lowers and halves it, so a second call on unchanged centers halves it AGAIN"""
    rng = np.random.Generator(np.random.PCG64(7))
    centers = _hist(rng, 6, 10)
    s = np.full(6, np.finfo(np.float32).max, dtype=np.float32)
    orc.kmeans_init_s(centers, s, orc.DIST_L2)
    want = np.array([min(orc.l2_dist(centers[i], centers[j]) for j in range(6) if j != i) / np.float32(2) for i in range(6)], dtype=np.float32)
    assert s.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    orc.kmeans_init_s(centers, s, orc.DIST_L2)
    assert s.view(np.uint32).tolist() == (want / np.float32(2)).astype(np.float32).view(np.uint32).tolist()


def test_oracle_growbatch_is_one_pass_over_the_batch():
    rng = np.random.Generator(np.random.PCG64(8))
    data = _hist(rng, 3000, 12)
    centers = data[rng.choice(3000, size=9, replace=False)].copy()
    order = rng.permutation(3000).astype(np.uint32)
    cl, cent, bounds, stats = orc.kmeans_fit_growbatch(data, order, 700, centers, orc.DIST_L2)
    want, _ = orc.kmeans_predict(data[order[:700]], centers, orc.DIST_L2)
    assert (cl == want).all()                                  # first round: a datum keeps cluster 0 only if it is within s[0] of it, else it scans every center
    assert np.isfinite(stats).all() and stats[1] > 0
    for j in range(9):
        if (cl == j).sum() == 0:
            assert (cent[j] == 0).all()                        # empty cluster: the mean of nothing is the zero histogram (`count > 0.0`, kmeans.rs:412)


def test_oracle_pick_restart_by_hand():
    """three candidate sets of three centers on a line (l2_dist): the mean pairwise distance by hand, the arg-max, and the last-of-equal-maxima rule of max_by"""
    def line(xs):
        return np.array([[x, 0.0] for x in xs], dtype=np.float32)
    cands = np.stack([line([0, 1, 2]), line([0, 3, 6]), line([1, 4, 7])])     # mean pairwise distances 4/3, 4, 4
    best, cd = orc.kmeans_pick_restart(cands, orc.DIST_L2)
    assert np.allclose(cd, [8.0 / 6.0, 4.0, 4.0]) and best == 2
