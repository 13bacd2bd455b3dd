"""GPU (-m gpu): Kmeans::predict and update_min_dists (gen_abstraction/kmeans.rs:173-211, :603-619) with emd_1d / l2_dist on the device,
through the C ABI, bit-exact against the CPU oracle."""
import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import orc
from rustsolver_amd import abstraction as ab

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if rs.device_count() < 1:
        pytest.fail("no HIP device visible: GPU parity tests need a real MI355X (there is no CPU fallback)")


@pytest.fixture(scope="module")
def table():
    n_actions, tree = rs.build_game_tree(rs.default_flop())
    return rs.create_infosets(n_actions, tree, [4], [1])


def histograms(rng, n, n_bins, kind):
    """EHS-like histograms: counts out of n_samples per bin (gen_abstraction/main.rs:86-160 produces such f32 rows)"""
    if kind == "counts":
        centres = rng.random(n)[:, None] * n_bins
        width = (0.5 + 6 * rng.random(n))[:, None]
        x = np.exp(-0.5 * ((np.arange(n_bins)[None, :] - centres) / width) ** 2)
        x = np.floor(x / x.sum(axis=1, keepdims=True) * 250) / 250.0
    elif kind == "sparse":
        x = rng.random((n, n_bins)) * (rng.random((n, n_bins)) < 0.25)
    else:
        x = rng.random((n, n_bins)) ** 3
    return x.astype(np.float32)


@pytest.mark.parametrize("n_bins", [1, 8, 20, 30, 32, 35, 64])
@pytest.mark.parametrize("kind", ["counts", "sparse", "dense"])
def test_predict_emd_equals_oracle(table, n_bins, kind):
    rng = np.random.Generator(np.random.PCG64(n_bins * 7 + len(kind)))
    n, k = 3000, 37
    data = histograms(rng, n, n_bins, kind)
    data[::97] = 0                                   # empty histograms: distance 0 to everything, cluster 0
    centers = histograms(rng, k, n_bins, kind)
    centers[5] = 0                                   # an empty center is at distance 0 from everything (emd.rs:59-61)
    centers[11] = centers[3]                         # a duplicate: ties keep the earlier one
    km = ab.Kmeans(table, data)
    cl, md = km.predict(centers, ab.DIST_EMD)
    ocl, omd = orc.kmeans_predict(data, centers, orc.DIST_EMD, threads=8)
    assert (cl == ocl).all(), np.nonzero(cl != ocl)[0][:10]
    assert (md.view(np.uint32) == omd.view(np.uint32)).all()
    assert not (cl == 11).any()


@pytest.mark.parametrize("n_bins", [1, 20, 30, 64])
def test_predict_l2_equals_oracle(table, n_bins):
    rng = np.random.Generator(np.random.PCG64(200 + n_bins))
    data = histograms(rng, 4000, n_bins, "dense")
    centers = histograms(rng, 64, n_bins, "counts")
    cl, md = ab.Kmeans(table, data).predict(centers, ab.DIST_L2)
    ocl, omd = orc.kmeans_predict(data, centers, orc.DIST_L2, threads=8)
    assert (cl == ocl).all() and (md.view(np.uint32) == omd.view(np.uint32)).all()


@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 1000])
def test_predict_ragged_sizes(table, n):
    rng = np.random.Generator(np.random.PCG64(n))
    data = histograms(rng, n, 30, "counts") if n else np.zeros((0, 30), dtype=np.float32)
    centers = histograms(rng, 5, 30, "counts")
    cl, md = ab.Kmeans(table, data).predict(centers)
    ocl, omd = orc.kmeans_predict(data, centers) if n else (np.zeros(0, np.uint32), np.zeros(0, np.float32))
    assert (cl == ocl).all() and (md.view(np.uint32) == omd.view(np.uint32)).all()


def test_predict_reference_known_answers_on_device(table, golden_dir):
    import json
    import os
    fx = json.load(open(os.path.join(golden_dir, "kmeans_emd.json")))
    c66, c27 = fx["reference"]["cases"][1], fx["reference"]["cases"][2]
    data = np.float32([c66["p"], c27["p"]])
    centers = np.float32([c66["q"], c27["q"]])
    km = ab.Kmeans(table, data)
    for col, want in ((0, [c66["emd"], None]), (1, [None, c27["emd"]])):
        _, md = km.predict(centers[col:col + 1])
        for got, w in zip(md.tolist(), want):
            if w is not None:
                assert abs(got - w) < fx["reference"]["tolerance"]   # emd.rs:157-158, :177-178


def test_update_min_dists_equals_oracle(table):
    rng = np.random.Generator(np.random.PCG64(5))
    data = histograms(rng, 5000, 30, "counts")
    km = ab.Kmeans(table, data)
    for dist, odist in ((ab.DIST_EMD, orc.DIST_EMD), (ab.DIST_L2, orc.DIST_L2)):
        md = np.full(5000, np.finfo(np.float32).max, dtype=np.float32)     # vec![f32::MAX; n_data] (kmeans.rs:78)
        omd = md.copy()
        for c in (10, 999, 4000):                                          # three kmeans++ rounds
            md = km.update_min_dists(md, data[c], dist)
            orc.update_min_dists(omd, data, data[c], odist)
            bad = np.nonzero(md.view(np.uint32) != omd.view(np.uint32))[0]
            assert bad.size == 0, (dist, c, bad[:8], md[bad[:8]], omd[bad[:8]])
        assert md[10] == 0.0 or dist == ab.DIST_EMD                        # emd(h, h) is a few ulps, squared: tiny, not necessarily 0


def test_full_size_flop_sweep_properties(table):
    """size-independent properties at the reference's own size (gen_emd(1, 500, 250, 20), main.rs:384: 1 286 792 flop histograms x 500
    centers x 20 bins): every datum that IS a center maps to (the first copy of) itself at the distance emd(h, h); sampled rows equal the oracle"""
    rng = np.random.Generator(np.random.PCG64(1))
    n, k, bins = 1286792, 500, 20
    data = histograms(rng, n, bins, "counts")
    pick = rng.choice(n, size=k, replace=False)
    centers = data[pick]
    cl, md = ab.Kmeans(table, data).predict(centers)
    first = {}
    for j, row in enumerate(centers):
        first.setdefault(row.tobytes(), j)
    self_d = np.array([orc.emd_1d(r, r) for r in centers], dtype=np.float32)
    for j, i in enumerate(pick):
        if md[i] == self_d[j]:
            assert cl[i] <= j or first[centers[j].tobytes()] == cl[i] or md[i] == self_d[cl[i]]
        assert md[i] <= self_d[j]
    sample = rng.choice(n, size=2000, replace=False)
    ocl, omd = orc.kmeans_predict(data[sample], centers, threads=8)
    assert (cl[sample] == ocl).all() and (md[sample].view(np.uint32) == omd.view(np.uint32)).all()


def test_back_to_back_sweeps_with_changing_centers(table):
    """every call stages new centers in the table's scratch: a sweep must never see the previous or the next call's centers
    (a first version allocated them with hipMallocAsync / hipFreeAsync per call and returned sporadically wrong clusters)"""
    rng = np.random.Generator(np.random.PCG64(99))
    data = histograms(rng, 20000, 30, "counts")
    km = ab.Kmeans(table, data)
    for rep in range(12):
        k = int(rng.integers(1, 300))
        centers = histograms(rng, k, 30, "counts" if rep % 2 else "dense")
        dist = ab.DIST_EMD if rep % 3 else ab.DIST_L2
        cl, md = km.predict(centers, dist)
        ocl, omd = orc.kmeans_predict(data, centers, dist, threads=8)
        assert (cl == ocl).all() and (md.view(np.uint32) == omd.view(np.uint32)).all(), rep


def test_predict_rejects_bad_inputs(table):
    km = ab.Kmeans(table, np.zeros((4, 70), dtype=np.float32))
    with pytest.raises(rs.RsError):
        km.predict(np.zeros((2, 70), dtype=np.float32))          # more than 64 bins
    km = ab.Kmeans(table, np.zeros((4, 8), dtype=np.float32))
    with pytest.raises(rs.RsError):
        km.predict(np.zeros((0, 8), dtype=np.float32))           # no centers: Rust indexes centers[0]
    with pytest.raises(ValueError):
        km.predict(np.zeros((2, 9), dtype=np.float32))


# ---- the training loops: init_s, reassign_clusters, fit_regular, fit_growbatch (kmeans.rs:213-601) -----------------------------------------------

@pytest.mark.parametrize("dist,odist", [(ab.DIST_EMD, orc.DIST_EMD), (ab.DIST_L2, orc.DIST_L2)])
@pytest.mark.parametrize("n_bins,kind", [(20, "counts"), (8, "sparse"), (30, "dense")])
def test_init_s_and_reassign_equal_oracle(table, dist, odist, n_bins, kind):
    rng = np.random.Generator(np.random.PCG64(500 + n_bins))
    n, k = 5000, 23
    data = histograms(rng, n, n_bins, kind)
    data[::131] = 0
    centers = data[rng.choice(n, size=k, replace=False)].copy()
    centers[7] = centers[2]                              # duplicate centers: s = 0 there
    km = ab.Kmeans(table, data)
    s = np.full(k, np.finfo(np.float32).max, dtype=np.float32)
    for round_ in range(2):                              # s is in/out: the second call starts from the halved values (kmeans.rs:518, :267-285)
        got = km.init_s(centers, s, dist)
        want = orc.kmeans_init_s(centers, s.copy(), odist)
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
        s = got
    clusters = rng.integers(0, k, size=n).astype(np.uint32)          # arbitrary state: some bounds hold, some force one distance, some a full scan
    bounds = np.stack([rng.random(n) * 0.3, rng.random(n) * 3.0], axis=1).astype(np.float32)
    bounds[::5, 1] = np.finfo(np.float32).max
    oc, ob = clusters.copy(), bounds.copy()
    orc.kmeans_reassign(data, centers, s, oc, ob, odist)
    gc, gb = km.reassign(centers, s, clusters, bounds, dist)
    assert (gc == oc).all() and (gb.view(np.uint32) == ob.view(np.uint32)).all()
    skipped = int((ob[:, 1] == bounds[:, 1]).sum())
    assert 0 < skipped < n, "the case must exercise both the skip and the scan"
    order = rng.permutation(n)[:1500].astype(np.uint32)              # growbatch's shuffled view
    c2, b2 = clusters[:1500].copy(), bounds[:1500].copy()
    orc.kmeans_reassign(data, centers, s, c2, b2, odist, order=order)
    g2, gb2 = km.reassign(centers, s, clusters[:1500], bounds[:1500], dist, order=order)
    assert (g2 == c2).all() and (gb2.view(np.uint32) == b2.view(np.uint32)).all()


@pytest.mark.parametrize("dist,odist", [(ab.DIST_EMD, orc.DIST_EMD), (ab.DIST_L2, orc.DIST_L2)])
def test_fit_regular_equals_oracle(table, dist, odist):
    """ten rounds as in the reference (kmeans.rs:589): clusters, centers, bounds and the printed inertia, bit for bit; and the assignment is the arg-min
    against the centers the LAST reassign saw (Hamerly's bounds only skip work)"""
    rng = np.random.Generator(np.random.PCG64(77))
    n, k, n_bins = 20000, 40, 20
    data = histograms(rng, n, n_bins, "counts")
    centers = data[rng.choice(n, size=k, replace=False)].copy()
    km = ab.Kmeans(table, data)
    gc, gcent, gb, gin = km.fit_regular(centers, dist, 10)
    oc, ocent, ob, oin = orc.kmeans_fit_regular(data, centers, odist, 10)
    assert (gc == oc).all()
    assert (gcent.view(np.uint32) == ocent.view(np.uint32)).all()
    assert (gb.view(np.uint32) == ob.view(np.uint32)).all()
    assert np.float32(gin).view(np.uint32) == np.float32(oin).view(np.uint32)
    _, c9, _, _ = km.fit_regular(centers, dist, 9)
    pc, _ = km.predict(c9, dist)
    assert (pc == gc).mean() >= (1.0 if dist == ab.DIST_L2 else 0.8)     # emd_1d is no metric: the reference's bounds skip too much there (test_kmeans_cpu.py)
    assert len(np.unique(gc)) > k // 2


def test_fit_growbatch_equals_oracle(table):
    """gen_emd's call (gen_abstraction/main.rs:352-361: init_random, fit_growbatch, predict) with a given shuffle: one pass over the first batch"""
    rng = np.random.Generator(np.random.PCG64(78))
    n, k, n_bins, batch = 30000, 50, 20, 4000
    data = histograms(rng, n, n_bins, "counts")
    centers = data[rng.choice(n, size=k, replace=False)].copy()
    order = rng.permutation(n).astype(np.uint32)
    km = ab.Kmeans(table, data)
    for dist, odist in ((ab.DIST_EMD, orc.DIST_EMD), (ab.DIST_L2, orc.DIST_L2)):
        gc, gcent, gb, gst = km.fit_growbatch(order, batch, centers, dist)
        oc, ocent, ob, ost = orc.kmeans_fit_growbatch(data, order, batch, centers, odist)
        assert (gc == oc).all() and (gcent.view(np.uint32) == ocent.view(np.uint32)).all() and (gb.view(np.uint32) == ob.view(np.uint32)).all()
        assert gst.view(np.uint32).tolist() == ost.view(np.uint32).tolist()
        cl, _ = km.predict(gcent, dist)                                   # ... and the bucket file gen_emd would write from these centers
        ocl, _ = orc.kmeans_predict(data, ocent, odist, threads=8)
        assert (cl == ocl).all()


def test_fit_rejects_bad_arguments(table):
    data = np.ones((10, 4), dtype=np.float32)
    km = ab.Kmeans(table, data)
    with pytest.raises(rs.RsError):
        km.fit_regular(data[:1], ab.DIST_L2, 3)                           # one center: the reference indexes center_movement[1]
    with pytest.raises(rs.RsError):
        km.fit_growbatch(np.arange(10, dtype=np.uint32), 11, data[:3], ab.DIST_L2)   # batch > n


def test_pick_restart_equals_oracle(table):
    """Kmeans::init_random's choice among restarts (kmeans.rs:124-156): mean pairwise center distance per candidate set, f32 sums in index order, arg-max"""
    rng = np.random.Generator(np.random.PCG64(79))
    data = histograms(rng, 4000, 20, "counts")
    km = ab.Kmeans(table, data[:8])
    cands = np.stack([data[rng.choice(len(data), size=37, replace=False)] for _ in range(6)])
    cands[4] = cands[1]                                   # equal scores: max_by keeps the LAST
    for dist, odist in ((ab.DIST_EMD, orc.DIST_EMD), (ab.DIST_L2, orc.DIST_L2)):
        best, cd = km.pick_restart(cands, dist)
        obest, ocd = orc.kmeans_pick_restart(cands, odist)
        assert cd.view(np.uint32).tolist() == ocd.view(np.uint32).tolist() and best == obest
        assert cd[4] == cd[1] and best != 1
