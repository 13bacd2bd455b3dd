"""Host mirror of the card-abstraction plumbing (card_abstraction.rs) on top of the C ABI: bucket files, index_to_cluster,
deterministic dense-id maps, the canonical hand index (rust_poker hand_indexer_s) and the per-round card abstractions."""
import ctypes as C

import numpy as np

from . import _lib as L


def read_cluster_file(path):
    """flat LE u32 per canonical hand index (card_abstraction.rs:227-229)"""
    out, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    L.check(L.load().rs_cluster_file_read(path.encode(), C.byref(out), C.byref(n)))
    try:
        return np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
    finally:
        L.load().rs_free_u32(out)


def write_cluster_file(path, clusters):
    """gen_abstraction/main.rs:372-380; refuses to overwrite (create_new)"""
    a = np.ascontiguousarray(clusters, dtype=np.uint32)
    L.check(L.load().rs_cluster_file_write(path.encode(), a.ctypes.data_as(C.POINTER(C.c_uint32)), len(a)))


def index_to_cluster(indices, cluster_arr=None):
    """card_abstraction.rs:20-29"""
    idx = np.ascontiguousarray(indices, dtype=np.uint64)
    out = np.empty(len(idx), dtype=np.uint64)
    if cluster_arr is None:
        rc = L.load().rs_index_to_cluster(None, 0, idx.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    else:
        arr = np.ascontiguousarray(cluster_arr, dtype=np.uint32)
        rc = L.load().rs_index_to_cluster(arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(arr), idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          len(idx), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc == L.ERR_OOB:
        raise IndexError(L.load().rs_last_error().decode())
    L.check(rc)
    return out


class DenseMap:
    """cluster_map[player] + size[player] of generate_maps (card_abstraction.rs:75-184), first-appearance order"""

    def __init__(self, buckets):
        b = np.ascontiguousarray(buckets, dtype=np.uint64)
        h = C.c_void_p()
        L.check(L.load().rs_dense_map_create(b.ctypes.data_as(C.POINTER(C.c_uint64)), len(b), C.byref(h)))
        self._h = h

    def __len__(self):          # get_size
        return L.load().rs_dense_map_size(self._h)

    def lookup(self, buckets):  # get_cluster's map step; KeyError where Rust would unwrap() a None
        b = np.ascontiguousarray(buckets, dtype=np.uint64)
        out = np.empty(len(b), dtype=np.uint32)
        rc = L.load().rs_dense_map_lookup(self._h, b.ctypes.data_as(C.POINTER(C.c_uint64)), len(b), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc == L.ERR_OOB:
            raise KeyError(L.load().rs_last_error().decode())
        L.check(rc)
        return out

    def keys(self):
        out = np.empty(len(self), dtype=np.uint64)
        L.check(L.load().rs_dense_map_keys(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def __del__(self):
        try:
            L.load().rs_dense_map_destroy(self._h)
        except Exception:
            pass


# ---- cards -------------------------------------------------------------------------------------------------------------------
RANKS = "23456789TJQKA"
SUITS = "shdc"   # only the NUMBER of distinct suits matters to anything in this package (suit isomorphism)


def card(text):
    """'As' -> 4*rank + suit (cfr.rs:592, gen_abstraction/ehs.rs:109: 48, 49 is AA)"""
    return 4 * RANKS.index(text[0].upper()) + SUITS.index(text[1].lower())


def card_mask(text):
    """rust_poker::hand_range::get_card_mask("4d5dAs3cKs") (options.rs:57)"""
    m = 0
    for i in range(0, len(text), 2):
        m |= 1 << card(text[i:i + 2])
    return m


def random_range(board_mask=0):
    """HandRange::from_string("random") after remove_invalid_combos(board_mask) (options.rs:63-64, cfr.rs:161-163):
    every two-card combo that avoids the board, as a uint8 [n][2] array"""
    return np.array([(a, b) for a in range(52) for b in range(a) if not ((board_mask >> a) & 1 or (board_mask >> b) & 1)], dtype=np.uint8)


class HandIndexer:
    """rust_poker::hand_indexer_s: init(rounds, cards_per_round) / size / get_index / get_hand"""

    def __init__(self, cards_per_round):
        cpr = (C.c_uint8 * len(cards_per_round))(*cards_per_round)
        h = C.c_void_p()
        L.check(L.load().rs_hand_indexer_create(len(cards_per_round), cpr, C.byref(h)))
        self._h = h
        self.cards_per_round = tuple(cards_per_round)
        self.rounds = len(cards_per_round)

    def size(self, round_):
        return int(L.load().rs_hand_indexer_size(self._h, round_))

    def n_cards(self, round_):
        return sum(self.cards_per_round[: round_ + 1])

    def get_index(self, cards, round_=None):
        """cards: uint8 [n][n_cards(round)] (or one hand) -> uint64 [n]; round_ defaults to the last, like get_index"""
        r = self.rounds - 1 if round_ is None else round_
        c = np.ascontiguousarray(cards, dtype=np.uint8)
        one = c.ndim == 1
        c = c.reshape(-1, c.shape[-1])
        if c.shape[1] < self.n_cards(r):
            raise ValueError("a hand of round %d has %d cards" % (r, self.n_cards(r)))
        c = np.ascontiguousarray(c[:, : self.n_cards(r)])   # like the reference, extra cards are ignored (card_abstraction.rs:319)
        out = np.empty(len(c), dtype=np.uint64)
        L.check(L.load().rs_hand_index(self._h, r, c.ctypes.data, len(c), out.ctypes.data))
        return int(out[0]) if one else out

    def get_hand(self, round_, indices):
        idx = np.ascontiguousarray(np.atleast_1d(indices), dtype=np.uint64)
        out = np.empty((len(idx), self.n_cards(round_)), dtype=np.uint8)
        rc = L.load().rs_hand_unindex(self._h, round_, idx.ctypes.data, len(idx), out.ctypes.data)
        if rc == L.ERR_OOB:
            raise IndexError(L.load().rs_last_error().decode())
        L.check(rc)
        return out

    def verify(self, cards, expect, round_=None):
        """rs_hand_index_verify: `expect` = the indices ANOTHER indexer (rust_poker's) gave for `cards`; returns (first_bad, got) -- (n, None) when all agree"""
        r = self.rounds - 1 if round_ is None else round_
        c = np.ascontiguousarray(cards, dtype=np.uint8)
        c = np.ascontiguousarray(c.reshape(-1, c.shape[-1])[:, : self.n_cards(r)])
        e = np.ascontiguousarray(expect, dtype=np.uint64)
        if len(e) != len(c):
            raise ValueError("one expected index per hand")
        bad, got = C.c_size_t(), C.c_uint64()
        rc = L.load().rs_hand_index_verify(self._h, r, c.ctypes.data, len(c), e.ctypes.data, C.byref(bad), C.byref(got))
        if rc == L.ERR_MISMATCH:
            return int(bad.value), int(got.value)
        L.check(rc)
        return len(c), None

    def get_index_device(self, table, cards, round_=None):
        """same on the GPU: cards uint8 [n][n_cards] -> uint64 [n] (uploads SoA rows, downloads the indices)"""
        from .solver import DeviceBuffer, deal_pitch
        r = self.rounds - 1 if round_ is None else round_
        c = np.ascontiguousarray(cards, dtype=np.uint8)
        n, nc = c.shape[0], self.n_cards(r)
        pitch = deal_pitch(n)
        host = np.zeros((nc, pitch), dtype=np.uint8)
        host[:, :n] = c[:, :nc].T
        dc = DeviceBuffer.from_numpy(table, host)
        do = DeviceBuffer(table, max(pitch, 1) * 8)
        L.check(L.load().rs_hand_index_device(table._h, self._h, r, dc.ptr, n, do.ptr))
        return do.download(np.uint64, pitch)[:n]

    def __del__(self):
        try:
            L.load().rs_hand_indexer_destroy(self._h)
        except Exception:
            pass


FLOP, TURN, RIVER = 0, 1, 2   # BettingRound (state.rs)


class CardAbstraction:
    """One round's ICardAbstraction (card_abstraction.rs:31-298): `ISOMORPHIC.init` / `EMD.init` / `OCHS.init` are
    `CardAbstraction.init(..., cluster_arr=None | the bucket file)`; get_cluster / get_size as in the reference."""

    def __init__(self, hand_ranges, initial_board_mask, round_, cluster_arr=None):
        h0 = np.ascontiguousarray(hand_ranges[0], dtype=np.uint8).reshape(-1, 2)
        h1 = np.ascontiguousarray(hand_ranges[1], dtype=np.uint8).reshape(-1, 2)
        arr = None if cluster_arr is None else np.ascontiguousarray(cluster_arr, dtype=np.uint32)
        h = C.c_void_p()
        rc = L.load().rs_card_abs_create(round_, h0.ctypes.data, len(h0), h1.ctypes.data, len(h1), initial_board_mask,
                                         None if arr is None else arr.ctypes.data, 0 if arr is None else len(arr), C.byref(h))
        if rc == L.ERR_OOB:
            raise IndexError(L.load().rs_last_error().decode())
        L.check(rc)
        self._h = h
        self.round = round_
        self.n_cards = 5 + round_

    init = classmethod(lambda cls, hand_ranges, board_mask, round_, cluster_arr=None: cls(hand_ranges, board_mask, round_, cluster_arr))

    def get_size(self, player):
        return int(L.load().rs_card_abs_size(self._h, player))

    def index_size(self):
        return int(L.load().rs_card_abs_index_size(self._h))

    def keys(self, player):
        out = np.empty(self.get_size(player), dtype=np.uint64)
        L.check(L.load().rs_card_abs_keys(self._h, player, out.ctypes.data))
        return out

    def get_cluster(self, cards, player):
        """cards: the player's two hole cards then the board (cfr.rs:357-365), one hand or [n][>= 5 + round]"""
        c = np.ascontiguousarray(cards, dtype=np.uint8)
        one = c.ndim == 1
        c = np.ascontiguousarray(c.reshape(-1, c.shape[-1])[:, : self.n_cards])
        out = np.empty(len(c), dtype=np.uint32)
        rc = L.load().rs_card_abs_get_cluster(self._h, c.ctypes.data, len(c), player, out.ctypes.data)
        if rc == L.ERR_OOB:
            raise KeyError(L.load().rs_last_error().decode())
        L.check(rc)
        return int(out[0]) if one else out

    def get_clusters_device(self, table, cards9):
        """cards9: uint8 [9][n] deal matrix (board x5, P0 x2, P1 x2) -> (ids of player 0, ids of player 1), computed on the GPU"""
        from .solver import DeviceBuffer, deal_pitch
        c = np.ascontiguousarray(cards9, dtype=np.uint8)
        n = c.shape[1]
        pitch = deal_pitch(n)
        host = np.zeros((9, pitch), dtype=np.uint8)
        host[:, :n] = c
        dc = DeviceBuffer.from_numpy(table, host)
        d0, d1 = DeviceBuffer(table, max(pitch, 1) * 4), DeviceBuffer(table, max(pitch, 1) * 4)
        L.check(L.load().rs_card_abs_clusters_device(self._h, table._h, dc.ptr, n, d0.ptr, d1.ptr))
        rc = L.load().rs_card_abs_status(self._h, table._h)
        if rc == L.ERR_OOB:
            raise KeyError(L.load().rs_last_error().decode())
        L.check(rc)
        return d0.download(np.uint32, pitch)[:n], d1.download(np.uint32, pitch)[:n]

    def destroy(self):
        if self._h:
            L.load().rs_card_abs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def sample_deals(table, seed, first_deal, board_mask, hand_ranges, n_deals):
    """generate_hand (cfr.rs:100-143) for n_deals deals on the GPU -> uint8 [9][n_deals]"""
    from .solver import DeviceBuffer, deal_pitch
    h0 = np.ascontiguousarray(hand_ranges[0], dtype=np.uint8).reshape(-1, 2)
    h1 = np.ascontiguousarray(hand_ranges[1], dtype=np.uint8).reshape(-1, 2)
    d0, d1 = DeviceBuffer.from_numpy(table, h0), DeviceBuffer.from_numpy(table, h1)
    pitch = deal_pitch(n_deals)
    dc = DeviceBuffer(table, 9 * max(pitch, 1))
    dc.zero()
    de = DeviceBuffer(table, 4)
    de.zero()
    L.check(L.load().rs_deals_sample(table._h, seed, first_deal, board_mask, d0.ptr, len(h0), d1.ptr, len(h1), n_deals, dc.ptr, de.ptr))
    err = int(de.download(np.uint32, 1)[0])
    if err:
        raise RuntimeError("generate_hand: no valid combo for some deal (error word %d)" % err)
    return dc.download(np.uint8, 9 * pitch).reshape(9, pitch)[:, :n_deals]


# ---- abstraction generator's distance sweep (gen_abstraction/kmeans.rs, emd.rs) -----------------------------------------------------
DIST_EMD, DIST_L2 = L.DIST_EMD, L.DIST_L2


def histogram_distance(p, q, dist=DIST_EMD):
    """emd::emd_1d (emd.rs:53-113) or kmeans::l2_dist (kmeans.rs:622-630) of two histograms, on the host"""
    a, b = np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)
    if a.shape != b.shape or a.ndim != 1:
        raise ValueError("two histograms of the same length")
    out = C.c_float()
    L.check(L.load().rs_histogram_distance(dist, a.ctypes.data, b.ctypes.data, len(a), C.byref(out)))
    return np.float32(out.value)


class Kmeans:
    """The distance sweeps of gen_abstraction/kmeans.rs on the GPU: `predict` (kmeans.rs:173-211) and the kmeans++ `update_min_dists`
    (kmeans.rs:603-619).  The dataset stays resident on the device between calls."""

    def __init__(self, table, dataset):
        from .solver import DeviceBuffer
        d = np.ascontiguousarray(dataset, dtype=np.float32)
        if d.ndim != 2:
            raise ValueError("dataset is [n][n_bins]")
        self.table, self.n, self.n_bins = table, d.shape[0], d.shape[1]
        self._data = DeviceBuffer.from_numpy(table, d) if d.size else None
        self._DeviceBuffer = DeviceBuffer

    def predict(self, centers, dist=DIST_EMD):
        """-> (clusters uint32 [n], distance to the chosen center float32 [n])"""
        c = np.ascontiguousarray(centers, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] != self.n_bins:
            raise ValueError("centers are [k][n_bins]")
        dc = self._DeviceBuffer(self.table, max(self.n, 1) * 4)
        dm = self._DeviceBuffer(self.table, max(self.n, 1) * 4)
        L.check(L.load().rs_kmeans_predict(self.table._h, dist, self._data.ptr if self._data else None, self.n, c.ctypes.data, len(c), self.n_bins,
                                           dc.ptr, dm.ptr))
        return dc.download(np.uint32, self.n), dm.download(np.float32, self.n)

    def update_min_dists(self, min_dists, new_center, dist=DIST_EMD):
        m = np.ascontiguousarray(min_dists, dtype=np.float32)
        c = np.ascontiguousarray(new_center, dtype=np.float32)
        dm = self._DeviceBuffer.from_numpy(self.table, m) if self.n else None
        L.check(L.load().rs_update_min_dists(self.table._h, dist, dm.ptr if dm else None, self._data.ptr if self._data else None, self.n, c.ctypes.data,
                                             self.n_bins))
        return dm.download(np.float32, self.n) if self.n else m

    # ---- the training loops (kmeans.rs:213-601) ------------------------------------------------------------------------------------------
    def init_s(self, centers, s, dist=DIST_EMD):
        """Kmeans::init_s (kmeans.rs:267-285): returns the updated s (float32 [k]); s is in/out in the reference"""
        c = np.ascontiguousarray(centers, dtype=np.float32)
        out = np.array(s, dtype=np.float32, order="C")
        L.check(L.load().rs_kmeans_init_s(self.table._h, dist, c.ctypes.data, len(c), self.n_bins, out.ctypes.data))
        return out

    def pick_restart(self, candidates, dist=DIST_EMD):
        """Kmeans::init_random's scoring (kmeans.rs:124-156): candidates [n_restarts][k][n_bins] -> (index of the most spread set, mean pairwise distances)"""
        import ctypes as C
        c = np.ascontiguousarray(candidates, dtype=np.float32)
        cd = np.zeros(c.shape[0], dtype=np.float32)
        best = C.c_int()
        L.check(L.load().rs_kmeans_pick_restart(self.table._h, dist, c.ctypes.data, c.shape[0], c.shape[1], c.shape[2], cd.ctypes.data, C.byref(best)))
        return best.value, cd

    def reassign(self, centers, s, clusters, bounds, dist=DIST_EMD, order=None):
        """Kmeans::reassign_clusters (kmeans.rs:287-334) -> (clusters uint32 [m], bounds float32 [m][2]); m = len(clusters) items, item i = dataset[order[i]]"""
        c = np.ascontiguousarray(centers, dtype=np.float32)
        s_ = np.ascontiguousarray(s, dtype=np.float32)
        dc = self._DeviceBuffer.from_numpy(self.table, np.ascontiguousarray(clusters, dtype=np.uint32))
        db = self._DeviceBuffer.from_numpy(self.table, np.ascontiguousarray(bounds, dtype=np.float32))
        do = None if order is None else self._DeviceBuffer.from_numpy(self.table, np.ascontiguousarray(order, dtype=np.uint32))
        m = len(clusters)
        L.check(L.load().rs_kmeans_reassign(self.table._h, dist, self._data.ptr, do.ptr if do else None, m, c.ctypes.data, len(c), self.n_bins, s_.ctypes.data,
                                            dc.ptr, db.ptr))
        return dc.download(np.uint32, m), db.download(np.float32, 2 * m).reshape(m, 2)

    def fit_regular(self, centers, dist=DIST_EMD, iterations=10):
        """Kmeans::fit_regular (kmeans.rs:497-600) -> (clusters uint32 [n], new centers [k][n_bins], bounds [n][2], inertia)"""
        import ctypes as C
        c = np.array(centers, dtype=np.float32, order="C")
        dc = self._DeviceBuffer(self.table, self.n * 4)
        db = self._DeviceBuffer(self.table, self.n * 8)
        inertia = C.c_float()
        L.check(L.load().rs_kmeans_fit_regular(self.table._h, dist, self._data.ptr, self.n, c.ctypes.data, len(c), self.n_bins, iterations, dc.ptr, db.ptr,
                                               C.byref(inertia)))
        return dc.download(np.uint32, self.n), c, db.download(np.float32, 2 * self.n).reshape(self.n, 2), np.float32(inertia.value)

    def fit_growbatch(self, order, batch, centers, dist=DIST_EMD):
        """Kmeans::fit_growbatch as coded (kmeans.rs:336-495) -> (clusters [batch], new centers, bounds [batch][2], stats = (p, inertia))"""
        c = np.array(centers, dtype=np.float32, order="C")
        do = self._DeviceBuffer.from_numpy(self.table, np.ascontiguousarray(order, dtype=np.uint32))
        dc = self._DeviceBuffer(self.table, batch * 4)
        db = self._DeviceBuffer(self.table, batch * 8)
        stats = np.zeros(2, dtype=np.float32)
        L.check(L.load().rs_kmeans_fit_growbatch(self.table._h, dist, self._data.ptr, self.n, do.ptr, batch, c.ctypes.data, len(c), self.n_bins, dc.ptr, db.ptr,
                                                 stats.ctypes.data))
        return dc.download(np.uint32, batch), c, db.download(np.float32, 2 * batch).reshape(batch, 2), stats
