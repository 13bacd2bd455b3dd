"""Host-side mirror of the reference solver's surface (src/solver) on top of the C ABI.

Names follow the Rust items they stand for so that tests read like tests of the reference:

    Options / default_flop()          options.rs:10-28, :52-81
    build_game_tree(options)          tree_builder.rs:9   -> (n_actions, tree)
    create_infosets(n_actions, tree, card_abs)   infoset.rs:8   -> InfosetTable
    table[an.index][cluster_idx]      cfr.rs:375 (get-infoset) -> Infoset view
    Infoset.get_strategy() / .get_final_strategy()   infoset.rs:83 / :104
    MCCFRTrainer.init(options, ...) / .train(iterations)   cfr.rs:159 / :188

All compute runs in the HIP library; this file only marshals numpy arrays through ctypes.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Options:
    """options.rs:10-28 (the fields build_game_tree reads)."""

    def __init__(self, stack_sizes=(500, 500), starting_pot=35, n_board_cards=5, bet_sizes=((0.5, 1.0),),
                 raise_sizes=((3.0,),)):
        self.stack_sizes = tuple(stack_sizes)
        self.starting_pot = starting_pot
        self.n_board_cards = n_board_cards  # board_mask.count_ones()
        self.bet_sizes = [list(b) for b in bet_sizes]
        self.raise_sizes = [list(r) for r in raise_sizes]

    def to_c(self):
        o = L.OptionsC()
        o.stack_sizes[0], o.stack_sizes[1] = self.stack_sizes
        o.starting_pot = self.starting_pot
        o.n_board_cards = self.n_board_cards
        o.n_rounds = len(self.bet_sizes)
        for r, (bs, rs) in enumerate(zip(self.bet_sizes, self.raise_sizes)):
            o.n_bet_sizes[r] = len(bs)
            for i, v in enumerate(bs):
                o.bet_sizes[r][i] = v
            o.n_raise_sizes[r] = len(rs)
            for i, v in enumerate(rs):
                o.raise_sizes[r][i] = v
        return o


def default_flop():
    """options::default_flop() (options.rs:52-81): despite the name, a 5-card (river) board."""
    o = L.OptionsC()
    L.check(L.load().rs_options_default(C.byref(o)))
    return Options((o.stack_sizes[0], o.stack_sizes[1]), o.starting_pot, o.n_board_cards,
                   [[o.bet_sizes[r][i] for i in range(o.n_bet_sizes[r])] for r in range(o.n_rounds)],
                   [[o.raise_sizes[r][i] for i in range(o.n_raise_sizes[r])] for r in range(o.n_rounds)])


def three_street_options():
    """options.rs:68-77 commented vectors (minus the 2.0 river bet) from a 3-card board."""
    return Options(n_board_cards=3, bet_sizes=((0.5, 1.0),) * 3, raise_sizes=((3.0,),) * 3)


class GameTree:
    """Tree<GameTreeNode> (tree.rs:14-17), host resident."""

    def __init__(self, handle):
        self._h = handle
        lib = L.load()
        self.n_nodes = lib.rs_tree_n_nodes(handle)
        self.n_action_nodes = lib.rs_tree_n_action_nodes(handle)
        self._nodes = None

    def get_node(self, idx):
        nd = L.TreeNode()
        L.check(L.load().rs_tree_get_node(self._h, idx, C.byref(nd)))
        return nd

    @property
    def nodes(self):
        if self._nodes is None:
            self._nodes = [self.get_node(i) for i in range(self.n_nodes)]
        return self._nodes

    def action_nodes(self):
        """ActionNodes ordered by .index"""
        out = [None] * self.n_action_nodes
        for nd in self.nodes:
            if nd.kind == L.NODE_ACTION:
                out[nd.index] = nd
        return out

    def __del__(self):
        try:
            L.load().rs_tree_destroy(self._h)
        except Exception:
            pass


def build_game_tree(options):
    """tree_builder.rs:9 -> (n_actions, tree)."""
    h = C.c_void_p()
    oc = options.to_c()
    L.check(L.load().rs_tree_build(C.byref(oc), C.byref(h)))
    t = GameTree(h)
    return t.n_action_nodes, t


def tree_from_nodes(nodes):
    arr = (L.TreeNode * len(nodes))(*nodes)
    h = C.c_void_p()
    L.check(L.load().rs_tree_from_nodes(arr, len(nodes), C.byref(h)))
    return GameTree(h)


class DeviceBuffer:
    """A device allocation owned through the table's stream (rs_dmalloc / rs_dfree)."""

    def __init__(self, table, nbytes):
        self.table = table
        self.nbytes = nbytes
        p = C.c_void_p()
        L.check(L.load().rs_dmalloc(table._h, nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, table, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(table, arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        L.check(L.load().rs_h2d(self.table._h, self.ptr, _vp(arr), arr.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        L.check(L.load().rs_d2h(self.table._h, _vp(out), self.ptr, out.nbytes))
        return out

    def zero(self):
        L.check(L.load().rs_dmemset(self.table._h, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            L.load().rs_dfree(self.table._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Infoset:
    """View of one info set: `&self.infosets[an.index][cluster_idx]` (cfr.rs:375, infoset.rs:63-67)."""

    def __init__(self, table, node, board, cluster):
        self._t, self._n, self._b, self._c = table, node, board, cluster
        self._a = table.node_desc(node).n_actions
        self._np = np.int32 if table.dtype == L.I32 else np.float32

    def _get(self):
        r = np.empty(self._a, dtype=self._np)
        s = np.empty(self._a, dtype=self._np)
        rc = L.load().rs_get_infoset(self._t._h, self._n, self._b, self._c, _vp(r), _vp(s))
        if rc == L.ERR_OOB:
            raise IndexError(L.load().rs_last_error().decode())
        L.check(rc)
        return r, s

    @property
    def regrets(self):
        return self._get()[0]

    @property
    def strategy_sum(self):
        return self._get()[1]

    def set(self, regrets=None, strategy_sum=None):
        r = None if regrets is None else np.ascontiguousarray(regrets, dtype=self._np)
        s = None if strategy_sum is None else np.ascontiguousarray(strategy_sum, dtype=self._np)
        rc = L.load().rs_set_infoset(self._t._h, self._n, self._b, self._c, None if r is None else _vp(r),
                                     None if s is None else _vp(s))
        if rc == L.ERR_OOB:
            raise IndexError(L.load().rs_last_error().decode())
        L.check(rc)

    def _strategy(self, fn):
        out = np.empty(self._a, dtype=np.float32)
        rc = fn(self._t._h, self._n, self._b, self._c, out.ctypes.data_as(C.POINTER(C.c_float)))
        if rc == L.ERR_OOB:
            raise IndexError(L.load().rs_last_error().decode())
        L.check(rc)
        return out

    def get_strategy(self):
        """infoset.rs:83-102"""
        return self._strategy(L.load().rs_get_strategy)

    def get_final_strategy(self):
        """infoset.rs:104-123"""
        return self._strategy(L.load().rs_get_final_strategy)


class _Row:
    def __init__(self, table, node):
        self._t, self._n = table, node

    def __len__(self):
        d = self._t.node_desc(self._n)
        return d.n_boards * d.n_clusters

    def __getitem__(self, key):
        """row[cluster_idx] (board 0) or row[board, cluster_idx]"""
        d = self._t.node_desc(self._n)
        board, cluster = key if isinstance(key, tuple) else (0, key)
        if not (0 <= cluster < d.n_clusters and 0 <= board < d.n_boards):
            raise IndexError("index out of bounds: the len is %d but the index is %d" % (d.n_clusters, cluster))
        return Infoset(self._t, self._n, board, cluster)


class InfosetTable:
    """`InfosetTable = Vec<Vec<Infoset>>` (infoset.rs:6), resident in HBM."""

    def __init__(self, handle, owned=True):
        self._h = handle
        self._owned = owned   # False: a view of a table that a native object (rs_deal_trainer) owns
        lib = L.load()
        self.n_nodes = lib.rs_table_n_nodes(handle)
        self.dtype = lib.rs_table_dtype(handle)
        self.np_dtype = np.int32 if self.dtype == L.I32 else np.float32
        self._descs = {}
        self._solvers = weakref.WeakSet()   # solvers built on this table must be destroyed before it

    @classmethod
    def create(cls, descs, dtype=L.I32, device=0):
        """descs: list of (n_actions, n_clusters, n_boards, player, round_idx) in ActionNode.index order"""
        arr = (L.NodeDesc * len(descs))()
        for i, (a, c, b, p, r) in enumerate(descs):
            arr[i].n_actions, arr[i].n_clusters, arr[i].n_boards, arr[i].player, arr[i].round_idx = a, c, b, p, r
        h = C.c_void_p()
        L.check(L.load().rs_table_create(arr, len(descs), dtype, device, C.byref(h)))
        return cls(h)

    def __len__(self):
        return self.n_nodes

    def __getitem__(self, node):
        if not 0 <= node < self.n_nodes:
            raise IndexError("index out of bounds: the len is %d but the index is %d" % (self.n_nodes, node))
        return _Row(self, node)

    def node_desc(self, node):
        if node not in self._descs:
            d = L.NodeDesc()
            L.check(L.load().rs_table_node_desc(self._h, node, C.byref(d)))
            self._descs[node] = d
        return self._descs[node]

    def lanes(self, node):
        d = self.node_desc(node)
        return d.n_boards * d.n_clusters

    def pitch(self, node):
        return L.load().rs_table_lane_pitch(self._h, node)

    def tile_lanes(self, node):
        """lanes per tile of the node's block ([pitch / T][A][T]); == pitch(node): the plain [A][pitch] block"""
        return int(L.load().rs_table_tile_lanes(self._h, node))

    @property
    def cells(self):
        return L.load().rs_table_cells(self._h)

    def cell_offset(self, node):
        return L.load().rs_table_cell_offset(self._h, node)

    @property
    def nbytes(self):
        return L.load().rs_table_bytes(self._h)

    # ---- bulk host <-> device -------------------------------------------------------------------
    def upload_node(self, node, regrets=None, strategy_sum=None):
        """arrays [A][n_boards*n_clusters]"""
        r = None if regrets is None else np.ascontiguousarray(regrets, dtype=self.np_dtype)
        s = None if strategy_sum is None else np.ascontiguousarray(strategy_sum, dtype=self.np_dtype)
        L.check(L.load().rs_table_upload_node(self._h, node, None if r is None else _vp(r), None if s is None else _vp(s)))

    def download_node(self, node):
        a, n = self.node_desc(node).n_actions, self.lanes(node)
        r = np.empty((a, n), dtype=self.np_dtype)
        s = np.empty((a, n), dtype=self.np_dtype)
        L.check(L.load().rs_table_download_node(self._h, node, _vp(r), _vp(s)))
        return r, s

    def upload(self, node, board, regrets=None, strategy_sum=None):
        r = None if regrets is None else np.ascontiguousarray(regrets, dtype=self.np_dtype)
        s = None if strategy_sum is None else np.ascontiguousarray(strategy_sum, dtype=self.np_dtype)
        L.check(L.load().rs_table_upload(self._h, node, board, None if r is None else _vp(r), None if s is None else _vp(s)))

    def download(self, node, board):
        d = self.node_desc(node)
        r = np.empty((d.n_actions, d.n_clusters), dtype=self.np_dtype)
        s = np.empty((d.n_actions, d.n_clusters), dtype=self.np_dtype)
        L.check(L.load().rs_table_download(self._h, node, board, _vp(r), _vp(s)))
        return r, s

    def fill_random(self, seed, regret_range=(-10**6, 10**6), ssum_range=(0, 10**6)):
        L.check(L.load().rs_table_fill_random(self._h, seed, regret_range[0], regret_range[1], ssum_range[0], ssum_range[1]))

    # ---- device vectors ----------------------------------------------------------------------------
    def lane_buffer(self, node, rows=1, data=None):
        """float32 device buffer [rows][pitch]; `data` = numpy [rows][lanes] (padded on upload)"""
        pitch = self.pitch(node)
        buf = DeviceBuffer(self, rows * pitch * 4)
        if data is None:
            buf.zero()
        else:
            host = np.zeros((rows, pitch), dtype=np.float32)
            host[:, : self.lanes(node)] = np.asarray(data, dtype=np.float32).reshape(rows, -1)
            buf.upload(host)
        return buf

    def read_lane_buffer(self, buf, node, rows=1):
        pitch = self.pitch(node)
        return buf.download(np.float32, rows * pitch).reshape(rows, pitch)[:, : self.lanes(node)]

    # ---- bulk kernels ---------------------------------------------------------------------------------
    def regret_match_node(self, node):
        """bulk get_strategy (infoset.rs:83-102) -> [A][lanes]"""
        a = self.node_desc(node).n_actions
        out = self.lane_buffer(node, a)
        L.check(L.load().rs_regret_match_node(self._h, node, out.ptr))
        return self.read_lane_buffer(out, node, a)

    def final_strategy_node(self, node):
        a = self.node_desc(node).n_actions
        out = self.lane_buffer(node, a)
        L.check(L.load().rs_final_strategy_node(self._h, node, out.ptr))
        return self.read_lane_buffer(out, node, a)

    def final_strategy_all(self):
        out = DeviceBuffer(self, self.cells * 4)
        L.check(L.load().rs_final_strategy_all(self._h, out.ptr))
        flat = out.download(np.float32, self.cells)
        res = []
        for n in range(self.n_nodes):
            a, p = self.node_desc(n).n_actions, self.pitch(n)
            off = self.cell_offset(n)
            res.append(flat[off: off + a * p].reshape(a, p)[:, : self.lanes(n)])
        return res

    def calc_br(self, tree):
        """MCCFRTrainer::calc_br as coded (cfr.rs:629-638): the pair train() prints at a discount tick"""
        out = np.zeros(2, dtype=np.float32)
        L.check(L.load().rs_calc_br(self._h, tree._h, _vp(out)))
        return out

    def best_response(self, tree, board, hands0, cluster0, hands1, cluster1, mode=L.BR_MAX):
        """value per deal of each player against the other's average strategy: mode BR_MAX a best response inside the abstraction,
        BR_AVERAGE its own average strategy; exploitability = best_response(...).sum() / 2"""
        b = np.ascontiguousarray(board, dtype=np.uint8)
        h0 = np.ascontiguousarray(hands0, dtype=np.uint8).reshape(-1, 2)
        h1 = np.ascontiguousarray(hands1, dtype=np.uint8).reshape(-1, 2)
        c0, c1 = np.ascontiguousarray(cluster0, dtype=np.uint32), np.ascontiguousarray(cluster1, dtype=np.uint32)
        if len(b) != 5 or len(c0) != len(h0) or len(c1) != len(h1):
            raise ValueError("board[5], one cluster id per hand")
        out = np.zeros(2, dtype=np.float64)
        L.check(L.load().rs_best_response(self._h, tree._h, _vp(b), _vp(h0), len(h0), _vp(c0), _vp(h1), len(h1), _vp(c1), mode, _vp(out)))
        return out

    def best_response_rounds(self, tree, board0, hands0, hands1, clusters, mode=L.BR_MAX):
        """multi-round best response (rs_best_response_rounds): clusters[r][p] = uint32 [prefixes of round r][n_hands_p] dense ids"""
        b = np.ascontiguousarray(board0, dtype=np.uint8)
        h0 = np.ascontiguousarray(hands0, dtype=np.uint8).reshape(-1, 2)
        h1 = np.ascontiguousarray(hands1, dtype=np.uint8).reshape(-1, 2)
        keep = [np.ascontiguousarray(clusters[r][p], dtype=np.uint32) for r in range(len(clusters)) for p in (0, 1)]
        arr = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
        out = np.zeros(2, dtype=np.float64)
        L.check(L.load().rs_best_response_rounds(self._h, tree._h, _vp(b), len(b), _vp(h0), len(h0), _vp(h1), len(h1), arr, len(clusters), mode,
                                                 out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def update_node(self, node, action_utils, reach=None, scale=100.0, mode=L.UPD_CLAMP_I64):
        """One traverser visit of every lane (cfr.rs:370-466 / :571-623).  Returns node util [lanes]."""
        a = self.node_desc(node).n_actions
        u = self.lane_buffer(node, a, action_utils)
        r = None if reach is None else self.lane_buffer(node, 1, reach)
        out = self.lane_buffer(node, 1)
        L.check(L.load().rs_update_node(self._h, node, u.ptr, None if r is None else r.ptr, scale, mode, out.ptr))
        return self.read_lane_buffer(out, node)[0]

    def node_util(self, node, action_utils):
        a = self.node_desc(node).n_actions
        u = self.lane_buffer(node, a, action_utils)
        out = self.lane_buffer(node, 1)
        L.check(L.load().rs_node_util(self._h, node, u.ptr, out.ptr))
        return self.read_lane_buffer(out, node)[0]

    def child_reach(self, node, reach=None):
        a = self.node_desc(node).n_actions
        r = None if reach is None else self.lane_buffer(node, 1, reach)
        out = self.lane_buffer(node, a)
        L.check(L.load().rs_child_reach(self._h, node, None if r is None else r.ptr, out.ptr))
        return self.read_lane_buffer(out, node, a)

    def get_infosets(self, node, lanes):
        """batched get-infoset: (regrets[A][n], strategy_sum[A][n]) of lanes (= board * n_clusters + cluster) of one node"""
        lanes = np.ascontiguousarray(lanes, dtype=np.uint32)
        a = self.node_desc(node).n_actions
        dt = np.int32 if self.dtype == L.I32 else np.float32
        r, s = np.zeros((a, len(lanes)), dtype=dt), np.zeros((a, len(lanes)), dtype=dt)
        L.check(L.load().rs_get_infosets(self._h, node, lanes.ctypes.data_as(C.c_void_p), len(lanes), r.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)))
        return r, s

    def checksum(self):
        """(regrets, strategy_sum) order-independent 64-bit checksums of the whole device arrays (diagnostics)"""
        out = (C.c_uint64 * 2)()
        L.check(L.load().rs_table_checksum(self._h, out))
        return int(out[0]), int(out[1])

    def save(self, path):
        """checkpoint ("RSTB" v1)"""
        L.check(L.load().rs_table_save(self._h, path.encode()))

    @classmethod
    def load(cls, path, device=0):
        h = C.c_void_p()
        L.check(L.load().rs_table_load(path.encode(), device, C.byref(h)))
        return cls(h)

    def discount(self, d):
        """cfr.rs:250-261"""
        L.check(L.load().rs_discount(self._h, float(d)))

    def sync(self):
        L.check(L.load().rs_sync(self._h))

    # ---- profiling ------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        L.check(L.load().rs_profile_enable(self._h, int(on)))

    def profile_mark(self):
        L.check(L.load().rs_profile_mark(self._h))

    def profile_marks(self, cap=65536):
        """durations (ms) between consecutive profile_mark() events on the table's stream; synchronises and forgets the marks"""
        buf, n = (C.c_float * cap)(), C.c_size_t()
        L.check(L.load().rs_profile_marks(self._h, buf, cap, C.byref(n)))
        return [float(buf[i]) for i in range(min(cap, n.value))]

    def profile_reset(self):
        L.check(L.load().rs_profile_reset(self._h))

    def profile_read(self):
        p = L.Profile()
        L.check(L.load().rs_profile_read(self._h, C.byref(p)))
        names = ["update", "node_util", "reach", "chance", "discount", "strategy", "tree"]
        return {n: dict(launches=int(p.launches[i]), ms=float(p.ms[i]), algo_bytes=float(p.algo_bytes[i]))
                for i, n in enumerate(names)}

    def destroy(self):
        if self._h:
            for sv in list(self._solvers):
                sv.destroy()
            if self._owned:
                L.load().rs_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def create_infosets(n_actions, tree, card_abs, n_boards=(1, 1, 1), dtype=L.I32, device=0):
    """infoset.rs:8.  `card_abs[round_idx]` stands for CardAbstraction::get_size: either an int (same for
    both players) or a (size_p0, size_p1) pair.  n_boards[round_idx] is the README's [board] axis."""
    if n_actions != tree.n_action_nodes:
        raise ValueError("n_actions does not match the tree")
    nc = ((C.c_uint32 * L.MAX_PLAYERS) * L.MAX_ROUNDS)()
    nb = (C.c_uint32 * L.MAX_ROUNDS)()
    for r in range(L.MAX_ROUNDS):
        sz = card_abs[r] if r < len(card_abs) else card_abs[-1]
        sz = (sz, sz) if isinstance(sz, int) else tuple(sz)
        nc[r][0], nc[r][1] = sz
        nb[r] = n_boards[r] if r < len(n_boards) else n_boards[-1]
    h = C.c_void_p()
    L.check(L.load().rs_create_infosets(tree._h, C.byref(nc), C.byref(nb), dtype, device, C.byref(h)))
    return InfosetTable(h)


DEFAULT_FORMS = {}   # rs_kernel_forms fields every MCCFRTrainer of this process starts from (the test-suite's fixtures set it; a caller passes forms=)


class MCCFRTrainer:
    """cfr.rs:146-297 for the batched lane model."""

    DISCOUNT_INTERVAL = 100_000   # cfr.rs:193
    DISCOUNT_CAP = 20_000_000     # cfr.rs:194

    def __init__(self, tree, infosets, leaves, scale=10000.0, mode=L.UPD_WRAP_I32, chance_mode=L.CHANCE_ENUM,
                 use_graph=False, leaves_p1=None, fuse_subtrees=None, opp_mode=L.OPP_FULL, sample_seed=0, deals=None, shard=None, prune_deal=None, forms=None):
        """leaves: dict tree-node-id -> (LEAF_* kind, DeviceBuffer) for every showdown / all-in terminal.
        deals: None (lane model) or dict (round_idx, player) -> uint32 array of dense cluster ids, one per deal
        (what get_cluster() returned, cfr.rs:361-365): batch-synchronous deal sweeps on the reference-shaped table."""
        self.game_tree, self.infosets = tree, infosets
        self._keep = [leaves, leaves_p1]
        if fuse_subtrees is None:   # tree-specialised kernels need hipRTC at run time; the level plan does not
            fuse_subtrees = bool(L.load().rs_jit_available())
        self.fused = bool(fuse_subtrees)
        self.n_deals = None
        batch = None
        if deals is not None:
            self.n_deals = len(next(iter(deals.values())))
            batch = L.DealBatch()
            batch.n_deals = self.n_deals
            pitch = deal_pitch(self.n_deals)
            for (r, pl), arr in deals.items():
                host = np.zeros(pitch, dtype=np.uint32)
                host[: self.n_deals] = np.asarray(arr, dtype=np.uint32)
                buf = DeviceBuffer.from_numpy(infosets, host)
                self._keep.append(buf)
                batch.d_cluster[r][pl] = buf.ptr
            if prune_deal is not None:   # UPD_PRUNE per deal (cfr.rs:213-221): uint8 flags, one per deal
                host = np.zeros(pitch, dtype=np.uint8)
                host[: self.n_deals] = np.asarray(prune_deal, dtype=np.uint8)
                buf = DeviceBuffer.from_numpy(infosets, host)
                self._keep.append(buf)
                batch.d_prune = buf.ptr
            chance_mode = L.CHANCE_PASS
        arrs = []
        for lv in (leaves, leaves if leaves_p1 is None else leaves_p1):
            arr = (L.LeafDesc * tree.n_nodes)()
            for nid, (kind, buf) in lv.items():
                arr[nid].kind = kind
                arr[nid].d_buf = buf.ptr
            arrs.append(arr)
        sw, sr, srd, sg = shard if shard else (0, 0, 0, 0)   # (world, rank, round, global boards of that round)
        p = L.SolverParams(scale, mode, chance_mode, int(use_graph), int(fuse_subtrees), opp_mode, sample_seed, sw, sr, srd, sg)
        for k, v in dict(DEFAULT_FORMS, **(forms or {})).items():   # rs_kernel_forms by field name: lane_fan, deals_per_thread, shadow, deal_order, delta_rows, direct_rows (0 = the engine's choice)
            setattr(p.forms, k, int(v))
        h = C.c_void_p()
        if batch is None:
            L.check(L.load().rs_solver_create(infosets._h, tree._h, arrs[0], arrs[1], C.byref(p), C.byref(h)))
        else:
            L.check(L.load().rs_solver_create_deals(infosets._h, tree._h, C.byref(batch), arrs[0], arrs[1], C.byref(p), C.byref(h)))
        self._h = h
        infosets._solvers.add(self)
        self.workspace_bytes = L.load().rs_solver_workspace_bytes(h)

    @classmethod
    def init(cls, options, card_abs, n_boards=(1, 1, 1), leaf_sign=None, dtype=L.I32, device=0, **kw):
        """MCCFRTrainer::init (cfr.rs:159-184): tree, table, plan.  leaf_sign: numpy [lanes of the last round]
        = sign(score0 - score1) per lane, shared by every showdown / all-in terminal of that round."""
        n_actions, tree = build_game_tree(options)
        infosets = create_infosets(n_actions, tree, card_abs, n_boards, dtype, device)
        leaves = {}
        bufs = {}
        for i, nd in enumerate(tree.nodes):
            if nd.kind == L.NODE_TERMINAL and nd.ttype != L.TERM_UNCONTESTED:
                parent = tree.nodes[nd.parent]
                r = parent.round_idx
                if r not in bufs:
                    sign = leaf_sign[r] if isinstance(leaf_sign, dict) else leaf_sign
                    bufs[r] = infosets.lane_buffer(parent.index, 1, sign)
                leaves[i] = (L.LEAF_SIGN, bufs[r])
        return cls(tree, infosets, leaves, **kw)

    def n_launches(self, player):
        return L.load().rs_solver_n_launches(self._h, player)

    def walk_counts(self, player):
        """(deal, round subtree) walks of the last sweep of `player`, per betting round (rs_solver_walk_counts)"""
        return walk_counts(self._h, player)

    @property
    def ordered(self):
        """deal sweeps walk the batch in the order of the traverser's last-round cluster (rs_kernel_forms.deal_order)"""
        return bool(L.load().rs_solver_forms(self._h) & 1)

    @property
    def delta_rows(self):
        """deal sweeps store their deltas by list position and sum them in one pass per sweep (rs_kernel_forms.delta_rows)"""
        return bool(L.load().rs_solver_forms(self._h) & 2)

    def iterate(self, player, want_root_util=False):
        """`self.cfr(0, player, hand, 1f32, ..)` for every lane (cfr.rs:217)"""
        root = self.game_tree.nodes[self.game_tree.nodes[0].children[0]]
        if not want_root_util:
            L.check(L.load().rs_iterate(self._h, player, None))
            return None
        if self.n_deals is not None:
            out = deal_buffer(self.infosets, self.n_deals)
            L.check(L.load().rs_iterate(self._h, player, out.ptr))
            return out.download(np.float32, deal_pitch(self.n_deals))[: self.n_deals]
        out = self.infosets.lane_buffer(root.index, 1)
        L.check(L.load().rs_iterate(self._h, player, out.ptr))
        return self.infosets.read_lane_buffer(out, root.index)[0]

    def iterate_phase(self, player, phase, want_root_util=False):
        """sharded sweeps driven by the host: phase 0, <exchange the slots>, phase 1"""
        root = self.game_tree.nodes[self.game_tree.nodes[0].children[0]]
        out = self.infosets.lane_buffer(root.index, 1) if (want_root_util and phase == 1) else None
        L.check(L.load().rs_iterate_phase(self._h, player, phase, None if out is None else out.ptr))
        return None if out is None else self.infosets.read_lane_buffer(out, root.index)[0]

    def exchange_info(self, player):
        """(device pointer, bytes per rank) of the exchange buffer [world][bytes per rank]"""
        buf, n = C.c_void_p(), C.c_size_t()
        L.check(L.load().rs_solver_exchange_info(self._h, player, C.byref(buf), C.byref(n)))
        return buf.value, n.value

    def attach_comm(self, comm_handle):
        L.check(L.load().rs_solver_attach_comm(self._h, comm_handle))

    def training_loop(self, on):
        """rs_solver_training_loop: bracket a hand-written loop of iterate() / table.discount() calls (nothing else may touch the table in between)"""
        L.check(L.load().rs_solver_training_loop(self._h, int(bool(on))))

    def train(self, iterations, discount_interval=None, discount_cap=None):
        """cfr.rs:188"""
        L.check(L.load().rs_train(self._h, iterations, discount_interval or self.DISCOUNT_INTERVAL,
                                  discount_cap or self.DISCOUNT_CAP))
        self.infosets.sync()

    def destroy(self):
        if self._h:
            L.load().rs_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def walk_counts(solver_handle, player):
    import ctypes as C
    out = (C.c_uint64 * 3)()
    L.check(L.load().rs_solver_walk_counts(solver_handle, player, out))
    return [int(x) for x in out]


class DealTrainer:
    """MCCFRTrainer::init + train (cfr.rs:159-297) with the whole deal pipeline on the GPU (rs_deal_trainer): generate_hand,
    get_cluster for every round and player, the showdown comparison and the sampled mccfr sweep, batch after batch."""

    def __init__(self, tree, card_abs, hand_ranges, board_mask, deals_per_batch, seed=0, scale=100.0, mode=L.UPD_CLAMP_I64,
                 opp_mode=L.OPP_SAMPLE, discount_interval=MCCFRTrainer.DISCOUNT_INTERVAL, discount_cap=MCCFRTrainer.DISCOUNT_CAP,
                 use_graph=False, fuse_subtrees=None, device=0, world=1, rank=0, prune_threshold=10_000_000, forms=None, prefetch=None, dtype=L.I32):
        """prune_threshold: cfr.rs:190 PRUNE_THRESHOLD (None = never prune).  forms: rs_kernel_forms fields by name, as for MCCFRTrainer.  world / rank: data-parallel training on replicated tables (one process per GPU): this rank deals its share of every global
        batch; attach_comm() makes every rank apply the deltas of the union batch."""
        if fuse_subtrees is None:
            fuse_subtrees = bool(L.load().rs_jit_available())
        self.game_tree, self.card_abs = tree, list(card_abs)
        p = L.DealTrainerParams()
        p.board_mask, p.deals_per_batch, p.seed = board_mask, deals_per_batch, seed
        p.world, p.rank = world, rank
        p.prune_threshold = 2**64 - 1 if prune_threshold is None else prune_threshold
        p.discount_interval, p.discount_cap = discount_interval, discount_cap
        p.table_dtype = dtype   # rs_deal_trainer_params.table_dtype: RS_I32 (the reference), or RS_F32 / RS_F16 (float deal sweeps: prune_threshold=None)
        p.prefetch = L.FORM_DEFAULT if prefetch is None else (L.FORM_ON if prefetch else L.FORM_OFF)   # rs_deal_trainer_params.prefetch: deal the next batch beside this one's sweeps
        p.solver.scale, p.solver.mode, p.solver.chance_mode = scale, mode, L.CHANCE_PASS
        p.solver.use_graph, p.solver.fuse_subtrees = int(use_graph), int(bool(fuse_subtrees))
        p.solver.opp_mode, p.solver.sample_seed = opp_mode, seed
        for k, v in dict(DEFAULT_FORMS, **(forms or {})).items():
            setattr(p.solver.forms, k, int(v))
        h0 = np.ascontiguousarray(hand_ranges[0], dtype=np.uint8).reshape(-1, 2)
        h1 = np.ascontiguousarray(hand_ranges[1], dtype=np.uint8).reshape(-1, 2)
        abs_arr = (C.c_void_p * len(self.card_abs))(*[a._h for a in self.card_abs])
        h = C.c_void_p()
        L.check(L.load().rs_deal_trainer_create(tree._h, abs_arr, len(self.card_abs), _vp(h0), len(h0), _vp(h1), len(h1), C.byref(p), device,
                                                C.byref(h)))
        self._h = h
        self.n_deals = deals_per_batch
        self.infosets = InfosetTable(C.c_void_p(L.load().rs_deal_trainer_table(h)), owned=False)

    def train(self, n_batches):
        L.check(L.load().rs_deal_trainer_train(self._h, n_batches))

    def deal(self):
        L.check(L.load().rs_deal_trainer_deal(self._h))

    def attach_comm(self, comm_handle):
        L.check(L.load().rs_deal_trainer_attach_comm(self._h, comm_handle))

    def iterate_phase(self, player, phase):
        """phase 0: the sweep (deltas accumulated), phase 1: table += delta; between them the ranks' deltas are summed"""
        L.check(L.load().rs_iterate_phase(L.load().rs_deal_trainer_solver(self._h), player, phase, None))

    def walk_counts(self, player):
        """(deal, round subtree) walks of the last batch's sweep of `player`, per betting round"""
        return walk_counts(L.load().rs_deal_trainer_solver(self._h), player)

    def finish_batch(self):
        L.check(L.load().rs_deal_trainer_finish_batch(self._h))

    def exchange_bytes(self):
        """bytes this rank has handed to the collectives of its data-parallel sweeps so far (rs_solver_exchange_bytes)"""
        b, n = C.c_uint64(), C.c_uint64()
        L.check(L.load().rs_solver_exchange_bytes(L.load().rs_deal_trainer_solver(self._h), C.byref(b), C.byref(n)))
        return b.value

    def deltas(self):
        """(regret deltas, strategy_sum deltas) as int32 arrays of rs_table_cells() elements"""
        a, b = C.c_void_p(), C.c_void_p()
        L.check(L.load().rs_table_deltas(self.infosets._h, C.byref(a), C.byref(b)))
        n = self.infosets.cells
        return self._download(a.value, np.int32, n), self._download(b.value, np.int32, n)

    def set_deltas(self, dreg, dssm):
        a, b = C.c_void_p(), C.c_void_p()
        L.check(L.load().rs_table_deltas(self.infosets._h, C.byref(a), C.byref(b)))
        for ptr, arr in ((a.value, dreg), (b.value, dssm)):
            arr = np.ascontiguousarray(arr, dtype=np.int32)
            L.check(L.load().rs_h2d(self.infosets._h, ptr, _vp(arr), arr.nbytes))

    def status(self):
        L.check(L.load().rs_deal_trainer_status(self._h))

    @property
    def iterations(self):
        return int(L.load().rs_deal_trainer_iterations(self._h))

    def set_tick_br(self, enable=True):
        """calc_br at every discount tick, as train() does (cfr.rs:244-246)"""
        L.check(L.load().rs_deal_trainer_set_tick_br(self._h, int(enable)))

    def last_br(self):
        """(pair of the latest tick, iteration count it was taken at)"""
        out, t = np.zeros(2, dtype=np.float32), C.c_uint64()
        L.check(L.load().rs_deal_trainer_last_br(self._h, _vp(out), C.byref(t)))
        return out, int(t.value)

    def calc_br(self):
        out = np.zeros(2, dtype=np.float32)
        L.check(L.load().rs_deal_trainer_calc_br(self._h, _vp(out)))
        return out

    def best_response(self, mode=L.BR_MAX):
        out = np.zeros(2, dtype=np.float64)
        L.check(L.load().rs_deal_trainer_best_response(self._h, mode, _vp(out)))
        return out

    def br_bytes(self):
        """device bytes held by the best-response objects the trainer keeps between calls"""
        return int(L.load().rs_deal_trainer_br_bytes(self._h))

    def br_release(self):
        L.check(L.load().rs_deal_trainer_br_release(self._h))

    def br_launches(self, sorted_showdowns=True):
        return int(L.load().rs_deal_trainer_br_launches(self._h, int(sorted_showdowns)))

    def exploitability(self, sorted_showdowns=True):
        """(BR value of player 0 + BR value of player 1) / 2 against the current average strategies, per deal, in pot units of the leaves.  sorted_showdowns: the leaves by
        rank order (RS_BR_SORTED: O(n log n) per run-out, equal to the pair loop of cfr.rs:323-347 within f64 rounding)"""
        return float(self.best_response(L.BR_MAX | (L.BR_SORTED if sorted_showdowns else 0)).sum() / 2.0)

    def _download(self, ptr, dtype, count):
        out = np.empty(count, dtype=dtype)
        L.check(L.load().rs_d2h(self.infosets._h, _vp(out), ptr, out.nbytes))
        return out

    def cards(self):
        """the current batch: uint8 [9][n_deals]"""
        pitch = deal_pitch(self.n_deals)
        return self._download(L.load().rs_deal_trainer_cards(self._h), np.uint8, 9 * pitch).reshape(9, pitch)[:, : self.n_deals]

    def signs(self):
        return self._download(L.load().rs_deal_trainer_signs(self._h), np.float32, deal_pitch(self.n_deals))[: self.n_deals]

    def prune_flags(self):
        """uint8 [n_deals]: the live batch's per-deal prune decision (cfr.rs:213-221); all zero before the threshold"""
        ptr = L.load().rs_deal_trainer_prune_flags(self._h)
        return self._download(ptr, np.uint8, deal_pitch(self.n_deals))[: self.n_deals]

    def clusters(self, round_idx, player):
        return self._download(L.load().rs_deal_trainer_clusters(self._h, round_idx, player), np.uint32, deal_pitch(self.n_deals))[: self.n_deals]

    def destroy(self):
        if self._h:
            self.infosets._h = None
            L.load().rs_deal_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def deal_pitch(n_deals):
    """per-deal device vectors are padded to a multiple of 64 lanes"""
    return (n_deals + 63) // 64 * 64


def deal_buffer(table, n_deals, data=None):
    """float32 device vector with one value per deal"""
    host = np.zeros(deal_pitch(n_deals), dtype=np.float32)
    if data is not None:
        host[:n_deals] = np.asarray(data, dtype=np.float32)
    return DeviceBuffer.from_numpy(table, host)


def showdown_sign(table, cards):
    """cards: uint8 [9][n_deals] (board x5, P0 hole x2, P1 hole x2) -> (DeviceBuffer of signs, numpy copy)"""
    cards = np.asarray(cards, dtype=np.uint8)
    n = cards.shape[1]
    pitch = deal_pitch(n)
    host = np.zeros((9, pitch), dtype=np.uint8)
    host[:, :n] = cards
    dc = DeviceBuffer.from_numpy(table, host)
    out = deal_buffer(table, n)
    L.check(L.load().rs_showdown_sign(table._h, dc.ptr, n, out.ptr))
    return out, out.download(np.float32, pitch)[:n]


def jit_check_tree(tree, dtype=L.I32, mode=L.UPD_CLAMP_I64, opp_mode=L.OPP_FULL):
    """compile (no GPU needed) every tree-specialised kernel of `tree`; returns the number of distinct kernels"""
    n = C.c_int()
    L.check(L.load().rs_jit_check_tree(tree._h, dtype, mode, opp_mode, C.byref(n)))
    return n.value


def jit_check_tree_deals(tree, mode=L.UPD_CLAMP_I64, opp_mode=L.OPP_SAMPLE):
    """compile (no GPU needed) every deal-batch kernel of `tree`: round subtrees, reach-down and walk, dense / listed, LDS / direct"""
    n = C.c_int()
    L.check(L.load().rs_jit_check_tree_deals(tree._h, mode, opp_mode, C.byref(n)))
    return n.value


def discount_factor(tc, interval=MCCFRTrainer.DISCOUNT_INTERVAL):
    """cfr.rs:248-249"""
    return np.float32(L.load().rs_discount_factor(tc, interval))


def device_count():
    n = C.c_int()
    rc = L.load().rs_device_count(C.byref(n))
    return n.value if rc == L.OK else 0
