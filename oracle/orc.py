"""ctypes binding of the CPU oracle (oracle/rs_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/rs_oracle.h).  Imported by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, never by ``rustsolver_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librs_oracle.so")

MAX_ACTIONS, MAX_ROUNDS, MAX_SIZES = 8, 3, 4
PRIVATE_CHANCE, PUBLIC_CHANCE, ACTION, TERMINAL = 0, 1, 2, 3
BR_SORTED = 0x100   # orc_best_response_rounds: | into the mode for the rank-order showdowns of rs_br.hip (same sums in the same order)
ALLIN, SHOWDOWN, UNCONTESTED = 0, 1, 2
ACT_BET, ACT_RAISE, ACT_CHECK, ACT_CALL, ACT_FOLD = 0, 1, 2, 3, 4
UPD_CLAMP_I64, UPD_WRAP_I32 = 0, 1
LEAF_UNCONTESTED, LEAF_SIGN, LEAF_UTIL = 0, 1, 2
CHANCE_PASS, CHANCE_ENUM = 0, 1
OPP_FULL, OPP_SAMPLE = 0, 1
T_I32, T_F32, T_F16 = 0, 1, 2
F_F32, F_F16 = 0, 1
PRUNE_THRESHOLD = -10000000


class Node(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("parent", C.c_int), ("n_children", C.c_int),
        ("children", C.c_int * MAX_ACTIONS),
        ("index", C.c_int), ("player", C.c_uint8), ("round_idx", C.c_uint8),
        ("action_kind", C.c_int * MAX_ACTIONS), ("action_amt", C.c_double * MAX_ACTIONS),
        ("value", C.c_uint32), ("ttype", C.c_int), ("last_to_act", C.c_uint8), ("round", C.c_int),
    ]


class Tree(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("n_nodes", C.c_int), ("cap", C.c_int), ("n_action_nodes", C.c_int)]


class Options(C.Structure):
    _fields_ = [
        ("stack_sizes", C.c_uint32 * 2), ("starting_pot", C.c_uint32), ("n_board_cards", C.c_int),
        ("n_rounds", C.c_int),
        ("n_bet_sizes", C.c_int * MAX_ROUNDS), ("bet_sizes", (C.c_double * MAX_SIZES) * MAX_ROUNDS),
        ("n_raise_sizes", C.c_int * MAX_ROUNDS), ("raise_sizes", (C.c_double * MAX_SIZES) * MAX_ROUNDS),
    ]


class Infoset(C.Structure):
    _fields_ = [("regrets", C.POINTER(C.c_int32)), ("strategy_sum", C.POINTER(C.c_int32)),
                ("fregrets", C.POINTER(C.c_float)), ("fstrategy_sum", C.POINTER(C.c_float)),
                ("n_actions", C.c_int)]


class Table(C.Structure):
    _fields_ = [("rows", C.POINTER(C.POINTER(Infoset))), ("row_len", C.POINTER(C.c_size_t)),
                ("n_rows", C.c_int), ("dtype", C.c_int)]


class Leaf(C.Structure):
    _fields_ = [("kind", C.c_int), ("buf", C.POINTER(C.c_float))]


class DealCtx(C.Structure):
    pass


class Ctx(C.Structure):
    _fields_ = [
        ("tree", C.POINTER(Tree)), ("table", C.POINTER(Table)),
        ("n_boards", C.c_uint32 * MAX_ROUNDS), ("n_clusters", C.c_uint32),
        ("leaves", C.POINTER(Leaf)),
        ("scale", C.c_float), ("mode", C.c_int), ("prune", C.c_int), ("rmplus", C.c_int),
        ("chance_mode", C.c_int), ("opp_mode", C.c_int), ("sample_seed", C.c_uint64), ("ref_alloc", C.c_int),
    ]


class CardsCtx(C.Structure):   # orc_cards_ctx (rs_oracle_mt.c)
    _fields_ = [("ix", C.c_void_p * MAX_ROUNDS), ("cluster_arr", C.c_void_p * MAX_ROUNDS),
                ("keys", (C.c_void_p * 2) * MAX_ROUNDS), ("ids", (C.c_void_p * 2) * MAX_ROUNDS), ("n_keys", (C.c_size_t * 2) * MAX_ROUNDS),
                ("n_rounds", C.c_int), ("first_street", C.c_int), ("board_mask", C.c_uint64), ("seed", C.c_uint64),
                ("hands", C.c_void_p * 2), ("n_hands", C.c_uint32 * 2), ("cidx", (C.c_void_p * 2) * MAX_ROUNDS), ("sign", C.c_void_p)]


class HandIndexerC(C.Structure):   # orc_hand_indexer (hand_index.h)
    _fields_ = [("rounds", C.c_int), ("cards_per_round", C.c_uint8 * 8), ("round_start", C.c_uint8 * 8),
                ("configurations", C.c_uint32 * 8), ("permutations", C.c_uint32 * 8), ("round_size", C.c_uint64 * 8),
                ("permutation_to_configuration", C.c_void_p * 8), ("permutation_to_pi", C.c_void_p * 8),
                ("configuration_to_equal", C.c_void_p * 8), ("configuration", C.c_void_p * 8),
                ("configuration_to_suit_size", C.c_void_p * 8), ("configuration_to_offset", C.c_void_p * 8)]


DealCtx._fields_ = [("ctx", C.POINTER(Ctx)), ("delta", C.POINTER(Table)),
                    ("cidx", (C.POINTER(C.c_uint32) * 2) * MAX_ROUNDS), ("n_deals", C.c_size_t), ("lane_base", C.c_size_t),
                    ("prune_deal", C.c_void_p)]


def build(force=False):
    """Compile oracle/librs_oracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("rs_oracle.c", "rs_oracle_mt.c", "rs_oracle.h", "hand_index.c", "hand_index.h", "kmeans_emd.c", "kmeans_fit.c", "best_response.c", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "librs_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.orc_options_default_river.argtypes = [C.POINTER(Options)]
    L.orc_options_three_street.argtypes = [C.POINTER(Options)]
    L.orc_tree_build.argtypes = [C.POINTER(Options), C.POINTER(Tree)]
    L.orc_tree_build.restype = C.c_int
    L.orc_tree_free.argtypes = [C.POINTER(Tree)]
    L.orc_table_create.argtypes = [C.POINTER(Tree), C.POINTER(C.c_uint32), C.c_uint32, C.c_int, C.POINTER(Table)]
    L.orc_table_create.restype = C.c_int
    L.orc_table_free.argtypes = [C.POINTER(Table)]
    L.orc_get_strategy.argtypes = [i32p, C.c_int, f32p]
    L.orc_get_final_strategy.argtypes = [i32p, C.c_int, f32p]
    L.orc_f32_as_i64.argtypes = [C.c_float]
    L.orc_f32_as_i64.restype = C.c_int64
    L.orc_f32_as_i32.argtypes = [C.c_float]
    L.orc_f32_as_i32.restype = C.c_int32
    L.orc_update_infoset.argtypes = [i32p, i32p, C.c_int, f32p, C.c_float, C.c_float, C.c_int, C.c_int]
    L.orc_update_infoset.restype = C.c_float
    L.orc_node_util.argtypes = [i32p, C.c_int, f32p]
    L.orc_node_util.restype = C.c_float
    L.orc_discount_factor.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_discount_factor.restype = C.c_float
    L.orc_discount_infoset.argtypes = [i32p, i32p, C.c_int, C.c_float]
    L.orc_discount_table.argtypes = [C.POINTER(Table), C.c_float]
    L.orc_traverse.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_float]
    L.orc_traverse.restype = C.c_float
    L.orc_iterate.argtypes = [C.POINTER(Ctx), C.c_int, f32p]
    L.orc_iterate_range.argtypes = [C.POINTER(Ctx), C.c_int, C.c_size_t, C.c_size_t, f32p]
    L.orc_train.argtypes = [C.POINTER(Ctx), C.c_size_t, C.c_size_t, C.c_size_t]
    L.orc_calc_br.argtypes = [C.POINTER(Tree), C.POINTER(Table), f32p]
    L.orc_best_response.argtypes = [C.POINTER(Tree), C.POINTER(Table), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_int, C.c_void_p]
    L.orc_get_strategy_f32.argtypes = [f32p, C.c_int, f32p]
    L.orc_update_infoset_f32.argtypes = [f32p, f32p, C.c_int, f32p, C.c_float, C.c_float, C.c_int, C.c_int]
    L.orc_update_infoset_f32.restype = C.c_float
    L.orc_node_util_f32.argtypes = [f32p, C.c_int, f32p]
    L.orc_node_util_f32.restype = C.c_float
    L.orc_discount_f32.argtypes = [f32p, f32p, C.c_int, C.c_float, C.c_int]
    L.orc_round_f16.argtypes = [C.c_float]
    L.orc_round_f16.restype = C.c_float
    L.orc_update_infoset_rmplus.argtypes = [i32p, i32p, C.c_int, f32p, C.c_float, C.c_float]
    L.orc_update_infoset_rmplus.restype = C.c_float
    for nm, pt in (("i32", i32p), ("f32", f32p)):
        getattr(L, "orc_table_set_node_" + nm).argtypes = [C.POINTER(Table), C.c_int, pt, pt]
        getattr(L, "orc_table_get_node_" + nm).argtypes = [C.POINTER(Table), C.c_int, pt, pt]
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_sample_bits.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    L.orc_sample_bits.restype = C.c_uint32
    L.orc_weighted_index.argtypes = [f32p, C.c_int, C.c_uint32]
    L.orc_weighted_index.restype = C.c_int
    L.orc_sweep_seed.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_sweep_seed.restype = C.c_uint64
    L.orc_traverse_deal.argtypes = [C.POINTER(DealCtx), C.c_int, C.c_int, C.c_size_t, C.c_float]
    L.orc_traverse_deal.restype = C.c_float
    L.orc_iterate_deals.argtypes = [C.POINTER(DealCtx), C.c_int, f32p]
    L.orc_table_create_sizes.argtypes = [C.POINTER(Tree), C.POINTER((C.c_uint32 * 2) * MAX_ROUNDS), C.c_int, C.POINTER(Table)]
    L.orc_table_create_sizes.restype = C.c_int
    L.orc_evaluate7.argtypes = [C.POINTER(C.c_uint8)]
    L.orc_evaluate7.restype = C.c_uint32
    L.orc_showdown_sign.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, f32p]
    L.orc_table_create_flat.argtypes = [C.POINTER(Tree), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Table)]
    L.orc_table_create_flat.restype = C.c_int
    L.orc_table_fill_flat.argtypes = [C.POINTER(Table), C.POINTER(Tree), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint64]
    L.orc_run_deal_sweeps_mt.argtypes = [C.POINTER(DealCtx), C.POINTER(Ctx), C.c_size_t, C.c_int]
    L.orc_iterate_mt.argtypes = [C.POINTER(Ctx), C.c_int, f32p, C.c_int]
    L.orc_run_iterations_mt.argtypes = [C.POINTER(Ctx), C.c_size_t, C.c_int]
    hp = C.POINTER(HandIndexerC)
    u8p = C.POINTER(C.c_uint8)
    L.orc_hand_indexer_init.argtypes = [C.c_int, u8p, hp]
    L.orc_hand_indexer_free.argtypes = [hp]
    L.orc_hand_indexer_size.argtypes = [hp, C.c_int]
    L.orc_hand_indexer_size.restype = C.c_uint64
    L.orc_hand_index_round.argtypes = [hp, C.c_int, C.c_void_p]
    L.orc_hand_index_round.restype = C.c_uint64
    L.orc_hand_index_last.argtypes = [hp, C.c_void_p]
    L.orc_hand_index_last.restype = C.c_uint64
    L.orc_hand_unindex.argtypes = [hp, C.c_int, C.c_uint64, C.c_void_p]
    L.orc_hand_canon.argtypes = [C.c_int, u8p, C.c_void_p, C.c_void_p]
    L.orc_generate_map.argtypes = [hp, C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_generate_map.restype = C.c_size_t
    L.orc_deal_bits.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.orc_deal_prune.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.orc_deal_bits.restype = C.c_uint64
    L.orc_generate_hand.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_emd_1d.argtypes = [f32p, f32p, C.c_int]
    L.orc_emd_1d.restype = C.c_float
    L.orc_l2_dist.argtypes = [f32p, f32p, C.c_int]
    L.orc_l2_dist.restype = C.c_float
    L.orc_kmeans_predict.argtypes = [C.c_int, f32p, C.c_size_t, C.c_size_t, f32p, C.c_int, C.c_int, C.POINTER(C.c_uint32), f32p]
    L.orc_kmeans_predict_mt.argtypes = [C.c_int, f32p, C.c_size_t, f32p, C.c_int, C.c_int, C.POINTER(C.c_uint32), f32p, C.c_int]
    L.orc_update_min_dists.argtypes = [C.c_int, f32p, f32p, C.c_size_t, f32p, C.c_int]
    L.orc_run_train_cards_mt.argtypes = [C.POINTER(DealCtx), C.POINTER(Ctx), C.POINTER(CardsCtx), C.c_size_t, C.c_int]
    L.orc_run_train_cards_mt.restype = C.c_int
    _lib = L
    return L


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- scalar wrappers (numpy in / numpy out) ------------------------------------------------------

def get_strategy(regrets):
    r = np.ascontiguousarray(regrets, dtype=np.int32)
    out = np.zeros(len(r), dtype=np.float32)
    lib().orc_get_strategy(_i32(r), len(r), _f32(out))
    return out


def get_final_strategy(ssum):
    s = np.ascontiguousarray(ssum, dtype=np.int32)
    out = np.zeros(len(s), dtype=np.float32)
    lib().orc_get_final_strategy(_i32(s), len(s), _f32(out))
    return out


def update_infoset(regrets, ssum, utils, reach, scale=100.0, mode=UPD_CLAMP_I64, prune=False):
    """Returns (util, new_regrets, new_strategy_sum); inputs are not modified."""
    r = np.array(regrets, dtype=np.int32)
    s = np.array(ssum, dtype=np.int32)
    u = np.ascontiguousarray(utils, dtype=np.float32)
    util = lib().orc_update_infoset(_i32(r), _i32(s), len(r), _f32(u), np.float32(reach), np.float32(scale),
                                    mode, int(prune))
    return np.float32(util), r, s


def update_infoset_rmplus(regrets, ssum, utils, reach, scale=100.0):
    r = np.array(regrets, dtype=np.int32)
    s = np.array(ssum, dtype=np.int32)
    u = np.ascontiguousarray(utils, dtype=np.float32)
    util = lib().orc_update_infoset_rmplus(_i32(r), _i32(s), len(r), _f32(u), np.float32(reach), np.float32(scale))
    return np.float32(util), r, s


def update_infoset_f32(regrets, ssum, utils, reach, scale=100.0, rmplus=False, storage=F_F32):
    r = np.array(regrets, dtype=np.float32)
    s = np.array(ssum, dtype=np.float32)
    u = np.ascontiguousarray(utils, dtype=np.float32)
    util = lib().orc_update_infoset_f32(_f32(r), _f32(s), len(r), _f32(u), np.float32(reach), np.float32(scale),
                                        int(rmplus), storage)
    return np.float32(util), r, s


def node_util(regrets, utils):
    r = np.ascontiguousarray(regrets, dtype=np.int32)
    u = np.ascontiguousarray(utils, dtype=np.float32)
    return np.float32(lib().orc_node_util(_i32(r), len(r), _f32(u)))


def discount(values_r, values_s, d):
    r = np.array(values_r, dtype=np.int32)
    s = np.array(values_s, dtype=np.int32)
    lib().orc_discount_infoset(_i32(r), _i32(s), len(r), np.float32(d))
    return r, s


def discount_factor(tc, interval=100000):
    return np.float32(lib().orc_discount_factor(tc, interval))


# ---- tree ----------------------------------------------------------------------------------------

def make_options(stacks=(500, 500), pot=35, n_board_cards=5, bet_sizes=((0.5, 1.0),), raise_sizes=((3.0,),)):
    o = Options()
    o.stack_sizes[0], o.stack_sizes[1] = stacks
    o.starting_pot = pot
    o.n_board_cards = n_board_cards
    o.n_rounds = len(bet_sizes)
    for r, (bs, rs) in enumerate(zip(bet_sizes, raise_sizes)):
        o.n_bet_sizes[r] = len(bs)
        for i, v in enumerate(bs):
            o.bet_sizes[r][i] = v
        o.n_raise_sizes[r] = len(rs)
        for i, v in enumerate(rs):
            o.raise_sizes[r][i] = v
    return o


def options_default_river():
    o = Options()
    lib().orc_options_default_river(C.byref(o))
    return o


def options_three_street():
    o = Options()
    lib().orc_options_three_street(C.byref(o))
    return o


class OracleTree:
    def __init__(self, options):
        self.t = Tree()
        if lib().orc_tree_build(C.byref(options), C.byref(self.t)) != 0:
            raise ValueError("invalid board mask")  # state.rs:64 panic
        self.n_nodes = self.t.n_nodes
        self.n_action_nodes = self.t.n_action_nodes

    @classmethod
    def from_nodes(cls, nodes):
        """a tree adopted from plain node records (the fields of rs_tree_node / orc_node: kind, parent, children, index, player, round_idx, value, ttype, last_to_act,
        round) instead of built from Options -- what rs_tree_from_nodes takes"""
        self = cls.__new__(cls)
        self._arr = (Node * len(nodes))()
        n_act = 0
        for i, src in enumerate(nodes):
            d = self._arr[i]
            d.kind, d.parent, d.n_children = int(src.kind), int(src.parent), int(src.n_children)
            for k in range(d.n_children):
                d.children[k] = int(src.children[k])
            d.index, d.player, d.round_idx = int(src.index), int(src.player), int(src.round_idx)
            d.value, d.ttype, d.last_to_act, d.round = int(src.value), int(src.ttype), int(src.last_to_act), int(src.round)
            n_act += 1 if d.kind == ACTION else 0
        self.t = Tree()
        self.t.nodes, self.t.n_nodes, self.t.cap, self.t.n_action_nodes = C.cast(self._arr, C.POINTER(Node)), len(nodes), len(nodes), n_act
        self.n_nodes, self.n_action_nodes = len(nodes), n_act
        self._borrowed = True   # the node array belongs to Python: orc_tree_free must not see it
        return self

    def node(self, i):
        return self.t.nodes[i]

    def as_dicts(self):
        out = []
        for i in range(self.n_nodes):
            n = self.t.nodes[i]
            d = {"id": i, "kind": n.kind, "parent": n.parent, "children": [n.children[k] for k in range(n.n_children)]}
            if n.kind == ACTION:
                d.update(index=n.index, player=n.player, round_idx=n.round_idx,
                         actions=[[n.action_kind[k], n.action_amt[k]] for k in range(n.n_children)])
            elif n.kind == TERMINAL:
                d.update(value=n.value, ttype=n.ttype, last_to_act=n.last_to_act, round=n.round)
            elif n.kind == PUBLIC_CHANCE:
                d.update(round=n.round)
            out.append(d)
        return out

    def __del__(self):
        try:
            if not getattr(self, "_borrowed", False):
                lib().orc_tree_free(C.byref(self.t))
        except Exception:
            pass


class OracleTable:
    """Reference-layout table (Vec<Vec<Infoset{Box<[i32]>, Box<[i32]>}>>, infoset.rs:6,63-67)."""

    def __init__(self, tree, n_boards, n_clusters, dtype=T_I32):
        self.tree = tree
        self.n_boards = list(n_boards) + [0] * (MAX_ROUNDS - len(n_boards))
        self.n_clusters = n_clusters
        self.dtype = dtype
        self.tb = Table()
        nb = (C.c_uint32 * MAX_ROUNDS)(*self.n_boards)
        if lib().orc_table_create(C.byref(tree.t), nb, n_clusters, dtype, C.byref(self.tb)) != 0:
            raise MemoryError
        self._node_by_index = {}
        for i in range(tree.n_nodes):
            n = tree.node(i)
            if n.kind == ACTION:
                self._node_by_index[n.index] = (n.n_children, n.round_idx, n.player)

    def node_shape(self, index):
        """(n_actions, n_lanes)"""
        a, r, _ = self._node_by_index[index]
        return a, self.n_boards[r] * self.n_clusters

    def set_node(self, index, regrets, ssum):
        """regrets/ssum: arrays [A][n_lanes]"""
        a, n = self.node_shape(index)
        if self.dtype == T_I32:
            R = np.ascontiguousarray(regrets, dtype=np.int32).reshape(a, n)
            S = np.ascontiguousarray(ssum, dtype=np.int32).reshape(a, n)
            lib().orc_table_set_node_i32(C.byref(self.tb), index, _i32(R), _i32(S))
        else:
            R = np.ascontiguousarray(regrets, dtype=np.float32).reshape(a, n)
            S = np.ascontiguousarray(ssum, dtype=np.float32).reshape(a, n)
            lib().orc_table_set_node_f32(C.byref(self.tb), index, _f32(R), _f32(S))

    def get_node(self, index):
        a, n = self.node_shape(index)
        if self.dtype == T_I32:
            R = np.zeros((a, n), dtype=np.int32)
            S = np.zeros((a, n), dtype=np.int32)
            lib().orc_table_get_node_i32(C.byref(self.tb), index, _i32(R), _i32(S))
        else:
            R = np.zeros((a, n), dtype=np.float32)
            S = np.zeros((a, n), dtype=np.float32)
            lib().orc_table_get_node_f32(C.byref(self.tb), index, _f32(R), _f32(S))
        return R, S

    def discount(self, d):
        lib().orc_discount_table(C.byref(self.tb), np.float32(d))

    def calc_br(self):
        """MCCFRTrainer::calc_br as coded (cfr.rs:629-638): the two numbers train() prints at a discount tick"""
        out = np.zeros(2, dtype=np.float32)
        lib().orc_calc_br(C.byref(self.tree.t), C.byref(self.tb), _f32(out))
        return out

    def best_response(self, board, hands0, cid0, hands1, cid1, mode=0):
        """value per deal of each player against the other's average strategy (mode 0: best response, 1: own average strategy)"""
        b = np.ascontiguousarray(board, dtype=np.uint8)
        h0, h1 = np.ascontiguousarray(hands0, dtype=np.uint8), np.ascontiguousarray(hands1, dtype=np.uint8)
        c0, c1 = np.ascontiguousarray(cid0, dtype=np.uint32), np.ascontiguousarray(cid1, dtype=np.uint32)
        out = np.zeros(2, dtype=np.float64)
        lib().orc_best_response(C.byref(self.tree.t), C.byref(self.tb), b.ctypes.data, h0.ctypes.data, len(c0), c0.ctypes.data, h1.ctypes.data, len(c1),
                                c1.ctypes.data, mode, out.ctypes.data)
        return out

    def best_response_rounds(self, board0, hands0, hands1, cids, mode=0):
        """multi-round best response (best_response.c orc_best_response_rounds): cids[r][p] = uint32 [prefixes of round r][n_p] dense cluster ids"""
        b0 = np.ascontiguousarray(board0, dtype=np.uint8)
        h0, h1 = np.ascontiguousarray(hands0, dtype=np.uint8), np.ascontiguousarray(hands1, dtype=np.uint8)
        keep = [np.ascontiguousarray(cids[r][p], dtype=np.uint32) for r in range(len(cids)) for p in (0, 1)]
        arr = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
        out = np.zeros(2, dtype=np.float64)
        fn = lib().orc_best_response_rounds
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(Tree), C.POINTER(Table), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        rc = fn(C.byref(self.tree.t), C.byref(self.tb), b0.ctypes.data, len(b0), h0.ctypes.data, len(h0), h1.ctypes.data, len(h1), arr, len(cids), mode, out.ctypes.data)
        if rc != 0:
            raise ValueError("orc_best_response_rounds: bad arguments")
        return out

    def __del__(self):
        try:
            lib().orc_table_free(C.byref(self.tb))
        except Exception:
            pass


class OracleSolver:
    """Per-lane cfr.rs:481-627 recursion over an OracleTable."""

    def __init__(self, tree, table, leaves, scale=10000.0, mode=UPD_WRAP_I32, prune=False, rmplus=False,
                 chance_mode=CHANCE_ENUM, ref_alloc=False, opp_mode=OPP_FULL, base_seed=0):
        """leaves: dict tree-node-id -> (kind, float32 array over that terminal's lanes or None)"""
        self.tree, self.table = tree, table
        self._keep = []
        arr = (Leaf * tree.n_nodes)()
        for i in range(tree.n_nodes):
            arr[i].kind = LEAF_UNCONTESTED
            arr[i].buf = None
        for nid, (kind, buf) in leaves.items():
            arr[nid].kind = kind
            if buf is not None:
                b = np.ascontiguousarray(buf, dtype=np.float32)
                self._keep.append(b)
                arr[nid].buf = _f32(b)
        self._leaves = arr
        c = Ctx()
        c.tree = C.pointer(tree.t)
        c.table = C.pointer(table.tb)
        for r in range(MAX_ROUNDS):
            c.n_boards[r] = table.n_boards[r]
        c.n_clusters = table.n_clusters
        c.leaves = arr
        c.scale = scale
        c.mode = mode
        c.prune = int(prune)
        c.rmplus = int(rmplus)
        c.chance_mode = chance_mode
        c.opp_mode = opp_mode
        c.sample_seed = 0
        c.ref_alloc = int(ref_alloc)
        self.ctx = c
        self.base_seed = base_seed
        self.calls = 0          # mirrors the GPU solver: sweep k uses orc_sweep_seed(base_seed, k)

    def set_leaves(self, leaves):
        for nid, (kind, buf) in leaves.items():
            self._leaves[nid].kind = kind
            if buf is not None:
                b = np.ascontiguousarray(buf, dtype=np.float32)
                self._keep.append(b)
                self._leaves[nid].buf = _f32(b)

    def iterate(self, player, threads=1, seed=None):
        self.ctx.sample_seed = lib().orc_sweep_seed(self.base_seed, self.calls) if seed is None else seed
        self.calls += 1
        n = self.table.n_boards[0] * self.table.n_clusters
        out = np.zeros(n, dtype=np.float32)
        if threads > 1:
            lib().orc_iterate_mt(C.byref(self.ctx), player, _f32(out), threads)
        else:
            lib().orc_iterate(C.byref(self.ctx), player, _f32(out))
        return out

    def run_iterations(self, iterations, threads=1):
        lib().orc_run_iterations_mt(C.byref(self.ctx), iterations, threads)

    def train(self, iterations, discount_interval=100000, discount_cap=20000000):
        lib().orc_train(C.byref(self.ctx), iterations, discount_interval, discount_cap)


class OracleDealTable(OracleTable):
    """The reference's own table shape: [action_node][cluster], sizes per (round_idx, player) (infoset.rs:28-32)."""

    def __init__(self, tree, sizes, dtype=T_I32):
        """sizes[round_idx] = (size_p0, size_p1)"""
        self.tree = tree
        self.sizes = [tuple(s) for s in sizes] + [(0, 0)] * (MAX_ROUNDS - len(sizes))
        self.n_boards = [1] * MAX_ROUNDS
        self.n_clusters = 0
        self.dtype = dtype
        self.tb = Table()
        arr = ((C.c_uint32 * 2) * MAX_ROUNDS)()
        for r in range(MAX_ROUNDS):
            arr[r][0], arr[r][1] = self.sizes[r]
        if lib().orc_table_create_sizes(C.byref(tree.t), C.byref(arr), dtype, C.byref(self.tb)) != 0:
            raise MemoryError
        self._node_by_index = {}
        for i in range(tree.n_nodes):
            n = tree.node(i)
            if n.kind == ACTION:
                self._node_by_index[n.index] = (n.n_children, n.round_idx, n.player)

    def node_shape(self, index):
        a, r, p = self._node_by_index[index]
        return a, self.sizes[r][p]


class OracleDealSolver(OracleSolver):
    """Batch-synchronous deal sweeps (orc_iterate_deals)."""

    def __init__(self, tree, table, leaves, cidx, n_deals, lane_base=0, prune_deal=None, **kw):
        """cidx: dict (round_idx, player) -> uint32 array [n_deals]; lane_base: global index of deal 0 (data-parallel batches);
        prune_deal: uint8 [n_deals] read at every sweep, 1 = the deal is traversed with prune = true (needs prune=True; cfr.rs:213-221)"""
        super().__init__(tree, table, leaves, chance_mode=CHANCE_PASS, **kw)
        self.delta = OracleDealTable(tree, table.sizes, table.dtype)
        self.n_deals = n_deals
        dc = DealCtx()
        dc.ctx = C.pointer(self.ctx)
        dc.delta = C.pointer(self.delta.tb)
        dc.n_deals = n_deals
        dc.lane_base = lane_base
        for (r, p), arr in cidx.items():
            a = np.ascontiguousarray(arr, dtype=np.uint32)
            self._keep.append(a)
            dc.cidx[r][p] = a.ctypes.data_as(C.POINTER(C.c_uint32))
        if prune_deal is not None:
            if prune_deal.dtype != np.uint8 or len(prune_deal) != n_deals or not prune_deal.flags.c_contiguous:
                raise ValueError("prune_deal: contiguous uint8 [n_deals]")
            self._keep.append(prune_deal)
            dc.prune_deal = prune_deal.ctypes.data
        self.dc = dc

    def iterate(self, player, threads=1, seed=None):
        self.ctx.sample_seed = lib().orc_sweep_seed(self.base_seed, self.calls) if seed is None else seed
        self.calls += 1
        out = np.zeros(self.n_deals, dtype=np.float32)
        lib().orc_iterate_deals(C.byref(self.dc), player, _f32(out))
        return out

    def run_sweeps(self, sweeps, threads):
        """timed CPU baseline: `sweeps` x (both players over all deals), reference-style allocations, `threads` workers"""
        lib().orc_run_deal_sweeps_mt(C.byref(self.dc), C.byref(self.ctx), sweeps, threads)


def run_train_from_cards(solver, cidx, sign, board_mask, ranges, seed, sweeps, threads, bucket_files=None):
    """timed CPU baseline of the device-trainer leg: `sweeps` x (deal a batch from cards, then both players over all deals).
    solver: an OracleDealSolver built on the SAME cidx / sign arrays (they are rewritten in place every sweep)."""
    n_rounds = len({r for (r, _p) in cidx})
    first = bin(board_mask).count("1") - 3
    cc = CardsCtx()
    keep = []
    for r in range(n_rounds):
        nb = 3 + first + r
        ix = HandIndexer([2, nb])
        keep.append(ix)
        cc.ix[r] = C.addressof(ix.ix)
        arr = None if bucket_files is None else bucket_files[r]
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=np.uint32)
            keep.append(arr)
            cc.cluster_arr[r] = arr.ctypes.data
        for p in (0, 1):
            keys = ix.generate_map(ranges[p], board_mask, nb, arr)
            order = np.argsort(keys)
            sk, ids = np.ascontiguousarray(keys[order]), np.ascontiguousarray(order.astype(np.uint32))
            keep += [sk, ids]
            cc.keys[r][p], cc.ids[r][p], cc.n_keys[r][p] = sk.ctypes.data, ids.ctypes.data, len(sk)
            cc.cidx[r][p] = cidx[(r, p)].ctypes.data
    cc.n_rounds, cc.first_street, cc.board_mask, cc.seed = n_rounds, first, board_mask, seed
    for p in (0, 1):
        h = np.ascontiguousarray(ranges[p], dtype=np.uint8).reshape(-1, 2)
        keep.append(h)
        cc.hands[p], cc.n_hands[p] = h.ctypes.data, len(h)
    cc.sign = sign.ctypes.data
    if lib().orc_run_train_cards_mt(C.byref(solver.dc), C.byref(solver.ctx), C.byref(cc), sweeps, threads) != 0:
        raise RuntimeError("orc_run_train_cards_mt: a deal could not be sampled or addressed")


class OracleFlatTable(OracleTable):
    """Tuned CPU layout (one contiguous pool per node, nothing boxed) for the non-strawman CPU baseline."""

    def __init__(self, tree, n_boards, n_clusters, seed=1):
        self.tree = tree
        self.n_boards = list(n_boards) + [0] * (MAX_ROUNDS - len(n_boards))
        self.n_clusters = n_clusters
        self.dtype = T_I32
        self.tb = Table()
        nb = (C.c_uint32 * MAX_ROUNDS)(*self.n_boards)
        if lib().orc_table_create_flat(C.byref(tree.t), nb, n_clusters, C.byref(self.tb)) != 0:
            raise MemoryError
        lib().orc_table_fill_flat(C.byref(self.tb), C.byref(tree.t), nb, n_clusters, seed)
        self._node_by_index = {}

    def __del__(self):
        pass   # pools are intentionally not freed (bounded baseline run)


def evaluate7(cards7):
    a = np.ascontiguousarray(cards7, dtype=np.uint8)
    return int(lib().orc_evaluate7(a.ctypes.data_as(C.POINTER(C.c_uint8))))


def showdown_sign(cards):
    """cards: uint8 [9][n] -> float32 [n], brute-force best-of-21 evaluator"""
    a = np.ascontiguousarray(cards, dtype=np.uint8)
    out = np.zeros(a.shape[1], dtype=np.float32)
    lib().orc_showdown_sign(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], _f32(out))
    return out


# ---- canonical hand index / card abstraction / deal sampler (hand_index.c) ---------------------------------------------------
class HandIndexer:
    """orc_hand_indexer: hand_indexer_s::init / size / get_index / get_hand restated on the CPU"""

    def __init__(self, cards_per_round):
        self.cards_per_round = tuple(cards_per_round)
        self.rounds = len(cards_per_round)
        self.ix = HandIndexerC()
        if lib().orc_hand_indexer_init(self.rounds, (C.c_uint8 * self.rounds)(*cards_per_round), C.byref(self.ix)) != 0:
            raise ValueError("orc_hand_indexer_init failed")

    def size(self, round_):
        return int(lib().orc_hand_indexer_size(C.byref(self.ix), round_))

    def n_cards(self, round_):
        return sum(self.cards_per_round[: round_ + 1])

    def get_index(self, cards, round_=None):
        r = self.rounds - 1 if round_ is None else round_
        c = np.ascontiguousarray(cards, dtype=np.uint8)
        if c.ndim == 1:
            return int(lib().orc_hand_index_round(C.byref(self.ix), r, c.ctypes.data))
        return np.array([lib().orc_hand_index_round(C.byref(self.ix), r, c[i].ctypes.data) for i in range(len(c))], dtype=np.uint64)

    def get_hand(self, round_, index):
        out = np.zeros(self.n_cards(round_), dtype=np.uint8)
        if lib().orc_hand_unindex(C.byref(self.ix), round_, int(index), out.ctypes.data) != 0:
            raise IndexError(index)
        return out

    def canon(self, cards, round_=None):
        """brute-force canonical form (minimum over the 24 suit relabellings), independent of the index arithmetic"""
        r = self.rounds - 1 if round_ is None else round_
        c = np.ascontiguousarray(cards, dtype=np.uint8)[: self.n_cards(r)].copy()
        out = np.zeros(len(c), dtype=np.uint8)
        lib().orc_hand_canon(r + 1, (C.c_uint8 * (r + 1))(*self.cards_per_round[: r + 1]), c.ctypes.data, out.ctypes.data)
        return bytes(out)

    def generate_map(self, hands, initial_board_mask, n_round_board_cards, cluster_arr=None):
        """generate_maps for one player (card_abstraction.rs:75-184): distinct buckets in first-appearance order"""
        h = np.ascontiguousarray(hands, dtype=np.uint8).reshape(-1, 2)
        arr = None if cluster_arr is None else np.ascontiguousarray(cluster_arr, dtype=np.uint32)
        ap = None if arr is None else arr.ctypes.data
        n = lib().orc_generate_map(C.byref(self.ix), h.ctypes.data, len(h), initial_board_mask, n_round_board_cards, ap, None, 0)
        if n == C.c_size_t(-1).value:
            raise ValueError("orc_generate_map failed")
        keys = np.zeros(n, dtype=np.uint64)
        lib().orc_generate_map(C.byref(self.ix), h.ctypes.data, len(h), initial_board_mask, n_round_board_cards, ap, keys.ctypes.data, n)
        return keys

    def __del__(self):
        try:
            lib().orc_hand_indexer_free(C.byref(self.ix))
        except Exception:
            pass


def deal_prune_flags(seed, first_deal, prune_threshold, n_deals):
    """train()'s per-deal prune decision (cfr.rs:213-221) for deals first_deal .. first_deal + n_deals - 1 -> uint8 [n_deals]"""
    return np.array([lib().orc_deal_prune(seed, first_deal + i, prune_threshold) for i in range(n_deals)], dtype=np.uint8)


def generate_hands(seed, first_deal, board_mask, hands0, hands1, n_deals):
    """orc_generate_hand for deals first_deal .. first_deal + n_deals - 1 -> uint8 [9][n_deals]"""
    h0 = np.ascontiguousarray(hands0, dtype=np.uint8).reshape(-1, 2)
    h1 = np.ascontiguousarray(hands1, dtype=np.uint8).reshape(-1, 2)
    out = np.zeros((n_deals, 9), dtype=np.uint8)
    for i in range(n_deals):
        if lib().orc_generate_hand(seed, first_deal + i, board_mask, h0.ctypes.data, len(h0), h1.ctypes.data, len(h1), out[i].ctypes.data) != 0:
            raise RuntimeError("orc_generate_hand: no valid deal")
    return np.ascontiguousarray(out.T)


# ---- abstraction generator's distance sweep (kmeans_emd.c) ------------------------------------------------------------------------
DIST_EMD, DIST_L2 = 0, 1


def emd_1d(p, q):
    """gen_abstraction/emd.rs:53-113"""
    a, b = np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)
    assert a.shape == b.shape and a.ndim == 1
    return np.float32(lib().orc_emd_1d(_f32(a), _f32(b), len(a)))


def l2_dist(p, q):
    """gen_abstraction/kmeans.rs:622-630"""
    a, b = np.ascontiguousarray(p, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)
    return np.float32(lib().orc_l2_dist(_f32(a), _f32(b), len(a)))


def kmeans_predict(dataset, centers, kind=DIST_EMD, threads=1):
    """Kmeans::predict (kmeans.rs:173-211) -> (clusters uint32 [n], min distances float32 [n])"""
    d, c = np.ascontiguousarray(dataset, dtype=np.float32), np.ascontiguousarray(centers, dtype=np.float32)
    assert d.ndim == 2 and c.ndim == 2 and d.shape[1] == c.shape[1]
    clusters, md = np.zeros(len(d), dtype=np.uint32), np.zeros(len(d), dtype=np.float32)
    lib().orc_kmeans_predict_mt(kind, _f32(d), len(d), _f32(c), len(c), d.shape[1], clusters.ctypes.data_as(C.POINTER(C.c_uint32)), _f32(md), threads)
    return clusters, md


def update_min_dists(min_dists, dataset, new_center, kind=DIST_EMD):
    """kmeans.rs:603-619, in place"""
    d, c = np.ascontiguousarray(dataset, dtype=np.float32), np.ascontiguousarray(new_center, dtype=np.float32)
    assert min_dists.dtype == np.float32 and min_dists.flags.c_contiguous
    lib().orc_update_min_dists(kind, _f32(min_dists), _f32(d), len(d), _f32(c), d.shape[1])
    return min_dists


def br_runouts(board0):
    """the run-outs of an initial board, in the enumeration order of the multi-round best response: uint8 [NB][5]"""
    b0 = np.ascontiguousarray(board0, dtype=np.uint8)
    fn = lib().orc_br_runouts
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    nb = fn(b0.ctypes.data, len(b0), None)
    out = np.zeros((nb, 5), dtype=np.uint8)
    fn(b0.ctypes.data, len(b0), out.ctypes.data)
    return out


# ---- k-means training loops (kmeans_fit.c) -----------------------------------------------------------------------------------------------
def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def kmeans_init_s(centers, s, kind=DIST_EMD):
    """Kmeans::init_s (kmeans.rs:267-285); s is updated in place (it is only ever lowered, then halved: see kmeans_fit.c)"""
    c = np.ascontiguousarray(centers, dtype=np.float32)
    assert s.dtype == np.float32 and s.flags.c_contiguous and len(s) == len(c)
    lib().orc_kmeans_init_s(kind, _f32(c), len(c), c.shape[1], _f32(s))
    return s


def kmeans_pick_restart(candidates, kind=DIST_EMD):
    """Kmeans::init_random's scoring (kmeans.rs:124-156): candidates [n_restarts][k][n_bins] -> (best index, cluster_dists)"""
    c = np.ascontiguousarray(candidates, dtype=np.float32)
    cd = np.zeros(c.shape[0], dtype=np.float32)
    fn = lib().orc_kmeans_pick_restart
    fn.restype = C.c_int
    best = fn(kind, _f32(c), c.shape[0], c.shape[1], c.shape[2], _f32(cd))
    return int(best), cd


def kmeans_reassign(dataset, centers, s, clusters, bounds, kind=DIST_EMD, order=None):
    """Kmeans::reassign_clusters (kmeans.rs:287-334); clusters (uint32 [n]) and bounds (float32 [n][2] = lower, upper) in place"""
    d, c = np.ascontiguousarray(dataset, dtype=np.float32), np.ascontiguousarray(centers, dtype=np.float32)
    assert clusters.dtype == np.uint32 and bounds.dtype == np.float32 and bounds.shape == (len(clusters), 2)
    o = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
    lib().orc_kmeans_reassign(kind, _f32(d), None if o is None else _u32(o), len(clusters), _f32(c), len(c), d.shape[1], _f32(np.ascontiguousarray(s, dtype=np.float32)),
                              _u32(clusters), _f32(bounds))


def kmeans_fit_regular(dataset, centers, kind=DIST_EMD, iterations=10):
    """Kmeans::fit_regular (kmeans.rs:497-600) -> (clusters, new centers, bounds, inertia)"""
    d, c = np.ascontiguousarray(dataset, dtype=np.float32), np.array(centers, dtype=np.float32, order="C")
    clusters, bounds = np.zeros(len(d), dtype=np.uint32), np.zeros((len(d), 2), dtype=np.float32)
    fn = lib().orc_kmeans_fit_regular
    fn.restype = C.c_float
    inertia = fn(kind, _f32(d), C.c_size_t(len(d)), _f32(c), len(c), d.shape[1], iterations, _u32(clusters), _f32(bounds))
    return clusters, c, bounds, np.float32(inertia)


def kmeans_fit_growbatch(dataset, order, batch, centers, kind=DIST_EMD):
    """Kmeans::fit_growbatch as coded (one pass over the first `batch` shuffled items, kmeans.rs:336-495) -> (clusters, new centers, bounds, (p, inertia))"""
    d, c = np.ascontiguousarray(dataset, dtype=np.float32), np.array(centers, dtype=np.float32, order="C")
    o = np.ascontiguousarray(order, dtype=np.uint32)
    clusters, bounds, stats = np.zeros(batch, dtype=np.uint32), np.zeros((batch, 2), dtype=np.float32), np.zeros(2, dtype=np.float32)
    lib().orc_kmeans_fit_growbatch(kind, _f32(d), C.c_size_t(len(d)), _u32(o), C.c_size_t(batch), _f32(c), len(c), d.shape[1], _u32(clusters), _f32(bounds), _f32(stats))
    return clusters, c, bounds, stats


# ---- cpu_soa: the non-strawman CPU baseline of bench.py (oracle/cpu_soa.c) -----------------------------------------------------------
def _cpu_tag():
    """a short hash of this host's CPU model + ISA flags: the SoA baseline is compiled -march=native, and the .so travels with the repository
    snapshot to a GPU box with a different CPU, where it must be rebuilt rather than executed"""
    import hashlib
    txt = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith(("model name", "flags")):
                txt += line
                if line.startswith("flags"):
                    break
    except OSError:
        pass
    return hashlib.sha1(txt.encode()).hexdigest()[:10]


_soa_lib = None


def soa_lib():
    global _soa_lib
    if _soa_lib is not None:
        return _soa_lib
    so = os.path.join(_HERE, "librs_soa_%s.so" % _cpu_tag())
    srcs = [os.path.join(_HERE, f) for f in ("cpu_soa.c", "rs_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        tmp = so + ".%d.tmp" % os.getpid()
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-trapping-math", "-Wall", "-pthread",
                               "-shared", "-o", tmp, os.path.join(_HERE, "cpu_soa.c"), "-lpthread"])
        os.replace(tmp, so)
    L = C.CDLL(so)
    L.soa_create.argtypes = [C.POINTER(Tree), C.c_size_t, C.POINTER(C.c_float), C.c_float, C.c_int]
    L.soa_create.restype = C.c_void_p
    L.soa_destroy.argtypes = [C.c_void_p]
    L.soa_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int]
    L.soa_get_node.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.soa_run.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.soa_table_bytes.argtypes = [C.c_void_p]
    L.soa_table_bytes.restype = C.c_size_t
    L.soae_create.argtypes = [C.POINTER(Tree), C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.c_float, C.c_int]
    L.soae_create.restype = C.c_void_p
    L.soae_destroy.argtypes = [C.c_void_p]
    L.soae_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int]
    L.soae_get_node.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.soae_run.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.soae_run.restype = C.c_int
    L.soae_table_bytes.argtypes = [C.c_void_p]
    L.soae_table_bytes.restype = C.c_size_t
    _soa_lib = L
    return L


class SoaSolver:
    """The lane-model sweep (full-width opponents, pass-through chance, i32 tables) on the GPU's SoA layout, vectorised over blocks of lanes,
    persistent threads with first-touch placement (oracle/cpu_soa.c).  Bench baseline only."""

    def __init__(self, tree, n_lanes, sign, scale=100.0, mode=UPD_CLAMP_I64):
        self.tree, self.n_lanes = tree, int(n_lanes)
        self._sign = np.ascontiguousarray(sign, dtype=np.float32)
        assert self._sign.size == self.n_lanes
        self._h = soa_lib().soa_create(C.byref(tree.t), self.n_lanes, _f32(self._sign), scale, mode)
        if not self._h:
            raise MemoryError("soa_create")

    def fill(self, seed, regret_range=(-10**6, 10**6), ssum_range=(0, 10**6), threads=1):
        soa_lib().soa_fill(self._h, seed, regret_range[0], regret_range[1], ssum_range[0], ssum_range[1], threads)

    def get_node(self, index, n_actions):
        r = np.zeros((n_actions, self.n_lanes), dtype=np.int32)
        s = np.zeros((n_actions, self.n_lanes), dtype=np.int32)
        soa_lib().soa_get_node(self._h, index, n_actions, _i32(r), _i32(s))
        return r, s

    def run(self, iterations, threads):
        soa_lib().soa_run(self._h, iterations, threads)

    @property
    def table_bytes(self):
        return int(soa_lib().soa_table_bytes(self._h))

    def destroy(self):
        if self._h:
            soa_lib().soa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class SoaEnumSolver:
    """The enumerating cfr() sweep over a multi-round tree (ENUM chance nodes, cfr.rs:502-522) on a block-major SoA layout: units of 16 clusters walked
    through the whole tree, boards enumerated in deal order (oracle/cpu_soa.c soae_*).  Bench baseline only (config 3's CPU denominator)."""

    def __init__(self, tree, boards, n_clusters, signs, scale=10000.0, mode=UPD_WRAP_I32):
        self.tree, self.boards, self.n_clusters = tree, [int(b) for b in boards], int(n_clusters)
        self._signs = [np.ascontiguousarray(x, dtype=np.float32) if x is not None else None for x in signs]
        for r, x in enumerate(self._signs):
            assert x is None or x.size == self.boards[r] * self.n_clusters
        arr = (C.POINTER(C.c_float) * len(self.boards))(*[_f32(x) if x is not None else C.POINTER(C.c_float)() for x in self._signs])
        self._h = soa_lib().soae_create(C.byref(tree.t), len(self.boards), (C.c_uint32 * len(self.boards))(*self.boards), self.n_clusters, arr, scale, mode)
        if not self._h:
            raise MemoryError("soae_create")

    def fill(self, seed, regret_range=(-10**6, 10**6), ssum_range=(0, 10**6), threads=1):
        return soa_lib().soae_fill(self._h, seed, regret_range[0], regret_range[1], ssum_range[0], ssum_range[1], threads)

    def get_node(self, index, n_actions, round_idx):
        n = self.boards[round_idx] * self.n_clusters
        r = np.zeros((n_actions, n), dtype=np.int32)
        s = np.zeros((n_actions, n), dtype=np.int32)
        soa_lib().soae_get_node(self._h, index, n_actions, round_idx, _i32(r), _i32(s))
        return r, s

    def run(self, iterations, threads):
        return soa_lib().soae_run(self._h, iterations, threads)

    @property
    def table_bytes(self):
        return int(soa_lib().soae_table_bytes(self._h))

    def destroy(self):
        if self._h:
            soa_lib().soae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
