"""ctypes declarations for include/rustsolver_amd.h (the C ABI) and include/rustsolver_amd_diag.h (diagnostics).  No torch, no CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RS_LIB_PATH") or os.path.join(HERE, "librustsolver_amd.so")

MAX_ACTIONS, MAX_ROUNDS, MAX_SIZES, MAX_PLAYERS = 8, 3, 4, 2
COMM_ID_BYTES = 128

OK, ERR_INVALID, ERR_OOB, ERR_OOM, ERR_HIP, ERR_UNSUPPORTED, ERR_COMM, ERR_MISMATCH = 0, -1, -2, -3, -4, -5, -6, -7
NODE_PRIVATE_CHANCE, NODE_PUBLIC_CHANCE, NODE_ACTION, NODE_TERMINAL = 0, 1, 2, 3
TERM_ALLIN, TERM_SHOWDOWN, TERM_UNCONTESTED = 0, 1, 2
ACT_BET, ACT_RAISE, ACT_CHECK, ACT_CALL, ACT_FOLD = 0, 1, 2, 3, 4
I32, F32, F16 = 0, 1, 2
UPD_CLAMP_I64, UPD_WRAP_I32, UPD_RMPLUS, UPD_PRUNE = 0, 1, 0x100, 0x200
LEAF_UNCONTESTED, LEAF_SIGN, LEAF_UTIL = 0, 1, 2
FORM_DEFAULT, FORM_ON, FORM_OFF = 0, 1, 2            # rs_kernel_forms values (RS_FORM_*)
FAN_DEFAULT, FAN_NONE, FAN_EXPAND = 0, 1, 2          # .lane_fan (RS_FAN_*)
SHADOW_RULE, SHADOW_ALL = 1, 2                       # .shadow (RS_SHADOW_*)
CHANCE_PASS, CHANCE_ENUM = 0, 1
OPP_FULL, OPP_SAMPLE = 0, 1
BR_MAX, BR_AVERAGE = 0, 1   # rs_best_response modes
BR_SORTED = 0x100           # | into the mode: showdowns by rank order (O(n log n) per run-out), sums in a fixed order of their own
DIST_EMD, DIST_L2 = 0, 1
K_UPDATE, K_NODE_UTIL, K_REACH, K_CHANCE, K_DISCOUNT, K_STRATEGY, K_TREE, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7


class TreeNode(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("parent", C.c_int32), ("n_children", C.c_int32),
        ("children", C.c_int32 * MAX_ACTIONS),
        ("index", C.c_int32), ("player", C.c_uint8), ("round_idx", C.c_uint8),
        ("action_kind", C.c_int32 * MAX_ACTIONS), ("action_amt", C.c_double * MAX_ACTIONS),
        ("value", C.c_uint32), ("ttype", C.c_int32), ("last_to_act", C.c_uint8), ("round", C.c_int32),
    ]


class OptionsC(C.Structure):
    _fields_ = [
        ("stack_sizes", C.c_uint32 * MAX_PLAYERS), ("starting_pot", C.c_uint32), ("n_board_cards", C.c_int32),
        ("n_rounds", C.c_int32),
        ("n_bet_sizes", C.c_int32 * MAX_ROUNDS), ("bet_sizes", (C.c_double * MAX_SIZES) * MAX_ROUNDS),
        ("n_raise_sizes", C.c_int32 * MAX_ROUNDS), ("raise_sizes", (C.c_double * MAX_SIZES) * MAX_ROUNDS),
    ]


class NodeDesc(C.Structure):
    _fields_ = [("n_actions", C.c_uint32), ("n_clusters", C.c_uint32), ("n_boards", C.c_uint32),
                ("player", C.c_uint8), ("round_idx", C.c_uint8)]


class LeafDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d_buf", C.c_void_p)]


class DealBatch(C.Structure):
    _fields_ = [("n_deals", C.c_uint32), ("d_cluster", (C.c_void_p * MAX_PLAYERS) * MAX_ROUNDS), ("d_prune", C.c_void_p)]


class KernelForms(C.Structure):   # rs_kernel_forms: every field 0 = the engine's own choice
    _fields_ = [("lane_fan", C.c_int32), ("deals_per_thread", C.c_int32), ("kept_records", C.c_int32), ("shadow", C.c_int32),
                ("deal_order", C.c_int32), ("delta_rows", C.c_int32), ("direct_rows", C.c_int32), ("reserved", C.c_int32 * 1)]


class TableParams(C.Structure):
    _fields_ = [("tile_lanes", C.c_uint32), ("tile_min_lanes", C.c_uint32)]


class SolverParams(C.Structure):
    _fields_ = [("scale", C.c_float), ("mode", C.c_int32), ("chance_mode", C.c_int32), ("use_graph", C.c_int32),
                ("fuse_subtrees", C.c_int32), ("opp_mode", C.c_int32), ("sample_seed", C.c_uint64),
                ("shard_world", C.c_int32), ("shard_rank", C.c_int32), ("shard_round", C.c_int32), ("shard_global_boards", C.c_uint32),
                ("deal_offset", C.c_uint32), ("forms", KernelForms)]


class DealTrainerParams(C.Structure):
    _fields_ = [("board_mask", C.c_uint64), ("deals_per_batch", C.c_uint32), ("seed", C.c_uint64), ("discount_interval", C.c_uint64),
                ("discount_cap", C.c_uint64), ("solver", SolverParams), ("world", C.c_uint32), ("rank", C.c_uint32),
                ("prune_threshold", C.c_uint64), ("prefetch", C.c_int32), ("table_dtype", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * K_COUNT), ("ms", C.c_double * K_COUNT), ("algo_bytes", C.c_double * K_COUNT)]


# every symbol the header declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "rs_last_error": (C.c_char_p, []),
    "rs_abi_version": (C.c_int, []),
    "rs_solver_training_loop": (C.c_int, [C.c_void_p, C.c_int]),
    "rs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rs_options_default": (C.c_int, [C.POINTER(OptionsC)]),
    "rs_tree_build": (C.c_int, [C.POINTER(OptionsC), _PP]),
    "rs_tree_from_nodes": (C.c_int, [C.POINTER(TreeNode), C.c_int, _PP]),
    "rs_tree_destroy": (None, [_P]),
    "rs_tree_n_nodes": (C.c_int, [_P]),
    "rs_tree_n_action_nodes": (C.c_int, [_P]),
    "rs_tree_get_node": (C.c_int, [_P, C.c_int, C.POINTER(TreeNode)]),
    "rs_table_create": (C.c_int, [C.POINTER(NodeDesc), C.c_int, C.c_int, C.c_int, _PP]),
    "rs_table_create_with": (C.c_int, [C.POINTER(NodeDesc), C.c_int, C.c_int, C.c_int, C.POINTER(TableParams), _PP]),
    "rs_create_infosets": (C.c_int, [_P, C.POINTER((C.c_uint32 * MAX_PLAYERS) * MAX_ROUNDS), C.POINTER(C.c_uint32 * MAX_ROUNDS),
                                     C.c_int, C.c_int, _PP]),
    "rs_table_destroy": (None, [_P]),
    "rs_table_n_nodes": (C.c_int, [_P]),
    "rs_table_node_desc": (C.c_int, [_P, C.c_int, C.POINTER(NodeDesc)]),
    "rs_table_dtype": (C.c_int, [_P]),
    "rs_table_device": (C.c_int, [_P]),
    "rs_table_tile_lanes": (C.c_size_t, [_P, C.c_int]),
    "rs_table_lane_pitch": (C.c_size_t, [_P, C.c_int]),
    "rs_table_cells": (C.c_size_t, [_P]),
    "rs_table_cell_offset": (C.c_size_t, [_P, C.c_int]),
    "rs_table_bytes": (C.c_size_t, [_P]),
    "rs_table_upload": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "rs_table_download": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "rs_table_upload_node": (C.c_int, [_P, C.c_int, _P, _P]),
    "rs_table_download_node": (C.c_int, [_P, C.c_int, _P, _P]),
    "rs_get_infoset": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rs_get_infosets": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, _P]),
    "rs_selftest_division": (C.c_int, [_P, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
    "rs_table_checksum": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "rs_set_infoset": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rs_get_strategy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rs_get_final_strategy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rs_table_fill_random": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "rs_fill_uniform_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_float, C.c_float]),
    "rs_table_plant_saturating": (C.c_int, [_P, C.c_uint64, C.c_uint32]),
    "rs_fill_uniform_f32_at": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_float, C.c_float, C.c_uint64]),
    "rs_table_fill_random_logical": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_uint64)]),
    "rs_table_checksum_logical": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rs_plant_outliers_f32": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_uint32, C.c_float]),
    "rs_dmalloc": (C.c_int, [_P, C.c_size_t, _PP]),
    "rs_dfree": (C.c_int, [_P, _P]),
    "rs_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rs_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rs_dmemset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "rs_sync": (C.c_int, [_P]),
    "rs_stream": (C.c_void_p, [_P]),
    "rs_regret_match_node": (C.c_int, [_P, C.c_int, _P]),
    "rs_final_strategy_node": (C.c_int, [_P, C.c_int, _P]),
    "rs_final_strategy_all": (C.c_int, [_P, _P]),
    "rs_stream_probe": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "rs_calc_br": (C.c_int, [_P, _P, _P]),
    "rs_best_response": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, _P, C.c_size_t, _P, C.c_int, _P]),
    "rs_update_node": (C.c_int, [_P, C.c_int, _P, _P, C.c_float, C.c_int, _P]),
    "rs_node_util": (C.c_int, [_P, C.c_int, _P, _P]),
    "rs_child_reach": (C.c_int, [_P, C.c_int, _P, _P]),
    "rs_discount": (C.c_int, [_P, C.c_float]),
    "rs_discount_factor": (C.c_float, [C.c_uint64, C.c_uint64]),
    "rs_solver_create": (C.c_int, [_P, _P, C.POINTER(LeafDesc), C.POINTER(LeafDesc), C.POINTER(SolverParams), _PP]),
    "rs_solver_create_deals": (C.c_int, [_P, _P, C.POINTER(DealBatch), C.POINTER(LeafDesc), C.POINTER(LeafDesc),
                                        C.POINTER(SolverParams), _PP]),
    "rs_solver_destroy": (None, [_P]),
    "rs_iterate": (C.c_int, [_P, C.c_int, _P]),
    "rs_train": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "rs_solver_attach_comm": (C.c_int, [_P, _P]),
    "rs_iterate_phase": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "rs_solver_exchange_info": (C.c_int, [_P, C.c_int, _PP, C.POINTER(C.c_size_t)]),
    "rs_comm_allgather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rs_comm_allreduce_deltas": (C.c_int, [_P, _P]),
    "rs_table_deltas": (C.c_int, [_P, _PP, _PP]),
    "rs_deal_trainer_attach_comm": (C.c_int, [_P, _P]),
    "rs_deal_trainer_finish_batch": (C.c_int, [_P]),
    "rs_solver_workspace_bytes": (C.c_size_t, [_P]),
    "rs_solver_n_launches": (C.c_int, [_P, C.c_int]),
    "rs_solver_walk_counts": (C.c_int, [_P, C.c_int, _P]),
    "rs_solver_exchange_bytes": (C.c_int, [_P, _P, _P]),
    "rs_solver_forms": (C.c_int, [_P]),
    "rs_jit_available": (C.c_int, []),
    "rs_jit_check_tree": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rs_jit_check_tree_deals": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rs_cluster_file_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]),
    "rs_cluster_file_write": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.c_size_t]),
    "rs_free_u32": (None, [C.POINTER(C.c_uint32)]),
    "rs_index_to_cluster": (C.c_int, [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64)]),
    "rs_dense_map_create": (C.c_int, [C.POINTER(C.c_uint64), C.c_size_t, _PP]),
    "rs_dense_map_destroy": (None, [_P]),
    "rs_dense_map_size": (C.c_size_t, [_P]),
    "rs_dense_map_lookup": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint32)]),
    "rs_dense_map_keys": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "rs_hand_indexer_create": (C.c_int, [C.c_int, C.POINTER(C.c_uint8), _PP]),
    "rs_hand_indexer_destroy": (None, [_P]),
    "rs_hand_indexer_size": (C.c_uint64, [_P, C.c_int]),
    "rs_hand_indexer_rounds": (C.c_int, [_P]),
    "rs_hand_indexer_n_cards": (C.c_int, [_P, C.c_int]),
    "rs_hand_index": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P]),
    "rs_hand_unindex": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P]),
    "rs_hand_index_verify": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, _P, _P]),
    "rs_hand_index_device": (C.c_int, [_P, _P, C.c_int, _P, C.c_uint32, _P]),
    "rs_card_abs_create": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.c_uint64, _P, C.c_size_t, _PP]),
    "rs_card_abs_destroy": (None, [_P]),
    "rs_card_abs_round": (C.c_int, [_P]),
    "rs_card_abs_size": (C.c_size_t, [_P, C.c_int]),
    "rs_card_abs_index_size": (C.c_uint64, [_P]),
    "rs_card_abs_keys": (C.c_int, [_P, C.c_int, _P]),
    "rs_card_abs_get_cluster": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    "rs_card_abs_clusters_device": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    "rs_card_abs_status": (C.c_int, [_P, _P]),
    "rs_deals_sample": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "rs_deals_prune_flags": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "rs_deal_trainer_create": (C.c_int, [_P, _PP, C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(DealTrainerParams), C.c_int, _PP]),
    "rs_deal_trainer_destroy": (None, [_P]),
    "rs_deal_trainer_table": (C.c_void_p, [_P]),
    "rs_deal_trainer_solver": (C.c_void_p, [_P]),
    "rs_deal_trainer_train": (C.c_int, [_P, C.c_uint64]),
    "rs_deal_trainer_deal": (C.c_int, [_P]),
    "rs_deal_trainer_status": (C.c_int, [_P]),
    "rs_deal_trainer_iterations": (C.c_uint64, [_P]),
    "rs_deal_trainer_set_tick_br": (C.c_int, [_P, C.c_int]),
    "rs_deal_trainer_last_br": (C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    "rs_deal_trainer_calc_br": (C.c_int, [_P, _P]),
    "rs_deal_trainer_best_response": (C.c_int, [_P, C.c_int, _P]),
    "rs_deal_trainer_br_bytes": (C.c_size_t, [_P]),
    "rs_deal_trainer_br_release": (C.c_int, [_P]),
    "rs_deal_trainer_br_launches": (C.c_int, [_P, C.c_int]),
    "rs_deal_trainer_cards": (C.c_void_p, [_P]),
    "rs_deal_trainer_signs": (C.c_void_p, [_P]),
    "rs_deal_trainer_prune_flags": (C.c_void_p, [_P]),
    "rs_deal_trainer_clusters": (C.c_void_p, [_P, C.c_int, C.c_int]),
    "rs_histogram_distance": (C.c_int, [C.c_int, _P, _P, C.c_int, C.POINTER(C.c_float)]),
    "rs_kmeans_predict": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, C.c_int, C.c_int, _P, _P]),
    "rs_update_min_dists": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_int]),
    "rs_showdown_sign": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "rs_best_response_rounds": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "rs_br_runouts": (C.c_size_t, [_P, C.c_int, _P]),
    "rs_kmeans_init_s": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P]),
    "rs_kmeans_pick_restart": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_int)]),
    "rs_kmeans_reassign": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_int, C.c_int, _P, _P, _P]),
    "rs_kmeans_fit_regular": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.POINTER(C.c_float)]),
    "rs_kmeans_fit_growbatch": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_int, C.c_int, _P, _P, _P]),
    "rs_table_save": (C.c_int, [_P, C.c_char_p]),
    "rs_table_load": (C.c_int, [C.c_char_p, C.c_int, _PP]),
    "rs_profile_enable": (C.c_int, [_P, C.c_int]),
    "rs_profile_read": (C.c_int, [_P, C.POINTER(Profile)]),
    "rs_profile_reset": (C.c_int, [_P]),
    "rs_profile_mark": (C.c_int, [_P]),
    "rs_profile_marks": (C.c_int, [_P, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]),
    "rs_comm_unique_id": (C.c_int, [_P]),
    "rs_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, _PP]),
    "rs_comm_destroy": (None, [_P]),
    "rs_replicated_begin": (C.c_int, [_P, C.c_uint32]),
    "rs_allreduce_replicated": (C.c_int, [_P, _P, C.c_uint32]),
}

_lib = None


class RsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rustsolver_amd error %d: %s" % (code, msg))
        self.code = code


def load():
    """Loads librustsolver_amd.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("rustsolver_amd: %s is missing -- build it with `python -m rustsolver_amd.build` "
                          "(hipcc, gfx950). There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise RsError(rc, load().rs_last_error().decode("utf-8", "replace"))
    return rc
