//! rust/ffi.rs -- the binding a RustSolver maintainer would add (src/solver/gpu.rs) to route the
//! info-set hot path through librustsolver_amd.so.  DOCUMENTATION ONLY: there is no rustc in the
//! build image, so this file is not compiled or tested here; the same ABI is exercised from
//! C++ and ctypes.  Generated-by-hand equivalent of `bindgen include/rustsolver_amd.h`.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RS_MAX_ACTIONS: usize = 8;
pub const RS_MAX_ROUNDS: usize = 3;
pub const RS_MAX_SIZES: usize = 4;

#[repr(C)] pub struct rs_table { _p: [u8; 0] }
#[repr(C)] pub struct rs_tree { _p: [u8; 0] }
#[repr(C)] pub struct rs_solver { _p: [u8; 0] }
#[repr(C)] pub struct rs_comm { _p: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_tree_node {
    pub kind: i32, pub parent: i32, pub n_children: i32, pub children: [i32; RS_MAX_ACTIONS],
    pub index: i32, pub player: u8, pub round_idx: u8,
    pub action_kind: [i32; RS_MAX_ACTIONS], pub action_amt: [f64; RS_MAX_ACTIONS],
    pub value: u32, pub ttype: i32, pub last_to_act: u8, pub round: i32,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_node_desc { pub n_actions: u32, pub n_clusters: u32, pub n_boards: u32, pub player: u8, pub round_idx: u8 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_leaf_desc { pub kind: i32, pub d_buf: *const f32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_solver_params { pub scale: f32, pub mode: i32, pub chance_mode: i32, pub use_graph: i32, pub fuse_subtrees: i32, pub opp_mode: i32, pub sample_seed: u64,
                               pub shard_world: i32, pub shard_rank: i32, pub shard_round: i32, pub shard_global_boards: u32,
                               pub deal_offset: u32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_deal_batch { pub n_deals: u32, pub d_cluster: [[*const u32; 2]; RS_MAX_ROUNDS], pub d_prune: *const u8 /* per-deal prune flags, cfr.rs:213-221 */ }

pub const RS_I32: c_int = 0;
pub const RS_UPD_CLAMP_I64: c_int = 0;   // cfr.rs:413-464
pub const RS_UPD_WRAP_I32: c_int = 1;    // cfr.rs:612-621
pub const RS_UPD_PRUNE: c_int = 0x200;   // cfr.rs:352
pub const RS_LEAF_SIGN: i32 = 1;
pub const RS_CHANCE_PASS: i32 = 0;

#[link(name = "rustsolver_amd")]
extern "C" {
    pub fn rs_last_error() -> *const c_char;
    pub fn rs_tree_from_nodes(nodes: *const rs_tree_node, n_nodes: c_int, out: *mut *mut rs_tree) -> c_int;
    pub fn rs_tree_destroy(tree: *mut rs_tree);
    pub fn rs_create_infosets(tree: *const rs_tree, n_clusters: *const [[u32; 2]; RS_MAX_ROUNDS],
                              n_boards: *const [u32; RS_MAX_ROUNDS], dtype: c_int, device: c_int,
                              out: *mut *mut rs_table) -> c_int;
    pub fn rs_table_destroy(table: *mut rs_table);
    pub fn rs_table_lane_pitch(table: *const rs_table, node: c_int) -> usize;
    pub fn rs_get_infoset(table: *mut rs_table, node: c_int, board: c_int, cluster: c_int,
                          regrets: *mut c_void, strategy_sum: *mut c_void) -> c_int;
    pub fn rs_get_strategy(table: *mut rs_table, node: c_int, board: c_int, cluster: c_int, out: *mut f32) -> c_int;
    pub fn rs_get_final_strategy(table: *mut rs_table, node: c_int, board: c_int, cluster: c_int, out: *mut f32) -> c_int;
    pub fn rs_dmalloc(table: *mut rs_table, bytes: usize, d_out: *mut *mut c_void) -> c_int;
    pub fn rs_h2d(table: *mut rs_table, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn rs_update_node(table: *mut rs_table, node: c_int, d_action_utils: *const f32, d_reach: *const f32,
                          scale: f32, mode: c_int, d_node_util: *mut f32) -> c_int;
    pub fn rs_discount(table: *mut rs_table, d: f32) -> c_int;
    pub fn rs_solver_create(table: *mut rs_table, tree: *const rs_tree, leaves_p0: *const rs_leaf_desc,
                            leaves_p1: *const rs_leaf_desc, params: *const rs_solver_params,
                            out: *mut *mut rs_solver) -> c_int;
    pub fn rs_solver_create_deals(table: *mut rs_table, tree: *const rs_tree, deals: *const rs_deal_batch,
                                  leaves_p0: *const rs_leaf_desc, leaves_p1: *const rs_leaf_desc,
                                  params: *const rs_solver_params, out: *mut *mut rs_solver) -> c_int;
    pub fn rs_solver_destroy(solver: *mut rs_solver);
    pub fn rs_iterate(solver: *mut rs_solver, traverser: c_int, d_root_util: *mut f32) -> c_int;
    pub fn rs_train(solver: *mut rs_solver, iterations: u64, discount_interval: u64, discount_cap: u64) -> c_int;

    // rust_poker::hand_indexer_s (card_abstraction.rs:88-90,205; gen_abstraction/main.rs:91,117)
    pub fn rs_hand_indexer_create(rounds: c_int, cards_per_round: *const u8, out: *mut *mut rs_hand_indexer) -> c_int;
    pub fn rs_hand_indexer_destroy(indexer: *mut rs_hand_indexer);
    pub fn rs_hand_indexer_size(indexer: *const rs_hand_indexer, round: c_int) -> u64;
    pub fn rs_hand_index(indexer: *const rs_hand_indexer, round: c_int, cards: *const u8, n: usize, out: *mut u64) -> c_int;
    pub fn rs_hand_unindex(indexer: *const rs_hand_indexer, round: c_int, indices: *const u64, n: usize, cards_out: *mut u8) -> c_int;
    // ISOMORPHIC / EMD / OCHS (card_abstraction.rs:186-298); hands = HandRange.hands as (u8,u8) pairs
    pub fn rs_card_abs_create(betting_round: c_int, hands_p0: *const u8, n_hands_p0: usize, hands_p1: *const u8, n_hands_p1: usize,
                              initial_board_mask: u64, cluster_arr: *const u32, arr_len: usize, out: *mut *mut rs_card_abs) -> c_int;
    pub fn rs_card_abs_destroy(abs_: *mut rs_card_abs);
    pub fn rs_card_abs_size(abs_: *const rs_card_abs, player: c_int) -> usize;
    pub fn rs_card_abs_get_cluster(abs_: *const rs_card_abs, cards: *const u8, n: usize, player: c_int, out: *mut u32) -> c_int;
    pub fn rs_card_abs_clusters_device(abs_: *mut rs_card_abs, table: *mut rs_table, d_cards: *const u8, n_deals: u32,
                                       d_cluster_p0: *mut u32, d_cluster_p1: *mut u32) -> c_int;
    pub fn rs_card_abs_status(abs_: *mut rs_card_abs, table: *mut rs_table) -> c_int;
    // gen_abstraction: Kmeans::predict (kmeans.rs:173), update_min_dists (kmeans.rs:603), emd_1d (emd.rs:53) / l2_dist (kmeans.rs:622)
    pub fn rs_histogram_distance(dist: c_int, p: *const f32, q: *const f32, n_bins: c_int, out: *mut f32) -> c_int;
    pub fn rs_kmeans_predict(table: *mut rs_table, dist: c_int, d_dataset: *const f32, n: usize, centers: *const f32, n_centers: c_int,
                             n_bins: c_int, d_clusters: *mut u32, d_min_dist: *mut f32) -> c_int;
    pub fn rs_update_min_dists(table: *mut rs_table, dist: c_int, d_min_dists: *mut f32, d_dataset: *const f32, n: usize,
                               new_center: *const f32, n_bins: c_int) -> c_int;
    // generate_hand (cfr.rs:100-143) and the whole MCCFRTrainer loop (cfr.rs:159-297) on the device
    pub fn rs_deals_sample(table: *mut rs_table, seed: u64, first_deal: u64, board_mask: u64, d_hands_p0: *const u8, n_hands_p0: u32,
                           d_hands_p1: *const u8, n_hands_p1: u32, n_deals: u32, d_cards: *mut u8, d_err: *mut u32) -> c_int;
    pub fn rs_deals_prune_flags(table: *mut rs_table, seed: u64, first_deal: u64, prune_threshold: u64, n_deals: u32, d_flags: *mut u8) -> c_int;
    pub fn rs_deal_trainer_create(tree: *const rs_tree, card_abs: *const *mut rs_card_abs, n_rounds: c_int, hands_p0: *const u8,
                                  n_hands_p0: usize, hands_p1: *const u8, n_hands_p1: usize, params: *const rs_deal_trainer_params,
                                  device: c_int, out: *mut *mut rs_deal_trainer) -> c_int;
    pub fn rs_deal_trainer_destroy(trainer: *mut rs_deal_trainer);
    pub fn rs_deal_trainer_table(trainer: *mut rs_deal_trainer) -> *mut rs_table;
    pub fn rs_deal_trainer_train(trainer: *mut rs_deal_trainer, n_batches: u64) -> c_int;
    pub fn rs_deal_trainer_status(trainer: *mut rs_deal_trainer) -> c_int;
    // calc_br as coded (cfr.rs:629-745; printed at every discount tick, cfr.rs:244-246) and the best response it stands in for
    pub fn rs_table_tile_lanes(table: *const rs_table, node: c_int) -> usize;   // block layout [pitch / T][A][T]; T == pitch: plain rows
    pub fn rs_stream_probe(table: *mut rs_table, bytes: usize, reps: c_int, gbps: *mut f64) -> c_int;
    pub fn rs_calc_br(table: *mut rs_table, tree: *const rs_tree, out: *mut f32) -> c_int;
    pub fn rs_best_response(table: *mut rs_table, tree: *const rs_tree, board: *const u8, hands_p0: *const u8, n_hands_p0: usize,
                            cluster_p0: *const u32, hands_p1: *const u8, n_hands_p1: usize, cluster_p1: *const u32, mode: c_int,
                            out: *mut f64) -> c_int;
    pub fn rs_deal_trainer_set_tick_br(trainer: *mut rs_deal_trainer, enable: c_int) -> c_int;
    pub fn rs_deal_trainer_last_br(trainer: *const rs_deal_trainer, out: *mut f32, iterations: *mut u64) -> c_int;
    pub fn rs_deal_trainer_calc_br(trainer: *mut rs_deal_trainer, out: *mut f32) -> c_int;
    pub fn rs_deal_trainer_best_response(trainer: *mut rs_deal_trainer, mode: c_int, out: *mut f64) -> c_int;
}

#[repr(C)] pub struct rs_hand_indexer { _private: [u8; 0] }
#[repr(C)] pub struct rs_card_abs { _private: [u8; 0] }
#[repr(C)] pub struct rs_deal_trainer { _private: [u8; 0] }
#[repr(C)]
pub struct rs_deal_trainer_params {
    pub board_mask: u64,          // Options.board_mask (options.rs:17)
    pub deals_per_batch: u32,
    pub seed: u64,
    pub discount_interval: u64,   // cfr.rs:190
    pub discount_cap: u64,        // cfr.rs:240
    pub solver: rs_solver_params,
    pub world: u32,               // data-parallel training on replicated tables
    pub rank: u32,
    pub prune_threshold: u64,     // cfr.rs:190 PRUNE_THRESHOLD; u64::MAX = never
}

/// Flatten `Tree<GameTreeNode>` (tree.rs:14-17, nodes.rs:46-52) into the ABI's node array.
/// In MCCFRTrainer::init (cfr.rs:159-184) this replaces `create_infosets(n_actions, &game_tree, &card_abs)`:
///
/// ```ignore
/// let flat: Vec<rs_tree_node> = (0..game_tree.len()).map(|id| flatten(game_tree.get_node(id))).collect();
/// let mut tree = std::ptr::null_mut();
/// check(unsafe { rs_tree_from_nodes(flat.as_ptr(), flat.len() as c_int, &mut tree) });
/// let sizes = [[card_abs[0].get_size(0) as u32, card_abs[0].get_size(1) as u32]; RS_MAX_ROUNDS];
/// let mut table = std::ptr::null_mut();
/// check(unsafe { rs_create_infosets(tree, &sizes, &[n_boards; RS_MAX_ROUNDS], RS_I32, 0, &mut table) });
/// ```
///
/// and `train()` (cfr.rs:188) becomes `rs_train(solver, iterations, 100_000, 20_000_000)`;
/// `self.infosets[an.index][cluster_idx].get_strategy()` (cfr.rs:375-376) becomes
/// `rs_get_strategy(table, an.index, board, cluster_idx, out.as_mut_ptr())`.
pub fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rs_last_error()) }.to_string_lossy().into_owned();
        panic!("rustsolver_amd error {}: {}", rc, msg); // the reference panics on every error path
    }
}
