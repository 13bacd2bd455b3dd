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
                               pub shard_world: i32, pub shard_rank: i32, pub shard_round: i32, pub shard_global_boards: u32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rs_deal_batch { pub n_deals: u32, pub d_cluster: [[*const u32; 2]; RS_MAX_ROUNDS] }

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
