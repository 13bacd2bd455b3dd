/*
 * rustsolver_amd.h -- C ABI of the MI355X-native CFR regret/strategy-update engine.
 *
 * This is the drop-in boundary for RustSolver's hot path.  The reference has no FFI of its own
 * (it is a single Rust crate); every entry point below replaces a crate-internal Rust item and
 * cites it (paths relative to the reference repository root).  The Rust-side binding a
 * maintainer would add is shown in INTEGRATION.md and rust/ffi.rs (generated from this header by
 * tools/gen_rust_ffi.py).  Bench / test diagnostics (synthetic fills, HIP-event profiling, the
 * stream probe, compile-only checks of the generated kernels, table checksums) are NOT part of
 * the drop-in surface: they live in rustsolver_amd_diag.h.
 *
 * Conventions
 *   - plain C: opaque handles, pointers and sizes only; no C++/torch types;
 *   - every call returns an int status (RS_OK = 0, negative = error); rs_last_error() gives the
 *     text for the calling thread.  Where the reference panics (index out of bounds, invalid
 *     board mask, ...) this library returns an error instead;
 *   - the caller owns every host buffer; pointers named d_* are DEVICE pointers (from
 *     rs_dmalloc or the caller's own hipMalloc on the same device);
 *   - one rs_table lives on one GPU and owns one HIP stream; all calls on a table (and on
 *     solvers built on it) are enqueued on that stream in call order.  Calls that return
 *     host data synchronise the stream; the others are asynchronous;
 *   - thread-safe across different tables, not for concurrent calls on the same table;
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point fails
 *     with RS_ERR_HIP.
 *
 * Data layout in HBM: per action node two arrays  regrets[A][pitch]  and
 * strategy_sum[A][pitch]; the fast axis is the LANE  lane = board * n_clusters + cluster  of
 * that node's round, pitch = rs_table_lane_pitch() >= n_boards*n_clusters (multiple of 64).
 * Every per-lane device vector passed through this ABI (utilities, reach, strategies) uses
 * the same pitch:  float buf[pitch]  or  float buf[A][pitch].
 */
#ifndef RUSTSOLVER_AMD_H
#define RUSTSOLVER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ABI_VERSION 6   /* 2: rs_deal_batch.d_prune, rs_deal_trainer_params.prune_threshold, tiled node blocks (rs_table_tile_lanes); 3: rs_get_infosets, diagnostics split into rustsolver_amd_diag.h;
                              4: rs_kernel_forms inside rs_solver_params, rs_table_params + rs_table_create_with, rs_deal_trainer_params.prefetch, f32 deal batches;
                              5: rs_kernel_forms without `worklist` and RS_FAN_LOOP / RS_SHADOW_WIDE, with direct_rows and kept_records (same size: a zeroed struct means what it meant);
                                 rs_hand_index_verify, rs_deal_trainer_br_bytes / _br_release / _br_launches, RS_ERR_MISMATCH;
                              6: rs_deal_trainer_params.table_dtype (was `reserved`: zero = RS_I32 means what it meant), float deal tables RS_F16 and RS_UPD_RMPLUS,
                                 rs_solver_exchange_bytes (diagnostics), data-parallel sweeps keep direct rows */
#define RS_MAX_ACTIONS 8
#define RS_MAX_ROUNDS 3
#define RS_MAX_SIZES 4
#define RS_MAX_PLAYERS 2

typedef struct rs_table rs_table;
typedef struct rs_tree rs_tree;
typedef struct rs_solver rs_solver;
typedef struct rs_comm rs_comm;

/* ---- status ------------------------------------------------------------------------------ */
enum {
    RS_OK = 0,
    RS_ERR_INVALID = -1,     /* bad argument (NULL, negative size, mismatched shapes) */
    RS_ERR_OOB = -2,         /* node / board / cluster index out of bounds (Rust: index panic) */
    RS_ERR_OOM = -3,         /* host or device allocation failed */
    RS_ERR_HIP = -4,         /* HIP runtime error or no usable GPU */
    RS_ERR_UNSUPPORTED = -5, /* combination not implemented (e.g. prune on float tables) */
    RS_ERR_COMM = -6,        /* RCCL error */
    RS_ERR_MISMATCH = -7     /* a self-check found a disagreement (rs_hand_index_verify) */
};
const char *rs_last_error(void);
int rs_abi_version(void);
int rs_device_count(int *out);

/* ---- public tree: tree.rs, nodes.rs, tree_builder.rs, state.rs, options.rs ------------------
 * The tree stays on the host.  Either adopt the tree the Rust side already built
 * (rs_tree_from_nodes) or let this library rebuild it from Options (rs_tree_build reproduces
 * build_game_tree, tree_builder.rs:9-143, including the pre-order ActionNode.index numbering
 * and the valid_actions child order Check, Call, Fold, Bet.., Raise.., state.rs:125-157). */
enum { RS_NODE_PRIVATE_CHANCE = 0, RS_NODE_PUBLIC_CHANCE = 1, RS_NODE_ACTION = 2, RS_NODE_TERMINAL = 3 }; /* nodes.rs:46-52 */
enum { RS_TERM_ALLIN = 0, RS_TERM_SHOWDOWN = 1, RS_TERM_UNCONTESTED = 2 };                                /* nodes.rs:16-21 */
enum { RS_ACT_BET = 0, RS_ACT_RAISE = 1, RS_ACT_CHECK = 2, RS_ACT_CALL = 3, RS_ACT_FOLD = 4 };            /* action_abstraction.rs:4-10 */

typedef struct rs_tree_node {
    int32_t kind;                       /* RS_NODE_* */
    int32_t parent;                     /* -1 for the root (tree.rs:22) */
    int32_t n_children;
    int32_t children[RS_MAX_ACTIONS];   /* tree.rs:21, child i = action i */
    int32_t index;                      /* ActionNode.index (nodes.rs:7), -1 otherwise */
    uint8_t player;                     /* ActionNode.player (nodes.rs:8) */
    uint8_t round_idx;                  /* ActionNode.round_idx (nodes.rs:9), relative to the first street */
    int32_t action_kind[RS_MAX_ACTIONS];/* ActionNode.actions (nodes.rs:6) */
    double action_amt[RS_MAX_ACTIONS];
    uint32_t value;                     /* TerminalNode.value = pot (nodes.rs:34) */
    int32_t ttype;                      /* TerminalNode.ttype (nodes.rs:35) */
    uint8_t last_to_act;                /* TerminalNode.last_to_act (nodes.rs:36) */
    int32_t round;                      /* TerminalNode.round / PublicChanceNode.round: 0 flop, 1 turn, 2 river */
} rs_tree_node;

typedef struct rs_options {             /* options.rs:10-28, the fields build_game_tree reads */
    uint32_t stack_sizes[RS_MAX_PLAYERS];
    uint32_t starting_pot;
    int32_t n_board_cards;              /* board_mask.count_ones(): 3 / 4 / 5 (state.rs:60-65) */
    int32_t n_rounds;                   /* action_abstraction.bet_sizes.len() */
    int32_t n_bet_sizes[RS_MAX_ROUNDS];
    double bet_sizes[RS_MAX_ROUNDS][RS_MAX_SIZES];
    int32_t n_raise_sizes[RS_MAX_ROUNDS];
    double raise_sizes[RS_MAX_ROUNDS][RS_MAX_SIZES];
} rs_options;

int rs_options_default(rs_options *out);                                   /* options::default_flop(), options.rs:52-81 */
int rs_tree_build(const rs_options *options, rs_tree **out);               /* build_game_tree, tree_builder.rs:9 */
int rs_tree_from_nodes(const rs_tree_node *nodes, int n_nodes, rs_tree **out); /* adopt a Tree<GameTreeNode> (tree.rs:14-17) */
void rs_tree_destroy(rs_tree *tree);
int rs_tree_n_nodes(const rs_tree *tree);
int rs_tree_n_action_nodes(const rs_tree *tree);                           /* the `n_actions` of tree_builder.rs:13 */
int rs_tree_get_node(const rs_tree *tree, int node_id, rs_tree_node *out); /* Tree::get_node, tree.rs:57 */

/* ---- info-set table: infoset.rs ---------------------------------------------------------- */
enum { RS_I32 = 0,   /* reference: Box<[i32]> regrets / strategy_sum (infoset.rs:63-67) */
       RS_F32 = 1,   /* extension: f32 tables */
       RS_F16 = 2 }; /* extension: binary16 tables, f32 arithmetic */

typedef struct rs_node_desc {           /* one row of InfosetTable = Vec<Vec<Infoset>> (infoset.rs:6) */
    uint32_t n_actions;                 /* node.children.len() (infoset.rs:33) */
    uint32_t n_clusters;                /* card_abs[round_idx].get_size(player) (infoset.rs:28-32) */
    uint32_t n_boards;                  /* README.md:31-54 [board] axis; 1 = the reference as coded */
    uint8_t player;
    uint8_t round_idx;
} rs_node_desc;

/* order of `nodes` = ActionNode.index.  Zero-initialised like Infoset::init (infoset.rs:76-81). */
int rs_table_create(const rs_node_desc *nodes, int n_nodes, int dtype, int device, rs_table **out);
/* the same with an explicit block layout: lanes per tile of a tiled node block (a power of two >= 64; 0 = the default, 16 384; UINT32_MAX = never tile) and the
 * narrowest node that is tiled (0 = the default, 2^20 lanes).  rs_table_create / rs_create_infosets use the defaults. */
typedef struct rs_table_params {
    uint32_t tile_lanes;
    uint32_t tile_min_lanes;
} rs_table_params;
int rs_table_create_with(const rs_node_desc *nodes, int n_nodes, int dtype, int device, const rs_table_params *params, rs_table **out);
/* create_infosets(n_actions, tree, card_abs) (infoset.rs:8-49): sizes from the tree, the per-round
 * cluster counts n_clusters[round_idx][player] and board counts n_boards[round_idx]. */
int rs_create_infosets(const rs_tree *tree, const uint32_t n_clusters[RS_MAX_ROUNDS][RS_MAX_PLAYERS],
                       const uint32_t n_boards[RS_MAX_ROUNDS], int dtype, int device, rs_table **out);
void rs_table_destroy(rs_table *table);

int rs_table_n_nodes(const rs_table *table);
int rs_table_node_desc(const rs_table *table, int node, rs_node_desc *out);
int rs_table_dtype(const rs_table *table);
int rs_table_device(const rs_table *table);
size_t rs_table_lane_pitch(const rs_table *table, int node);  /* elements; 0 on error */
/* Layout of a node's block inside the two table arrays, for callers that address cells themselves (everything in this ABI does it for them):
 * T = rs_table_tile_lanes(node).  T == pitch: the plain block [A][pitch].  T < pitch (nodes of at least 2^20 lanes; T = 16 384): the rows are
 * interleaved tile by tile, [pitch / T][A][T], i.e. cell (action a, lane l) is element rs_table_cell_offset(node) + ((l / T) * A + a) * T + l % T --
 * a sweep streams all rows of a node together, and rows that sit next to each other move 22-35 % faster than rows tens of MB apart. */
size_t rs_table_tile_lanes(const rs_table *table, int node);
size_t rs_table_cells(const rs_table *table);                 /* sum over nodes of n_actions * pitch */
size_t rs_table_cell_offset(const rs_table *table, int node); /* element offset of a node's block (A * pitch elements) */
size_t rs_table_bytes(const rs_table *table);                 /* device bytes of both arrays */

/* Host <-> device copies.  Element type on the host: int32_t for RS_I32, float for RS_F32 and
 * RS_F16 (converted with round-to-nearest-even).  Either pointer may be NULL to skip that array.
 * Per (node, board): host layout [A][n_clusters].  Per node: host layout [A][n_boards*n_clusters]. */
int rs_table_upload(rs_table *table, int node, int board, const void *regrets, const void *strategy_sum);
int rs_table_download(rs_table *table, int node, int board, void *regrets, void *strategy_sum);
int rs_table_upload_node(rs_table *table, int node, const void *regrets, const void *strategy_sum);
int rs_table_download_node(rs_table *table, int node, void *regrets, void *strategy_sum);

/* get-infoset: `&self.infosets[an.index][cluster_idx]` (cfr.rs:375) with its two pub fields
 * (infoset.rs:65-66); out arrays of n_actions elements (host type as above). */
int rs_get_infoset(rs_table *table, int node, int board, int cluster, void *regrets, void *strategy_sum);
int rs_set_infoset(rs_table *table, int node, int board, int cluster, const void *regrets, const void *strategy_sum);
int rs_get_strategy(rs_table *table, int node, int board, int cluster, float *out);       /* Infoset::get_strategy, infoset.rs:83-102 */
int rs_get_final_strategy(rs_table *table, int node, int board, int cluster, float *out); /* Infoset::get_final_strategy, infoset.rs:104-123 */

/* get-infoset for a batch: the info sets of lanes[0..n) (lane = board * n_clusters + cluster, HOST array) of one node in one call -- what n calls of
 * rs_get_infoset would return, host layout [A][n] per array (host type as above).  Either out pointer may be NULL. */
int rs_get_infosets(rs_table *table, int node, const uint32_t *lanes, size_t n, void *regrets, void *strategy_sum);

/* ---- device memory helpers (for hosts without their own HIP binding) ------------------------ */
int rs_dmalloc(rs_table *table, size_t bytes, void **d_out);
int rs_dfree(rs_table *table, void *d_ptr);
int rs_h2d(rs_table *table, void *d_dst, const void *src, size_t bytes);
int rs_d2h(rs_table *table, void *dst, const void *d_src, size_t bytes);
int rs_dmemset(rs_table *table, void *d_dst, int byte, size_t bytes);
int rs_sync(rs_table *table);
void *rs_stream(rs_table *table);   /* the table's hipStream_t */

/* ---- bulk kernels on one action node ---------------------------------------------------------- */
enum {
    RS_UPD_CLAMP_I64 = 0,       /* mccfr traverser block, cfr.rs:413-464: i64 add, clamp to i32 (reference scale 100.0) */
    RS_UPD_WRAP_I32 = 1,        /* cfr action block, cfr.rs:612-621: saturating `as i32`, wrapping += (reference scale 10000.0) */
    RS_UPD_ARITH_MASK = 0xff,
    RS_UPD_RMPLUS = 0x100,      /* extension: regrets floored at 0 on write (regret matching+) */
    RS_UPD_PRUNE = 0x200        /* cfr.rs:352,:379-386,:419-441: skip actions with regret <= -10 000 000 */
};

/* bulk get_strategy over every lane of a node (infoset.rs:83-102): d_strategy[A][pitch] */
int rs_regret_match_node(rs_table *table, int node, float *d_strategy);
/* bulk get_final_strategy (infoset.rs:104-123): d_strategy[A][pitch] */
int rs_final_strategy_node(rs_table *table, int node, float *d_strategy);
/* every node: d_out[rs_table_cells()] with node blocks at rs_table_cell_offset(), each a plain [A][pitch] block whatever the table's own tiling
 * (calc_br's reader, cfr.rs:669-672) */
int rs_final_strategy_all(rs_table *table, float *d_out);
/* MCCFRTrainer::calc_br exactly AS CODED (cfr.rs:629-745): out[0], out[1] = the two numbers train() prints at every discount tick
 * (cfr.rs:244-246).  Its op vectors have length 1 (cfr.rs:631), so only get_final_strategy() of bucket 0 of every action node
 * enters (cfr.rs:677-679) and a terminal pays op * (+-)pot / op (cfr.rs:703-741): a placeholder, reproduced in its f32 operation
 * order.  One gather kernel over the table, a few floats back.  Synchronises the table's stream. */
int rs_calc_br(rs_table *table, const rs_tree *tree, float *out /*[2]*/);
/* What that placeholder stands in for: out[p] = expected utility per deal, in the trainer's leaf
 * utilities (cfr.rs:314-348), of player p against the opponent's AVERAGE strategy when p plays
 *   RS_BR_MAX      a best response inside the abstraction (one action per info set = cluster), or
 *   RS_BR_AVERAGE  its own average strategy (out[0] + out[1] = 0: the game is zero-sum),
 * so (out[0] + out[1]) / 2 under RS_BR_MAX is the exploitability of the average strategy profile.  Vector form over both ranges
 * of a single-round tree on a full board (the configuration main.rs runs): deals weigh as generate_hand draws them (cfr.rs:124-137),
 * hands[n][2] hole cards, cluster[n] = get_cluster(hole cards + board, player) of every hand, board[5].  f64 on the device, every sum
 * in a fixed order.  Host pointers; synchronises the table's stream.  RS_ERR_UNSUPPORTED for trees with public chance nodes. */
enum { RS_BR_MAX = 0, RS_BR_AVERAGE = 1,
       RS_BR_SORTED = 0x100 };   /* OR into the mode: showdown and fold leaves by RANK ORDER -- the opponent's hands of a run-out sorted by score once per call, a leaf = a
                                    difference of prefix sums of the opponent's reach over that order, corrected for the hands that hold one of the traverser's cards --
                                    O(n log n) per run-out instead of the pair loop of cfr.rs:323-347 (full 1 176-combo ranges from a flop: 3.1 s -> ~0.2 s per call).  The
                                    sums run in a fixed order of their own: equal to the pair loop within f64 rounding (1e-12 relative), identical to the oracle's sorted mode */
int rs_best_response(rs_table *table, const rs_tree *tree, const uint8_t *board, const uint8_t *hands_p0, size_t n_hands_p0, const uint32_t *cluster_p0,
                     const uint8_t *hands_p1, size_t n_hands_p1, const uint32_t *cluster_p1, int mode, double *out /*[2]*/);
/* The same over MULTI-ROUND trees (flop or turn start).  A lane is (run-out b, hand h): generate_hand (cfr.rs:100-143) completes the board to five cards first --
 * ordered sequences without replacement, uniform -- then draws the hands; every showdown compares seven-card hands on the full board and chance nodes pass through
 * (cfr.rs:306-313).  cluster[r * 2 + p] (HOST) = dense cluster ids of player p in betting round r, [prefixes of round r][n_hands_p]: the prefix of a run-out is its
 * first r new cards, prefixes and run-outs enumerated with the first new card most significant, cards ascending among those still in the deck (rs_br_runouts writes
 * the run-outs, [NB][5], and returns NB = 1, 48 or 2 352); entries of (prefix, hand) pairs that share a card are ignored.  RS_BR_MAX: at each of p's nodes every cluster
 * takes the action with the largest SUM of its lanes' counterfactual values over all run-outs and hands -- the best response inside the abstraction when it has perfect
 * recall, otherwise the value of a valid pure strategy of the abstracted game (a lower bound).  f64, fixed-order sums; synchronises. */
int rs_best_response_rounds(rs_table *table, const rs_tree *tree, const uint8_t *board0, int n_board0, const uint8_t *hands_p0, size_t n_hands_p0,
                            const uint8_t *hands_p1, size_t n_hands_p1, const uint32_t *const *cluster, int n_rounds, int mode, double *out /*[2]*/);
size_t rs_br_runouts(const uint8_t *board0, int n_board0, uint8_t *out_cards /* [NB][5], may be NULL */);

/* One traverser visit of every lane of `node` (cfr.rs:370-466 / :571-623):
 *   sigma = get_strategy(); util = sum_a utils[a]*sigma[a]; regrets / strategy_sum updated with
 *   (scale*reach)*(utils[a]-util) and (scale*reach)*sigma[a] in the arithmetic of `mode`.
 * d_action_utils[A][pitch], d_reach[pitch] (NULL = 1.0 for every lane), d_node_util[pitch] (NULL = discard). */
int rs_update_node(rs_table *table, int node, const float *d_action_utils, const float *d_reach, float scale, int mode,
                   float *d_node_util);
/* opponent / read-only visit: util = sum_a utils[a]*sigma[a] (cfr.rs:574,:588,:608-610), no table write */
int rs_node_util(rs_table *table, int node, const float *d_action_utils, float *d_node_util);
/* opponent reach: d_child_reach[A][pitch] = sigma[a] * d_reach (cfr.rs:585; NULL reach = 1.0) */
int rs_child_reach(rs_table *table, int node, const float *d_reach, float *d_child_reach);

/* discount sweep, cfr.rs:250-261: x = ((x as f32) * d) as i32 for every regret and strategy_sum */
int rs_discount(rs_table *table, float d);
/* cfr.rs:248-249: p = (tc / interval) as f32; d = p / (p + 1.0) */
float rs_discount_factor(uint64_t tc, uint64_t interval);

/* ---- iterate: MCCFRTrainer (cfr.rs) ------------------------------------------------------------
 * Lane model: lane (board b, cluster c) is one scalar cfr() traversal (cfr.rs:481-627)
 * in which get_cluster() (cfr.rs:564-568) returns c for either player and evaluate() is replaced by
 * the leaf inputs below.  The host walks the public tree once to build a launch plan (per tree
 * depth: opponent-reach kernels top-down, node-util / update kernels bottom-up) and replays it. */
enum { RS_LEAF_UNCONTESTED = 0, /* +-pot from TerminalNode.last_to_act (cfr.rs:316-322); no buffer */
       RS_LEAF_SIGN = 1,        /* d_buf[pitch] = sign(score[0]-score[1]) of evaluate() (cfr.rs:323-347): value = +-pot / 0 */
       RS_LEAF_UTIL = 2 };      /* d_buf[pitch] = utility from the traverser's point of view, used verbatim */
enum { RS_OPP_FULL = 0,         /* cfr(): every opponent action is recursed, reach * sigma[i], util = sum (cfr.rs:576-589) */
       RS_OPP_SAMPLE = 1 };     /* mccfr(): ONE opponent action sampled from sigma with rand's WeightedIndex (cfr.rs:467-476);
                                   the random bits are a counter hash of (sweep seed, ActionNode.index, lane), the sweep seed
                                   advances by one on every rs_iterate call (sample_seed, call index) */
enum { RS_CHANCE_PASS = 0,      /* mccfr: PublicChance goes to child 0 (cfr.rs:306-309); needs equal board counts */
       RS_CHANCE_ENUM = 1 };    /* cfr: reach *= 1/len, util = sum over the len = n_boards[r+1]/n_boards[r] deals (cfr.rs:502-522) */

typedef struct rs_leaf_desc {
    int32_t kind;
    const float *d_buf;
} rs_leaf_desc;

/* Kernel-form choices a caller may legitimately make.  EVERY field: 0 = the engine's own choice (what a zeroed struct gets), so a caller that never looks at
 * this struct loses nothing.  Results are bit-identical whatever is chosen here (the GPU tests run the forms against each other and against the oracle); only
 * time and workspace change.  The remaining A/B switches of the kernel generator are test-only and come from the environment, read in one place
 * (csrc/rs_knobs.cpp). */
enum { RS_FORM_DEFAULT = 0, RS_FORM_ON = 1, RS_FORM_OFF = 2 };
enum { RS_FAN_DEFAULT = 0,   /* = RS_FAN_EXPAND */
       RS_FAN_NONE = 1,      /* lane sweeps, subtree directly below an ENUM chance node (cfr.rs:502-522): separate expand / reduce launches */
       RS_FAN_EXPAND = 2 };  /* the subtree's kernel scales the chance node's own reach row by 1/len itself: no expand launch, no per-deal reach rows */
enum { RS_SHADOW_DEFAULT = 0,   /* = RS_SHADOW_RULE */
       RS_SHADOW_RULE = 1,      /* deal sweeps: a node keeps an AoS shadow only while the batch is likely to read it (n_deals * 8 >= its cells * round subtrees) */
       RS_SHADOW_ALL = 2 };     /* a shadow for every node */
typedef struct rs_kernel_forms {
    int32_t lane_fan;           /* RS_FAN_* */
    int32_t deals_per_thread;   /* deal sweeps: 1, 2 or 4 deals per thread of the generated kernels */
    int32_t kept_records;       /* RS_FORM_*: the nodes of the rounds that `direct_rows` covers KEEP their shadow records between sweeps (wide records, one set for both
                                   traversers): the pass that adds the delta rows into the table adds them to the records too, rs_discount sweeps them as well, any other
                                   write to the table has them rebuilt before the next sweep; inside rs_train / rs_deal_trainer_train they are the working copy and the table's
                                   rows of those nodes are written back when the loop returns.  The walks then read staged rows instead of gathering a table that is too large
                                   to transpose per sweep (solve_three_street, 64 K deals against 2 GB: 1.60 -> 1.23 ms per batch, 1.28 -> 0.81 without discount ticks).
                                   Costs the records' memory (1.3x the nodes' table rows) -- and, for a host that drives rs_iterate and rs_discount ITSELF, a discount
                                   sweep over table and records both (1.6 -> 1.9 ms per batch while a tick comes every 1.5 batches, 1.3 -> 0.95 after the discount
                                   phase): such a loop should call rs_train / rs_deal_trainer_train for its batches, or switch this off for short runs.  Default: on.  (ABI 4 had `worklist` in this slot.) */
    int32_t shadow;             /* RS_SHADOW_* */
    int32_t deal_order;         /* RS_FORM_*: sampled deal sweeps walk the batch in the order of the traverser's last-round cluster id, and the last round's subtrees
                                   sum their deltas along the runs of equal cluster (DPP segmented scan) instead of LDS tiles or delta rows (default: on for multi-round
                                   trees beyond 24 K deals per batch, profiles/r04_deals.md) */
    int32_t delta_rows;         /* RS_FORM_*: i32 deal sweeps keep no delta tiles and issue no atomics inside the walk: a visit stores its deltas at the deal's LIST POSITION
                                   ([2A][batch pitch] i32 rows per traverser node, coalesced), and one streaming pass per sweep sums every row by cluster (LDS histogram of
                                   one row at a time) into the delta tables.  Applies to the round subtrees whose traverser nodes have at most 16 384 clusters */
    int32_t direct_rows;        /* RS_FORM_*: round subtrees whose traverser nodes have MORE than 16 384 clusters (lossless abstractions: 180 234 on the river) store delta rows too, and
                                   one pass per round adds them straight into the TABLE once the round's walks are done (atomic adds at the key row's clusters) -- no delta-table
                                   entries, no share of the apply pass for those nodes.  Default: on; data-parallel deal batches exchange the delta TABLES between sweep and apply
                                   and must switch it off (rs_solver_attach_comm refuses a solver that has it on; rs_deal_trainer does so for world > 1) */
    int32_t reserved[1];        /* zero */
} rs_kernel_forms;

typedef struct rs_solver_params {
    float scale;            /* 100.0 (cfr.rs:424) or 10000.0 (cfr.rs:617) */
    int32_t mode;           /* RS_UPD_* */
    int32_t chance_mode;    /* RS_CHANCE_* */
    int32_t use_graph;      /* 1: replay each traverser's plan as one hipGraph launch */
    int32_t fuse_subtrees;  /* 1: every chance-free subtree is walked by ONE tree-specialised straight-line kernel per
                               traverser, generated from the tree and compiled at create time with hipRTC: utilities and
                               reaches stay in registers, only table rows and leaf rows touch HBM.  0: level-by-level node
                               kernels.  Results are bit-identical either way. */
    int32_t opp_mode;       /* RS_OPP_* */
    uint64_t sample_seed;   /* RS_OPP_SAMPLE: base seed of the sweep-seed sequence */
    /* Multi-GPU sharding of a multi-round RS_CHANCE_ENUM sweep (BASELINE configs[3]); shard_world <= 1 = off.
     * The boards of round `shard_round` (and of every later round) are split contiguously over the ranks, sizes differing
     * by at most one; earlier rounds are REPLICATED.  This table holds only the rank's boards:
     * n_boards[shard_round] = hi - lo with lo = rank*base + min(rank, rem), base = shard_global_boards / shard_world.
     * At every chance node entering shard_round the ranks exchange the utility rows of their boards (one all-gather),
     * after which each rank sums all deals in the reference order and applies IDENTICAL updates to the replicated
     * rounds: results equal the single-GPU sweep bit for bit, for any rank count. */
    int32_t shard_world;
    int32_t shard_rank;
    int32_t shard_round;
    uint32_t shard_global_boards;   /* boards of shard_round over all ranks */
    /* Data-parallel deal batches (rs_solver_create_deals on a REPLICATED table, one rank per GPU): the global index of this rank's
     * first deal inside the union batch.  Only the opponent-sampling hash sees it, so that a deal draws the same bits whichever
     * rank owns it.  With a communicator attached rs_iterate sweeps, all-reduces the i32 deltas (ncclInt32 sum) and applies the
     * union: N ranks x n deals equal ONE GPU with N*n deals per batch, bit for bit. */
    uint32_t deal_offset;
    rs_kernel_forms forms;  /* zeroed = the engine's choices */
} rs_solver_params;

/* leaves_p0 / leaves_p1: one entry per TREE node id (only terminals are read) for traverser 0 / 1;
 * pass the same array twice unless RS_LEAF_UTIL buffers differ per traverser. */
int rs_solver_create(rs_table *table, const rs_tree *tree, const rs_leaf_desc *leaves_p0, const rs_leaf_desc *leaves_p1,
                     const rs_solver_params *params, rs_solver **out);   /* MCCFRTrainer::init, cfr.rs:159-184 */
/* Deal batches: lanes are DEALS.  Every deal carries, per (round_idx, player), the dense cluster id that
 * ICardAbstraction::get_cluster() returned for it (card_abstraction.rs:204-209, :245-251, :287-293; cfr.rs:361-365) and
 * the table keeps the reference's own shape  infosets[an.index][cluster_idx]  (n_boards = 1; cluster counts may differ
 * per player and round).  Several deals of a batch may address the same info set, which the reference lets race
 * (cfr.rs:414); here the sweep is BATCH-SYNCHRONOUS and deterministic: every deal reads the table as it was when the
 * sweep started, its update becomes the i32 delta (new - old) against the value it read, deltas are accumulated with
 * atomic adds (wrapping, order-independent) and applied when the sweep ends.  RS_CHANCE_PASS (one run-out per deal,
 * cfr.rs:306-313).  Float tables (RS_F32, RS_F16: extensions, fuse_subtrees = 1, one GPU, no RS_UPD_PRUNE): a visit's deltas
 * (scale*reach)*(u - util) and (scale*reach)*sigma are f32, every cell's deltas are added IN DEAL ORDER from 0.0 and then to
 * the cell -- one rounding to the table's type per cell and sweep, and with RS_UPD_RMPLUS the traverser's regrets that do
 * not end the sweep above 0 end it at 0.  Leaf buffers and d_root_util hold one float per deal, pitch = round_up(n_deals, 64);
 * cluster-id vectors have the same pitch (padding ignored). */
typedef struct rs_deal_batch {
    uint32_t n_deals;
    const uint32_t *d_cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS];   /* device; [round_idx][player] */
    const uint8_t *d_prune;   /* RS_UPD_PRUNE only: device [pitch] bytes, 1 = this deal is traversed with prune = true.  In train() pruning is a
                                 property of the DEAL (cfr.rs:213-221: t > PRUNE_THRESHOLD && q > 0.05); NULL = every deal */
} rs_deal_batch;
int rs_solver_create_deals(rs_table *table, const rs_tree *tree, const rs_deal_batch *deals, const rs_leaf_desc *leaves_p0,
                           const rs_leaf_desc *leaves_p1, const rs_solver_params *params, rs_solver **out);
void rs_solver_destroy(rs_solver *solver);
/* one traverser sweep over every lane: `self.cfr(0, player, hand, 1f32, ..)` (cfr.rs:217) for all lanes.
 * d_root_util[pitch of the root round] (NULL = discard) receives the value returned at node 0. */
int rs_iterate(rs_solver *solver, int traverser, float *d_root_util);
/* MCCFRTrainer::train (cfr.rs:188-265), deterministic: per iteration both traversers sweep, t += 1, then
 * the discount check `t > threshold` (d = p/(p+1), p = t/interval) until t > discount_cap. */
int rs_train(rs_solver *solver, uint64_t iterations, uint64_t discount_interval, uint64_t discount_cap);
/* For a host that writes MCCFRTrainer::train's loop itself (rs_iterate / rs_iterate_phase per batch, rs_discount at its ticks): on = 1 in front of the loop, on = 0 behind
 * it.  In between the solver's kept shadow records (rs_kernel_forms.kept_records) are the working copy, as they are inside rs_train and rs_deal_trainer_train: the table's rows
 * of those nodes are neither added to nor discounted until on = 0 writes them back.  Any call that reads or writes table contents in between (download, upload, strategy
 * query, best response, checkpoint, fill) writes them back itself first and ends the working-copy state -- correct, but the rest of the loop then updates table and records
 * both; a second solver sweeping the same table in between is not covered.  A solver without kept records: both calls do nothing. */
int rs_solver_training_loop(rs_solver *solver, int on);
/* Deal-batch solvers: rs_iterate_phase(.., 0) = the sweep (deltas accumulated, table untouched), (.., 1) = table += delta, delta = 0;
 * between the two the ranks' delta tables are summed (rs_comm_allreduce_deltas; tests emulate ranks and add them on the host).
 * Sharded sweeps: rs_iterate = phase 0 (everything inside the sharded rounds) + exchange + phase 1 (the replicated rounds).
 * With a communicator attached the exchange is one in-place ncclAllGather over xGMI; without one the host runs the phases
 * itself and moves the ranks' slots (tests emulate several ranks on one GPU this way). */
int rs_solver_attach_comm(rs_solver *solver, rs_comm *comm);
int rs_iterate_phase(rs_solver *solver, int traverser, int phase, float *d_root_util /* phase 1 only, may be NULL */);
/* the exchange buffer of a traverser: [shard_world][bytes_per_rank]; rank r's slot is the r-th */
int rs_solver_exchange_info(rs_solver *solver, int traverser, void **d_buf, size_t *bytes_per_rank);
int rs_comm_allgather(rs_comm *comm, rs_table *table, void *d_buf, size_t bytes_per_rank);
int rs_comm_allreduce_deltas(rs_comm *comm, rs_table *table);   /* in-place ncclInt32 sum of the table's two delta arrays */
/* the delta tables of a deal-batch table: two DEVICE i32 arrays of rs_table_cells() elements laid out like the table itself */
int rs_table_deltas(rs_table *table, int32_t **d_dregrets, int32_t **d_dstrategy_sum);
/* device memory a solver holds beside the table, EVERYTHING it allocated: the utility / reach arena; for deal sweeps the AoS shadow of the table, the packed (or ordered)
   per-deal records, the live-deal lists with their reach / position rows and counters, the work lists; the job descriptors of every launch; the exchange buffer of a
   sharded sweep.  (The table's own delta arrays of a deal solver belong to the table: rs_table_deltas.) */
size_t rs_solver_workspace_bytes(const rs_solver *solver);
int rs_jit_available(void);   /* 1 if libhiprtc.so can be loaded (needed for fuse_subtrees) */
int rs_solver_n_launches(const rs_solver *solver, int traverser);
/* which kernel forms the solver chose (rs_kernel_forms): bit 0 = deal sweeps walk the batch in last-round-cluster order (deal_order), bit 1 = some round subtree
   stores delta rows (delta_rows); negative = bad solver */
int rs_solver_forms(const rs_solver *solver);

/* ---- card-abstraction plumbing in front of get-infoset (host only; card_abstraction.rs) -----------------------------
 * These entry points take canonical hand indices as INPUT (e.g. from the Rust side's own hand_indexer_s); the index itself is
 * computed by rs_hand_indexer / rs_card_abs further down. */
typedef struct rs_dense_map rs_dense_map;
/* bucket files: flat little-endian u32 per canonical hand index (gen_abstraction/main.rs:372-380 writes,
 * card_abstraction.rs:227-229 / :269-271 read).  *out is malloc'ed: release it with rs_free_u32. */
int rs_cluster_file_read(const char *path, uint32_t **out, size_t *n_out);
int rs_cluster_file_write(const char *path, const uint32_t *clusters, size_t n);   /* fails if the file exists (create_new) */
void rs_free_u32(uint32_t *p);
/* index_to_cluster (card_abstraction.rs:20-29); cluster_arr = NULL is the ISOMORPHIC abstraction (bucket = index) */
int rs_index_to_cluster(const uint32_t *cluster_arr, size_t arr_len, const uint64_t *indices, size_t n, uint64_t *out);
/* dense ids 0..size: generate_maps (card_abstraction.rs:75-184) in first-appearance order of `buckets` (the reference's
 * channel-arrival order is nondeterministic) */
int rs_dense_map_create(const uint64_t *buckets, size_t n, rs_dense_map **out);
void rs_dense_map_destroy(rs_dense_map *map);
size_t rs_dense_map_size(const rs_dense_map *map);                                  /* get_size, card_abstraction.rs:211-213 */
int rs_dense_map_lookup(const rs_dense_map *map, const uint64_t *buckets, size_t n, uint32_t *dense_out); /* get_cluster's map step */
int rs_dense_map_keys(const rs_dense_map *map, uint64_t *keys_out);                 /* dense id -> bucket, size() entries */

/* ---- canonical hand index: rust_poker::hand_indexer_s (Cargo.toml:18, crate not vendored) -------------------------------------
 * The suit-isomorphic index of K. Waugh, "A Fast and Optimal Hand Isomorphism Algorithm" (AAAI 2013 workshop), which the crate
 * wraps.  card = 4*rank + suit, rank 0..12 = 2..A.  A hand is its rounds' cards in order (hole cards first); cards inside a round
 * may come in any order.  The partition into classes is pinned by the reference's own numbers (1 286 792 flop hands, out.txt:1;
 * 12 888 turn clusters, card_abstraction.rs:315); the ORDER of indices follows the published algorithm and is not pinned by any
 * reference fixture. */
typedef struct rs_hand_indexer rs_hand_indexer;
/* hand_indexer_s::init(rounds, cards_per_round) (card_abstraction.rs:88-90, gen_abstraction/ehs.rs:30-33) */
int rs_hand_indexer_create(int rounds, const uint8_t *cards_per_round, rs_hand_indexer **out);
void rs_hand_indexer_destroy(rs_hand_indexer *indexer);
uint64_t rs_hand_indexer_size(const rs_hand_indexer *indexer, int round);      /* .size(round) (gen_abstraction/main.rs:91,359) */
int rs_hand_indexer_rounds(const rs_hand_indexer *indexer);
int rs_hand_indexer_n_cards(const rs_hand_indexer *indexer, int round);        /* cards of rounds 0..round */
/* .get_index(cards) for n hands (card_abstraction.rs:205); cards[n][n_cards(round)]; `round` = rounds-1 is what get_index returns */
int rs_hand_index(const rs_hand_indexer *indexer, int round, const uint8_t *cards, size_t n, uint64_t *out);
/* The index-ORDER self-check (card_abstraction.rs:204-209 get_cluster -> hand_indexer.get_index; :227-229 bucket files indexed by it): expect[i] = what the
 * caller's own indexer (rust_poker's hand_indexer_s::get_index) returned for cards[i].  RS_OK when every hand agrees; RS_ERR_MISMATCH at the first that does not
 * (*first_bad = its position, n when all agree; *got_at_first_bad = this library's index for it; either may be NULL).  Run it on ~1 000 random hands per round before
 * loading bucket files written by the reference's gen_abstraction (rust/verify_index.rs). */
int rs_hand_index_verify(const rs_hand_indexer *indexer, int round, const uint8_t *cards, size_t n, const uint64_t *expect, size_t *first_bad,
                         uint64_t *got_at_first_bad);
/* .get_hand(round, index, cards) (gen_abstraction/main.rs:117,195): one representative hand per index; cards_out[n][n_cards(round)] */
int rs_hand_unindex(const rs_hand_indexer *indexer, int round, const uint64_t *indices, size_t n, uint8_t *cards_out);
/* get_index on the GPU: d_cards[n_cards(round)][pitch] (u8, row i = card i, pitch = round_up(n, 64)), d_out[n]; asynchronous */
int rs_hand_index_device(rs_table *table, rs_hand_indexer *indexer, int round, const uint8_t *d_cards, uint32_t n, uint64_t *d_out);

/* ---- one round's card abstraction: ISOMORPHIC / EMD / OCHS (card_abstraction.rs:31-298) ----------------------------------------
 * ::init = generate_maps (card_abstraction.rs:75-184): every hand of every range x every way to complete the board to the round is
 * indexed with hand_indexer_s::init(2, [2, 3 + betting_round]), mapped through the bucket file if there is one (index_to_cluster,
 * :20-29) and given the next dense id on first sight.  hands_p*: (u8,u8) combos = HandRange.hands after remove_invalid_combos
 * (cfr.rs:161-163).  cluster_arr = NULL: ISOMORPHIC (bucket = index); otherwise the round_{r}_emd.dat / _ochs.dat contents
 * (rs_cluster_file_read), copied.  Dense ids follow first appearance in the reference's loop order (its own order is a race). */
typedef struct rs_card_abs rs_card_abs;
int rs_card_abs_create(int betting_round /* 0 flop, 1 turn, 2 river */, const uint8_t *hands_p0, size_t n_hands_p0, const uint8_t *hands_p1,
                       size_t n_hands_p1, uint64_t initial_board_mask, const uint32_t *cluster_arr, size_t arr_len, rs_card_abs **out);
void rs_card_abs_destroy(rs_card_abs *abs);
int rs_card_abs_round(const rs_card_abs *abs);
size_t rs_card_abs_size(const rs_card_abs *abs, int player);                     /* get_size (card_abstraction.rs:211-213) */
uint64_t rs_card_abs_index_size(const rs_card_abs *abs);                         /* hand_indexer.size(1): length a bucket file must have */
int rs_card_abs_keys(const rs_card_abs *abs, int player, uint64_t *keys_out);    /* dense id -> bucket, size(player) entries */
/* get_cluster(&cards, player) (card_abstraction.rs:204-209, :245-251, :287-293) for n hands on the host;
 * cards[n][5 + betting_round] = the player's hole cards, then the board (cfr.rs:357-365) */
int rs_card_abs_get_cluster(const rs_card_abs *abs, const uint8_t *cards, size_t n, int player, uint32_t *out);
/* get_cluster for every deal of a batch on the GPU.  d_cards[9][pitch] u8 as for rs_showdown_sign: rows 0-4 the board, 5-6 player 0's
 * hole cards, 7-8 player 1's.  d_cluster_p0 / d_cluster_p1 [pitch] receive the dense ids (either may be NULL): ready to be an
 * rs_deal_batch.d_cluster[round_idx][player].  Asynchronous on the table's stream.  A deal whose bucket has no dense id (Rust:
 * unwrap on None) gets id 0 and raises an error word that rs_card_abs_status reports. */
int rs_card_abs_clusters_device(rs_card_abs *abs, rs_table *table, const uint8_t *d_cards, uint32_t n_deals, uint32_t *d_cluster_p0,
                                uint32_t *d_cluster_p1);
int rs_card_abs_status(rs_card_abs *abs, rs_table *table);   /* synchronises; RS_ERR_OOB if a deal since the last call had no cluster */

/* ---- generate_hand for a batch of deals (cfr.rs:100-143) ---------------------------------------------------------------------------
 * Board cards missing from board_mask are drawn uniformly without replacement by rejection (cfr.rs:115-122), then one combo per range
 * by rejection against everything dealt (cfr.rs:126-137); Uniform::from(0..52) and slice::choose are rand 0.7's widening-multiply
 * samplers.  The random bits are a counter hash of (seed, first_deal + i) -- the reference's SmallRng is seeded from thread_rng and is
 * not reproducible.  d_hands_p*: DEVICE arrays of (u8,u8) combos.  d_cards[9][pitch] as above.  d_err (device u32, may be NULL): bit 2 is
 * raised when a deal found no valid combo within 4096 draws (the reference would loop forever). */
int rs_deals_sample(rs_table *table, uint64_t seed, uint64_t first_deal, uint64_t board_mask, const uint8_t *d_hands_p0,
                    uint32_t n_hands_p0, const uint8_t *d_hands_p1, uint32_t n_hands_p1, uint32_t n_deals, uint8_t *d_cards, uint32_t *d_err);

/* train()'s per-deal prune decision (cfr.rs:213-221): d_flags[pitch] u8, 1 where deal number first_deal + i exceeds prune_threshold
 * (PRUNE_THRESHOLD = 10 000 000, cfr.rs:190) AND its `q: f32 = rng.gen()` -- counter 4096 of the deal's hash, after everything
 * generate_hand can draw -- is > 0.05.  Ready to be an rs_deal_batch.d_prune.  Asynchronous on the table's stream. */
int rs_deals_prune_flags(rs_table *table, uint64_t seed, uint64_t first_deal, uint64_t prune_threshold, uint32_t n_deals, uint8_t *d_flags);

/* ---- MCCFRTrainer on the GPU: init + train over sampled deals (cfr.rs:159-297) -----------------------------------------------------
 * Per batch, all on the table's stream: rs_deals_sample -> rs_card_abs_clusters_device for every round -> rs_showdown_sign -> one
 * sampled-opponent sweep per traverser (rs_solver_create_deals).  One deal = one reference iteration (cfr.rs:209-226). */
typedef struct rs_deal_trainer rs_deal_trainer;
typedef struct rs_deal_trainer_params {
    uint64_t board_mask;           /* Options.board_mask (options.rs:17): 3, 4 or 5 cards */
    uint32_t deals_per_batch;
    uint64_t seed;
    uint64_t discount_interval;    /* cfr.rs:190 DISCOUNT_INTERVAL (0 = never discount) */
    uint64_t discount_cap;         /* cfr.rs:240: no discount once t exceeds it */
    rs_solver_params solver;       /* scale 100, RS_UPD_CLAMP_I64, RS_OPP_SAMPLE = the reference's mccfr(); chance_mode is forced to PASS,
                                      deal_offset is set from rank */
    uint32_t world, rank;          /* data-parallel training on replicated tables: this rank deals numbers (b*world + rank)*n .. + n of
                                      global batch b; 0 / 0 or 1 / 0 = single GPU.  t advances by world * deals_per_batch per batch */
    uint64_t prune_threshold;      /* cfr.rs:190 PRUNE_THRESHOLD (10 000 000): deals numbered beyond it are traversed with prune = true when their
                                      q > 0.05 (cfr.rs:213-221, rs_deals_prune_flags); UINT64_MAX = never.  With a finite threshold the solver runs
                                      in RS_UPD_PRUNE mode with per-deal flags that stay zero (= unpruned, bit for bit) before it */
    int32_t prefetch;              /* RS_FORM_*: deal the next batch on a second stream while the current one is swept (default: on from 65 536 deals per batch for multi-round games, beyond 262 144 for one-round games) */
    int32_t table_dtype;           /* element type of the table the trainer creates: RS_I32 (0: the reference's), or -- extensions -- RS_F32 / RS_F16: float deal sweeps
                                      (rs_solver_create_deals: f32 per-deal deltas summed in deal order, one rounding per cell and sweep); these need prune_threshold =
                                      UINT64_MAX and world <= 1.  (The field was `reserved`, zero, until ABI 5) */
} rs_deal_trainer_params;
/* MCCFRTrainer::init (cfr.rs:159-184): card_abs[round_idx] for the tree's rounds (borrowed: keep them alive), ranges as above;
 * creates the zero-filled table from the abstractions' sizes (create_infosets, cfr.rs:176) on `device`. */
int rs_deal_trainer_create(const rs_tree *tree, rs_card_abs *const *card_abs, int n_rounds, const uint8_t *hands_p0, size_t n_hands_p0,
                           const uint8_t *hands_p1, size_t n_hands_p1, const rs_deal_trainer_params *params, int device,
                           rs_deal_trainer **out);
void rs_deal_trainer_destroy(rs_deal_trainer *trainer);
rs_table *rs_deal_trainer_table(rs_deal_trainer *trainer);      /* trainer.infosets; owned by the trainer */
rs_solver *rs_deal_trainer_solver(rs_deal_trainer *trainer);
int rs_deal_trainer_train(rs_deal_trainer *trainer, uint64_t n_batches);   /* train(): deals_per_batch iterations per batch */
/* world > 1: the communicator whose all-reduce makes every rank apply the deltas of the union batch (rs_comm_create on this
 * trainer's table) */
int rs_deal_trainer_attach_comm(rs_deal_trainer *trainer, rs_comm *comm);
/* the bookkeeping that ends a batch (t += world * deals_per_batch, discount check of cfr.rs:240-262); rs_deal_trainer_train calls it,
 * callers that drive rs_deal_trainer_deal + rs_iterate_phase themselves call it once per batch */
int rs_deal_trainer_finish_batch(rs_deal_trainer *trainer);
int rs_deal_trainer_deal(rs_deal_trainer *trainer);             /* only the dealing half of a batch (cards, cluster ids, signs) */
/* synchronises; error if a deal could not be sampled / addressed since the last call.  A trainer that deals ahead (prefetch) has dealt the batch AFTER the last one it trained
 * as well -- it waits in the staging buffers for the next call -- so the report covers that batch too. */
int rs_deal_trainer_status(rs_deal_trainer *trainer);
uint64_t rs_deal_trainer_iterations(const rs_deal_trainer *trainer);   /* t of cfr.rs:200 */
/* calc_br at the discount ticks (cfr.rs:244-246 runs it before the sweep and prints the two numbers): enable != 0 makes every tick
 * call rs_calc_br first; rs_deal_trainer_last_br returns the latest pair and the iteration count it was taken at (RS_ERR_INVALID
 * before the first tick).  rs_deal_trainer_calc_br / _best_response run the two readers on the trainer's own tree, ranges and
 * abstraction at any time (best response: single-round trainers on a full board). */
int rs_deal_trainer_set_tick_br(rs_deal_trainer *trainer, int enable);
int rs_deal_trainer_last_br(const rs_deal_trainer *trainer, float *out /*[2]*/, uint64_t *iterations);
int rs_deal_trainer_calc_br(rs_deal_trainer *trainer, float *out /*[2]*/);
int rs_deal_trainer_best_response(rs_deal_trainer *trainer, int mode, double *out /*[2]*/);
/* The best-response objects a trainer keeps between calls (one per showdown mode: the game-only index, and the walk's workspace -- a buffer per tree edge while that stays
 * below 16 GB, two per tree depth otherwise): the device bytes they hold, a call that gives the workspaces back (the next best response allocates them again), and the kernel
 * launches the last call made (-1: the depth-first walk ran, one or two launches per tree node). */
size_t rs_deal_trainer_br_bytes(const rs_deal_trainer *trainer);
int rs_deal_trainer_br_release(rs_deal_trainer *trainer);
int rs_deal_trainer_br_launches(const rs_deal_trainer *trainer, int sorted);
const uint8_t *rs_deal_trainer_cards(const rs_deal_trainer *trainer);  /* device: the current batch's d_cards[9][pitch] */
const float *rs_deal_trainer_signs(const rs_deal_trainer *trainer);    /* device: its showdown signs [pitch] */
const uint8_t *rs_deal_trainer_prune_flags(const rs_deal_trainer *trainer);   /* device: its per-deal prune flags [pitch] (all 0 before the threshold) */
const uint32_t *rs_deal_trainer_clusters(const rs_deal_trainer *trainer, int round_idx, int player);   /* device: its cluster ids [pitch] */

/* ---- abstraction generator's distance sweep: gen_abstraction/kmeans.rs, emd.rs -----------------------------
 * The sweep that WRITES the bucket files read above: Kmeans::predict over every canonical hand's histogram (main.rs:361-380).
 * Histograms are f32[n_bins] (type Histogram = Vec<f32>), 1..64 bins.  Arithmetic is the reference's, bit for bit: f32, no FMA, the
 * operation order of emd_1d (emd.rs:53-113) / l2_dist (kmeans.rs:622-630). */
enum { RS_DIST_EMD = 0,   /* emd::emd_1d */
       RS_DIST_L2 = 1 };  /* kmeans::l2_dist */
/* one dist_func(p, q) on the host */
int rs_histogram_distance(int dist, const float *p, const float *q, int n_bins, float *out);
/* Kmeans::predict (kmeans.rs:173-211): d_dataset = DEVICE [n][n_bins] row-major, centers = HOST [n_centers][n_bins];
 * d_clusters[n] (u32: first center with the strictly smallest distance) and d_min_dist[n] (that distance) are DEVICE buffers, either may
 * be NULL.  The reference's returned `inertia` is a racy load+store (kmeans.rs:205); sum d_min_dist instead.  Asynchronous on the table's
 * stream; `table` only lends its device and stream. */
int rs_kmeans_predict(rs_table *table, int dist, const float *d_dataset, size_t n, const float *centers, int n_centers, int n_bins,
                      uint32_t *d_clusters, float *d_min_dist);
/* update_min_dists (kmeans.rs:603-619), the kmeans++ step: d_min_dists[i] = min(d_min_dists[i], dist(dataset[i], new_center)^2) */
int rs_update_min_dists(rs_table *table, int dist, float *d_min_dists, const float *d_dataset, size_t n, const float *new_center, int n_bins);

/* The training loops that produce the centers (kmeans.rs:213-601), Hamerly-bounded: d_clusters (u32 [n]) and d_bounds (float [n][2] = (lower, upper) per datum) are
 * DEVICE arrays; centers and s live on the HOST; every f32 sum runs in the reference's order (means are summed in data order), so results are bit-identical to the
 * sequential Rust loops (rayon only ever parallelises loops whose iterations write their own element).  All of them synchronise.
 *   rs_kmeans_init_s        Kmeans::init_s (:267-285): s[i] = min(s[i], min_{j != i} dist(c_i, c_j)) / 2 -- s is IN/OUT, as coded (created once, :518)
 *   rs_kmeans_reassign      Kmeans::reassign_clusters (:287-334) = assignment_with_bounds (:213-265); d_order (may be NULL): datum i = dataset[d_order[i]]
 *   rs_kmeans_fit_regular   Kmeans::fit_regular (:497-600), `iterations` = 10 in the reference; centers in/out, d_clusters out, d_bounds / inertia out (may be NULL)
 *   rs_kmeans_fit_growbatch Kmeans::fit_growbatch AS CODED (:336-495: one pass over the first `batch` shuffled items, then `break`); the shuffle (:352) is the
 *                           caller's: d_order; stats (may be NULL) = {min_change, inertia as printed} */
int rs_kmeans_init_s(rs_table *table, int dist, const float *centers, int n_centers, int n_bins, float *s);
/* Kmeans::init_random's choice among restarts (kmeans.rs:104-165): the caller draws the candidate center sets with its rng (`choose_multiple`, :120), this scores them
 * as coded (mean pairwise distance, f32 sums in index order, :133-148) and returns the most spread one (*best; the last of equal maxima, :151-156).
 * centers: HOST [n_restarts][n_centers][n_bins]; cluster_dists: HOST out [n_restarts], may be NULL.  (init_pp's heavy step is rs_update_min_dists.) */
int rs_kmeans_pick_restart(rs_table *table, int dist, const float *centers, int n_restarts, int n_centers, int n_bins, float *cluster_dists, int *best);
int rs_kmeans_reassign(rs_table *table, int dist, const float *d_dataset, const uint32_t *d_order, size_t n, const float *centers, int n_centers, int n_bins,
                       const float *s, uint32_t *d_clusters, float *d_bounds);
int rs_kmeans_fit_regular(rs_table *table, int dist, const float *d_dataset, size_t n, float *centers, int n_centers, int n_bins, int iterations, uint32_t *d_clusters,
                          float *d_bounds, float *inertia);
int rs_kmeans_fit_growbatch(rs_table *table, int dist, const float *d_dataset, size_t n, const uint32_t *d_order, size_t batch, float *centers, int n_centers,
                            int n_bins, uint32_t *d_clusters, float *d_bounds, float *stats);

/* ---- showdown evaluation on the device ---------------------------------------------------------------
 * d_cards[9][pitch] (u8, pitch = round_up(n_deals, 64)): rows 0-4 the board, 5-6 player 0's hole cards, 7-8 player 1's;
 * card = 4*rank + suit, rank 0..12 = 2..A (cfr.rs:592).  d_sign[lane] = sign(evaluate(hand0) - evaluate(hand1)) exactly as
 * cfr.rs:324-333 compares the two scores: ready to be used as an RS_LEAF_SIGN buffer. */
int rs_showdown_sign(rs_table *table, const uint8_t *d_cards, uint32_t n_deals, float *d_sign);

/* ---- table checkpoints (the reference never persists the trained table) ----------------------------
 * "RSTB" v1: header, node descriptors, then per node regrets[A][lanes] and strategy_sum[A][lanes] (no pitch padding),
 * then a 64-bit FNV-1a checksum.  rs_table_load creates the table on `device`. */
int rs_table_save(rs_table *table, const char *path);
int rs_table_load(const char *path, int device, rs_table **out);

/* ---- multi-GPU: boards shard across GPUs, one process per GPU -------------------------
 * Replicated tables (rounds whose boards are not sharded) accumulate rank-local deltas; one RCCL
 * all-reduce (sum) over xGMI brings every rank to the same values. */
#define RS_COMM_ID_BYTES 128
int rs_comm_unique_id(void *id_out /* RS_COMM_ID_BYTES */);   /* rank 0; ship the bytes to the other ranks */
int rs_comm_create(rs_table *table, const void *id, int rank, int n_ranks, rs_comm **out);
void rs_comm_destroy(rs_comm *comm);
/* snapshot the replicated rounds (bit r of round_mask) before a local iteration ... */
int rs_replicated_begin(rs_table *table, uint32_t round_mask);
/* ... then x = snapshot + allreduce_sum(x - snapshot) for regrets and strategy_sum of those rounds */
int rs_allreduce_replicated(rs_table *table, rs_comm *comm, uint32_t round_mask);

#ifdef __cplusplus
}
#endif
#endif /* RUSTSOLVER_AMD_H */
