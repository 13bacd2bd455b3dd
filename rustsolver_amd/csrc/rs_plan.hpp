// rs_plan.hpp -- what the plan builder (rs_plan.cpp) and the executor (rs_solver.cpp) share: the launch plan of one traverser and the solver object.  Not part of the ABI.
#pragma once

#include <map>
#include <vector>

#include "rs_internal.hpp"

namespace rs {

constexpr size_t kCountStride = 64;   // u32 elements between two live-deal counters (256 B)
constexpr uint32_t kScanParentMin = 2048;   // deal batches beyond this size compact a round subtree's live deals from its parent's lists (rs_solver.cpp scan_parent); 65 536 until the end of
                                            // round 4: re-measured with three auxiliary streams 4 K / 16 K / 64 K deals 0.48-0.54 / 0.63-0.69 / 0.75-0.76 against 0.51-0.58 / 0.66-0.71 / 0.82-0.84 ms
// Rows that many workgroups stream at the same offsets at the same time (delta rows, live-deal lists) must not start a power of two apart: a 4 M-deal batch puts them 16 MiB
// apart, every stream then sits on the same memory channel at the same moment, and the row-summing pass ran 4x slower than with 4 196 416 deals (12.7 against 8.9 ms per batch)
constexpr size_t kRowStagger = 1088;   // elements between the natural pitch and the one used (4 352 B: off every power-of-two interleave up to 4 KiB, rows stay 256-B aligned)
constexpr uint32_t kSiblingsMinDeals = 524288;   // deal batches beyond this size compact the live deals of sibling roots in one scan of their source (rs_plan_deals.cpp)
// Deal batches beyond kRowsMinDeals store delta rows in their list walkers, beyond kOrderMinDeals they also walk the batch in the order of the traverser's last-round cluster
// (rs_solver.cpp).  Three streets, 5 000-bucket files, ms per batch with LDS tiles / rows / rows + order (profiles/r04_deals.md): 1 K deals 0.55 / 0.46 / 0.48, 4 K 0.60 / 0.52 /
// 0.53, 16 K 0.70-0.81 / 0.65 / 0.72, 32 K 0.80 / 0.77 / 0.60-0.70, 64 K - / 0.83-0.91 / 0.74-0.84 (until the end of round 4 both forms started at 48 K deals).
constexpr uint32_t kDownRowsMinDeals = 262144;   // batches beyond this keep a row copy of the first round for their dense reach-down kernel, which stages it (one deal per lane)
constexpr uint32_t kRowApplyMaxDeals = 32768;   // up to here the delta rows of a round go into the delta tables by atomics (k_row_apply) instead of k_row_sums' LDS tiles (rs_solver.cpp)
// Round 5 (merged launches, one stream; ms per batch with LDS tiles / rows / rows + order): 256 deals 0.54 / 0.22 / -, 1 K 0.49 / 0.26 / -, 4 K - / 0.34 / 0.33, 8 K - / 0.41 / 0.38,
// 16 K - / 0.50 / 0.44: rows from 64 deals, order from 6 K.
constexpr uint32_t kRowsMinDeals = 63;
constexpr uint32_t kOrderMinDeals = 6144;
                                            // Round 3 (gathering walks): 4 M deals 1.13-1.19x, 256 K a wash -> 512 K.  Round 4 (staged rows, runs summed by DPP): 4 M 1.16x, 1 M 1.18x, 512 K 1.16x,
                                            // 256 K 1.10x, 128 K 1.11x, 64 K 1.24x over the tile kernels on one card (profiles/r04_deals.md)
constexpr size_t kWorklistLdsBytes = 64;   // in front of the tiles of a work-list kernel: lds_all[0] holds the ticket (rs_jit.cpp)
enum LaunchKind { L_REACH, L_PRUNE_REACH, L_EXPAND, L_UPDATE, L_NODE_UTIL, L_REDUCE, L_TREE, L_SEED, L_APPLY, L_SHADOW, L_COMPACT, L_NANFILL, L_PACK, L_ORDER, L_ROWSUM };

struct Launch {
    int group = 0;                      // > 0: consecutive launches of one group are independent of each other (round subtrees) and may overlap
    int kind;
    int n_actions = 0;
    int first_job = 0, n_jobs = 0;
    int first_group = 0, n_groups = 0;  // L_COMPACT: the launch's jobs as sibling groups (k_compact_siblings), 0 = one job per root
    uint32_t max_n_vec = 0;
    size_t max_lanes = 0;   // chance launches: largest lane count among the jobs
    double bytes = 0.0;
};

struct ReachSrc {
    const float *ptr = nullptr;
    float cst = 1.0f;
    bool valid = false;
};

// one launch of a tree-specialised (hipRTC) kernel: blockIdx.y indexes the argument blobs
struct JitLaunch {
    hipFunction_t fn = nullptr;         // resolved by rs_solver_create once the whole plan is known (all kernels compiled together)
    std::string source, entry;          // the generated source and its entry point (released after the compile)
    size_t src_off[3] = {0, 0, 0};      // JitSubtree::src_struct / src_entry / src_body of `source`
    // Small deal batches (rs_solver.cpp merge_small_groups): the kernels of one group of independent round subtrees -- different shapes, different code -- run as ONE launch
    // whose entry point dispatches on blockIdx.y; `members` = the launches it stands for (their blobs stay where they are), this launch holds the merged kernel alone.
    std::vector<int> members;
    bool absorbed = false;              // a member of a merged launch: never compiled or launched on its own
    MergedArgs margs{};
    std::vector<unsigned char> blob;   // n_jobs * stride bytes, layout = JArgs of the generated source
    size_t stride = 0;
    int n_jobs = 0;
    uint32_t max_n_vec = 0;
    unsigned char *d_blob = nullptr;
    double bytes = 0.0;
    int threads = 256;
    size_t lds_bytes = 0;
    bool persistent = false;            // resident LDS tiles: one long-lived workgroup per CU, flushes once
    bool seg = false;                   // ordered sweeps, last round: no LDS, 256-thread workgroups, many per CU
    bool rows = false;                  // delta rows: no LDS, no barrier, 256-thread workgroups, many per CU
    bool staged = false;                // staged rows: lds_bytes is the waves' staging area, not delta tiles (grid as for the forms without LDS)
    bool worklist = false;              // list-walking kernels with LDS tiles: a 1-D grid of resident workgroups pulls (job, trip) items; k_worklist runs right before
    uint32_t *d_wl = nullptr;           // [2 + n_jobs + 1]
    uint32_t off_count = 0, deals_per_trip = 0;
};

struct Plan {
    std::vector<ChanceJob> chance_jobs;   // L_EXPAND / L_REDUCE launches index into this (first_job, n_jobs)
    ChanceJob *d_chance_jobs = nullptr;
    std::vector<JitLaunch> jit;
    std::vector<NodeJob> jobs;
    NodeJob *d_jobs = nullptr;
    std::vector<Launch> launches;
    // sparse deal sweeps: per subtree root the list of live deals (reach not NaN), rebuilt by k_compact_live after the top-down pass
    uint32_t *d_lists = nullptr;        // [n_compact][pitch]
    float *d_rlists = nullptr;          // position-indexed rows: the reach of every list entry, same shape as d_lists
    uint32_t *d_plists = nullptr;       // position-indexed rows: where the parent subtree reads every list entry's utility, same shape as d_lists
    float *d_hrows = nullptr;           // hand-off rows of the round subtrees with a reach-down kernel: [opponent nodes handed + 1][batch pitch + stagger] floats per root
    uint32_t *d_klists = nullptr;       // delta rows: the traverser's cluster of every list entry (the key row of k_row_sums), same shape as d_lists
    std::vector<size_t> drow_off;       // delta rows: per table node the int offset of its [2A][batch pitch] rows inside the solver's d_drows (SIZE_MAX: none)
    std::vector<RowSumJob> row_jobs;
    RowSumJob *d_row_jobs = nullptr;
    uint32_t row_max_cells = 0;
    ApplyJob *d_apply_jobs = nullptr;   // deal sweeps: the cell ranges of the traverser's own nodes (where its deltas are)
    // deal sweeps on f32 tables: per-deal delta rows of every traverser node, the traverser's deals listed per cluster and round (rebuilt every sweep), the ordered apply
    float *d_frows = nullptr;
    std::vector<size_t> frow_off;       // per table node: float offset of its [2A][pitch] rows inside d_frows (SIZE_MAX: not this traverser's)
    ApplyF32Job *d_f32_jobs = nullptr;
    int n_f32_jobs = 0;
    uint32_t f32_max_clusters = 0;
    uint32_t *d_member_start[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr}, *d_members[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};
    uint32_t *d_member_scratch = nullptr;   // tile histograms, then totals
    int n_apply_jobs = 0;
    size_t apply_max_vec = 0;
    bool apply_whole = false;           // the apply pass runs over the whole table (cell ranges that are no multiple of four: never with 64-lane padded pitches)
    size_t *d_pack_off = nullptr;       // data-parallel sweeps: where every apply job's cells start in the packed buffer the ranks sum (vectors of 4 cells)
    size_t pack_vec = 0;                // ... and its length per array
    size_t aux_bytes = 0;               // device memory of this plan beside the arena: live-deal lists and the reach rows of the round subtrees
    size_t n_count_words = 0;           // u32 words of d_counts (all counters, kCountStride apart)
    bool counts_zeroed_by_shadow = false;   // the sweep opens with a k_build_shadow launch, which zeroes d_counts too (else: a memset in front of every compaction)
    uint32_t *d_counts = nullptr;       // [n_compact]
    CompactJob *d_compact_jobs = nullptr;
    std::vector<CompactJob> compact_jobs;
    std::vector<size_t> count_off;      // per compact job: index of its first counter (a job has one per cluster range)
    std::vector<int> compact_round;     // per compact job: betting round of its root
    std::vector<rs::CompactGroup> compact_groups;   // runs of sibling jobs (same source, no cluster ranges), in job order
    rs::CompactGroup *d_compact_groups = nullptr;
    int dense_roots[RS_MAX_ROUNDS] = {0, 0, 0};   // round subtrees that walk the whole batch, per round
    uint32_t compact_max_lanes = 0;
    uint32_t *d_bmask = nullptr;        // liveness masks of the round subtrees' reach-down kernels (CompactJob.mask), one row per parent root
    float *d_reach_nan = nullptr;       // round subtrees: reach buffers of every root but the first, all NaN at the start of a sweep
    size_t reach_nan_bytes = 0;
    size_t split = 0;                   // sharded sweeps: launches [0, split) = phase 0, [split, end) = phase 1
    int n_boundary = 0;                 // chance nodes entering the sharded round
    size_t arena_bytes = 0;
    const float *root_util = nullptr;   // inside the arena
    size_t root_lanes = 0;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
};


}  // namespace rs

using rs::Knobs;
using rs::OrderJob;
using rs::PackJob;
using rs::RowSumJob;
using rs::Plan;
using rs::ShadowJob;

struct rs_solver {
    rs_table *table = nullptr;
    rs_tree tree;
    rs_solver_params params{};
    Knobs knobs;                        // kernel-form switches, resolved once at creation (rs_knobs.cpp)
    int lds_limit = 64 * 1024;          // LDS bytes the device gives ONE workgroup (MI355X: 160 KiB), queried at creation
    std::vector<rs_leaf_desc> leaves[2];
    Plan plan[2];
    char *d_arena = nullptr;
    size_t arena_bytes = 0;
    size_t other_bytes = 0;             // every other device allocation of the solver: table shadow, packed / ordered per-deal records, job blobs, work lists, counters, exchange buffer
    uint32_t n_boards[RS_MAX_ROUNDS] = {0, 0, 0};
    uint32_t n_clusters = 0;
    size_t pitch[RS_MAX_ROUNDS] = {0, 0, 0};
    int n_rounds = 0;
    // multi-GPU sharding (rs_solver_params.shard_*)
    bool sharded = false;
    uint32_t shard_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // first global board of every rank at shard_round, then the total
    size_t slot_lanes = 0;              // floats per (rank, boundary node) in the exchange buffer
    float *d_exchange = nullptr;        // [world][n_boundary][slot_lanes]
    size_t exchange_floats_per_rank = 0;
    rs_comm *comm = nullptr;
    int n_cus = 256;                    // multiprocessors of the device (grid of the persistent deal kernels)
    // round subtrees of one round are independent: their launches are spread over a few auxiliary streams (fork / join with events)
    // Three: with the table's own stream that makes four, one per hardware queue of the device (GPU_MAX_HW_QUEUES defaults to 4) -- a fifth stream shares a queue with another
    // and which two collide differs from process to process: with four auxiliary streams a 64 K-deal batch took 0.81 or 0.97 ms depending on the run, with three 0.76-0.77 in
    // every run (16 K deals 0.95 -> 0.82, 256 K 1.40 -> 1.30, the 2 GB lossless table 1.43 -> 1.36; 1 M and 4 M deals unchanged).  (8: 4 M deals 6.95 -> 7.24 ms, 64 K 0.91 -> 1.10)
    static constexpr int kAux = 3;
    hipStream_t aux[kAux] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {};
    // deal sweeps through tree-specialised kernels read the table from an AoS shadow rebuilt at the start of every sweep
    int32_t *d_shadow = nullptr;
    ShadowJob *d_shadow_jobs = nullptr;   // the jobs of traverser 0's sweep, then those of traverser 1's (the same nodes, different record widths)
    int n_shadow_jobs = 0;                // per traverser
    std::vector<size_t> shadow_off_p[2];  // per traverser and table node, in ints (SIZE_MAX: no shadow): where cluster 0's record of the node starts
    std::vector<uint32_t> shadow_stride_p[2];   // ints between two clusters' records of the node: the ROW of its round subtree and role (rs_solver.cpp setup_table_shadow)
    std::vector<uint32_t> shadow_rec_p[2];      // ints of the node's own record: 2 * half at the sweep's traverser nodes, half at the opponent's
    std::vector<uint32_t> shadow_rowoff_p[2];   // ints from the start of the row to the node's record
    uint32_t shadow_max_clusters = 0;
    // The FIRST round's nodes a second time, as rows of narrow records (regrets / strategy only), for the dense reach-down kernel of a big batch alone (round 5): its 4 M deals
    // read every node of the round, 28 gathers of 64 lines each per thread pair; staged like the list walkers' rows (rs_device.hpp stage_rows) the wave touches 4-8 rows per
    // load.  The dense WALK keeps the node arrays above (its deltas go to LDS tiles: no staging area to spare, and node arrays serve it better: NOTES round 4).
    std::vector<size_t> down_off_p[2];          // per traverser and table node (SIZE_MAX: none)
    std::vector<uint32_t> down_stride_p[2], down_rowoff_p[2];
    // KEPT records (setup_table_shadow): the nodes of the rounds whose delta rows go straight into the table keep their records between sweeps -- wide records ({regrets,
    // strategy sums}, no strategies) at the front of d_shadow, shared by both traversers' sweeps, built when the table has moved on without them (rs_table.epoch), and taking
    // every addition k_row_apply makes to the table and every discount sweep (solver_table_discounted)
    size_t kept_ints = 0;                 // d_shadow[0 .. kept_ints)
    ShadowJob *d_kept_jobs = nullptr;
    int n_kept_jobs = 0;
    uint32_t kept_max_clusters = 0;
    uint64_t kept_epoch = ~uint64_t(0);   // the table epoch the kept records are in step with
    std::vector<char> kept_node;          // per table node
    uint32_t *d_kept_primary = nullptr;   // device flag k_row_apply reads: the kept records alone take the additions (solver_kept_primary)
    bool primary = false;
    rs::DiscountJob *d_disc_jobs = nullptr;   // the table without the kept nodes, as stretches of consecutive nodes
    int n_disc_jobs = 0;
    size_t disc_max_vec = 0;
    // sparse deal sweeps fetch the per-deal inputs of a round (both cluster ids, leaf value, prune flag) as ONE packed 16-byte record per live deal
    void *d_attr[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};
    PackJob *d_pack_jobs = nullptr;
    int n_pack_jobs = 0;
    unsigned attr_used = 0;             // bit r: some generated kernel reads the packed records of round r (only the list-walking forms do)
    // ordered sweeps (rs_kernel_forms.deal_order): traverser p's sweep walks the batch sorted by p's cluster id on the last round; d_arec holds the 32-byte per-deal
    // records in that order (rebuilt at the start of every sweep by k_order_*), d_attr[r] all point at it
    bool ordered = false;
    int order_round = 0;                // the last betting round of the tree
    void *d_arec = nullptr;             // == d_arec_p[0]
    void *d_arec_p[2] = {nullptr, nullptr};   // one set of records per traverser (round 5): traverser 1's can be sorted while traverser 0's sweep still reads its own
    uint32_t *d_order_tot = nullptr;    // [2 traversers][2][n_bins]: counts and cursors of the counting sorts
    OrderJob order_job[2];              // per traverser
    // The records depend on the deals alone, not on the table: a caller that knows the next batch early (rs_deal_trainer deals ahead on a second stream) sorts them itself,
    // beside the sweeps (solver_order_on), and the plans' own L_ORDER launches do nothing.  Set before the first sweep (a captured graph keeps what it was captured with).
    bool order_ahead = false;
    // data-parallel deal sweeps (a communicator attached): what the ranks exchange between sweep and apply (solver_exchange_deltas)
    int32_t *d_packed = nullptr;        // [2][max pack_vec * 4] the traverser's delta cells of the rounds that sum through the delta tables, one ncclInt32 all-reduce
    uint32_t *d_items = nullptr;        // [3 * item_cap] this rank's (job, row, cluster, delta) items of the rounds whose rows go straight into the table
    uint32_t *d_items_all = nullptr;    // [world][3 * item_cap]
    uint32_t *d_item_count = nullptr;   // [1 + world]: this rank's cursor, then every rank's count
    uint32_t item_cap = 0, items_world = 0;   // items_world: ranks d_items_all and d_item_count were sized for
    uint64_t dp_bytes_total = 0, dp_sweeps = 0;
    uint64_t dp_bytes_last = 0;         // bytes this rank handed to the collectives in its last sweep (all-reduce buffer + every rank's items)
    int (*before_sweep)(void *ctx, int traverser) = nullptr;   // ... and is asked in front of every sweep whether the records are the live batch's (rs_iterate, rs_iterate_phase 0)
    void *before_sweep_ctx = nullptr;
    // delta rows (rs_kernel_forms.delta_rows): one buffer for both traversers' sweeps (they never overlap), [2A][batch pitch] i32 per traverser node of an eligible round
    bool rows = false;
    bool direct_rows = false;           // rounds whose traverser nodes outgrow the summing pass's LDS tile store delta rows too, added straight into the table (k_row_apply)
    int first_round = 0;                // betting round of the first action node: its subtree walks the whole batch and keeps its tiles unless RS_JIT_ROWS = 2
    int32_t *d_drows = nullptr;
    bool deal_mode = false;             // lanes are deals (rs_solver_create_deals)
    rs_deal_batch deals{};
    uint64_t *d_seed_state = nullptr;   // RS_OPP_SAMPLE: {base seed, call index, seed of the current sweep}
    const uint64_t *d_seed() const { return d_seed_state ? d_seed_state + 2 : nullptr; }
};

namespace rs {

#define RS_HIP(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

// rs_plan.cpp: shapes and sharding derived from table + tree (validation included), then one PlanBuilder per traverser: layout() decides which buffers exist
// (offsets into the arena the caller then allocates), emit() writes the jobs and launches
// rs_plan_deals.cpp: one ROW of the sweep's table shadow = the records of the nodes of one player in one round subtree (action counts `n_actions`, in ActionNode.index order),
// `wide`: records hold regrets and strategy sums (the sweep's traverser), else regrets / strategy only.  Returns the row's ints (a multiple of 4: rows start 16-byte aligned);
// rec[k] = ints of node k's record, off[k] = where it starts in the row (16-byte records first, the 8-byte ones of two-action opponent nodes behind them)
uint32_t shadow_row_layout(const std::vector<uint32_t> &n_actions, bool wide, std::vector<uint32_t> &rec, std::vector<uint32_t> &off);
int derive_geometry(rs_solver *s);
bool rows_round_direct(const rs_solver *s, int p, int round);   // ... and are its rows added straight into the table (more clusters than the summing pass's LDS tile holds)?
bool rows_round_ok(const rs_solver *s, int p, int round);   // rs_plan_deals.cpp: do traverser p's nodes of this round take the delta-rows form?
size_t drows_ints(const rs_solver *s, int p);               // ints of delta rows traverser p's sweep needs
struct PlanBuilder;
PlanBuilder *plan_builder_new(rs_solver *s, int traverser);
int plan_builder_layout(PlanBuilder *b);
int plan_builder_emit(PlanBuilder *b);
void plan_builder_free(PlanBuilder *b);

}  // namespace rs
