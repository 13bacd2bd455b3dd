// rs_internal.hpp -- structures shared by the host side (rs_table.cpp, rs_solver.cpp) and the
// HIP kernels (rs_kernels.hip).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "rustsolver_amd.h"
#include "rustsolver_amd_diag.h"

namespace rs {

constexpr int kLanePad = 64;      // lane pitch granularity (elements): keeps every row 256-B aligned
constexpr int kVec = 4;           // lanes per thread: one 16-B access per f32/i32 row
constexpr int kBlock = 256;       // threads per workgroup (4 waves of 64)
constexpr int32_t kPruneThreshold = -10000000;  // cfr.rs:352

// ---- job descriptors (device-visible, plain data) --------------------------------------------

// where the utility of one action comes from
enum ChildKind : int32_t {
    CH_BUF = 0,    // u = buf[lane]                      child action/chance node util, or RS_LEAF_UTIL
    CH_CONST = 1,  // u = value                          UNCONTESTED terminal (cfr.rs:316-322)
    CH_SIGN = 2,   // u = sign-compare(buf[lane]) * pot  SHOWDOWN / ALLIN terminal (cfr.rs:323-347)
};

struct ChildSrc {
    const float *buf;
    float value;      // CH_CONST: the utility; CH_SIGN: the pot as f32
    int32_t kind;     // ChildKind; for CH_SIGN bit 8 set = traverser is player 1 (flip the comparison)
};

// One action node visit over all lanes.  Used by the update, node-util and reach kernels.
struct NodeJob {
    void *regrets;            // [A][pitch] table element type
    void *ssum;               // [A][pitch]
    ChildSrc child[RS_MAX_ACTIONS];
    const float *reach;       // [pitch] or nullptr
    float reach_const;        // used when reach == nullptr
    float *out_util;          // [pitch] or nullptr
    float *out_reach[RS_MAX_ACTIONS];  // reach kernel: per child [pitch] or nullptr (not needed)
    uint32_t n_vec;           // pitch / kVec
    uint32_t pitch;           // elements
    float scale;
    int32_t n_actions;
    uint32_t node_index;      // ActionNode.index: part of the opponent-sampling hash
    // deal batches (lanes = deals): table rows are gathered through cidx and updated through atomic deltas
    const uint32_t *cidx;     // [lane pitch] dense cluster id of the acting player on this round, or nullptr
    void *dreg;               // delta rows matching regrets / ssum
    void *dssm;
    uint32_t n_lanes;         // deals in the batch (lanes beyond are padding)
    uint32_t lane_base;       // data-parallel deal batches: global index of this rank's first deal (opponent-sampling hash only)
    // tiled node blocks (big lane tables, rs_table.cpp): the rows of a node are interleaved in tiles of row_stride lanes, [tile][action][row_stride]; the
    // thread's vector v of row a sits at element a * row_stride + 4 * (((v >> tile_shift) * A << tile_shift) + (v & mask)).  Untiled: row_stride = pitch,
    // tile_shift = 31 (one tile)
    uint32_t row_stride;
    uint32_t tile_shift;
    const uint8_t *prune_lane; // RS_UPD_PRUNE on deal batches: [lane pitch] 1 = this deal is traversed with prune = true (cfr.rs:219), nullptr = every lane
};

// chance node: expand (top-down) / reduce (bottom-up) between a parent round and a child round
struct ChanceJob {
    const float *src;   // expand: parent reach [pitch_parent] or nullptr (const); reduce: child util [pitch_child]
    float *dst;         // expand: child reach [pitch_child]; reduce: parent util [pitch_parent]
    float src_const;    // expand with src == nullptr
    float inv;          // expand: 1/len multiplier (cfr.rs:510)
    uint32_t fan;       // deals per parent board
    uint32_t n_clusters;
    uint32_t n_parent_lanes;  // n_boards_parent * n_clusters
    // board axis of the child round sharded over ranks (shard_world == 0: not sharded)
    uint32_t board_off;       // expand: global index of this rank's first child board
    uint32_t n_child_lanes;   // expand: child lanes to write (local boards * n_clusters)
    uint32_t shard_world;     // reduce: src is the exchange buffer [rank][..] instead of one contiguous [boards][C]
    uint32_t rank_stride;     // reduce: floats between two ranks' slots
    uint32_t shard_lo[9];     // reduce: first global board of every rank, then the total
};

// ---- kernel launchers (rs_kernels.hip) ----------------------------------------------------------
struct KernelCfg {
    int dtype;       // RS_I32 / RS_F32 / RS_F16
    int mode;        // RS_UPD_* bits (update only)
};

hipError_t launch_update(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                         KernelCfg cfg, hipStream_t stream);
// d_seed != nullptr selects the sampled-opponent form (cfr.rs:467-476) with *d_seed as the sweep seed
hipError_t launch_node_util(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                            KernelCfg cfg, const uint64_t *d_seed, hipStream_t stream);
hipError_t launch_reach(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                        KernelCfg cfg, const uint64_t *d_seed, hipStream_t stream);
// deal sweeps: AoS shadow of the table nodes (rs_device.hpp gather_rec): one job per action node
struct ShadowJob {
    const int32_t *regrets;   // [A][pitch]
    const int32_t *ssum;
    int32_t *dst;             // [n_clusters][stride]: regrets (half ints), then strategy_sum (half ints) when stride == 2 * half
    uint32_t pitch, n_clusters, n_actions, half;   // half = 4 (A <= 4) or 8
    uint32_t stride, sigma;   // ints of the RECORD: half: regrets only (a node of the sweep's opponent); 2 * half: both arrays (a traverser node).  sigma (with stride == half): the
                              // record holds get_strategy() of the regrets (f32 bits) instead of the regrets: the sweep only reads an opponent node to sample from it
    uint32_t row_stride, pad_;   // ints between two clusters' records: the records of all nodes of one round subtree and role sit side by side in one ROW per cluster
};
// sparse deal sweeps: live deals (reach not NaN) of one subtree root, compacted (order irrelevant: every use commutes)
struct CompactJob {
    const float *reach;   // [n_lanes], or nullptr: every lane is live (the first root)
    uint32_t *list;       // [n_parts][list_stride]
    uint32_t *count;      // [n_parts], count_stride apart
    uint32_t n_lanes;
    // cluster-partitioned workgroups: a live deal goes to the list of part key[lane] / part_size (key = its traverser cluster id)
    const uint32_t *key;  // nullptr when n_parts == 1
    uint32_t part_size, n_parts, list_stride, count_stride;
    // Only the deals the PARENT subtree walked can be live here: scan the parent's live lists (one per cluster range of the parent) instead of the whole batch.
    // nullptr: the parent walks every deal (an unlisted first root): scan lanes 0 .. n_lanes.
    const uint32_t *src_list;
    const uint32_t *src_count;
    uint32_t src_parts, src_list_stride, src_count_stride;
    // Position-indexed rows: the parent's reach-down kernel wrote `reach` at the parent's LIST POSITION (part * src_list_stride + entry), so this scan reads it coalesced;
    // the reach of every live deal is then stored beside its new list entry (rlist, same layout as list), where the subtree's kernels read it coalesced too.
    uint32_t pos_rows, key_stride;   // key_stride: words between two deals' keys (1: a plain id vector; 8: the 32-byte records of an ordered sweep)
    float *rlist;         // [n_parts][list_stride] or nullptr
    uint32_t *plist;      // [n_parts][list_stride] or nullptr: where the parent subtree will read this deal's utility (its list position; the deal id below an unlisted parent)
    // Liveness by mask: the parent's reach-down kernel wrote ONE word per entry with a bit per next-round root it hands a reach to (and the reach rows of those alone): the
    // scan reads 4 bytes per entry instead of a float per (entry, sibling), most of them NaN.  nullptr: liveness = the reach is not NaN.
    const uint32_t *mask;
    uint32_t bit, pad_;
};
// best response over run-outs (rs_br.hip): what depends on the game only is prepared once, the walk runs against the table as it stands
struct BrRun;
int br_prepare(rs_table *t, const rs_tree *tree, const uint8_t *board0, int n_board0, const uint8_t *hands_p0, size_t n_hands_p0, const uint8_t *hands_p1, size_t n_hands_p1,
               const uint32_t *const *cluster, int n_rounds, bool sorted, BrRun **prepared);
int br_execute(BrRun *prepared, int mode /* RS_BR_MAX / RS_BR_AVERAGE */, double *out /* [2] */);
void br_free(BrRun *prepared);
size_t br_held_bytes(const BrRun *prepared);        // device bytes the object holds: the game-only half + the walk's workspace (kept from the first br_execute on)
void br_release_workspace(BrRun *prepared);         // gives the workspace back; the next br_execute allocates it again
int br_last_launches(const BrRun *prepared);        // launches of the last br_execute when it ran the level plan, else -1
// compact jobs [first, first + n) scan one source for n sibling roots (k_compact_siblings): n <= 16, no cluster ranges
struct CompactGroup {
    uint32_t first, n;
};
hipError_t launch_compact_siblings(const CompactJob *d_jobs, const CompactGroup *d_groups, int n_groups, uint32_t max_lanes, hipStream_t stream);
// deal sweeps: packed per-deal inputs of one round (rs_kernels.hip k_pack_attr)
struct u32x4_host { uint32_t x, y, z, w; };
struct PackJob {
    const uint32_t *cid0, *cid1;   // [n] dense cluster ids of the two players on this round (either may be null: a player without nodes there)
    const float *leaf;             // [n] the one leaf buffer every showdown / all-in terminal shares
    const uint8_t *prune;          // [n] or null
    u32x4_host *out;               // [pitch]
    uint32_t n;
};
hipError_t launch_pack_attr(const PackJob *d_jobs, int n_jobs, uint32_t max_n, hipStream_t stream);
// ordered deal sweeps (rs_kernels.hip k_order_*): counting sort of the batch by `key` and the 32-byte per-deal records in that order
constexpr uint32_t kOrderMaxBins = 16384;   // LDS counters of one workgroup (twice: starts and counts)
constexpr int kOrderThreads = 512;
struct OrderJob {
    const uint32_t *key;       // [n] the traverser's cluster id on the last round
    const uint32_t *cid[6];    // [n] cluster ids [2 * round + player], null = a player without nodes there
    const float *leaf;         // [n]
    const uint8_t *prune;      // [n] or null
    uint32_t *tot;             // [n_bins] scratch: deals per bin ...
    uint32_t *cursor;          // [n_bins] ... and how many of them the workgroups have reserved so far (tot + n_bins: one memset clears both)
    void *arec;                // [pitch] x 32 bytes, out
    uint32_t n, n_bins, n_chunks, chunk;
};
hipError_t launch_order(const OrderJob &job, hipStream_t stream);
hipError_t launch_unpermute_f32(const float *in, const void *arec, float *out, uint32_t n, hipStream_t stream);
hipError_t launch_compact_live(const CompactJob *d_jobs, int n_jobs, uint32_t max_lanes, hipStream_t stream);
// the work list of one launch of list-walking kernels: wl[0] = ticket counter (zeroed), wl[1] = n_jobs, wl[2 + j] = trips of jobs 0 .. j-1, wl[2 + n_jobs] = all trips.
// The count of job j is read through the pointer at blob + j * stride + off_count; a trip covers deals_per_trip list entries.
struct WorklistDesc {
    const unsigned char *blob;
    uint32_t *wl;
    uint32_t stride, off_count, n_jobs, deals_per_trip;
};
constexpr int kWorklistBatch = 16;   // work lists one launch of k_worklist builds (one workgroup each): the launches of one group of independent round subtrees
struct WorklistBatch {
    WorklistDesc d[kWorklistBatch];
};
hipError_t launch_worklist(const WorklistBatch &batch, int n, hipStream_t stream);
// d_seed_state != nullptr: the same launch advances the sweep seed (k_next_seed's work), one launch less per sampled deal sweep
hipError_t launch_build_shadow(const ShadowJob *d_jobs, int n_jobs, uint32_t max_clusters, hipStream_t stream, uint64_t *d_seed_state, uint32_t *d_zero = nullptr, uint32_t n_zero = 0);   // d_zero: words zeroed by the same launch (the sweep's list counters)
hipError_t launch_apply_delta(void *regrets, void *dregrets, void *ssum, void *dssum, size_t n_cells, hipStream_t stream);
// the same over the cell ranges of some nodes only (a traverser's sweep writes deltas at ITS nodes; the other half of the delta arrays is zero and need not be read)
struct ApplyJob {
    size_t first_vec;   // in vectors of 4 cells from the start of the table
    size_t n_vec;
};
hipError_t launch_apply_delta_jobs(void *regrets, void *dregrets, void *ssum, void *dssum, const ApplyJob *d_jobs, int n_jobs, size_t max_vec, hipStream_t stream);
// deal sweeps on f32 tables: the per-deal delta rows of one traverser node ([2A][pitch]: regret deltas, then strategy-sum deltas) are summed per cluster over the cluster's
// deals IN DEAL ORDER (members[start[c] .. start[c + 1]) ascending) from 0.0 and the sums added to the node's table rows
struct ApplyF32Job {
    void *reg, *ssm;           // [A][tpitch] table elements (binary32 or binary16)
    const float *rows;         // [2A][pitch]
    const uint32_t *start, *members;
    uint32_t n_actions, tpitch, n_clusters, pad_;
};
hipError_t launch_apply_f32_rows(const ApplyF32Job *d_jobs, int n_jobs, uint32_t max_clusters, uint32_t pitch, int dtype, bool rmplus, hipStream_t stream);
// i32 deal sweeps with delta rows: the walk of a round subtree stored, for every traverser node, [2A][pitch] i32 deltas at the list position of each walked deal (zero where
// the deal did not come by).  One job = the A rows of one node and array + the key row beside them (the traverser's cluster of every position); a workgroup takes one chunk
// of positions, sums it per cluster in an LDS tile [n_rows][n_clusters] and adds the non-zero cells to the delta table's rows -- integer adds: any order, same bits
struct RowSumJob {
    const int32_t *rows;       // [n_rows][pitch]
    const uint32_t *key;       // [n]
    const uint32_t *count;     // device count of list entries, or null: n_const
    int32_t *dst;              // [n_rows][tpitch] inside the delta table
    uint32_t n_rows, pitch, tpitch, n_clusters, n_const, direct;   // direct: dst is the TABLE's own rows of the node (k_row_apply: no LDS tile, any cluster count)
    int32_t *mirror;           // direct: the node's KEPT shadow records (this array's half of them), which take the same additions as the table -- or null
    const uint32_t *primary;   // ... and ONLY they while *primary != 0 (the working copy of a training loop: rs_solver.cpp solver_kept_primary)
    uint32_t mstride, pad_;    // ints between two clusters' records there
};
struct DiscountJob {
    void *regrets, *ssum;
    size_t n_vec;
};
hipError_t launch_discount_jobs(const DiscountJob *d_jobs, int n_jobs, size_t max_vec, float d, int dtype, hipStream_t stream);
hipError_t launch_unbuild_shadow(const ShadowJob *d_jobs, int n_jobs, uint32_t max_clusters, hipStream_t stream);
hipError_t launch_row_apply(const RowSumJob *d_jobs, int n_jobs, uint32_t max_entries, hipStream_t stream);
// data-parallel deal batches: the rows of the direct rounds as 12-byte (job, row, cluster, delta) items for the ranks to exchange, and every rank's items applied
hipError_t launch_rows_to_items(const RowSumJob *d_jobs, int first_job, int n_jobs, uint32_t max_entries, uint32_t *d_items, uint32_t *d_cursor, uint32_t cap, hipStream_t stream);
hipError_t launch_apply_items(const RowSumJob *d_jobs, const uint32_t *d_items, uint32_t n, hipStream_t stream);
hipError_t launch_pack_cells(void *dregrets, void *dssum, const ApplyJob *d_jobs, const size_t *d_pack_off, int n_jobs, size_t max_vec, void *packed, size_t total_vec, bool unpack, hipStream_t stream);
hipError_t launch_row_sums(const RowSumJob *d_jobs, int n_jobs, uint32_t max_entries, uint32_t chunk, uint32_t max_cells, hipStream_t stream);
constexpr uint32_t kRowSumMaxCells = 16384;   // ints of one job's LDS tile (64 KiB: two workgroups per CU)
constexpr uint32_t kRowSumChunk = 262144;     // list positions per workgroup
// stable counting sort of 0 .. n-1 by key (rs_kmeans.hip): members[start[c] .. start[c + 1]) = the indices with key c, ascending; scratch: tile_hist [ceil(n / 512)][k], total [k]
hipError_t launch_member_lists(const uint32_t *keys, size_t n, int k, uint32_t *tile_hist, uint32_t *total, uint32_t *start /* [k + 1] */, uint32_t *members /* [n] */,
                               hipStream_t stream);
size_t member_list_tiles(size_t n);
hipError_t launch_showdown_sign(const uint8_t *cards, float *sign, uint32_t n, uint32_t pitch, hipStream_t stream);
hipError_t launch_next_seed(uint64_t *d_state /* {base, call_index, seed} */, hipStream_t stream);
hipError_t launch_probe_copy(const void *in, void *out, size_t bytes, unsigned blocks, hipStream_t stream);   // rs_stream_probe
hipError_t launch_prune_reach(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec,
                              int n_actions, KernelCfg cfg, hipStream_t stream);
hipError_t launch_strategy(const void *src /*[A][pitch], or tiled*/, float *dst /*[A][pitch]*/, uint32_t pitch, uint32_t row_stride, uint32_t tile_shift, int n_actions,
                           int dtype, hipStream_t stream);
hipError_t launch_chance_expand(const ChanceJob *d_jobs, int n_jobs, size_t max_child_lanes, bool vec4, hipStream_t stream);
hipError_t launch_chance_reduce(const ChanceJob *d_jobs, int n_jobs, size_t max_parent_lanes, bool vec4, hipStream_t stream);
hipError_t launch_discount(void *regrets, void *ssum, size_t n_cells, float d, int dtype, hipStream_t stream);
hipError_t launch_fill_random(void *dst, size_t n_cells, uint64_t seed, int64_t lo, int64_t hi, int dtype,
                              hipStream_t stream);
hipError_t launch_fill_uniform(float *dst, size_t n, uint64_t seed, float lo, float hi, hipStream_t stream, size_t index_offset = 0);
// op 0: fill i32 / binary16 cells with lo + hash(seed, node, action, lane_off + lane) % span; op 1: add the order-independent checksum of the same keys and the cells' bits to *d_out
hipError_t launch_logical(void *x, size_t n, uint32_t node, uint32_t A, size_t tile, size_t lanes, size_t lane_off, size_t es, int op, uint64_t seed, int64_t lo, int64_t hi,
                          unsigned long long *d_out, hipStream_t stream);
hipError_t launch_plant_saturating(void *regrets, size_t n_cells, uint64_t seed, uint32_t one_in, hipStream_t stream);
hipError_t launch_plant_outliers(float *dst, size_t n, uint64_t seed, uint32_t one_in, float magnitude, hipStream_t stream);
hipError_t launch_delta_sub(void *x, const void *snap, size_t n, int dtype, hipStream_t stream);  // x -= snap
hipError_t launch_delta_add(void *x, const void *snap, size_t n, int dtype, hipStream_t stream);  // x += snap
hipError_t launch_gather_lanes(const void *block, const uint32_t *d_lanes, size_t n, uint32_t A, size_t tile, size_t es, void *d_out, hipStream_t stream);
hipError_t launch_checksum(const void *x /* one node block */, size_t n, size_t cell_off, uint32_t A, size_t tile, size_t lanes, size_t es, unsigned long long *d_out, hipStream_t stream);
hipError_t launch_selftest_division(size_t n, uint64_t seed, unsigned long long *d_mismatches, float *d_first_bad, hipStream_t stream);
hipError_t launch_delta_swap(void *x, void *snap, size_t n, int dtype, hipStream_t stream);      // d = snap - x; x = snap; snap = d

// ---- kernel-form switches (rs_knobs.cpp) -------------------------------------------------------------
// What a solver / table / trainer was asked to do differently from the engine's own choices: rs_kernel_forms / rs_table_params / rs_deal_trainer_params.prefetch
// (the caller's API), then the environment (tests and the A/B tools only), resolved ONCE per object in knobs_resolve -- the only place of the library that reads
// a knob from the environment.  kUnset = nobody said anything: the code's own rule decides.
constexpr int kUnset = -2147483647 - 1;
struct Knobs {
    // API-backed (rs_kernel_forms); the environment overrides three of them for tests
    int fan = kUnset;               // rs_kernel_forms.lane_fan: 0 none / 1 the expand step inside the subtree kernel
    int lanes = kUnset;             // RS_JIT_LANES / deals_per_thread: 1 / 2 / 4
    int shadow_all = 0;             // rs_kernel_forms.shadow = RS_SHADOW_ALL
    int no_kept = 0;                // rs_kernel_forms.kept_records = RS_FORM_OFF
    int ordered = kUnset;           // RS_JIT_ORDERED / deal_order: 1 on / 0 off
    int rows = kUnset;              // RS_JIT_ROWS / delta_rows: 1 on / 0 off (delta rows by list position + one summing pass per round)
    int direct_rows = kUnset;       // RS_JIT_DIRECT_ROWS / direct_rows: 1 on / 0 off
    // test-only: forms the engine picks by size, forced onto small inputs
    int rows_chunk = kUnset;        // RS_JIT_ROWS_CHUNK: list entries one workgroup of the summing pass takes (small values: several chunks per row)
    int scan_all = kUnset;          // RS_JIT_SCAN_ALL: 0 = a root's live deals are compacted from its parent's lists (what batches beyond 64 K deals get), 1 = from the whole batch
    int lds_max = kUnset;           // RS_JIT_LDS_MAX: bytes of LDS a workgroup may take (small values: cluster ranges on every round)
    int max_blocks = kUnset;        // RS_JIT_MAX_BLOCKS: grid cap of the generated kernels (small values: several trips per workgroup)
    int no_siblings = kUnset;       // RS_JIT_NO_SIBLINGS: 1 = one compaction job per root (k_compact_live), 0 = one per parent (k_compact_siblings: batches beyond 512 K deals)
    long tile_lanes = kUnset;       // RS_TABLE_TILE_LANES: lanes per tile of a tiled node block; every node wider than that is tiled (0: never tile)
    // test-only: facilities with a fallback
    int no_stage = 0;               // RS_JIT_NO_STAGE: the list walkers gather their records per node instead of staging their deals' rows in LDS
    int jit_no_procs = 0;           // RS_JIT_NO_PROCS: the kernels of a plan are compiled in this process one by one (what happens anyway when the rs_jitc helper is missing)
    int br_depth_first = 0;         // RS_BR_DEPTH_FIRST: the best response walks the tree depth first (what happens anyway when the level plan's buffers do not fit)
    int no_merge = 0;               // RS_JIT_NO_MERGE: small deal batches keep one launch per subtree shape, spread over the auxiliary streams (what larger batches get)
    int no_overlap = 0;             // RS_JIT_NO_OVERLAP: the launches of a round run one after the other on the table's stream (profiling: overlapped kernels stretch each other's durations)
    int dump = 0;                   // RS_JIT_DUMP: every generated source is written to /tmp/rs_tree_kernel_<hash>.hip
};
Knobs knobs_resolve(const rs_kernel_forms *forms);
// Several generated kernels as ONE: every member's body becomes a device function in a namespace of its own (its JArgs with it), the entry point dispatches on blockIdx.y
// ranges.  offs[k] = {src_struct, src_entry, src_body} of sources[k].  Empty string: the members' preludes differ (deals per thread), they cannot share a translation unit.
constexpr int kMergeMax = 16;
constexpr int kMergeCountJobs = 640;   // jobs whose list counters the merged kernel finds through its arguments (2.5 KB of the 4 KB a kernel may take)
struct MergedArgs {                // the merged kernel's first argument, by value
    const void *blob[kMergeMax];
    unsigned first[kMergeMax + 1];
    unsigned per_block;            // deals a workgroup takes per trip (256 threads x deals per thread)
    const unsigned *counts;        // the plan's list counters, or nullptr: no early exit
    unsigned cidx[kMergeCountJobs];   // per job (grid row): its counter's index in `counts`, 0xffffffff = the job walks the whole batch
};
std::string jit_merge_sources(const std::vector<const std::string *> &sources, const std::vector<const size_t *> &offs, const std::string &entry);
const char *rccl_library_override();   // $RS_RCCL_LIB (tests: tests/stub_rccl.c), or nullptr
std::string jit_cache_dir();   // $RS_JIT_CACHE (empty string: no disk cache), else ~/.cache/rustsolver_amd

// ---- tree-specialised kernels (rs_jit.cpp) ---------------------------------------------------------
struct JitSubtree {
    std::string source;
    std::string entry;             // kernel name: rs_tree_p{traverser}_{lanes|deals|deals_lds}[_sampled]
    size_t off_bmask = 0, off_bbit = 0;   // deals: the reach-down kernel's liveness mask row and the bit of every next-round root (JArgs.bmask, JArgs.bbit[])
    size_t src_struct = 0, src_entry = 0, src_body = 0;   // where `struct JArgs`, the kernel's signature and its body start inside `source` (jit_merge_sources)
    int threads = 256;             // workgroup size the kernel was generated for
    bool worklist = false;         // the kernel takes a third argument (the work list of its launch, rs_kernels.hip k_worklist) and a 1-D grid
    int lanes = 4;                 // lanes (deals) per thread the kernel was generated for: n_vec = pitch / lanes
    std::vector<int> node_ids;     // tree node id of every action node, in the order the kernel indexes reg[] / ssm[]
    std::vector<int> leaf_terms;   // one terminal id per distinct leaf buffer, in the order of leaf[]
    std::vector<int> const_terms;  // terminal ids in the order of cval[]
    int max_actions = 0;
    size_t off_reg = 0, off_ssm = 0, off_leaf = 0, off_reach = 0, off_out = 0, off_seed = 0, off_cval = 0, off_nidx = 0,
           off_reach_const = 0, off_scale = 0, off_n_vec = 0, off_pitch = 0, off_row_stride = 0, off_tile_shift = 0, args_size = 0;
    size_t off_dreg = 0, off_dssm = 0, off_cidx = 0, off_tpitch = 0, off_n_lanes = 0;   // deal batches only
    size_t off_loff = 0, off_sstride = 0, off_resident = 0, off_trans = 0;                               // deal batches: LDS tile placement
    size_t off_shd = 0;                                                                   // deal batches: AoS shadow of every node
    size_t off_list = 0, off_count = 0;                                                   // sparse deal sweeps: list of live deals and its length
    size_t off_butil = 0, off_breach = 0;                                                 // round subtrees: utility / reach buffers of the next round's roots
    size_t off_prune = 0;                                                                 // deal batches: per-deal prune flags (u8), may be null
    size_t off_attr = 0;                                                                  // sparse deal sweeps: packed per-deal inputs of the subtree's round, may be null
    size_t off_rlist = 0;                                                                 // the reach of every entry of the live list (position-indexed rows), may be null
    size_t off_plist = 0;                                                                 // the parent-subtree position of every entry of the live list, may be null
    size_t off_klist = 0;                                                                 // delta rows: the key row of the job's list (the traverser's cluster of every entry), written by the walk
    size_t off_hrow = 0, off_hpitch = 0;                                                  // hand-off rows of the job (reach-down kernel -> walk), pitch between them
    size_t off_rowp = 0;                                                                  // staged rows: the shadow rows of player 0's and player 1's nodes of the subtree
    bool staged = false;
    size_t stage_lds_bytes = 0;                                                           // per workgroup
    int n_handed = 0;                                                                     // opponent nodes whose draw the reach-down kernel hands to the walk (+ 1 row of packed actions)
    size_t off_c0 = 0, off_rcount = 0, off_rp = 0;                                         // the cluster range a job's LDS tiles cover
    size_t off_fan = 0, off_inv = 0, off_cvec = 0;                                        // lane sweeps: deals below the ENUM chance node the kernel walks itself
    std::vector<int> boundary_roots;   // tree id of every next-round root below this subtree, in the order of butil[] / breach[]
};
// staged rows (rs_device.hpp stage_rows): the shadow rows of a round subtree as the generated kernel needs to know them -- structure only, no addresses
struct JitStage {
    int ch[2] = {0, 0};       // per PLAYER: 16-byte chunks of a row of that player's nodes in this round subtree (0: the kernel reads none)
    int chp[2] = {0, 0};      // the same in LDS: ch rounded up to an odd number
    std::vector<int> off;     // per tree node: ints from the start of its row to its record
};
constexpr int kStageMaxChunks = 16;   // rows beyond 256 bytes keep their gathers (64 deals x 17 chunks x 16 B = 17 KB of LDS per wave)
void jit_emit_subtree(const std::vector<rs_tree_node> &nodes, int root, int p, const std::vector<char> &has_own,
                      const std::vector<int> &leaf_buf, const std::vector<int> &leaf_flags, int dtype, int arith, bool sampled,
                      bool deals, bool lds, bool sparse, bool down, bool prune, int lanes, const std::vector<char> *cut, JitSubtree &out, const Knobs &knobs, int fan = 0, bool packed = false,
                      bool posrows = false, bool worklist = false, bool ordered = false, bool seg = false, bool rows = false,
                      const std::vector<char> *sigma = nullptr /* per tree node: its shadow record holds the strategy (opponent nodes of a deal sweep) */,
                      bool handoff = false /* deal sweeps: the reach-down kernel stores its draws by list position, the walk reads them (both kernels of a root alike) */,
                      const JitStage *stage = nullptr /* list walkers, one deal per lane: the subtree's records come from rows the wave stages in LDS */);
bool jit_available();
int jit_get_kernel(const std::string &source, const std::string &entry, int device, hipFunction_t *fn, bool dump = false);
uint64_t jit_source_key(const std::string &source);   // what the caches are keyed by (source + compiler version + options)
struct JitRequest {
    const std::string *source, *entry;
    hipFunction_t fn;
};
// the same for many kernels: the ones no cache holds are compiled concurrently on host threads, then loaded one after the other
int jit_get_kernels(std::vector<JitRequest> &reqs, int device, bool dump = false);
int jit_compile_only(const std::string &source, bool dump = false);
int jit_compile_many(const std::map<std::string, int> &sources, bool dump = false);   // the keys, by helper processes side by side (compile-only check of the CPU suite)
const char *jit_device_source();

// ---- host-side objects ----------------------------------------------------------------------------
int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);

inline size_t elem_size(int dtype) { return dtype == RS_F16 ? 2 : 4; }
inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

struct Profile {
    bool on = false;
    struct Pending {
        hipEvent_t a, b;
        int kind;
        double bytes;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::vector<hipEvent_t> marks;     // rs_profile_mark: events on the table's stream, read (and recycled) by rs_profile_marks
    rs_profile acc{};
};

}  // namespace rs

struct rs_tree {
    std::vector<rs_tree_node> nodes;
    int n_action_nodes = 0;
};

struct rs_table {
    int device = 0;
    int dtype = RS_I32;
    hipStream_t stream = nullptr;
    std::vector<rs_node_desc> nodes;
    std::vector<size_t> pitch;        // per node, elements
    std::vector<size_t> tile;         // per node: lanes per tile of its block [pitch / tile][A][tile]; == pitch: the plain [A][pitch] block
    std::vector<size_t> cell_off;     // per node, element offset of its [A][pitch] block
    size_t n_cells = 0;
    void *d_regrets = nullptr;        // n_cells elements
    void *d_ssum = nullptr;
    void *d_snap_regrets = nullptr;   // replicated-round snapshots (multi-GPU), lazily allocated, compact
    void *d_snap_ssum = nullptr;
    uint32_t rep_mask = 0;            // rounds covered by the snapshot
    std::vector<int> rep_nodes;       // replicated node indices
    std::vector<size_t> rep_off;      // element offset of each of them inside the compact snapshot
    size_t rep_cells = 0;
    void *d_dregrets = nullptr;       // deal batches: delta tables (same layout as the table), zero between sweeps
    void *d_dssum = nullptr;
    std::vector<struct rs_solver *> solvers;   // live solvers built on this table: released before the table goes away
    uint64_t epoch = 0;                        // counts the calls that wrote regrets / strategy sums (uploads, fills, sweeps, discounts, ...): a deal solver's KEPT shadow
                                               // records (rs_solver.cpp setup_table_shadow) are rebuilt when the table moved on without them
    void *d_query = nullptr;          // scratch of the single-info-set strategy queries (rs_get_strategy)
    uint32_t *d_err_sink = nullptr;   // error word for card kernels whose caller passed none (rs_deals_sample)
    void *d_km_scratch = nullptr;     // staged k-means centers (rs_kmeans_predict), grow-only
    size_t km_scratch_bytes = 0;
    rs::NodeJob *d_job = nullptr;     // one device job slot for the per-node ABI calls (stream-ordered reuse)
    rs::Profile prof;

    bool tiled(int node) const { return tile[size_t(node)] != pitch[size_t(node)]; }
    uint32_t tile_shift(int node) const {   // log2 of the VECTORS (4 lanes) per tile; 31 = one tile
        if (!tiled(node)) return 31;
        uint32_t sh = 0;
        while ((size_t(rs::kVec) << sh) < tile[size_t(node)]) ++sh;
        return sh;
    }
    // element offset inside the node's block of (action a, lane l)
    size_t elem_index(int node, size_t a, size_t l) const {
        const size_t T = tile[size_t(node)];
        return ((l / T) * nodes[size_t(node)].n_actions + a) * T + l % T;
    }
    void *regrets_ptr(int node) const { return (char *)d_regrets + cell_off[node] * rs::elem_size(dtype); }
    void *ssum_ptr(int node) const { return (char *)d_ssum + cell_off[node] * rs::elem_size(dtype); }
};

struct rs_card_abs;
namespace rs {
// card kernels on a stream of the caller's choice (rs_cards.hip); the C ABI forms use the table's stream
int card_abs_clusters_on(rs_card_abs *abs, rs_table *t, hipStream_t stream, const uint8_t *d_cards, uint32_t n_deals, uint32_t *d_cluster_p0,
                         uint32_t *d_cluster_p1);
int deals_sample_on(rs_table *t, hipStream_t stream, uint64_t seed, uint64_t first_deal, uint64_t board_mask, const uint8_t *d_hands_p0, uint32_t n_hands_p0,
                    const uint8_t *d_hands_p1, uint32_t n_hands_p1, uint32_t n_deals, uint8_t *d_cards, uint32_t *d_err, float *d_sign /* fused showdown signs, may be null */,
                    uint8_t *d_flags /* fused per-deal prune flags, may be null */, uint64_t prune_threshold);
int deal_prune_flags_on(rs_table *t, hipStream_t stream, uint64_t seed, uint64_t first_deal, uint64_t prune_threshold, uint32_t n_deals, uint8_t *d_flags);
int table_copy_node_raw(rs_table *t, int node, int which /* 0 regrets, 1 strategy_sum */, void *host /* [A][lanes], table element type */, int dir /* 0 up, 1 down */);
void solver_release_device(struct rs_solver *s);   // frees a solver's device state and detaches it from its table
void solver_table_discounted(struct rs_solver *s, float d, uint64_t epoch_before);   // rs_discount ran on the solver's table: the same sweep over its kept shadow records
// Training loops (rs_train, rs_deal_trainer_train) make the kept records the WORKING COPY for their duration: on = 1 at the start (records rebuilt if out of step), 0 before
// returning (the table's rows written back from the records).  While on, nothing but the solver's sweeps and solver_discount_primary may touch the table.
int solver_kept_primary(struct rs_solver *s, bool on);
constexpr uint64_t kKeptPrimaryMinTrips = 16;   // training loops shorter than this leave the table's rows the working copy (the write-back at the end would cost more than it saves)
bool solver_is_primary(const struct rs_solver *s);
int table_settle(rs_table *t);   // rs_table.cpp: the table's rows up to date before anything reads or writes them (a training loop's working copy written back)
// rs_comm.cpp: the collectives of a data-parallel deal sweep (rs_solver.cpp solver_exchange_deltas)
int comm_world(const struct rs_comm *c);
int comm_rank(const struct rs_comm *c);
int comm_allreduce_i32(struct rs_comm *c, rs_table *t, void *d_buf, size_t n);
int comm_allgather_u32(struct rs_comm *c, rs_table *t, const void *d_send, void *d_recv, size_t n);
// ordered deal sweeps: the caller sorts the per-deal records of every batch itself, ahead of the sweep (true: accepted -- an ordered solver on one GPU that has not swept yet)
bool solver_order_ahead(struct rs_solver *s, bool on, int (*before_sweep)(void *ctx, int traverser), void *ctx);
int solver_order_on(struct rs_solver *s, int traverser, hipStream_t stream, const uint32_t *const cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS], const float *leaf, const uint8_t *prune);
int solver_discount_primary(struct rs_solver *s, float d);   // rs_discount while on: the kept records and the table WITHOUT their nodes
// profiling hooks used around launches
void prof_begin(rs_table *t, int kind, double bytes);
void prof_end(rs_table *t);
double algo_bytes_update(const rs_table *t, int node, int n_buf_children, bool has_reach, bool has_out);
}  // namespace rs
