// rs_solver.cpp -- the host-side traversal scheduler: MCCFRTrainer (cfr.rs:146-297) for the
// batched lane model.  The branchy public-tree walk happens ONCE, here on the host, and yields a
// static launch plan per traverser:
//   top-down, by tree depth   : opponent nodes write reach[child] = sigma[a]*reach (cfr.rs:585),
//                               ENUM chance nodes expand reach * 1/len to the child round's boards
//   bottom-up, by tree depth  : opponent nodes write util = sum sigma*u (cfr.rs:588), traverser nodes
//                               run the regret / strategy_sum update (cfr.rs:612-621 or :413-464),
//                               ENUM chance nodes sum their deals (cfr.rs:519)
// Nodes of one depth, kernel kind and action count share one launch (blockIdx.y = node).  Terminal
// children cost no launch and no buffer: their utility is a constant or a sign lookup folded into
// the consuming kernel.  rs_iterate replays the plan (optionally as one hipGraph).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "rs_internal.hpp"

using namespace rs;

namespace {

constexpr size_t kCountStride = 64;   // u32 elements between two live-deal counters (256 B)
constexpr uint32_t kScanParentMin = 65536;   // deal batches beyond this size compact a round subtree's live deals from its parent's lists (rs_solver.cpp scan_parent)
constexpr size_t kWorklistLdsBytes = 64;   // in front of the tiles of a work-list kernel: lds_all[0] holds the ticket (rs_jit.cpp)
enum LaunchKind { L_REACH, L_PRUNE_REACH, L_EXPAND, L_UPDATE, L_NODE_UTIL, L_REDUCE, L_TREE, L_SEED, L_APPLY, L_SHADOW, L_COMPACT, L_NANFILL, L_PACK, L_ORDER };

struct Launch {
    int group = 0;                      // > 0: consecutive launches of one group are independent of each other (round subtrees) and may overlap
    int kind;
    int n_actions = 0;
    int first_job = 0, n_jobs = 0;
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
    hipFunction_t fn = nullptr;
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
    ApplyJob *d_apply_jobs = nullptr;   // deal sweeps: the cell ranges of the traverser's own nodes (where its deltas are)
    int n_apply_jobs = 0;
    size_t apply_max_vec = 0;
    size_t aux_bytes = 0;               // device memory of this plan beside the arena: live-deal lists and the reach rows of the round subtrees
    size_t n_count_words = 0;           // u32 words of d_counts (all counters, kCountStride apart)
    uint32_t *d_counts = nullptr;       // [n_compact]
    CompactJob *d_compact_jobs = nullptr;
    std::vector<CompactJob> compact_jobs;
    std::vector<size_t> count_off;      // per compact job: index of its first counter (a job has one per cluster range)
    uint32_t compact_max_lanes = 0;
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

}  // namespace

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
    static constexpr int kAux = 4;
    hipStream_t aux[kAux] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {nullptr, nullptr, nullptr, nullptr};
    // deal sweeps through tree-specialised kernels read the table from an AoS shadow rebuilt at the start of every sweep
    int32_t *d_shadow = nullptr;
    ShadowJob *d_shadow_jobs = nullptr;   // the jobs of traverser 0's sweep, then those of traverser 1's (the same nodes, different record widths)
    int n_shadow_jobs = 0;                // per traverser
    std::vector<size_t> shadow_off_p[2];  // per traverser and table node, in ints (SIZE_MAX: no shadow)
    std::vector<uint32_t> shadow_stride_p[2];
    uint32_t shadow_max_clusters = 0;
    // sparse deal sweeps fetch the per-deal inputs of a round (both cluster ids, leaf value, prune flag) as ONE packed 16-byte record per live deal
    void *d_attr[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};
    PackJob *d_pack_jobs = nullptr;
    int n_pack_jobs = 0;
    unsigned attr_used = 0;             // bit r: some generated kernel reads the packed records of round r (only the list-walking forms do)
    // ordered sweeps (rs_kernel_forms.deal_order): traverser p's sweep walks the batch sorted by p's cluster id on the last round; d_arec holds the 32-byte per-deal
    // records in that order (rebuilt at the start of every sweep by k_order_*), d_attr[r] all point at it
    bool ordered = false;
    int order_round = 0;                // the last betting round of the tree
    void *d_arec = nullptr;
    uint32_t *d_order_tot = nullptr;    // [2][n_bins]: counts and cursors of the counting sort
    OrderJob order_job[2];              // per traverser
    bool deal_mode = false;             // lanes are deals (rs_solver_create_deals)
    rs_deal_batch deals{};
    uint64_t *d_seed_state = nullptr;   // RS_OPP_SAMPLE: {base seed, call index, seed of the current sweep}
    const uint64_t *d_seed() const { return d_seed_state ? d_seed_state + 2 : nullptr; }
};

namespace {

#define RS_HIP(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

// ---- geometry / validation ------------------------------------------------------------------------------
int derive_geometry(rs_solver *s) {
    const rs_table *t = s->table;
    const rs_tree &tr = s->tree;
    if (int(t->nodes.size()) != tr.n_action_nodes)
        return fail(RS_ERR_INVALID, "rs_solver_create: table has " + std::to_string(t->nodes.size()) +
                                        " rows but the tree has " + std::to_string(tr.n_action_nodes) + " action nodes");
    if (s->deal_mode) {
        // lanes = deals on every round; the table keeps the reference's [action_node][cluster] shape
        const size_t deal_pitch = round_up(s->deals.n_deals, kLanePad);
        for (const rs_tree_node &nd : tr.nodes) {
            if (nd.kind != RS_NODE_ACTION) continue;
            const rs_node_desc &d = t->nodes[nd.index];
            if (d.n_actions != uint32_t(nd.n_children) || d.player != nd.player || d.round_idx != nd.round_idx)
                return fail(RS_ERR_INVALID, "rs_solver_create_deals: table row " + std::to_string(nd.index) + " does not match the tree");
            if (d.n_boards != 1)
                return fail(RS_ERR_INVALID, "rs_solver_create_deals: the table must have n_boards = 1 (deals index clusters, not boards)");
            if (t->tiled(nd.index))
                return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: deal sweeps gather from plain [action][cluster] node blocks; this table's node " +
                                                    std::to_string(nd.index) + " is tiled (RS_TABLE_TILE_LANES)");
            if (!s->deals.d_cluster[nd.round_idx][nd.player])
                return fail(RS_ERR_INVALID, "rs_solver_create_deals: no cluster ids for round " + std::to_string(nd.round_idx) +
                                                " player " + std::to_string(nd.player));
            s->n_rounds = std::max(s->n_rounds, nd.round_idx + 1);
        }
        for (int r = 0; r < RS_MAX_ROUNDS; ++r) {
            s->n_boards[r] = 1;
            s->pitch[r] = deal_pitch;
        }
        s->n_clusters = s->deals.n_deals;   // lanes per round = n_boards * n_clusters = n_deals
        return RS_OK;
    }
    bool seen[RS_MAX_ROUNDS] = {false, false, false};
    for (const rs_tree_node &nd : tr.nodes) {
        if (nd.kind != RS_NODE_ACTION) continue;
        const rs_node_desc &d = t->nodes[nd.index];
        if (d.n_actions != uint32_t(nd.n_children) || d.player != nd.player || d.round_idx != nd.round_idx)
            return fail(RS_ERR_INVALID, "rs_solver_create: table row " + std::to_string(nd.index) + " does not match the tree");
        const int r = nd.round_idx;
        if (!seen[r]) {
            seen[r] = true;
            s->n_boards[r] = d.n_boards;
            s->pitch[r] = t->pitch[nd.index];
            if (s->n_clusters == 0) s->n_clusters = d.n_clusters;
            s->n_rounds = std::max(s->n_rounds, r + 1);
        }
        // lane model: lane (b, c) addresses row c of BOTH players' tables on every street
        if (d.n_boards != s->n_boards[r] || d.n_clusters != s->n_clusters)
            return fail(RS_ERR_UNSUPPORTED,
                        "rs_solver_create: the lane model needs one cluster count for both players and all rounds, and one "
                        "board count per round (node " + std::to_string(nd.index) + ")");
    }
    for (int r = 0; r < s->n_rounds; ++r)
        if (!seen[r]) return fail(RS_ERR_INVALID, "rs_solver_create: no action node in round " + std::to_string(r));
    if (s->params.shard_world > 1) {
        const int W = s->params.shard_world, g = s->params.shard_rank, sr = s->params.shard_round;
        if (W > 8 || g < 0 || g >= W || sr < 1 || sr >= s->n_rounds)
            return fail(RS_ERR_INVALID, "rs_solver_create: bad shard_world / shard_rank / shard_round (world <= 8, 1 <= round < n_rounds)");
        if (s->params.chance_mode != RS_CHANCE_ENUM)
            return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: board sharding of a multi-round sweep needs RS_CHANCE_ENUM");
        const uint32_t G = s->params.shard_global_boards, base = G / W, rem = G % W;
        if (G % s->n_boards[sr - 1] != 0) return fail(RS_ERR_INVALID, "rs_solver_create: shard_global_boards must be a multiple of the parent round's boards");
        for (int q = 0; q <= W; ++q) s->shard_lo[q] = uint32_t(q) * base + std::min<uint32_t>(uint32_t(q), rem);
        if (s->n_boards[sr] != s->shard_lo[g + 1] - s->shard_lo[g] || s->n_boards[sr] == 0)
            return fail(RS_ERR_INVALID, "rs_solver_create: this rank must hold boards [" + std::to_string(s->shard_lo[g]) + ", " +
                                            std::to_string(s->shard_lo[g + 1]) + ") of the sharded round");
        s->sharded = true;
        s->slot_lanes = round_up(size_t(base + (rem ? 1 : 0)) * s->n_clusters, kLanePad);
    }
    for (int r = 1; r < s->n_rounds; ++r) {
        if (s->sharded && r == s->params.shard_round) continue;   // local boards of the sharded round are a slice
        if (s->n_boards[r] % s->n_boards[r - 1] != 0)
            return fail(RS_ERR_INVALID, "rs_solver_create: n_boards of a round must be a multiple of the previous round's");
        if (s->params.chance_mode == RS_CHANCE_PASS && s->n_boards[r] != s->n_boards[r - 1])
            return fail(RS_ERR_INVALID, "rs_solver_create: RS_CHANCE_PASS needs the same board count on every round");
    }
    return RS_OK;
}

struct Builder {
    rs_solver *s;
    int p;  // traverser
    Plan &plan;
    const std::vector<rs_tree_node> &nodes;
    std::vector<int> depth, lane_round;
    std::vector<char> has_own, closed, fused_root, inside;
    // Lane sweeps with ENUM chance nodes: a round's action nodes that have chance nodes below them form a ROUND SUBTREE cut at those chance nodes: a reach-down kernel
    // (rows for the chance nodes that need one) and a walk-up kernel that reads the chance nodes' utility rows -- instead of one level kernel per depth, kind and action count
    std::vector<char> lr_root;        // root of such a round subtree
    std::vector<char> next_root;      // action node directly below a chance node: where the generated kernels stop (`cut`)
    std::vector<std::vector<int>> lr_bnd;   // per lr root: the ENUM chance nodes below it (inside its round)
    bool lane_rounds = false;
    std::vector<char> fan_root;       // fused root directly below an ENUM chance node whose deals its kernel walks itself (no expand / reduce launch, no child-round rows)
    std::vector<ReachSrc> reach;      // reach source feeding each node
    std::vector<size_t> util_off;     // arena offset of a node's util buffer (+1; 0 = none)
    std::vector<size_t> reach_off;    // arena offset of a node's own reach buffer (+1; 0 = alias / const)
    size_t arena = 0;
    int max_depth = 0;
    std::map<const float *, int> leaf_ids;   // leaf buffer -> id, so that kernels see the sharing pattern, not pointers
    // deal batches: ONE generated subtree per betting round, cut at the chance nodes (a deal has one run-out, cfr.rs:306-313).  A round
    // subtree has a DOWN kernel (reach for the next round's roots) and the usual kernel that walks back up and updates the table.
    bool round_mode = false;
    int jit_lanes = 4;                   // deals per thread of the generated deal kernels: the first round's subtree (1 for small batches) ...
    int jit_lanes_below = 4;             // ... and the subtrees of later rounds, which walk short live-deal lists
    static constexpr uint32_t kSmallDealBatch = 1u << 18;   // measured on the river game: 64 K deals 0.161 -> 0.101 ms per batch, 256 K 0.175 -> 0.152, 1 M 0.295 -> 0.314
    std::vector<std::vector<int>> bnd;       // per round root: the next-round roots below it
    std::vector<int> nan_slot;               // per round root (except the first): slot of its reach buffer in the NaN-prefilled arena
    int n_nan = 0;
    int first_root = -1;
    std::vector<std::vector<int>> roots_of_round;   // round subtrees by betting round
    int next_group = 0;
    int lds_limit = 64 * 1024;               // what the device gives ONE workgroup (MI355X: 160 KiB)
    bool want_lists = false, want_parts = false;
    std::vector<int> parent_root_;           // round subtrees: the root of the round subtree above a root
    bool scan_parent = false;                // the compaction of a root's live deals scans its parent's lists, not the whole batch
    bool pos_rows = false;                   // scan_parent only: the reach rows between a listed root's reach-down kernel and its children's compaction are indexed by the root's
                                             // LIST POSITION (written and read coalesced) and the compaction stores every live deal's reach beside its list entry
    std::vector<size_t> nan_off;             // per reach row of the round subtrees: float offset in plan.d_reach_nan (rows of listed parents hold one segment per cluster range)
    // Cluster-partitioned workgroups: when the LDS tiles of ALL traverser nodes of a round subtree do not fit together, the cluster axis is cut
    // into n_parts ranges of part_size clusters such that inside one range they do; every live deal is listed under the range of its traverser
    // cluster and each (root, range) becomes its own kernel job whose tiles cover that range only -- all resident, zeroed and flushed once.
    struct Parts {
        uint32_t first, second;   // number of cluster ranges, clusters per range
        uint32_t n_clusters, pitch;   // of the TRAVERSER's nodes in this round subtree (the root may be the opponent's)
    };
    bool seg_root(int root) const { return s->ordered && nodes[size_t(root)].round_idx == s->order_round; }   // its deltas are summed by wave segments: no LDS tiles
    Parts parts_of(int root) const {
        const rs_table *t = s->table;
        size_t sum_a = 0;
        uint32_t n_cl = 0, pitch = 0;
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int q = stack.back();
            stack.pop_back();
            const rs_tree_node &qn = nodes[q];
            if (qn.kind == RS_NODE_ACTION && qn.player == p && qn.n_children > 0) {
                sum_a += size_t(qn.n_children);
                n_cl = t->nodes[size_t(qn.index)].n_clusters;
                pitch = uint32_t(t->pitch[size_t(qn.index)]);
            }
            for (int k = 0; k < qn.n_children; ++k) {
                const int c = qn.children[k];
                if (nodes[c].kind != RS_NODE_PRIVATE_CHANCE && nodes[c].kind != RS_NODE_PUBLIC_CHANCE) stack.push_back(c);
            }
        }
        const size_t limit = size_t(lds_limit) / 4;
        if (!want_parts || sum_a == 0 || 2 * sum_a * pitch <= limit || seg_root(root)) return Parts{1u, pitch, n_cl, pitch};
        // Partitioning costs list indirection (gathers instead of row loads, a bucketing pass).  When most tiles would be resident anyway --
        // 1 081 clusters miss the budget by 1 % and keep 5 of 7 -- it loses (measured 1.33 against 0.84 ms per batch): only partition when
        // fewer than half of the tile bytes fit.
        if (limit * 2 >= 2 * sum_a * pitch) return Parts{1u, pitch, n_cl, pitch};
        const uint32_t r = uint32_t(limit / (2 * sum_a)) / 64u * 64u;
        if (r < 64u || (n_cl + r - 1) / r > 64u) return Parts{1u, pitch, n_cl, pitch};   // k_compact_live handles up to 64 ranges
        return Parts{(n_cl + r - 1) / r, r, n_cl, pitch};
    }

    Builder(rs_solver *s_, int p_) : s(s_), p(p_), plan(s_->plan[p_]), nodes(s_->tree.nodes) {}

    size_t alloc(int round, size_t segments = 1) {
        const size_t off = arena;
        arena += round_up(s->pitch[round] * segments * sizeof(float), 256);
        return off + 1;
    }
    float *aptr(size_t off1) const { return reinterpret_cast<float *>(s->d_arena + (off1 - 1)); }
    std::vector<float *> util_override;   // sharded: the utility rows of boundary children live in the exchange buffer
    std::vector<int> boundary_k;          // chance node id -> index among the boundary nodes, -1 otherwise
    float *uptr(int id) const { return util_override[id] ? util_override[id] : aptr(util_off[id]); }
    float *nan_ptr(int id) const { return plan.d_reach_nan + nan_off[size_t(nan_slot[size_t(id)])]; }   // round mode: a root's reach buffer
    // will `root` walk a live-deal list?  (every round root but the first does once lists are wanted; the first only when its tiles had to be partitioned)
    bool listed_root(int root) { return want_lists && (root != first_root || parts_of(first_root).first > 1); }
    // deals below a chance node: global count when its child round is the sharded one
    uint32_t fan_of(int chance_id) const {
        const int c = nodes[chance_id].children[0];
        if (s->sharded && lane_round[c] == s->params.shard_round) return s->params.shard_global_boards / s->n_boards[lane_round[chance_id]];
        return s->n_boards[lane_round[c]] / s->n_boards[lane_round[chance_id]];
    }
    bool boundary(int chance_id) const {
        return s->sharded && nodes[chance_id].kind == RS_NODE_PUBLIC_CHANCE && lane_round[nodes[chance_id].children[0]] == s->params.shard_round;
    }

    void annotate(int id, int d, int round) {
        depth[id] = d;
        max_depth = std::max(max_depth, d);
        const rs_tree_node &nd = nodes[id];
        if (nd.kind == RS_NODE_ACTION) round = nd.round_idx;
        lane_round[id] = round;
        bool own = nd.kind == RS_NODE_ACTION && nd.player == p && nd.n_children > 0;
        bool cl = nd.kind != RS_NODE_PUBLIC_CHANCE && nd.kind != RS_NODE_PRIVATE_CHANCE;
        for (int k = 0; k < nd.n_children; ++k) {
            const int c = nd.children[k];
            // the child of a public chance node lives on the next round's boards
            annotate(c, d + 1, nd.kind == RS_NODE_PUBLIC_CHANCE ? round + 1 : round);
            own = own || has_own[c];
            cl = cl && closed[c];
        }
        has_own[id] = own;
        closed[id] = cl;   // no chance node at or below: one lane geometry, fusable into a single kernel
    }

    // ---- fused subtrees: every topmost chance-free subtree becomes ONE tree-specialised kernel ---------------------
    void mark_inside(int id) {
        for (int k = 0; k < nodes[id].n_children; ++k) {
            inside[nodes[id].children[k]] = 1;
            mark_inside(nodes[id].children[k]);
        }
    }
    void fan_mode(int id) {   // may the kernel of root `id` take over work of the ENUM chance node above it?
        const int par = nodes[id].parent;
        if (par >= 0 && chance_enum(nodes[par]) && !boundary(par) && s->n_clusters % 4 == 0 && !s->deal_mode) {
            const int mode = s->knobs.fan != kUnset ? s->knobs.fan : 1;
            if (mode == 1 || (mode == 2 && closed[id])) fan_root[id] = char(mode);
            else if (mode == 2) fan_root[id] = 1;   // a round subtree with chance nodes below cannot walk its own deals (its rows are per deal of the round above): expand step only
        }
    }
    void mark_lane_round_inside(int root, int id) {
        for (int k = 0; k < nodes[id].n_children; ++k) {
            const int c = nodes[id].children[k];
            const rs_tree_node &cn = nodes[c];
            if (cn.kind == RS_NODE_ACTION) {
                inside[c] = 1;
                mark_lane_round_inside(root, c);
            } else if (chance_enum(cn)) {
                lr_bnd[size_t(root)].push_back(c);
                mark_fused(cn.children[0]);
            } else if (cn.kind != RS_NODE_TERMINAL) {
                mark_fused(c);   // cannot happen below the root (private chance only there), kept for safety
            }
        }
    }
    void mark_fused(int id) {
        const rs_tree_node &nd = nodes[id];
        if (lane_rounds && nd.kind == RS_NODE_ACTION && nd.n_children > 0 && !closed[id]) {
            fused_root[id] = 1;
            lr_root[id] = 1;
            fan_mode(id);
            mark_lane_round_inside(id, id);
            return;
        }
        // (pruned lane sweeps kept the level plan in round 1; since round 2 the generated lane kernels have the pruned forms the deal kernels always had)
        if (s->params.fuse_subtrees && nd.kind == RS_NODE_ACTION && nd.n_children > 0 && closed[id]) {
            fused_root[id] = 1;
            mark_inside(id);
            // Directly below an ENUM chance node (cfr.rs:502-522) the subtree's kernel can take over the chance node's work.  Needs whole vectors per board
            // (n_clusters % 4 == 0) and the node's deals in one contiguous [boards][C] block (not the chance node entering a SHARDED round, whose deals live
            // in other ranks' slots).  RS_JIT_FAN: 0 = never; 1 (default) = the kernel scales the chance node's own incoming reach itself (no expand launch,
            // no per-deal reach rows); 2 = it also walks the node's deals itself and sums them in order (no reduce launch, no per-deal rows at all: 7 GB
            // less workspace at config-3 size, but measured 11 % slower there, so only on request)
            fan_mode(id);
            return;
        }
        for (int k = 0; k < nd.n_children; ++k) mark_fused(nd.children[k]);
    }

    int resolve(int c) const {
        while (nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE) c = nodes[c].children[0];
        return c;
    }
    void mark_round_inside(int root, int id) {
        for (int k = 0; k < nodes[id].n_children; ++k) {
            int c = nodes[id].children[k];
            bool through_chance = false;
            while (nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE) {
                inside[c] = 1;   // nothing is launched for a pass-through chance node
                through_chance = true;
                c = nodes[c].children[0];
            }
            if (through_chance && nodes[c].kind == RS_NODE_ACTION && nodes[c].n_children > 0) {
                bnd[size_t(root)].push_back(c);
                mark_round(c);
            } else {
                inside[c] = 1;
                if (nodes[c].kind == RS_NODE_ACTION) mark_round_inside(root, c);
            }
        }
    }
    void mark_round(int root) {
        fused_root[root] = 1;
        mark_round_inside(root, root);
    }
    // `segments`: the utility row of a root below a LISTED parent is addressed by the parent's list position, one segment per cluster range of the parent (pos_rows)
    void layout_round(int root, size_t segments = 1) {
        util_off[root] = alloc(lane_round[root], segments);
        const size_t below = (pos_rows && listed_root(root)) ? size_t(parts_of(root).first) : size_t(1);
        for (int b : bnd[size_t(root)]) {
            nan_slot[size_t(b)] = n_nan++;
            layout_round(b, below);
        }
    }

    // an action node without valid actions (state.rs:125-157 can return none): worth 0, owns nothing, launches nothing
    bool dead_end(int id) const { return nodes[id].kind == RS_NODE_ACTION && nodes[id].n_children == 0; }

    bool chance_enum(const rs_tree_node &nd) const {
        return nd.kind == RS_NODE_PUBLIC_CHANCE && s->params.chance_mode == RS_CHANCE_ENUM;
    }

    // pass 1: decide which buffers exist (offsets only; the arena is allocated afterwards)
    void layout(int id) {
        const rs_tree_node &nd = nodes[id];
        const bool prune = (s->params.mode & RS_UPD_PRUNE) != 0;
        if (((nd.kind == RS_NODE_ACTION && nd.n_children > 0) && fan_root[id] != 2) || chance_enum(nd)) util_off[id] = alloc(lane_round[id]);   // a deal-walking root returns through its chance node's row
        if (lr_root[id]) {   // a round subtree: rows only for its chance nodes (utility up, reach down where a traverser node lies below), then the next round
            for (int ch : lr_bnd[size_t(id)]) {
                util_off[ch] = alloc(lane_round[ch]);
                if (has_own[ch]) reach_off[ch] = alloc(lane_round[ch]);
                layout(nodes[ch].children[0]);
            }
            return;
        }
        if (fused_root[id]) return;   // everything below lives in registers / LDS of k_subtree
        for (int k = 0; k < nd.n_children; ++k) {
            const int c = nd.children[k];
            if (nodes[c].kind != RS_NODE_TERMINAL && has_own[c]) {
                const bool opp = nd.kind == RS_NODE_ACTION && nd.player != p;
                const bool own_prune = nd.kind == RS_NODE_ACTION && nd.player == p && prune;
                if (opp || own_prune) reach_off[c] = alloc(lane_round[c]);
                // ENUM chance: a buffer only if the incoming reach is itself a buffer (decided in pass 2)
            }
            layout(c);
        }
    }

    ChildSrc child_source(int c) const {
        const rs_tree_node &cn = nodes[c];
        switch (cn.kind) {
        case RS_NODE_ACTION:
            if (cn.n_children == 0) return ChildSrc{nullptr, 0.0f, CH_CONST};   // util = 0f32 and empty loops, cfr.rs:571-589
            return ChildSrc{uptr(c), 0.0f, CH_BUF};
        case RS_NODE_PRIVATE_CHANCE: return child_source(cn.children[0]);
        case RS_NODE_PUBLIC_CHANCE:
            if (chance_enum(cn)) return ChildSrc{uptr(c), 0.0f, CH_BUF};
            return child_source(cn.children[0]);  // cfr.rs:306-309
        default: break;
        }
        const float pot = float(cn.value);  // `tn.value as f32`
        if (cn.ttype == RS_TERM_UNCONTESTED)  // cfr.rs:316-322
            return ChildSrc{nullptr, (p == cn.last_to_act) ? -1.0f * pot : 1.0f * pot, CH_CONST};
        const rs_leaf_desc &lf = s->leaves[p][c];
        if (lf.kind == RS_LEAF_UTIL) return ChildSrc{lf.d_buf, 0.0f, CH_BUF};
        return ChildSrc{lf.d_buf, pot, CH_SIGN | (p == 1 ? 0x100 : 0)};  // cfr.rs:323-347
    }

    void node_job(int id, NodeJob &job) const {
        const rs_tree_node &nd = nodes[id];
        const rs_table *t = s->table;
        std::memset(&job, 0, sizeof(job));
        job.regrets = t->regrets_ptr(nd.index);
        job.ssum = t->ssum_ptr(nd.index);
        job.pitch = uint32_t(t->pitch[nd.index]);
        job.row_stride = uint32_t(t->tile[size_t(nd.index)]);
        job.tile_shift = t->tile_shift(nd.index);
        job.n_vec = job.pitch / kVec;
        job.n_actions = nd.n_children;
        job.scale = s->params.scale;
        job.reach = reach[id].ptr;
        job.reach_const = reach[id].cst;
        job.node_index = uint32_t(nd.index);
        if (s->deal_mode) {
            job.n_vec = uint32_t(s->pitch[0] / kVec);
            job.cidx = s->deals.d_cluster[nd.round_idx][nd.player];
            job.dreg = (char *)t->d_dregrets + t->cell_off[nd.index] * 4;
            job.dssm = (char *)t->d_dssum + t->cell_off[nd.index] * 4;
            job.n_lanes = s->deals.n_deals;
            job.lane_base = s->params.deal_offset;
            if (s->params.mode & RS_UPD_PRUNE) job.prune_lane = s->deals.d_prune;
        }
    }

    double lanes(int id) const { return double(s->n_boards[lane_round[id]]) * s->n_clusters; }

    int build() {
        const size_t n = nodes.size();
        depth.assign(n, 0);
        lane_round.assign(n, 0);
        has_own.assign(n, 0);
        closed.assign(n, 0);
        fused_root.assign(n, 0);
        fan_root.assign(n, 0);
        lr_root.assign(n, 0);
        next_root.assign(n, 0);
        lr_bnd.assign(n, {});
        inside.assign(n, 0);
        reach.assign(n, ReachSrc{});
        util_override.assign(n, nullptr);
        boundary_k.assign(n, -1);
        util_off.assign(n, 0);
        reach_off.assign(n, 0);
        annotate(0, 0, 0);
        plan.n_boundary = 0;
        for (size_t id = 0; id < n; ++id)
            if (boundary(int(id))) boundary_k[id] = plan.n_boundary++;
        bnd.assign(n, {});
        nan_slot.assign(n, -1);
        {
            const Knobs &kn = s->knobs;
            const bool round_off = kn.no_rounds != 0;
            first_root = resolve(0);
            // Small deal batches leave most SIMDs without a wave, and a generated kernel is a long dependent instruction stream: one deal per
            // thread puts four times as many waves on the chip, each walking a quarter of the code (RS_JIT_LANES = 1 / 4 overrides)
            // Only the first round's subtree sees the whole batch; the subtrees behind chance nodes walk the live-deal lists of their roots, a small
            // share of it each (three streets, 1 M deals per batch: 6.46 -> 5.87 ms with one deal per thread everywhere; 128 K deals: 3.81 -> 2.45)
            // Round 2: kernels with LDS delta tiles take one deal per thread at every batch size -- their tiles hold the CU to ONE workgroup, which is 1 024 threads for
            // the one-deal forms and 512 for the four-deal forms (registers), and 16 lean waves per CU beat 8 fat ones (river game, 4 M deals per batch: 0.858 -> 0.768 ms,
            // 1 M: 0.295 -> 0.274, 256 K: 0.143 -> 0.123; gpurun_out/r02z/ab_l1.log).  Beyond 256 K deals per batch the kernel that walks the WHOLE batch (the first round's) takes TWO deals per thread
            // (still 1 024 threads, twice the gathers in flight per wave): river game 4 M deals 0.770 -> 0.738 ms, 1 M 0.269 -> 0.259; at 64 K it loses (0.078 -> 0.102), and so
            // do the list-walking kernels of later rounds at 1 M deals (three streets 3.70 -> 4.01 ms), which stay at one (gpurun_out/r03o/ab_l2.log, r03p/times.log).  The rule below is left for the kernels without tiles.
            jit_lanes = (s->deal_mode && s->deals.n_deals <= kSmallDealBatch) ? 1 : 4;
            jit_lanes_below = s->deal_mode ? 1 : 4;
            if (kn.lanes != kUnset) jit_lanes = jit_lanes_below = (kn.lanes == 1 || kn.lanes == 2) ? kn.lanes : 4;
            round_mode = s->deal_mode && s->params.fuse_subtrees && !round_off && nodes[first_root].kind == RS_NODE_ACTION &&
                         nodes[first_root].n_children > 0;
            const bool sparse_off = kn.no_sparse != 0, parts_off = kn.no_parts != 0;
            want_lists = s->deal_mode && s->params.opp_mode == RS_OPP_SAMPLE && !sparse_off;
            want_parts = round_mode && want_lists && !parts_off;
            // three streets, 4 M deals per batch: 13.49 -> 12.51 ms; 1 M: 5.98 -> 5.74; but 64 K: 2.14 -> 2.34 (latency-bound: the list adds a dependent load per entry),
            // so only batches beyond the small-batch switch (RS_JIT_SCAN_ALL = 1 / 0 forces either)
            scan_parent = round_mode && want_lists && s->deals.n_deals > kScanParentMin;   // with list-position rows it pays from 128 K deals per batch on (1.21 -> 1.07 ms; 64 K: 1.00 -> 1.02,
                                                                                           // lossless abstractions at 64 K 2.5 -> 2.9: gpurun_out/r04d/ab_scan_small.log)
            if (kn.scan_all != kUnset) scan_parent = round_mode && want_lists && kn.scan_all == 0;
            pos_rows = scan_parent && !kn.no_posrows;
            lds_limit = s->lds_limit;
            if (kn.lds_max != kUnset) lds_limit = std::min(lds_limit, kn.lds_max);
            lds_limit -= int(kWorklistLdsBytes);   // the work-list kernels keep their ticket in front of the tiles
        }
        if (round_mode) {
            mark_round(first_root);
            layout_round(first_root);
        } else {
            // lane sweeps: round subtrees above the last round (full-width cfr() with ENUM chance nodes; prune keeps the level plan's NaN bookkeeping)
            lane_rounds = !s->deal_mode && s->params.fuse_subtrees && s->params.chance_mode == RS_CHANCE_ENUM && s->params.opp_mode == RS_OPP_FULL &&
                          !s->knobs.no_lane_rounds;
            for (size_t id = 0; id < n; ++id)
                if (nodes[id].kind == RS_NODE_PUBLIC_CHANCE) {
                    const int c = nodes[id].children[0];
                    if (nodes[c].kind == RS_NODE_ACTION && nodes[c].n_children > 0) next_root[size_t(c)] = 1;
                }
            mark_fused(0);
            layout(0);
        }
        // ENUM chance children: need their own reach buffer when the chance node's reach is a buffer.
        // Resolve top-down in id order (parents have smaller ids than children).
        std::vector<char> reach_is_buf(n, 0);
        for (size_t id = 0; id < n; ++id)
            if (reach_off[id]) reach_is_buf[id] = 1;   // the chance nodes of lane round subtrees: their reach row is written by the subtree's reach-down kernel
        for (size_t id = 0; id < n; ++id) {
            const rs_tree_node &nd = nodes[id];
            if (fused_root[id] || inside[id]) continue;
            for (int k = 0; k < nd.n_children; ++k) {
                const int c = nd.children[k];
                if (nodes[c].kind == RS_NODE_TERMINAL || !has_own[c]) continue;
                if (reach_off[c]) reach_is_buf[c] = 1;
                else if (chance_enum(nd)) {
                    if (fan_root[c]) continue;   // its kernel reads the chance node's own incoming reach and scales it in registers
                    if (reach_is_buf[id]) {
                        reach_off[c] = alloc(lane_round[c]);
                        reach_is_buf[c] = 1;
                    }
                } else reach_is_buf[c] = reach_is_buf[id];
            }
        }
        plan.arena_bytes = arena;
        return RS_OK;
    }

    // one subtree job of a tree-specialised kernel: `down` = the top-down half of a round subtree (deal batches), else the walk that updates the table
    int add_jit_job(int id, bool down, const std::vector<int> &sparse_slot, std::map<hipFunction_t, int> &by_fn) {
        const size_t n = nodes.size();
        const rs_table *t = s->table;
        const double es = double(elem_size(t->dtype));
        if (has_own[id] && !reach[id].valid) return fail(RS_ERR_INVALID, "rs_solver_create: internal: no reach for a fused subtree");
        std::vector<int> leaf_buf(n, -1), leaf_flags(n, 0);
        for (size_t t2 = 0; t2 < n; ++t2) {
            const rs_tree_node &tn = nodes[t2];
            if (tn.kind != RS_NODE_TERMINAL || tn.ttype == RS_TERM_UNCONTESTED) continue;
            const rs_leaf_desc &lf = s->leaves[p][t2];
            auto it = leaf_ids.find(lf.d_buf);
            if (it == leaf_ids.end()) it = leaf_ids.emplace(lf.d_buf, int(leaf_ids.size())).first;
            leaf_buf[t2] = it->second;
            leaf_flags[t2] = lf.kind == RS_LEAF_UTIL ? 0 : 1;
        }
        // deal batches: privatise the deltas of a traverser node in LDS when [2][A][table pitch] ints fit in what the device gives
        // ONE workgroup (MI355X: 160 KiB, launchable without any attribute -- probed; a 5 000-cluster node needs 121 KiB)
        size_t lds_need = 0;
        if (s->deal_mode) {
            std::vector<int> stack{id};
            while (!stack.empty()) {
                const int q = stack.back();
                stack.pop_back();
                const rs_tree_node &qn = nodes[q];
                if (qn.kind == RS_NODE_ACTION && qn.player == p && qn.n_children > 0)
                    lds_need = std::max(lds_need, size_t(2) * qn.n_children * t->pitch[qn.index] * 4);
                for (int k = 0; k < qn.n_children; ++k) {
                    const int c = qn.children[k];
                    const bool chance = nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE;
                    if (!(round_mode && chance)) stack.push_back(c);   // a round subtree ends at the chance nodes
                }
            }
        }
        const bool lds_off = s->knobs.no_lds != 0;
        const bool sparse = sparse_slot[id] >= 0;
        const Parts parts = sparse ? parts_of(id) : Parts{1u, 0u, 0u, 0u};
        if (parts.first > 1) lds_need = lds_need / std::max<size_t>(1, parts.pitch) * parts.second;   // tiles cover one range
        const bool seg = seg_root(id) && !down && t->dtype == RS_I32;
        const bool use_lds = s->deal_mode && lds_need > 0 && lds_need <= size_t(lds_limit) && !lds_off && !down && !seg;
        JitSubtree js;
        jit_emit_subtree(nodes, id, p, has_own, leaf_buf, leaf_flags, t->dtype, s->params.mode & RS_UPD_ARITH_MASK,
                         s->params.opp_mode == RS_OPP_SAMPLE, s->deal_mode, use_lds, sparse, down, (s->params.mode & RS_UPD_PRUNE) != 0,
                         (use_lds && s->knobs.lanes == kUnset) ? ((!sparse && s->deals.n_deals > kSmallDealBatch) ? 2 : 1) : ((id == first_root || nodes[id].round_idx == nodes[first_root].round_idx) ? jit_lanes : jit_lanes_below),
                         round_mode ? &fused_root : (lane_rounds ? &next_root : nullptr), js, s->knobs,
                         int(fan_root[id]), sparse && s->d_attr[nodes[id].round_idx] != nullptr, pos_rows, !s->knobs.no_worklist, s->ordered, seg);
        const bool fan = fan_root[id] == 2, xfan = fan_root[id] == 1;
        const int fan_par = fan_root[id] ? nodes[id].parent : -1;   // the ENUM chance node whose work this kernel takes over
        hipFunction_t fn = nullptr;
        if (int rc = jit_get_kernel(js.source, js.entry, t->device, &fn, s->knobs.dump != 0)) return rc;
        auto bi = by_fn.find(fn);
        if (bi == by_fn.end()) {
            bi = by_fn.emplace(fn, int(plan.jit.size())).first;
            plan.jit.emplace_back();
            plan.jit.back().fn = fn;
            plan.jit.back().stride = js.args_size;
            plan.jit.back().threads = js.threads;
            plan.jit.back().worklist = js.worklist;
            plan.jit.back().seg = seg;
            plan.jit.back().off_count = uint32_t(js.off_count);
            plan.jit.back().deals_per_trip = uint32_t(js.threads * js.lanes);
        }
        JitLaunch &JL = plan.jit[bi->second];
        for (uint32_t part = 0; part < parts.first; ++part) {   // one job per cluster range (one in all unless the tiles had to be partitioned)
        const size_t base = JL.blob.size();
        JL.blob.resize(base + js.args_size, 0);
        unsigned char *a = JL.blob.data() + base;
        auto put_ptr = [&](size_t off, const void *ptr) { std::memcpy(a + off, &ptr, 8); };
        auto put_f32 = [&](size_t off, float f) { std::memcpy(a + off, &f, 4); };
        auto put_u32 = [&](size_t off, uint32_t u) { std::memcpy(a + off, &u, 4); };
        double bytes = 0.0;
        for (size_t k = 0; k < js.node_ids.size(); ++k) {
            const rs_tree_node &an = nodes[js.node_ids[k]];
            put_ptr(js.off_reg + 8 * k, t->regrets_ptr(an.index));
            put_ptr(js.off_ssm + 8 * k, t->ssum_ptr(an.index));
            put_u32(js.off_nidx + 4 * k, uint32_t(an.index));
            bytes += lanes(js.node_ids[k]) * an.n_children * es * (an.player == p ? 4.0 : 1.0);
        }
        for (size_t k = 0; k < js.leaf_terms.size(); ++k) put_ptr(js.off_leaf + 8 * k, s->leaves[p][js.leaf_terms[k]].d_buf);
        for (size_t k = 0; k < js.const_terms.size(); ++k) {
            const rs_tree_node &tn = nodes[js.const_terms[k]];
            const float pot = float(tn.value);   // `tn.value as f32`
            put_f32(js.off_cval + 4 * k, tn.ttype == RS_TERM_UNCONTESTED ? ((p == tn.last_to_act) ? -1.0f * pot : 1.0f * pot) : pot);
        }
        put_ptr(js.off_reach, reach[id].ptr);
        put_ptr(js.off_out, fan ? uptr(fan_par) : uptr(id));
        put_ptr(js.off_seed, s->d_seed());
        put_f32(js.off_reach_const, reach[id].cst);
        put_f32(js.off_scale, s->params.scale);
        // fan: one thread per 4 clusters of a PARENT board (n_clusters % 4 == 0: exactly lanes / 4 vectors, no padding lanes)
        const uint32_t n_vec = fan ? uint32_t(size_t(s->n_boards[lane_round[fan_par]]) * s->n_clusters / 4)
                               : (xfan ? uint32_t(size_t(s->n_boards[lane_round[id]]) * s->n_clusters / 4) : uint32_t(s->pitch[lane_round[id]] / size_t(js.lanes)));
        put_u32(js.off_n_vec, n_vec);
        if (!s->deal_mode) {
            const uint32_t f = fan_par >= 0 ? fan_of(fan_par) : 1u;
            put_u32(js.off_fan, f);
            put_f32(js.off_inv, 1.0f / float(f));   // the same f32 quotient k_chance_expand multiplies by
            put_u32(js.off_cvec, s->n_clusters / 4);
            for (size_t k = 0; k < js.boundary_roots.size(); ++k) {   // lane round subtrees: the rows of the ENUM chance node above every next-round root
                const int ch = nodes[size_t(js.boundary_roots[k])].parent;
                put_ptr(js.off_butil + 8 * k, uptr(ch));
                put_ptr(js.off_breach + 8 * k, reach_off[size_t(ch)] ? aptr(reach_off[size_t(ch)]) : nullptr);
                bytes += lanes(id) * 4.0;
            }
        }
        put_u32(js.off_pitch, uint32_t(s->pitch[lane_round[id]]));
        {   // every node of a fused subtree lives on one round: same lanes, same tiling
            const int n0 = nodes[size_t(js.node_ids.empty() ? id : js.node_ids[0])].index;
            put_u32(js.off_row_stride, uint32_t(t->tile[size_t(n0)]));
            put_u32(js.off_tile_shift, t->tile_shift(n0));
        }
        if (s->deal_mode) {
            const int r = nodes[id].round_idx;
            const uint32_t *cx[2] = {s->deals.d_cluster[r][0], s->deals.d_cluster[r][1]};
            uint32_t tp[2] = {0, 0};
            for (size_t k = 0; k < js.node_ids.size(); ++k) {
                const rs_tree_node &an = nodes[js.node_ids[k]];
                put_ptr(js.off_dreg + 8 * k, (char *)t->d_dregrets + t->cell_off[an.index] * 4);
                put_ptr(js.off_dssm + 8 * k, (char *)t->d_dssum + t->cell_off[an.index] * 4);
                put_ptr(js.off_shd + 8 * k, s->shadow_off_p[p][an.index] == SIZE_MAX ? nullptr : s->d_shadow + s->shadow_off_p[p][an.index]);
                put_u32(js.off_sstride + 4 * k, s->shadow_stride_p[p][an.index]);
                tp[an.player] = uint32_t(t->pitch[an.index]);
            }
            for (int q = 0; q < 2; ++q)   // a player without nodes in this subtree: any valid vector will do
                put_ptr(js.off_cidx + 8 * q, cx[q] ? cx[q] : cx[1 - q]);
            put_u32(js.off_tpitch, tp[0]);
            put_u32(js.off_tpitch + 4, tp[1]);
            put_u32(js.off_n_lanes, s->deals.n_deals);
            put_u32(js.off_n_lanes + 4, s->params.deal_offset);   // JArgs.lane_base
            for (size_t k = 0; k < js.boundary_roots.size(); ++k) {   // round subtrees: what the next round's roots return / are handed
                const int b = js.boundary_roots[k];
                put_ptr(js.off_butil + 8 * k, uptr(b) + ((pos_rows && sparse) ? size_t(part) * s->pitch[lane_round[id]] : size_t(0)));
                // position-indexed rows: this job's segment of the row starts where its list does
                put_ptr(js.off_breach + 8 * k, nan_ptr(b) + ((pos_rows && sparse && down) ? size_t(part) * s->pitch[lane_round[id]] : size_t(0)));
            }
            if (sparse) {   // the subtree walks only its live deals
                const CompactJob &cj = plan.compact_jobs[size_t(sparse_slot[id])];
                put_ptr(js.off_list, cj.list + size_t(part) * cj.list_stride);
                put_ptr(js.off_count, cj.count + size_t(part) * cj.count_stride);
                // position-indexed rows: the entries of every list but the first root's (all of whose deals are live, with the constant root reach) carry their reach
                put_ptr(js.off_rlist, (pos_rows && id != first_root) ? plan.d_rlists + (cj.list - plan.d_lists) + size_t(part) * cj.list_stride : nullptr);
                put_ptr(js.off_plist, (pos_rows && id != first_root) ? plan.d_plists + (cj.list - plan.d_lists) + size_t(part) * cj.list_stride : nullptr);
            }
            // LDS tile placement: as many traverser nodes as fit keep a RESIDENT tile (zeroed / flushed once per workgroup), the
            // rest share one transient area.  Smallest tiles first; the transient area must hold the largest tile left out.
            size_t lds_total = 0;
            for (size_t k = 0; k < js.node_ids.size(); ++k) put_u32(js.off_loff + 4 * k, 0xffffffffu);
            // the cluster range the tiles of this job cover: everything (rows as far apart as the table's), or one part
            const uint32_t own_pitch = tp[p] ? tp[p] : tp[1 - p];
            const uint32_t rp = parts.first > 1 ? parts.second : own_pitch;
            const uint32_t c0 = parts.first > 1 ? part * parts.second : 0u;
            const uint32_t n_cl = parts.n_clusters;
            put_u32(js.off_c0, c0);
            put_u32(js.off_rcount, parts.first > 1 ? std::min(parts.second, n_cl > c0 ? n_cl - c0 : 0u) : own_pitch);
            put_u32(js.off_rp, rp);
            put_ptr(js.off_prune, (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr);
            put_ptr(js.off_attr, (sparse || s->ordered) ? s->d_attr[nodes[id].round_idx] : nullptr);
            if (sparse && s->d_attr[nodes[id].round_idx]) s->attr_used |= 1u << nodes[id].round_idx;
            if (use_lds) {
                const bool resident_off = s->knobs.no_resident != 0;
                std::vector<std::pair<size_t, size_t>> tiles;   // (ints, k)
                for (size_t k = 0; k < js.node_ids.size(); ++k) {
                    const rs_tree_node &an = nodes[js.node_ids[k]];
                    if (an.player == p && an.n_children > 0) tiles.emplace_back(size_t(2) * an.n_children * rp, k);
                }
                std::sort(tiles.begin(), tiles.end());
                const size_t limit = size_t(lds_limit) / 4;
                size_t resident = 0, n_res = 0;
                while (!resident_off && n_res < tiles.size()) {
                    const size_t rest = n_res + 1 < tiles.size() ? tiles.back().first : 0;   // largest tile that would stay transient
                    if (resident + tiles[n_res].first + rest > limit) break;
                    resident += tiles[n_res].first;
                    ++n_res;
                }
                size_t at = 0;
                for (size_t i = 0; i < n_res; ++i) {
                    put_u32(js.off_loff + 4 * tiles[i].second, uint32_t(at));
                    at += tiles[i].first;
                }
                put_u32(js.off_resident, uint32_t(resident));
                put_u32(js.off_trans, uint32_t(resident));
                lds_total = (resident + (n_res < tiles.size() ? tiles.back().first : 0)) * 4;
                if (n_res) JL.persistent = true;
            }
            if (use_lds) JL.lds_bytes = std::max(JL.lds_bytes, lds_total);
        }
        JL.n_jobs += 1;
        JL.max_n_vec = std::max(JL.max_n_vec, n_vec);
        if (fan) JL.bytes += bytes + lanes(id) * 4.0 * js.leaf_terms.size() + lanes(fan_par) * ((reach[id].ptr ? 4.0 : 0.0) + 4.0);
        else if (xfan) JL.bytes += bytes + lanes(id) * (4.0 * js.leaf_terms.size() + 4.0) + lanes(fan_par) * (reach[id].ptr ? 4.0 : 0.0);
        else JL.bytes += (bytes + lanes(id) * (4.0 * js.leaf_terms.size() + (reach[id].ptr ? 4.0 : 0.0) + 4.0)) / parts.first;
        }   // parts
        return RS_OK;
    }

    // pass 2 (after the arena exists): emit jobs and launches
    int emit() {
        const size_t n = nodes.size();
        const rs_table *t = s->table;
        const bool prune = (s->params.mode & RS_UPD_PRUNE) != 0;
        const double es = double(elem_size(t->dtype));
        // The plan walks the tree in dependency order: by tree depth for the level plan -- or, when every action node lives in a generated subtree (lane round
        // subtrees), by ROUND: root (0), round-0 subtrees (1), the chance nodes below them (2), round-1 subtrees (3), ...  Subtrees of one round never depend on
        // each other whatever their depth, so same-shape subtrees of ALL depths share one launch and a round's chance nodes one expand / reduce launch.
        if (lane_rounds) {
            max_depth = 2 * s->n_rounds;
            for (size_t id = 0; id < n; ++id) {
                const rs_tree_node &nd = nodes[id];
                if (nd.kind == RS_NODE_PRIVATE_CHANCE) depth[id] = 0;
                else if (nd.kind == RS_NODE_PUBLIC_CHANCE) depth[id] = 2 * lane_round[id] + 2;
                else depth[id] = 2 * lane_round[id] + 1;   // action nodes and terminals of round r
            }
        }
        std::vector<std::vector<int>> by_depth(max_depth + 1);
        for (size_t id = 0; id < n; ++id) by_depth[depth[id]].push_back(int(id));
        if (s->sharded)
            for (size_t id = 0; id < n; ++id)
                if (boundary_k[id] >= 0)   // this rank's slot for the k-th boundary node
                    util_override[nodes[id].children[0]] =
                        s->d_exchange + (size_t(s->params.shard_rank) * plan.n_boundary + boundary_k[id]) * s->slot_lanes;

        if (s->ordered) {   // the batch sorted by this traverser's last-round cluster, per-deal inputs of all rounds as 32-byte records in that order (every sweep)
            Launch L;
            L.kind = L_ORDER;
            L.bytes = double(s->deals.n_deals) * (4.0 + 29.0 + 32.0);
            plan.launches.push_back(L);
        } else if (s->n_pack_jobs) {   // the batch's per-deal inputs, packed per round (the deals of a trainer change from batch to batch: every sweep)
            Launch L;
            L.kind = L_PACK;
            L.bytes = double(s->deals.n_deals) * s->n_pack_jobs * 29.0;
            plan.launches.push_back(L);
        }
        if (s->d_shadow) {   // the table as of sweep start, transposed for the deal kernels' gathers; sampled sweeps: the same launch advances the seed
            Launch L;
            L.kind = L_SHADOW;
            L.bytes = double(t->n_cells) * 16.0;
            L.n_jobs = s->params.opp_mode == RS_OPP_SAMPLE ? 1 : 0;   // 1 = with the seed
            plan.launches.push_back(L);
        } else if (s->params.opp_mode == RS_OPP_SAMPLE) {   // advance the sweep seed (part of the plan, hence of the hipGraph)
            Launch L;
            L.kind = L_SEED;
            plan.launches.push_back(L);
        }
        reach[0] = ReachSrc{nullptr, 1.0f, true};  // self.cfr(0, player, hand, 1f32, ..), cfr.rs:217
        const std::vector<int> sparse_slot_none(n, -1);
        // ---- top-down ------------------------------------------------------------------------------
        for (int d = 0; d <= max_depth; ++d) {
            std::map<int, std::vector<int>> reach_groups, prune_groups;  // by n_actions
            Launch LE;   // every ENUM chance node of this depth that has to expand a reach buffer
            LE.kind = L_EXPAND;
            LE.first_job = int(plan.chance_jobs.size());
            std::vector<int> down_roots;
            for (int id : by_depth[d]) {
                const rs_tree_node &nd = nodes[id];
                if (lr_root[id]) {   // lane round subtree: its reach-down kernel writes the reach row of every chance node below that has a traverser node under it
                    bool any = false;
                    for (int ch : lr_bnd[size_t(id)])
                        if (reach_off[size_t(ch)]) {
                            reach[size_t(ch)] = ReachSrc{aptr(reach_off[size_t(ch)]), 0.0f, true};
                            any = true;
                        }
                    if (any) down_roots.push_back(id);
                }
                if (nd.kind == RS_NODE_TERMINAL || fused_root[id] || inside[id] || dead_end(id)) continue;
                const bool opp = nd.kind == RS_NODE_ACTION && nd.player != p;
                const bool own = nd.kind == RS_NODE_ACTION && nd.player == p;
                bool any_child_buf = false;
                for (int k = 0; k < nd.n_children; ++k) {
                    const int c = nd.children[k];
                    if (nodes[c].kind == RS_NODE_TERMINAL || !has_own[c]) continue;
                    if (fan_root[c]) {
                        reach[c] = reach[id];   // the subtree's kernel multiplies by 1 / len itself (cfr.rs:510), once per deal
                    } else if (reach_off[c]) {
                        reach[c] = ReachSrc{aptr(reach_off[c]), 0.0f, true};
                        any_child_buf = true;
                    } else if (chance_enum(nd)) {
                        // constant incoming reach: fold cfr_reach * (1.0 / len) on the host (same f32 ops)
                        const uint32_t fan = fan_of(id);
                        reach[c] = ReachSrc{nullptr, reach[id].cst * (1.0f / float(fan)), true};
                    } else reach[c] = reach[id];  // own node (cfr.rs:580) or pass-through chance
                }
                if (!any_child_buf) continue;
                if (opp) reach_groups[nd.n_children].push_back(id);
                else if (own && prune) prune_groups[nd.n_children].push_back(id);
                else if (chance_enum(nd)) {
                    const int c = nd.children[0];
                    const uint32_t fan = fan_of(id);
                    ChanceJob cj{};
                    cj.src = reach[id].ptr;
                    cj.dst = aptr(reach_off[c]);
                    cj.src_const = reach[id].cst;
                    cj.inv = 1.0f / float(fan);
                    cj.fan = fan;
                    cj.n_clusters = s->n_clusters;
                    cj.n_parent_lanes = uint32_t(s->n_boards[lane_round[id]] * s->n_clusters);
                    cj.n_child_lanes = uint32_t(s->n_boards[lane_round[c]] * s->n_clusters);   // local boards when sharded
                    cj.board_off = boundary(id) ? s->shard_lo[s->params.shard_rank] : 0;
                    plan.chance_jobs.push_back(cj);
                    LE.max_lanes = std::max(LE.max_lanes, size_t(lanes(c)));
                    LE.bytes += lanes(c) * 4.0 + lanes(id) * 4.0;
                }
            }
            LE.n_jobs = int(plan.chance_jobs.size()) - LE.first_job;
            if (LE.n_jobs) plan.launches.push_back(LE);
            if (!down_roots.empty()) {
                std::map<hipFunction_t, int> by_fn;
                for (int id : down_roots)
                    if (int rc = add_jit_job(id, true, sparse_slot_none, by_fn)) return rc;
                const int group = by_fn.size() > 1 ? ++next_group : 0;
                for (auto &kv : by_fn) {
                    Launch L;
                    L.kind = L_TREE;
                    L.group = group;
                    L.first_job = kv.second;
                    L.bytes = plan.jit[kv.second].bytes;
                    plan.launches.push_back(L);
                }
            }
            for (int which = 0; which < 2; ++which) {
                for (auto &g : (which == 0 ? reach_groups : prune_groups)) {
                    Launch L;
                    L.kind = which == 0 ? L_REACH : L_PRUNE_REACH;
                    L.n_actions = g.first;
                    L.first_job = int(plan.jobs.size());
                    for (int id : g.second) {
                        NodeJob job;
                        node_job(id, job);
                        const rs_tree_node &nd = nodes[id];
                        int n_out = 0;
                        for (int k = 0; k < nd.n_children; ++k) {
                            const int c = nd.children[k];
                            if (nodes[c].kind != RS_NODE_TERMINAL && has_own[c] && reach_off[c]) {
                                job.out_reach[k] = aptr(reach_off[c]);
                                ++n_out;
                            }
                        }
                        L.max_n_vec = std::max(L.max_n_vec, job.n_vec);
                        L.bytes += lanes(id) * (nd.n_children * es + (job.reach ? 4.0 : 0.0) + 4.0 * n_out);
                        plan.jobs.push_back(job);
                    }
                    L.n_jobs = int(plan.jobs.size()) - L.first_job;
                    plan.launches.push_back(L);
                }
            }
        }
        // ---- sparse deal sweeps: which subtree roots get a compacted list of their live deals ---------------------------------
        // mccfr() follows ONE opponent action per node (cfr.rs:467-476): below a sampled node most deals are off their path (NaN reach).  A
        // subtree kernel that walks every deal would compute nothing for them; instead the live ones are compacted and only they are walked.
        std::vector<int> sparse_slot(n, -1);
        // the compact jobs of `ids` (in that order); reach_of(id) = the buffer whose non-NaN lanes are the live deals
        auto make_lists = [&](const std::vector<int> &ids, auto reach_of) -> int {
            const size_t n_sparse = ids.size();
            if (!n_sparse) return RS_OK;
            size_t list_elems = 0, n_counts = 0;
            std::vector<uint32_t> n_parts(n_sparse, 1), part_size(n_sparse, 0);
            for (size_t k = 0; k < n_sparse; ++k) {
                const Parts pr = parts_of(ids[k]);
                n_parts[k] = pr.first;
                part_size[k] = pr.second;
                list_elems += size_t(pr.first) * s->pitch[lane_round[ids[k]]];
                n_counts += pr.first;
            }
            hipError_t ea = hipMalloc((void **)&plan.d_lists, list_elems * sizeof(uint32_t));
            if (ea == hipSuccess && pos_rows) ea = hipMalloc((void **)&plan.d_rlists, list_elems * sizeof(float));
            if (ea == hipSuccess && pos_rows) ea = hipMalloc((void **)&plan.d_plists, list_elems * sizeof(uint32_t));
            plan.aux_bytes += list_elems * sizeof(uint32_t) * (1 + (pos_rows ? 2 : 0));
            plan.n_count_words = n_counts * kCountStride;
            if (ea == hipSuccess) ea = hipMalloc((void **)&plan.d_counts, n_counts * kCountStride * sizeof(uint32_t));
            if (ea == hipSuccess) ea = hipMemsetAsync(plan.d_counts, 0, n_counts * kCountStride * sizeof(uint32_t), t->stream);
            if (ea == hipSuccess) ea = hipMalloc((void **)&plan.d_compact_jobs, n_sparse * sizeof(CompactJob));
            if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: live-deal lists");
            plan.compact_jobs.resize(n_sparse);
            plan.count_off.assign(n_sparse + 1, 0);
            size_t at = 0, cat = 0;
            for (size_t k = 0; k < n_sparse; ++k) {
                const int id = ids[k];
                sparse_slot[id] = int(k);
                CompactJob &cj = plan.compact_jobs[k];
                cj = CompactJob{};
                cj.reach = reach_of(id);
                cj.list = plan.d_lists + at;
                cj.count = plan.d_counts + cat * kCountStride;   // one cache line each: atomics on neighbours would serialise
                cj.n_lanes = s->deals.n_deals;
                cj.n_parts = n_parts[k];
                cj.part_size = std::max<uint32_t>(1, part_size[k]);
                cj.list_stride = uint32_t(s->pitch[lane_round[id]]);
                cj.count_stride = uint32_t(kCountStride);
                cj.key = n_parts[k] > 1 ? s->deals.d_cluster[nodes[id].round_idx][p] : nullptr;   // the traverser's cluster on this round
                cj.key_stride = 1;
                if (cj.key && s->ordered) {   // list entries are ranks: the key sits in the rank's record
                    cj.key = static_cast<const uint32_t *>(s->d_arec) + 2 * nodes[id].round_idx + p;
                    cj.key_stride = 8;
                }
                plan.count_off[k] = cat;
                at += size_t(n_parts[k]) * s->pitch[lane_round[id]];
                cat += n_parts[k];
                plan.compact_max_lanes = std::max(plan.compact_max_lanes, cj.n_lanes);
            }
            plan.count_off[n_sparse] = cat;
            // a deal can only be live in a round subtree if its PARENT subtree walked it: scan the parent's live lists rather than the whole batch (the parent's reach-down
            // kernel writes every boundary row for every deal it walks -- reach or NaN -- so nothing stale is ever read and the rows need no NaN fill per sweep)
            if (scan_parent)
                for (size_t k = 0; k < n_sparse; ++k) {
                    const int par = parent_root_.empty() ? -1 : parent_root_[size_t(ids[k])];
                    if (par < 0 || sparse_slot[size_t(par)] < 0) continue;
                    const CompactJob &pj = plan.compact_jobs[size_t(sparse_slot[size_t(par)])];
                    CompactJob &cj = plan.compact_jobs[k];
                    cj.src_list = pj.list;
                    cj.src_count = pj.count;
                    cj.src_parts = pj.n_parts;
                    cj.src_list_stride = pj.list_stride;
                    cj.src_count_stride = pj.count_stride;
                    cj.pos_rows = pos_rows ? 1u : 0u;
                }
            if (pos_rows)   // every list but the first root's carries the reach of its entries (the first root's deals are all live, with the constant root reach)
                for (size_t k = 0; k < n_sparse; ++k)
                    if (ids[k] != first_root) {
                        plan.compact_jobs[k].rlist = plan.d_rlists + (plan.compact_jobs[k].list - plan.d_lists);
                        plan.compact_jobs[k].plist = plan.d_plists + (plan.compact_jobs[k].list - plan.d_lists);
                    }
            ea = hipMemcpy(plan.d_compact_jobs, plan.compact_jobs.data(), n_sparse * sizeof(CompactJob), hipMemcpyHostToDevice);
            if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: live-deal lists");
            return RS_OK;
        };
        auto push_compact = [&](int first, int count) {
            if (count <= 0) return;
            Launch L;
            L.kind = L_COMPACT;
            L.first_job = first;
            L.n_jobs = count;
            L.bytes = 8.0 * double(s->deals.n_deals) * count;
            plan.launches.push_back(L);
        };
        if (round_mode) {
            // ---- round subtrees, top-down: NaN-fill every root's reach buffer, then round by round compact the live deals of the round's
            // roots and let their DOWN kernels hand reach to the next round's roots
            roots_of_round.assign(1, std::vector<int>{first_root});
            for (size_t r = 0; r < roots_of_round.size(); ++r)
                for (int root : roots_of_round[r])
                    for (int b : bnd[size_t(root)]) {
                        if (roots_of_round.size() <= r + 1) roots_of_round.emplace_back();
                        roots_of_round[r + 1].push_back(b);
                    }
            std::vector<int> &parent_root = parent_root_;
            parent_root.assign(n, -1);
            for (size_t r = 0; r < roots_of_round.size(); ++r)
                for (int root : roots_of_round[r])
                    for (int b : bnd[size_t(root)]) parent_root[size_t(b)] = root;
            nan_off.assign(size_t(n_nan) + 1, 0);
            for (size_t b = 0; b < n; ++b)
                if (nan_slot[b] >= 0) {   // rows of a listed parent hold one list-position segment per cluster range of the parent
                    const int par = parent_root[b];
                    nan_off[size_t(nan_slot[b]) + 1] = s->pitch[0] * ((pos_rows && par >= 0 && listed_root(par)) ? size_t(parts_of(par).first) : size_t(1));
                }
            for (size_t k = 0; k < size_t(n_nan); ++k) nan_off[k + 1] += nan_off[k];
            if (n_nan) {
                plan.reach_nan_bytes = nan_off[size_t(n_nan)] * sizeof(float);
                plan.aux_bytes += plan.reach_nan_bytes;
                hipError_t en = hipMalloc((void **)&plan.d_reach_nan, plan.reach_nan_bytes);
                if (en == hipSuccess) en = hipMemsetAsync(plan.d_reach_nan, 0xff, plan.reach_nan_bytes, t->stream);
                if (en != hipSuccess) return hip_fail(en, "rs_solver_create: reach buffers of the round subtrees");
                // dense sweeps read every lane of a root's row: lanes nobody handed a reach to must hold NaN.  List sweeps only ever read what the parent's reach-down
                // kernel wrote in THIS sweep (the compaction scans the parent's lists), so they need no fill -- unless the old whole-batch scan is asked for
                if (!scan_parent) {
                    Launch L;
                    L.kind = L_NANFILL;
                    plan.launches.push_back(L);
                }
            }
            std::vector<int> listed;
            std::vector<std::pair<int, int>> slots_of_round(roots_of_round.size(), {0, 0});   // (first compact job, count)
            if (want_lists && parts_of(first_root).first > 1) {   // its tiles had to be partitioned: every deal is live, listed by cluster range
                slots_of_round[0] = {0, 1};
                listed.push_back(first_root);
            }
            if (want_lists)
                for (size_t r = 1; r < roots_of_round.size(); ++r) {
                    slots_of_round[r] = {int(listed.size()), int(roots_of_round[r].size())};
                    listed.insert(listed.end(), roots_of_round[r].begin(), roots_of_round[r].end());
                }
            if (int rc = make_lists(listed, [&](int id) { return id == first_root ? (const float *)nullptr : (const float *)nan_ptr(id); })) return rc;
            for (size_t r = 0; r < roots_of_round.size(); ++r) {
                push_compact(slots_of_round[r].first, slots_of_round[r].second);
                std::map<hipFunction_t, int> by_fn;
                for (int root : roots_of_round[r]) {
                    if (bnd[size_t(root)].empty()) continue;
                    for (int b : bnd[size_t(root)]) reach[b] = ReachSrc{nan_ptr(b), 0.0f, true};
                    if (int rc = add_jit_job(root, true, sparse_slot, by_fn)) return rc;
                }
                const int group = ++next_group;   // the DOWN kernels of one round write different reach buffers
                for (auto &kv : by_fn) {
                    Launch L;
                    L.kind = L_TREE;
                    L.group = group;
                    L.first_job = kv.second;
                    L.bytes = plan.jit[kv.second].bytes;
                    plan.launches.push_back(L);
                }
            }
        } else if (want_lists) {
            std::vector<int> listed;
            for (size_t id = 0; id < n; ++id)
                if (fused_root[id] && !inside[id] && !dead_end(int(id)) && reach[id].ptr) listed.push_back(int(id));
            if (int rc = make_lists(listed, [&](int id) { return reach[id].ptr; })) return rc;
            push_compact(0, int(listed.size()));
        }
        // ---- bottom-up -----------------------------------------------------------------------------
        // sharded sweeps: pass 0 = everything inside the sharded rounds (phase 0, before the exchange), pass 1 = the
        // replicated rounds including the boundary reduces (phase 1); unsharded: one pass
        for (int pass = 0; pass < (s->sharded ? 2 : 1); ++pass) {
        if (pass == 1) plan.split = plan.launches.size();
        auto in_pass = [&](int id) { return !s->sharded || (lane_round[id] >= s->params.shard_round) == (pass == 0); };
        for (int d = max_depth; d >= 0; --d) {
            std::map<int, std::vector<int>> upd_groups, util_groups;
            std::vector<int> sub_roots;
            Launch LR;   // every ENUM chance node of this depth
            LR.kind = L_REDUCE;
            LR.first_job = int(plan.chance_jobs.size());
            for (int id : by_depth[d]) {
                const rs_tree_node &nd = nodes[id];
                if (inside[id] || dead_end(id) || !in_pass(id)) continue;
                if (fused_root[id]) {
                    if (!round_mode) sub_roots.push_back(id);   // round subtrees walk back up round by round, below
                    continue;
                }
                if (nd.kind == RS_NODE_ACTION) (nd.player == p ? upd_groups : util_groups)[nd.n_children].push_back(id);
                else if (chance_enum(nd)) {
                    const int c = nd.children[0];
                    if (fan_root[c] == 2) continue;   // the child's kernel sums its deals in order and writes this node's row itself (cfr.rs:519)
                    const ChildSrc src = child_source(c);
                    if (src.kind != CH_BUF) return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: chance node above a terminal");
                    ChanceJob cj{};
                    cj.src = src.buf;
                    cj.dst = aptr(util_off[id]);
                    cj.fan = fan_of(id);
                    cj.n_clusters = s->n_clusters;
                    cj.n_parent_lanes = uint32_t(s->n_boards[lane_round[id]] * s->n_clusters);
                    if (boundary(id)) {   // the deals live in the exchange buffer, one slot per rank
                        cj.src = s->d_exchange + size_t(boundary_k[id]) * s->slot_lanes;
                        cj.shard_world = uint32_t(s->params.shard_world);
                        cj.rank_stride = uint32_t(size_t(plan.n_boundary) * s->slot_lanes);
                        std::memcpy(cj.shard_lo, s->shard_lo, sizeof(cj.shard_lo));
                    }
                    plan.chance_jobs.push_back(cj);
                    LR.max_lanes = std::max(LR.max_lanes, size_t(lanes(id)));
                    LR.bytes += lanes(c) * 4.0 + lanes(id) * 4.0;
                }
            }
            LR.n_jobs = int(plan.chance_jobs.size()) - LR.first_job;
            if (LR.n_jobs) plan.launches.push_back(LR);
            if (!sub_roots.empty()) {
                // tree-specialised kernels: subtrees of one shape share a kernel and a launch (blockIdx.y = subtree)
                std::map<hipFunction_t, int> by_fn;
                for (int id : sub_roots)
                    if (int rc = add_jit_job(id, false, sparse_slot, by_fn)) return rc;
                const int group = by_fn.size() > 1 ? ++next_group : 0;   // the subtrees of one depth touch disjoint nodes and buffers: their launches may overlap
                for (auto &kv : by_fn) {
                    Launch L;
                    L.kind = L_TREE;
                    L.group = group;
                    L.first_job = kv.second;
                    L.bytes = plan.jit[kv.second].bytes;
                    plan.launches.push_back(L);
                }
            }
            for (int which = 0; which < 2; ++which) {
                for (auto &g : (which == 0 ? upd_groups : util_groups)) {
                    Launch L;
                    L.kind = which == 0 ? L_UPDATE : L_NODE_UTIL;
                    L.n_actions = g.first;
                    L.first_job = int(plan.jobs.size());
                    for (int id : g.second) {
                        NodeJob job;
                        node_job(id, job);
                        const rs_tree_node &nd = nodes[id];
                        int n_buf = 0;
                        for (int k = 0; k < nd.n_children; ++k) {
                            job.child[k] = child_source(nd.children[k]);
                            if ((job.child[k].kind & 0xff) != CH_CONST) ++n_buf;
                        }
                        job.out_util = uptr(id);
                        if (which == 0 && !reach[id].valid) return fail(RS_ERR_INVALID, "rs_solver_create: internal: no reach for a traverser node");
                        L.max_n_vec = std::max(L.max_n_vec, job.n_vec);
                        if (which == 0) L.bytes += lanes(id) * (nd.n_children * 4.0 * es + 4.0 * n_buf + (job.reach ? 4.0 : 0.0) + 4.0);
                        else L.bytes += lanes(id) * (nd.n_children * es + 4.0 * n_buf + 4.0);
                        plan.jobs.push_back(job);
                    }
                    L.n_jobs = int(plan.jobs.size()) - L.first_job;
                    plan.launches.push_back(L);
                }
            }
        }
        }   // pass
        if (round_mode)   // ---- round subtrees, bottom-up: last round first; the subtrees of one round are independent of each other
            for (size_t r = roots_of_round.size(); r-- > 0;) {
                std::map<hipFunction_t, int> by_fn;
                for (int root : roots_of_round[r])
                    if (int rc = add_jit_job(root, false, sparse_slot, by_fn)) return rc;
                const int group = ++next_group;
                for (auto &kv : by_fn) {
                    Launch L;
                    L.kind = L_TREE;
                    L.group = group;
                    L.first_job = kv.second;
                    L.bytes = plan.jit[kv.second].bytes;
                    plan.launches.push_back(L);
                }
            }
        if (!s->sharded) plan.split = plan.launches.size();   // deal batches: phase 0 = the sweep, phase 1 = the apply below
        if (s->deal_mode) {   // table += delta, delta = 0: over the traverser's own nodes (nobody else's deltas were written: the other half of the delta arrays stays unread)
            std::vector<ApplyJob> aj;
            double cells = 0.0;
            for (size_t i = 0; i < t->nodes.size(); ++i) {
                const rs_node_desc &d = t->nodes[i];
                if (d.n_actions == 0 || d.player != p) continue;
                const size_t nc = size_t(d.n_actions) * t->pitch[i];
                if ((t->cell_off[i] % kVec) || (nc % kVec)) { aj.clear(); break; }   // never with 64-lane padded pitches; the whole-table form is the fallback
                aj.push_back(ApplyJob{t->cell_off[i] / kVec, nc / kVec});
                plan.apply_max_vec = std::max(plan.apply_max_vec, nc / kVec);
                cells += double(nc);
            }
            if (!aj.empty() && !s->knobs.apply_whole_table) {
                hipError_t ea = hipMalloc((void **)&plan.d_apply_jobs, aj.size() * sizeof(ApplyJob));
                if (ea == hipSuccess) ea = hipMemcpy(plan.d_apply_jobs, aj.data(), aj.size() * sizeof(ApplyJob), hipMemcpyHostToDevice);
                if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: apply jobs");
                plan.n_apply_jobs = int(aj.size());
            }
            Launch L;
            L.kind = L_APPLY;
            L.bytes = (plan.n_apply_jobs ? cells : double(t->n_cells)) * 32.0;
            plan.launches.push_back(L);
        }
        // value returned at node 0
        const ChildSrc root = child_source(0);
        if (root.kind != CH_BUF) return fail(RS_ERR_INVALID, "rs_solver_create: the root has no action node below it");
        plan.root_util = root.buf;
        plan.root_lanes = s->pitch[0];
        return RS_OK;
    }
};

WorklistDesc worklist_desc(const JitLaunch &JL) { return WorklistDesc{JL.d_blob, JL.d_wl, uint32_t(JL.stride), JL.off_count, uint32_t(JL.n_jobs), JL.deals_per_trip}; }

// wl_done: the work list of this launch was built by its group's shared k_worklist launch (run_range)
int run_launch(rs_solver *s, const Plan &plan, const Launch &L, hipStream_t tree_stream = nullptr, bool wl_done = false) {
    rs_table *t = s->table;
    if (!tree_stream) tree_stream = t->stream;
    static const int prof_kind[] = {RS_K_REACH, RS_K_REACH, RS_K_CHANCE, RS_K_UPDATE, RS_K_NODE_UTIL, RS_K_CHANCE, RS_K_TREE, RS_K_CHANCE, RS_K_DISCOUNT};
    if (L.kind == L_APPLY) {
        prof_begin(t, RS_K_DISCOUNT, L.bytes);
        hipError_t ea = plan.n_apply_jobs ? launch_apply_delta_jobs(t->d_regrets, t->d_dregrets, t->d_ssum, t->d_dssum, plan.d_apply_jobs, plan.n_apply_jobs, plan.apply_max_vec, t->stream)
                                          : launch_apply_delta(t->d_regrets, t->d_dregrets, t->d_ssum, t->d_dssum, t->n_cells, t->stream);
        prof_end(t);
        RS_HIP(ea, "k_apply_delta");
        return RS_OK;
    }
    if (L.kind == L_SEED) {
        RS_HIP(launch_next_seed(s->d_seed_state, t->stream), "k_next_seed");
        return RS_OK;
    }
    if (L.kind == L_NANFILL) {   // 0xffffffff is a NaN: lanes nobody hands a reach to stay off every path
        RS_HIP(hipMemsetAsync(plan.d_reach_nan, 0xff, plan.reach_nan_bytes, t->stream), "reach NaN fill");
        return RS_OK;
    }
    if (L.kind == L_COMPACT) {   // jobs [first_job, first_job + n_jobs)
        prof_begin(t, RS_K_REACH, L.bytes);
        const size_t c_lo = plan.count_off[size_t(L.first_job)], c_hi = plan.count_off[size_t(L.first_job + L.n_jobs)];
        hipError_t ec = hipMemsetAsync(plan.d_counts + c_lo * kCountStride, 0, (c_hi - c_lo) * kCountStride * sizeof(uint32_t), t->stream);
        if (ec == hipSuccess) ec = launch_compact_live(plan.d_compact_jobs + L.first_job, L.n_jobs, plan.compact_max_lanes, t->stream);
        prof_end(t);
        RS_HIP(ec, "k_compact_live");
        return RS_OK;
    }
    if (L.kind == L_ORDER) {
        prof_begin(t, RS_K_REACH, L.bytes);
        hipError_t eo = launch_order(s->order_job[&plan == &s->plan[1] ? 1 : 0], t->stream);
        prof_end(t);
        RS_HIP(eo, "k_order");
        return RS_OK;
    }
    if (L.kind == L_PACK) {
        RS_HIP(launch_pack_attr(s->d_pack_jobs, s->n_pack_jobs, s->deals.n_deals, t->stream), "k_pack_attr");
        return RS_OK;
    }
    if (L.kind == L_SHADOW) {
        prof_begin(t, RS_K_STRATEGY, L.bytes);
        const int which = &plan == &s->plan[1] ? 1 : 0;   // the shadow holds what THIS traverser's sweep reads: regrets everywhere, strategy sums at its own nodes
        hipError_t es = launch_build_shadow(s->d_shadow_jobs + size_t(which) * s->n_shadow_jobs, s->n_shadow_jobs, s->shadow_max_clusters, t->stream, L.n_jobs ? s->d_seed_state : nullptr);
        prof_end(t);
        RS_HIP(es, "k_build_shadow");
        return RS_OK;
    }
    prof_begin(t, prof_kind[L.kind], L.bytes);
    hipError_t e = hipSuccess;
    const NodeJob *jobs = plan.d_jobs + L.first_job;
    const KernelCfg cfg{t->dtype, s->params.mode};
    switch (L.kind) {
    case L_REACH: e = launch_reach(jobs, L.n_jobs, L.max_n_vec, L.n_actions, cfg, s->d_seed(), t->stream); break;
    case L_PRUNE_REACH: e = launch_prune_reach(jobs, L.n_jobs, L.max_n_vec, L.n_actions, cfg, t->stream); break;
    case L_EXPAND:
        e = launch_chance_expand(plan.d_chance_jobs + L.first_job, L.n_jobs, L.max_lanes, s->n_clusters % 4 == 0, t->stream);
        break;
    case L_UPDATE: e = launch_update(jobs, L.n_jobs, L.max_n_vec, L.n_actions, cfg, t->stream); break;
    case L_NODE_UTIL: e = launch_node_util(jobs, L.n_jobs, L.max_n_vec, L.n_actions, cfg, s->d_seed(), t->stream); break;
    case L_REDUCE:
        e = launch_chance_reduce(plan.d_chance_jobs + L.first_job, L.n_jobs, L.max_lanes, s->n_clusters % 4 == 0, t->stream);
        break;
    case L_TREE: {
        const JitLaunch &JL = plan.jit[L.first_job];
        size_t blocks = (size_t(JL.max_n_vec) + JL.threads - 1) / JL.threads;   // interleaved A/B: block size and grid cap are irrelevant for the lane kernels
        blocks = std::min<size_t>(std::max<size_t>(blocks, 1), JL.lds_bytes ? 256 * 4 : 256 * 16);   // LDS form: fewer, longer-lived workgroups
        if (JL.persistent) blocks = std::min<size_t>(blocks, size_t(s->n_cus));   // 224 VGPRs: one workgroup per CU is all that fits; more would only flush more
        if (JL.seg && JL.n_jobs > 1) blocks = std::min<size_t>(blocks, std::max<size_t>(64, size_t(s->n_cus) * 32 / size_t(JL.n_jobs) * 4));   // list walkers: a list holds a share of the batch
        if (s->knobs.max_blocks != kUnset)   // tests: force several trips per workgroup
            blocks = std::max<size_t>(1, std::min<size_t>(blocks, size_t(std::max(1, s->knobs.max_blocks))));
        const void *d_blob = JL.d_blob;
        int flags = s->params.mode & ~RS_UPD_ARITH_MASK;
        if (JL.worklist) {   // trips per job from the live-list counts, then resident workgroups that pull them
            if (!wl_done) {
                WorklistBatch one{};
                one.d[0] = worklist_desc(JL);
                e = launch_worklist(one, 1, tree_stream);
                if (e != hipSuccess) break;
            }
            size_t grid = size_t(s->n_cus) * ((JL.lds_bytes + kWorklistLdsBytes) * 2 <= size_t(s->lds_limit) ? 2 : 1);
            if (s->knobs.max_blocks != kUnset) grid = std::max<size_t>(1, std::min<size_t>(grid, size_t(std::max(1, s->knobs.max_blocks))));
            uint32_t *d_wl = JL.d_wl;
            void *wparams[] = {&d_blob, &flags, &d_wl};
            e = hipModuleLaunchKernel(JL.fn, (unsigned)grid, 1, 1, (unsigned)JL.threads, 1, 1, (unsigned)(JL.lds_bytes + kWorklistLdsBytes), tree_stream, wparams, nullptr);
            break;
        }
        void *params[] = {&d_blob, &flags};
        e = hipModuleLaunchKernel(JL.fn, (unsigned)blocks, (unsigned)JL.n_jobs, 1, (unsigned)JL.threads, 1, 1, (unsigned)JL.lds_bytes, tree_stream, params, nullptr);
        break;
    }
    }
    prof_end(t);
    RS_HIP(e, "plan launch");
    return RS_OK;
}

// launches [lo, hi) in order; consecutive launches of one group (independent round subtrees) are spread over the auxiliary streams:
// the main stream records a fork event, every used stream waits for it, and the main stream waits for all of them afterwards.  The same
// calls inside a stream capture become the forks and joins of the graph.
int run_range(rs_solver *s, Plan &plan, size_t lo, size_t hi) {
    rs_table *t = s->table;
    for (size_t i = lo; i < hi;) {
        size_t j = i + 1;
        if (plan.launches[i].group > 0)
            while (j < hi && plan.launches[j].group == plan.launches[i].group) ++j;
        if (j - i < 2 || !s->ev_fork || t->prof.on) {
            for (; i < j; ++i)
                if (int rc = run_launch(s, plan, plan.launches[i])) return rc;
            continue;
        }
        {   // the work lists of the group's launches: one k_worklist launch (a workgroup per list) on the main stream, before the fork
            WorklistBatch batch{};
            int nb = 0;
            for (size_t k = i; k < j; ++k) {
                if (plan.launches[k].kind != L_TREE || !plan.jit[size_t(plan.launches[k].first_job)].worklist) continue;
                batch.d[nb++] = worklist_desc(plan.jit[size_t(plan.launches[k].first_job)]);
                if (nb == kWorklistBatch) {
                    RS_HIP(launch_worklist(batch, nb, t->stream), "k_worklist");
                    nb = 0;
                }
            }
            RS_HIP(launch_worklist(batch, nb, t->stream), "k_worklist");
        }
        RS_HIP(hipEventRecord(s->ev_fork, t->stream), "fork");
        const int used = int(std::min<size_t>(rs_solver::kAux, j - i));
        for (int k = 0; k < used; ++k) RS_HIP(hipStreamWaitEvent(s->aux[k], s->ev_fork, 0), "fork wait");
        for (size_t k = i; k < j; ++k)
            if (int rc = run_launch(s, plan, plan.launches[k], s->aux[(k - i) % rs_solver::kAux], true)) return rc;
        for (int k = 0; k < used; ++k) {
            RS_HIP(hipEventRecord(s->ev_join[k], s->aux[k]), "join");
            RS_HIP(hipStreamWaitEvent(t->stream, s->ev_join[k], 0), "join wait");
        }
        i = j;
    }
    return RS_OK;
}

// phase -1: the whole plan; 0: launches before the exchange; 1: after it (sharded sweeps only)
int run_plan(rs_solver *s, int p, int phase = -1) {
    Plan &plan = s->plan[p];
    rs_table *t = s->table;
    if (phase < 0 && s->params.use_graph && !t->prof.on && !s->sharded && !s->comm) {
        if (!plan.graph_exec) {
            RS_HIP(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
            const int rc = run_range(s, plan, 0, plan.launches.size());
            hipError_t e = hipStreamEndCapture(t->stream, &plan.graph);
            if (rc != RS_OK) return rc;
            RS_HIP(e, "hipStreamEndCapture");
            RS_HIP(hipGraphInstantiate(&plan.graph_exec, plan.graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        }
        RS_HIP(hipGraphLaunch(plan.graph_exec, t->stream), "hipGraphLaunch");
        return RS_OK;
    }
    const size_t lo = phase == 1 ? plan.split : 0, hi = phase == 0 ? plan.split : plan.launches.size();
    return run_range(s, plan, lo, hi);
}

}  // namespace

extern "C" {

static int solver_create_impl(rs_table *table, const rs_tree *tree, const rs_deal_batch *deals, const rs_leaf_desc *leaves_p0,
                              const rs_leaf_desc *leaves_p1, const rs_solver_params *params, rs_solver **out);

int rs_solver_create(rs_table *table, const rs_tree *tree, const rs_leaf_desc *leaves_p0, const rs_leaf_desc *leaves_p1,
                     const rs_solver_params *params, rs_solver **out) {
    return solver_create_impl(table, tree, nullptr, leaves_p0, leaves_p1, params, out);
}

int rs_solver_create_deals(rs_table *table, const rs_tree *tree, const rs_deal_batch *deals, const rs_leaf_desc *leaves_p0,
                           const rs_leaf_desc *leaves_p1, const rs_solver_params *params, rs_solver **out) {
    if (!deals || deals->n_deals == 0) return fail(RS_ERR_INVALID, "rs_solver_create_deals: empty deal batch");
    if (table && table->dtype != RS_I32)
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: deal batches accumulate i32 deltas; use an RS_I32 table");
    if (params && params->chance_mode != RS_CHANCE_PASS)
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: a deal has one run-out: use RS_CHANCE_PASS (cfr.rs:306-313)");
    if (table && (!table->d_dregrets || !table->d_dssum)) {
        hipError_t e = hipSetDevice(table->device);
        const size_t bytes = table->n_cells * 4;
        if (e == hipSuccess && !table->d_dregrets) e = hipMalloc(&table->d_dregrets, bytes);
        if (e == hipSuccess && !table->d_dssum) e = hipMalloc(&table->d_dssum, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(table->d_dregrets, 0, bytes, table->stream);
        if (e == hipSuccess) e = hipMemsetAsync(table->d_dssum, 0, bytes, table->stream);
        if (e != hipSuccess) return hip_fail(e, "rs_solver_create_deals: delta tables");
    }
    return solver_create_impl(table, tree, deals, leaves_p0, leaves_p1, params, out);
}

}  // extern "C"

void rs::solver_release_device(rs_solver *s) {
    if (!s || !s->table) return;   // already detached (its table was destroyed first)
    rs_table *t = s->table;
    (void)hipSetDevice(t->device);
    (void)hipStreamSynchronize(t->stream);
    for (int p = 0; p < 2; ++p) {
        Plan &pl = s->plan[p];
        if (pl.graph_exec) (void)hipGraphExecDestroy(pl.graph_exec);
        if (pl.graph) (void)hipGraphDestroy(pl.graph);
        if (pl.d_jobs) (void)hipFree(pl.d_jobs);
        if (pl.d_chance_jobs) (void)hipFree(pl.d_chance_jobs);
        if (pl.d_reach_nan) (void)hipFree(pl.d_reach_nan);
        if (pl.d_lists) (void)hipFree(pl.d_lists);
        if (pl.d_rlists) (void)hipFree(pl.d_rlists);
        if (pl.d_plists) (void)hipFree(pl.d_plists);
        if (pl.d_apply_jobs) (void)hipFree(pl.d_apply_jobs);
        if (pl.d_counts) (void)hipFree(pl.d_counts);
        if (pl.d_compact_jobs) (void)hipFree(pl.d_compact_jobs);
        for (JitLaunch &JL : pl.jit) {
            if (JL.d_blob) (void)hipFree(JL.d_blob);
            if (JL.d_wl) (void)hipFree(JL.d_wl);
        }
        pl = Plan{};
    }
    if (s->d_arena) (void)hipFree(s->d_arena);
    for (int k = 0; k < rs_solver::kAux; ++k) {
        if (s->aux[k]) (void)hipStreamDestroy(s->aux[k]);
        if (s->ev_join[k]) (void)hipEventDestroy(s->ev_join[k]);
        s->aux[k] = nullptr;
        s->ev_join[k] = nullptr;
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    s->ev_fork = nullptr;
    if (s->d_shadow) (void)hipFree(s->d_shadow);
    if (s->d_shadow_jobs) (void)hipFree(s->d_shadow_jobs);
    for (int r = 0; r < RS_MAX_ROUNDS; ++r) {
        if (s->d_attr[r] && s->d_attr[r] != s->d_arec) (void)hipFree(s->d_attr[r]);
        s->d_attr[r] = nullptr;
    }
    if (s->d_arec) (void)hipFree(s->d_arec);
    if (s->d_order_tot) (void)hipFree(s->d_order_tot);
    s->d_arec = nullptr;
    s->d_order_tot = nullptr;
    if (s->d_pack_jobs) (void)hipFree(s->d_pack_jobs);
    s->d_pack_jobs = nullptr;
    s->n_pack_jobs = 0;
    s->d_shadow = nullptr;
    s->d_shadow_jobs = nullptr;
    if (s->d_seed_state) (void)hipFree(s->d_seed_state);
    if (s->d_exchange) (void)hipFree(s->d_exchange);
    s->d_exchange = nullptr;
    s->d_arena = nullptr;
    s->d_seed_state = nullptr;
    t->solvers.erase(std::remove(t->solvers.begin(), t->solvers.end(), s), t->solvers.end());
    s->table = nullptr;
}

static int solver_create_impl(rs_table *table, const rs_tree *tree, const rs_deal_batch *deals, const rs_leaf_desc *leaves_p0,
                              const rs_leaf_desc *leaves_p1, const rs_solver_params *params, rs_solver **out) {
    if (!table || !tree || !leaves_p0 || !leaves_p1 || !params || !out)
        return fail(RS_ERR_INVALID, "rs_solver_create: NULL argument");
    const int arith = params->mode & RS_UPD_ARITH_MASK;
    if (arith != RS_UPD_CLAMP_I64 && arith != RS_UPD_WRAP_I32) return fail(RS_ERR_INVALID, "rs_solver_create: bad update mode");
    if ((params->mode & RS_UPD_PRUNE) && table->dtype != RS_I32)
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: RS_UPD_PRUNE needs an RS_I32 table");
    if ((params->mode & RS_UPD_RMPLUS) && table->dtype == RS_I32 && arith != RS_UPD_CLAMP_I64)
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: RS_UPD_RMPLUS on i32 tables uses the clamp arithmetic");
    if (params->chance_mode != RS_CHANCE_PASS && params->chance_mode != RS_CHANCE_ENUM)
        return fail(RS_ERR_INVALID, "rs_solver_create: bad chance mode");
    if (params->fuse_subtrees < 0 || params->fuse_subtrees > 1) return fail(RS_ERR_INVALID, "rs_solver_create: fuse_subtrees must be 0 or 1");
    if (params->fuse_subtrees && !jit_available())
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: fuse_subtrees needs libhiprtc.so (tree-specialised kernels); pass 0 for the level plan");
    if (params->opp_mode != RS_OPP_FULL && params->opp_mode != RS_OPP_SAMPLE) return fail(RS_ERR_INVALID, "rs_solver_create: bad opp_mode");
    if (params->opp_mode == RS_OPP_SAMPLE && params->chance_mode != RS_CHANCE_PASS)
        return fail(RS_ERR_UNSUPPORTED, "rs_solver_create: RS_OPP_SAMPLE is mccfr(), whose chance nodes pass through (cfr.rs:306-313): use RS_CHANCE_PASS");
    if (tree->nodes.empty() || tree->nodes[0].kind != RS_NODE_PRIVATE_CHANCE)
        return fail(RS_ERR_INVALID, "rs_solver_create: node 0 must be the private chance root (tree_builder.rs:60-66)");
    if (params->opp_mode == RS_OPP_SAMPLE)
        for (const rs_tree_node &nd : tree->nodes)
            if (nd.kind == RS_NODE_ACTION && nd.n_children == 0)
                return fail(RS_ERR_INVALID, "rs_solver_create: action node " + std::to_string(nd.index) +
                                                " has no valid action; mccfr would panic in WeightedIndex::new(&[]).unwrap() (cfr.rs:471)");

    rs_solver *s = new (std::nothrow) rs_solver();
    if (!s) return fail(RS_ERR_OOM, "rs_solver_create: out of host memory");
    s->table = table;
    table->solvers.push_back(s);
    s->tree = *tree;
    s->params = *params;
    s->knobs = knobs_resolve(&params->forms);
    if (deals) {
        s->deal_mode = true;
        s->deals = *deals;
    }
    const size_t n = tree->nodes.size();
    s->leaves[0].assign(leaves_p0, leaves_p0 + n);
    s->leaves[1].assign(leaves_p1, leaves_p1 + n);
    int rc = derive_geometry(s);
    for (size_t i = 0; rc == RS_OK && i < n; ++i) {
        const rs_tree_node &nd = tree->nodes[i];
        if (nd.kind != RS_NODE_TERMINAL || nd.ttype == RS_TERM_UNCONTESTED) continue;
        for (int p = 0; p < 2; ++p) {
            const rs_leaf_desc &lf = s->leaves[p][i];
            if ((lf.kind != RS_LEAF_SIGN && lf.kind != RS_LEAF_UTIL) || !lf.d_buf)
                rc = fail(RS_ERR_INVALID, "rs_solver_create: terminal " + std::to_string(i) +
                                              " is a showdown / all-in and needs an RS_LEAF_SIGN or RS_LEAF_UTIL buffer");
        }
    }
    if (rc != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    hipError_t e = hipSetDevice(table->device);
    if (e != hipSuccess) {
        rs_solver_destroy(s);
        return hip_fail(e, "hipSetDevice");
    }
    if (params->opp_mode == RS_OPP_SAMPLE) {
        const uint64_t init[3] = {params->sample_seed, 0, 0};
        if ((e = hipMalloc((void **)&s->d_seed_state, sizeof(init))) != hipSuccess ||
            (e = hipMemcpy(s->d_seed_state, init, sizeof(init), hipMemcpyHostToDevice)) != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: seed state");
            rs_solver_destroy(s);
            return rc;
        }
    }
    // streams for independent round subtrees (deal sweeps) and, on request (RS_LANE_OVERLAP=1), for the independent subtree launches of one tree depth in lane sweeps
    if (hipDeviceGetAttribute(&s->lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, table->device) != hipSuccess || s->lds_limit < 1024) s->lds_limit = 64 * 1024;
    if (hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, table->device) != hipSuccess || s->n_cus < 1) s->n_cus = 256;
    if (((s->deal_mode && !s->knobs.no_overlap) || (!s->deal_mode && s->knobs.lane_overlap)) && s->params.fuse_subtrees) {
        e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
        for (int k = 0; e == hipSuccess && k < rs_solver::kAux; ++k) {
            e = hipStreamCreateWithFlags(&s->aux[k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: auxiliary streams");
            rs_solver_destroy(s);
            return rc;
        }
    }
    if (s->deal_mode && s->params.fuse_subtrees) {   // AoS shadow of every table node for the deal kernels' gathers (rs_device.hpp gather_rec)
        std::vector<ShadowJob> jobs;
        size_t ints = 0;
        // A record is worth transposing when the sweep reads it: sampled sweeps reach a node of a later round with probability ~ 1 / (round subtrees of that round), so a
        // node gets a shadow only while n_deals / roots * 8 >= its cells -- 64 K deals against 180 234 river clusters (lossless abstraction, 2 GB table) spent 0.9 ms per
        // sweep transposing records nobody read.  Without one the kernels gather the table's own rows (rs_device.hpp gather_node).  RS_JIT_SHADOW_ALL keeps every shadow.
        std::vector<size_t> round_roots(size_t(s->n_rounds) + 1, 0);
        {
            const int first = [&] { int c = 0; while (tree->nodes[size_t(c)].kind == RS_NODE_PRIVATE_CHANCE || tree->nodes[size_t(c)].kind == RS_NODE_PUBLIC_CHANCE) c = tree->nodes[size_t(c)].children[0]; return c; }();
            if (tree->nodes[size_t(first)].kind == RS_NODE_ACTION) round_roots[size_t(std::min<int>(tree->nodes[size_t(first)].round_idx, s->n_rounds))] += 1;
            for (size_t i = 1; i < n; ++i)
                if (tree->nodes[i].kind == RS_NODE_PUBLIC_CHANCE && tree->nodes[i].n_children > 0) {
                    const rs_tree_node &c = tree->nodes[size_t(tree->nodes[i].children[0])];
                    if (c.kind == RS_NODE_ACTION) round_roots[size_t(std::min<int>(c.round_idx, s->n_rounds))] += 1;
                }
        }
        const bool shadow_all = s->knobs.shadow_all != 0 || s->params.opp_mode != RS_OPP_SAMPLE;
        for (int tp = 0; tp < 2; ++tp) {   // traverser tp's sweep: 2 * half ints per record at its own nodes, half at the opponent's
            s->shadow_off_p[tp].assign(table->nodes.size(), SIZE_MAX);
            s->shadow_stride_p[tp].assign(table->nodes.size(), 0);
            for (size_t i = 0; i < table->nodes.size(); ++i) {
                const rs_node_desc &d = table->nodes[i];
                if (d.n_actions == 0) continue;
                const size_t roots = std::max<size_t>(1, round_roots[size_t(std::min<int>(d.round_idx, s->n_rounds))]);
                if (!shadow_all && !table->tiled(int(i)) && size_t(s->deals.n_deals) * 8 < size_t(d.n_clusters) * d.n_actions * roots) continue;   // no shadow: J.shd = nullptr
                const uint32_t half = d.n_actions <= 4 ? 4 : 8;
                const uint32_t stride = (d.player == tp || s->knobs.shadow_wide) ? 2 * half : half;
                s->shadow_off_p[tp][i] = ints;
                s->shadow_stride_p[tp][i] = stride;
                ShadowJob j{};
                j.regrets = static_cast<const int32_t *>(table->regrets_ptr(int(i)));
                j.ssum = static_cast<const int32_t *>(table->ssum_ptr(int(i)));
                j.pitch = uint32_t(table->pitch[i]);
                j.n_clusters = d.n_clusters;
                j.n_actions = d.n_actions;
                j.half = half;
                j.stride = stride;
                jobs.push_back(j);
                ints += round_up(size_t(d.n_clusters) * stride, 64);
                s->shadow_max_clusters = std::max(s->shadow_max_clusters, d.n_clusters);
            }
        }
        s->other_bytes += std::max<size_t>(ints * 4, 256) + std::max<size_t>(jobs.size() * sizeof(ShadowJob), 256);
        e = hipMalloc((void **)&s->d_shadow, std::max<size_t>(ints * 4, 256));
        if (e == hipSuccess) e = hipMemsetAsync(s->d_shadow, 0, std::max<size_t>(ints * 4, 256), table->stream);
        {
            size_t k = 0;
            for (int tp = 0; tp < 2; ++tp)
                for (size_t i = 0; i < table->nodes.size(); ++i)
                    if (s->shadow_off_p[tp][i] != SIZE_MAX) jobs[k++].dst = s->d_shadow + s->shadow_off_p[tp][i];
        }
        s->n_shadow_jobs = int(jobs.size() / 2);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_shadow_jobs, std::max<size_t>(jobs.size() * sizeof(ShadowJob), 256));
        if (e == hipSuccess && !jobs.empty()) e = hipMemcpy(s->d_shadow_jobs, jobs.data(), jobs.size() * sizeof(ShadowJob), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: table shadow");
            rs_solver_destroy(s);
            return rc;
        }
    }
    // sparse (live-deal list) sweeps: pack the per-deal inputs of every round when ALL showdown / all-in leaves of both traversers share one buffer (the trainer's
    // d_sign; otherwise the kernels keep their separate gathers).  RS_JIT_NO_PACK turns it off (A/B knob)
    if (s->deal_mode && s->params.fuse_subtrees && s->params.opp_mode == RS_OPP_SAMPLE && !s->knobs.no_sparse && !s->knobs.no_pack) {
        const float *leaf = nullptr;
        bool one = true;
        for (size_t i = 0; i < n && one; ++i) {
            const rs_tree_node &nd = tree->nodes[i];
            if (nd.kind != RS_NODE_TERMINAL || nd.ttype == RS_TERM_UNCONTESTED) continue;
            for (int p = 0; p < 2; ++p) {
                if (!leaf) leaf = s->leaves[p][i].d_buf;
                one = one && leaf == s->leaves[p][i].d_buf;
            }
        }
        // Ordered sweeps: worth it when a wave of 64 consecutive deals of the sorted batch mostly shares its last-round cluster, i.e. from about 64 deals per cluster
        // (three streets, 5 000-bucket files, 4 M deals per batch: 838 per cluster) -- small batches against big abstractions keep the unordered forms.
        if (one && leaf) {
            const rs_tree_node &fr = tree->nodes[size_t([&] { int c = 0; while (tree->nodes[size_t(c)].kind == RS_NODE_PRIVATE_CHANCE || tree->nodes[size_t(c)].kind == RS_NODE_PUBLIC_CHANCE) c = tree->nodes[size_t(c)].children[0]; return c; }())];
            const bool round_mode = fr.kind == RS_NODE_ACTION && fr.n_children > 0 && !s->knobs.no_rounds;
            s->order_round = s->n_rounds - 1;
            uint32_t bins[2] = {0, 0};
            for (size_t i = 0; i < table->nodes.size(); ++i)
                if (table->nodes[i].round_idx == s->order_round && table->nodes[i].n_actions > 0) bins[table->nodes[i].player] = std::max(bins[table->nodes[i].player], table->nodes[i].n_clusters);
            const bool fits = bins[0] >= 1 && bins[1] >= 1 && bins[0] <= kOrderMaxBins && bins[1] <= kOrderMaxBins && s->deals.d_cluster[s->order_round][0] &&
                              s->deals.d_cluster[s->order_round][1] && s->deals.n_deals < (1u << 31) && table->dtype == RS_I32;
            // Measured (round 3, one MI355X, profiles/r03_deals_ab.md): the segment-summing river kernels are 1.3-1.4x faster than the tile kernels (three streets, 5 000-bucket files,
            // 4 M deals per batch: 20.7 + 16.1 ms against 29.6 + 19.7 ms over seven batches), but the sort and the records cost 0.15 ms per sweep: 8.36 against 8.56 ms per batch there,
            // 3.57 against 3.40 at 1 M deals, and on the river game 0.84 against 0.69 -- a wash at best, so the form is opt-in (rs_kernel_forms.deal_order = RS_FORM_ON)
            s->ordered = round_mode && fits && s->knobs.ordered != kUnset && s->knobs.ordered != 0;
            if (s->ordered) {
                const size_t pitch = round_up(s->deals.n_deals, kLanePad);
                const uint32_t n = s->deals.n_deals;
                const uint32_t n_chunks = uint32_t(std::min<size_t>(size_t(s->n_cus) * 2, (size_t(n) + kOrderThreads - 1) / kOrderThreads));
                const uint32_t chunk = uint32_t(round_up((size_t(n) + n_chunks - 1) / n_chunks, kOrderThreads));
                const uint32_t max_bins = std::max(bins[0], bins[1]);
                s->other_bytes += pitch * 32 + size_t(2) * max_bins * sizeof(uint32_t);
                e = hipMalloc(&s->d_arec, pitch * 32);
                if (e == hipSuccess) e = hipMemsetAsync(s->d_arec, 0, pitch * 32, table->stream);
                if (e == hipSuccess) e = hipMalloc((void **)&s->d_order_tot, size_t(2) * max_bins * sizeof(uint32_t));
                if (e != hipSuccess) {
                    rc = hip_fail(e, "rs_solver_create: ordered deal records");
                    rs_solver_destroy(s);
                    return rc;
                }
                for (int tp = 0; tp < 2; ++tp) {
                    OrderJob &oj = s->order_job[tp];
                    oj = OrderJob{};
                    oj.key = s->deals.d_cluster[s->order_round][tp];
                    for (int r = 0; r < s->n_rounds; ++r)
                        for (int pl = 0; pl < 2; ++pl) oj.cid[2 * r + pl] = s->deals.d_cluster[r][pl];
                    oj.leaf = leaf;
                    oj.prune = (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr;
                    oj.tot = s->d_order_tot;
                    oj.cursor = s->d_order_tot + bins[tp];
                    oj.arec = s->d_arec;
                    oj.n = n;
                    oj.n_bins = bins[tp];
                    oj.n_chunks = n_chunks;
                    oj.chunk = chunk;
                }
                for (int r = 0; r < s->n_rounds; ++r) s->d_attr[r] = s->d_arec;   // what the generated kernels are handed as J.attr on every round
            }
        }
        if (one && !s->ordered) {
            std::vector<PackJob> jobs;
            const size_t pitch = round_up(s->deals.n_deals, kLanePad);
            for (int r = 0; r < s->n_rounds && e == hipSuccess; ++r) {
                s->other_bytes += pitch * 16;
                e = hipMalloc(&s->d_attr[r], pitch * 16);
                if (e == hipSuccess) e = hipMemsetAsync(s->d_attr[r], 0, pitch * 16, table->stream);
                PackJob j{};
                j.cid0 = s->deals.d_cluster[r][0];
                j.cid1 = s->deals.d_cluster[r][1];
                j.leaf = leaf;
                j.prune = (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr;
                j.out = static_cast<u32x4_host *>(s->d_attr[r]);
                j.n = s->deals.n_deals;
                jobs.push_back(j);
            }
            if (e == hipSuccess) e = hipMalloc((void **)&s->d_pack_jobs, std::max<size_t>(jobs.size() * sizeof(PackJob), 256));
            if (e == hipSuccess && !jobs.empty()) e = hipMemcpy(s->d_pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                rc = hip_fail(e, "rs_solver_create: packed deal inputs");
                rs_solver_destroy(s);
                return rc;
            }
            s->n_pack_jobs = int(jobs.size());
        }
    }
    Builder b0(s, 0), b1(s, 1);
    if ((rc = b0.build()) != RS_OK || (rc = b1.build()) != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    // Only the list-walking kernels read the packed records: a round whose subtrees all walk the whole batch (the first round; every round of a one-round game) needs none
    if (s->n_pack_jobs) {
        std::vector<PackJob> keep;
        for (int r = 0; r < s->n_rounds; ++r) {
            if (!s->d_attr[r]) continue;
            if (s->attr_used & (1u << r)) {
                PackJob j{};
                j.cid0 = s->deals.d_cluster[r][0];
                j.cid1 = s->deals.d_cluster[r][1];
                j.prune = (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr;
                j.out = static_cast<u32x4_host *>(s->d_attr[r]);
                j.n = s->deals.n_deals;
                keep.push_back(j);
            } else {
                (void)hipFree(s->d_attr[r]);
                s->d_attr[r] = nullptr;
                s->other_bytes -= round_up(s->deals.n_deals, kLanePad) * 16;
            }
        }
        if (int(keep.size()) != s->n_pack_jobs) {
            std::vector<PackJob> all(size_t(s->n_pack_jobs));
            e = hipMemcpy(all.data(), s->d_pack_jobs, all.size() * sizeof(PackJob), hipMemcpyDeviceToHost);
            for (PackJob &j : keep) j.leaf = all[0].leaf;   // one leaf buffer for every round (the condition under which records are packed at all)
            if (e == hipSuccess && !keep.empty()) e = hipMemcpy(s->d_pack_jobs, keep.data(), keep.size() * sizeof(PackJob), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                rc = hip_fail(e, "rs_solver_create: packed deal inputs");
                rs_solver_destroy(s);
                return rc;
            }
            s->n_pack_jobs = int(keep.size());
            if (keep.empty())
                for (int p = 0; p < 2; ++p) {   // the pack launch goes too (plan.split counts launches: keep it pointing at the same one)
                    Plan &pl = s->plan[p];
                    std::vector<Launch> kept;
                    size_t split = pl.split;
                    for (size_t i = 0; i < pl.launches.size(); ++i) {
                        if (pl.launches[i].kind == L_PACK) {
                            if (i < pl.split) --split;
                            continue;
                        }
                        kept.push_back(pl.launches[i]);
                    }
                    pl.launches.swap(kept);
                    pl.split = split;
                }
        }
    }
    s->arena_bytes = std::max(s->plan[0].arena_bytes, s->plan[1].arena_bytes);
    if ((e = hipMalloc((void **)&s->d_arena, std::max<size_t>(s->arena_bytes, 256))) != hipSuccess) {
        rc = hip_fail(e, "rs_solver_create: workspace hipMalloc");
        rs_solver_destroy(s);
        return rc;
    }
    // padding lanes are read by the vector kernels: keep them finite
    (void)hipMemsetAsync(s->d_arena, 0, std::max<size_t>(s->arena_bytes, 256), table->stream);
    if (s->sharded) {
        const size_t nb = size_t(std::max(s->plan[0].n_boundary, s->plan[1].n_boundary));
        s->exchange_floats_per_rank = nb * s->slot_lanes;
        const size_t bytes = std::max<size_t>(size_t(s->params.shard_world) * s->exchange_floats_per_rank * sizeof(float), 256);
        s->other_bytes += bytes;
        if ((e = hipMalloc((void **)&s->d_exchange, bytes)) != hipSuccess || (e = hipMemsetAsync(s->d_exchange, 0, bytes, table->stream)) != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: exchange buffer");
            rs_solver_destroy(s);
            return rc;
        }
    }
    if ((rc = b0.emit()) != RS_OK || (rc = b1.emit()) != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    for (int p = 0; p < 2; ++p) {
        Plan &pl = s->plan[p];
        const size_t bytes = std::max<size_t>(pl.jobs.size(), 1) * sizeof(NodeJob);
        s->other_bytes += bytes + std::max<size_t>(pl.chance_jobs.size(), 1) * sizeof(ChanceJob) + pl.n_count_words * sizeof(uint32_t) + pl.compact_jobs.size() * sizeof(CompactJob) +
                          size_t(pl.n_apply_jobs) * sizeof(ApplyJob);
        for (const JitLaunch &JL : pl.jit) s->other_bytes += JL.blob.size() + (JL.worklist ? (size_t(JL.n_jobs) + 3) * sizeof(uint32_t) : 0);
        if ((e = hipMalloc((void **)&pl.d_jobs, bytes)) != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: job allocation");
            rs_solver_destroy(s);
            return rc;
        }
        if ((e = hipMalloc((void **)&pl.d_chance_jobs, std::max<size_t>(pl.chance_jobs.size(), 1) * sizeof(ChanceJob))) != hipSuccess ||
            (e = hipMemcpyAsync(pl.d_chance_jobs, pl.chance_jobs.data(), pl.chance_jobs.size() * sizeof(ChanceJob),
                                hipMemcpyHostToDevice, table->stream)) != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: chance job upload");
            rs_solver_destroy(s);
            return rc;
        }
        for (JitLaunch &JL : pl.jit) {
            if (JL.worklist && (e = hipMalloc((void **)&JL.d_wl, (size_t(JL.n_jobs) + 3) * sizeof(uint32_t))) != hipSuccess) {
                rc = hip_fail(e, "rs_solver_create: work list");
                rs_solver_destroy(s);
                return rc;
            }
            if ((e = hipMalloc((void **)&JL.d_blob, JL.blob.size())) != hipSuccess ||
                (e = hipMemcpyAsync(JL.d_blob, JL.blob.data(), JL.blob.size(), hipMemcpyHostToDevice, table->stream)) != hipSuccess) {
                rc = hip_fail(e, "rs_solver_create: tree-kernel argument upload");
                rs_solver_destroy(s);
                return rc;
            }
        }
        if ((e = hipMemcpyAsync(pl.d_jobs, pl.jobs.data(), pl.jobs.size() * sizeof(NodeJob), hipMemcpyHostToDevice,
                                table->stream)) != hipSuccess) {
            rc = hip_fail(e, "rs_solver_create: job upload");
            rs_solver_destroy(s);
            return rc;
        }
    }
    if ((e = hipStreamSynchronize(table->stream)) != hipSuccess) {
        rc = hip_fail(e, "rs_solver_create: sync");
        rs_solver_destroy(s);
        return rc;
    }
    *out = s;
    return RS_OK;
}

extern "C" {

void rs_solver_destroy(rs_solver *s) {
    if (!s) return;
    solver_release_device(s);
    delete s;
}

static int copy_root(rs_solver *s, int traverser, float *d_root_util) {
    if (d_root_util && s->ordered) {   // the sweep's lanes are ranks of its order: hand the utilities out by deal id
        RS_HIP(launch_unpermute_f32(s->plan[traverser].root_util, s->d_arec, d_root_util, s->deals.n_deals, s->table->stream), "rs_iterate: root util by deal id");
        return RS_OK;
    }
    if (d_root_util)
        RS_HIP(hipMemcpyAsync(d_root_util, s->plan[traverser].root_util, s->plan[traverser].root_lanes * sizeof(float),
                              hipMemcpyDeviceToDevice, s->table->stream),
               "rs_iterate: root util copy");
    return RS_OK;
}

int rs_iterate(rs_solver *s, int traverser, float *d_root_util) {
    if (!s) return fail(RS_ERR_INVALID, "rs_iterate: solver is NULL");
    if (!s->table) return fail(RS_ERR_INVALID, "rs_iterate: the solver's table has been destroyed");
    if (traverser != 0 && traverser != 1) return fail(RS_ERR_INVALID, "rs_iterate: traverser must be 0 or 1");
    RS_HIP(hipSetDevice(s->table->device), "hipSetDevice");
    if (s->sharded) {
        if (!s->comm) return fail(RS_ERR_INVALID, "rs_iterate: sharded solver without a communicator: rs_solver_attach_comm, or drive the "
                                                  "phases with rs_iterate_phase and exchange the slots yourself");
        if (int rc = run_plan(s, traverser, 0)) return rc;
        if (int rc = rs_comm_allgather(s->comm, s->table, s->d_exchange, size_t(s->plan[traverser].n_boundary) * s->slot_lanes * sizeof(float)))
            return rc;
        if (int rc = run_plan(s, traverser, 1)) return rc;
        return copy_root(s, traverser, d_root_util);
    }
    if (s->deal_mode && s->comm) {   // data-parallel deal batches: sweep, sum the deltas over the ranks, apply the union
        if (int rc = run_plan(s, traverser, 0)) return rc;
        if (int rc = rs_comm_allreduce_deltas(s->comm, s->table)) return rc;
        if (int rc = run_plan(s, traverser, 1)) return rc;
        return copy_root(s, traverser, d_root_util);
    }
    if (int rc = run_plan(s, traverser)) return rc;
    return copy_root(s, traverser, d_root_util);
}

int rs_iterate_phase(rs_solver *s, int traverser, int phase, float *d_root_util) {
    if (!s || !s->table) return fail(RS_ERR_INVALID, "rs_iterate_phase: bad solver");
    if ((traverser != 0 && traverser != 1) || (phase != 0 && phase != 1)) return fail(RS_ERR_INVALID, "rs_iterate_phase: bad traverser / phase");
    if (!s->sharded && !s->deal_mode) return fail(RS_ERR_INVALID, "rs_iterate_phase: the solver is neither sharded nor a deal-batch solver");
    RS_HIP(hipSetDevice(s->table->device), "hipSetDevice");
    if (int rc = run_plan(s, traverser, phase)) return rc;
    return phase == 1 ? copy_root(s, traverser, d_root_util) : RS_OK;
}

int rs_solver_attach_comm(rs_solver *s, rs_comm *comm) {
    if (!s) return fail(RS_ERR_INVALID, "rs_solver_attach_comm: solver is NULL");
    s->comm = comm;
    return RS_OK;
}

int rs_solver_exchange_info(rs_solver *s, int traverser, void **d_buf, size_t *bytes_per_rank) {
    if (!s || !d_buf || !bytes_per_rank || traverser < 0 || traverser > 1) return fail(RS_ERR_INVALID, "rs_solver_exchange_info: bad argument");
    if (!s->sharded) return fail(RS_ERR_INVALID, "rs_solver_exchange_info: the solver is not sharded");
    *d_buf = s->d_exchange;
    *bytes_per_rank = size_t(s->plan[traverser].n_boundary) * s->slot_lanes * sizeof(float);
    return RS_OK;
}

int rs_train(rs_solver *s, uint64_t iterations, uint64_t discount_interval, uint64_t discount_cap) {
    if (!s) return fail(RS_ERR_INVALID, "rs_train: solver is NULL");
    if (!s->table) return fail(RS_ERR_INVALID, "rs_train: the solver's table has been destroyed");
    if (discount_interval == 0) return fail(RS_ERR_INVALID, "rs_train: discount_interval must be > 0");
    uint64_t t = 0, threshold = discount_interval;
    while (t < iterations) {                       // cfr.rs:207
        for (int player = 0; player < 2; ++player)  // cfr.rs:216-224
            if (int rc = rs_iterate(s, player, nullptr)) return rc;
        t += 1;                                     // cfr.rs:226
        if (t > discount_cap) continue;             // cfr.rs:240-242
        if (t > threshold) {                        // cfr.rs:243
            if (int rc = rs_discount(s->table, rs_discount_factor(t, discount_interval))) return rc;  // cfr.rs:248-261
            threshold = t + discount_interval;      // cfr.rs:262
        }
    }
    return RS_OK;
}

size_t rs_solver_workspace_bytes(const rs_solver *s) { return s ? s->arena_bytes + s->plan[0].aux_bytes + s->plan[1].aux_bytes + s->other_bytes : 0; }

int rs_jit_available(void) { return jit_available() ? 1 : 0; }

// Generates and compiles (no GPU needed) the tree-specialised kernel of every chance-free subtree of `tree`, for
// both traversers, assuming one shared sign buffer per round.  *n_kernels = distinct kernels.
int rs_jit_check_tree(const rs_tree *tree, int dtype, int mode, int opp_mode, int *n_kernels) {
    if (!tree || tree->nodes.empty()) return fail(RS_ERR_INVALID, "rs_jit_check_tree: bad tree");
    const std::vector<rs_tree_node> &nodes = tree->nodes;
    const size_t n = nodes.size();
    std::map<std::string, int> seen;
    const Knobs knobs = knobs_resolve(nullptr);
    for (int p = 0; p < 2; ++p) {
        std::vector<char> has_own(n, 0), closed(n, 0);
        std::vector<int> leaf_buf(n, -1), leaf_flags(n, 0);
        for (size_t i = n; i-- > 0;) {   // children have larger ids than parents
            const rs_tree_node &nd = nodes[i];
            bool own = nd.kind == RS_NODE_ACTION && nd.player == p;
            bool cl = nd.kind != RS_NODE_PUBLIC_CHANCE && nd.kind != RS_NODE_PRIVATE_CHANCE;
            for (int k = 0; k < nd.n_children; ++k) {
                own = own || has_own[nd.children[k]];
                cl = cl && closed[nd.children[k]];
            }
            has_own[i] = own;
            closed[i] = cl;
            if (nd.kind == RS_NODE_TERMINAL && nd.ttype != RS_TERM_UNCONTESTED) {
                leaf_buf[i] = nd.round;
                leaf_flags[i] = 1;
            }
        }
        std::vector<char> next_root(n, 0);
        for (size_t i = 0; i < n; ++i)
            if (nodes[i].kind == RS_NODE_PUBLIC_CHANCE && nodes[size_t(nodes[i].children[0])].kind == RS_NODE_ACTION && nodes[size_t(nodes[i].children[0])].n_children > 0)
                next_root[size_t(nodes[i].children[0])] = 1;
        for (size_t i = 0; i < n && opp_mode == RS_OPP_FULL; ++i) {   // lane round subtrees (ENUM sweeps): reach-down and walk-up kernel of every non-closed round root
            const rs_tree_node &nd = nodes[i];
            if (nd.kind != RS_NODE_ACTION || closed[i] || nd.n_children == 0) continue;
            if (nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_ACTION) continue;   // inside a round subtree
            for (int form = 0; form < 4; ++form) {   // walk-up / reach-down, each reading its own reach row or the chance node's above (expand step taken over)
                const bool down = (form & 1) != 0, xr = (form & 2) != 0;
                if (xr && !(nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_PUBLIC_CHANCE)) continue;
                JitSubtree js;
                jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, dtype, mode & RS_UPD_ARITH_MASK, false, false, false, false, down, (mode & RS_UPD_PRUNE) != 0, 4, &next_root, js, knobs,
                                 xr ? 1 : 0);
                if (down && js.boundary_roots.empty()) continue;
                if (seen.count(js.source)) continue;
                seen[js.source] = 1;
                if (int rc = jit_compile_only(js.source, knobs.dump != 0)) return rc;
            }
        }
        for (size_t i = 0; i < n; ++i) {
            const rs_tree_node &nd = nodes[i];
            if (nd.kind != RS_NODE_ACTION || !closed[i] || nd.n_children == 0) continue;
            if (nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_ACTION && closed[nd.parent]) continue;   // not topmost
            for (int fan = 0; fan < 3; ++fan) {   // below a public chance node also the forms that take over the node's expand (1) and its deal loop (2)
                if (fan && !(nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_PUBLIC_CHANCE)) continue;
                JitSubtree js;
                jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, dtype, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, false, false, false, false,
                                 (mode & RS_UPD_PRUNE) != 0, 4, nullptr, js, knobs, fan);
                if (seen.count(js.source)) continue;
                seen[js.source] = 1;
                if (int rc = jit_compile_only(js.source, knobs.dump != 0)) return rc;
            }
        }
    }
    if (n_kernels) *n_kernels = int(seen.size());
    return RS_OK;
}
// the same for deal batches: every round subtree (cut at the chance nodes) in the forms rs_solver_create_deals can pick -- the DOWN half and the
// table-updating walk, each dense and over a live-deal list, the walk with and without LDS tiles
int rs_jit_check_tree_deals(const rs_tree *tree, int mode, int opp_mode, int *n_kernels) {
    if (!tree || tree->nodes.empty()) return fail(RS_ERR_INVALID, "rs_jit_check_tree_deals: bad tree");
    const std::vector<rs_tree_node> &nodes = tree->nodes;
    const size_t n = nodes.size();
    auto resolve = [&](int c) {
        while (nodes[size_t(c)].kind == RS_NODE_PRIVATE_CHANCE || nodes[size_t(c)].kind == RS_NODE_PUBLIC_CHANCE) c = nodes[size_t(c)].children[0];
        return c;
    };
    std::vector<char> root(n, 0);   // first action node, and every action node reached through a chance node
    const int first = resolve(0);
    if (nodes[size_t(first)].kind == RS_NODE_ACTION && nodes[size_t(first)].n_children > 0) root[size_t(first)] = 1;
    for (size_t i = 0; i < n; ++i)
        if (i > 0 && (nodes[i].kind == RS_NODE_PRIVATE_CHANCE || nodes[i].kind == RS_NODE_PUBLIC_CHANCE)) {
            const int c = resolve(int(i));
            if (nodes[size_t(c)].kind == RS_NODE_ACTION && nodes[size_t(c)].n_children > 0) root[size_t(c)] = 1;
        }
    std::map<std::string, int> seen;
    const Knobs knobs = knobs_resolve(nullptr);
    for (int p = 0; p < 2; ++p) {
        std::vector<char> has_own(n, 0);
        std::vector<int> leaf_buf(n, -1), leaf_flags(n, 0);
        for (size_t i = n; i-- > 0;) {   // children have larger ids than parents
            const rs_tree_node &nd = nodes[i];
            bool own = nd.kind == RS_NODE_ACTION && nd.player == p;
            for (int k = 0; k < nd.n_children; ++k) own = own || has_own[size_t(nd.children[k])];
            has_own[i] = own;
            if (nd.kind == RS_NODE_TERMINAL && nd.ttype != RS_TERM_UNCONTESTED) {
                leaf_buf[i] = 0;
                leaf_flags[i] = 1;
            }
        }
        for (size_t i = 0; i < n; ++i) {
            if (!root[i]) continue;
            for (int form = 0; form < 12; ++form) {   // down dense / sparse, walk lds dense / sparse, walk direct dense / sparse; four deals per thread, then one
                const int lanes = form < 6 ? 4 : 1, f6 = form % 6;
                const bool down = f6 < 2, sparse = (f6 & 1) != 0, lds = f6 >= 2 && f6 < 4;
                if (lanes == 4 && nodes[i].round_idx != nodes[size_t(first)].round_idx) continue;   // later rounds always run one deal per thread
                if (sparse && opp_mode != RS_OPP_SAMPLE) continue;
                JitSubtree js;
                jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                 (mode & RS_UPD_PRUNE) != 0, lanes, &root, js, knobs, 0, sparse);   // sparse forms fetch packed per-deal records (the separate gathers remain as the
                                                                                               // fallback for solvers whose leaves do not share one buffer: compiled by the GPU tests)
                if (down && js.boundary_roots.empty()) continue;   // a last-round subtree hands no reach on
                if (!seen.count(js.source)) {
                    seen[js.source] = 1;
                    if (int rc = jit_compile_only(js.source, knobs.dump != 0)) return rc;
                }
                if (opp_mode == RS_OPP_SAMPLE && (f6 == 0 || f6 == 1 || f6 == 2 || f6 == 3) && (mode & RS_UPD_ARITH_MASK) == RS_UPD_CLAMP_I64) {
                    // ordered sweeps (rs_kernel_forms.deal_order): 32-byte records by rank; the last round's walk sums its deltas by wave segments instead of LDS tiles
                    int last_round = 0;
                    for (const rs_tree_node &q : nodes)
                        if (q.kind == RS_NODE_ACTION) last_round = std::max(last_round, int(q.round_idx));
                    JitSubtree jo;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, true, true, lds, sparse, down, (mode & RS_UPD_PRUNE) != 0, lanes, &root, jo, knobs,
                                     0, true, sparse, sparse && lds && !down, true, nodes[i].round_idx == last_round);
                    if (!(down && jo.boundary_roots.empty()) && !seen.count(jo.source)) {
                        seen[jo.source] = 1;
                        if (int rc = jit_compile_only(jo.source, knobs.dump != 0)) return rc;
                    }
                }
                if (sparse && lds && !down) {   // the work-list form every list-walking kernel with LDS tiles is launched in
                    JitSubtree jw;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                     (mode & RS_UPD_PRUNE) != 0, lanes, &root, jw, knobs, 0, sparse, true, true);
                    if (!seen.count(jw.source)) {
                        seen[jw.source] = 1;
                        if (int rc = jit_compile_only(jw.source, knobs.dump != 0)) return rc;
                    }
                }
                if (sparse && !js.boundary_roots.empty()) {   // the forms that address the rows shared with the next round by list position (large batches)
                    JitSubtree jp;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                     (mode & RS_UPD_PRUNE) != 0, lanes, &root, jp, knobs, 0, sparse, true);
                    if (!seen.count(jp.source)) {
                        seen[jp.source] = 1;
                        if (int rc = jit_compile_only(jp.source, knobs.dump != 0)) return rc;
                    }
                }
            }
        }
    }
    if (n_kernels) *n_kernels = int(seen.size());
    return RS_OK;
}
int rs_solver_forms(const rs_solver *s) { return s ? (s->ordered ? 1 : 0) : RS_ERR_INVALID; }
int rs_solver_n_launches(const rs_solver *s, int traverser) {
    if (!s || traverser < 0 || traverser > 1) return RS_ERR_INVALID;
    return int(s->plan[traverser].launches.size());
}

}  // extern "C"
