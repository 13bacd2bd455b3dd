// rs_plan_builder.hpp -- the plan builder's state and interface, shared by rs_plan.cpp (lane sweeps and everything common) and rs_plan_deals.cpp (what only deal sweeps
// need: cluster ranges, live-deal lists, round subtrees, the delta apply).  Internal.
#pragma once

#include "rs_plan.hpp"

namespace rs {

struct PlanBuilder {
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
    std::vector<uint32_t *> mask_of_root;    // per round root whose reach-down kernel writes a liveness mask for its next-round roots (CompactJob.mask): its mask row
    std::vector<int> mask_bit_of_root;       // per next-round root: its bit in the parent's mask words, -1 = none
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
    bool seg_root(int root) const { return s->ordered && nodes[size_t(root)].round_idx == s->order_round && !rows_root(root); }   // its deltas are summed by wave segments: no LDS tiles
    // delta rows (rs_kernel_forms.delta_rows): the walk of this root stores its deltas by list position, k_row_sums adds them up per cluster: no tiles, no cluster ranges
    // (the list walkers only: the first round's dense walk keeps its tiles)
    bool rows_root(int root) const { return round_mode && rows_round_ok(s, p, nodes[size_t(root)].round_idx) && want_lists && root != first_root; }
    Parts parts_of(int root) const;
    // hand-off rows (rs_jit.cpp): the reach-down kernel of `root` stores its draws by list position, the walk of `root` reads them
    std::vector<size_t> hrow_off;   // per tree node: float offset of the root's rows in plan.d_hrows, SIZE_MAX = none
    bool handoff_root(int root) const { return !hrow_off.empty() && hrow_off[size_t(root)] != SIZE_MAX; }

    PlanBuilder(rs_solver *s_, int p_) : s(s_), p(p_), plan(s_->plan[p_]), nodes(s_->tree.nodes) {}

    size_t alloc(int round, size_t segments = 1);
    float *aptr(size_t off1) const { return reinterpret_cast<float *>(s->d_arena + (off1 - 1)); }
    std::vector<float *> util_override;   // sharded: the utility rows of boundary children live in the exchange buffer
    std::vector<int> boundary_k;          // chance node id -> index among the boundary nodes, -1 otherwise
    float *uptr(int id) const { return util_override[id] ? util_override[id] : aptr(util_off[id]); }
    float *nan_ptr(int id) const { return plan.d_reach_nan + nan_off[size_t(nan_slot[size_t(id)])]; }   // round mode: a root's reach buffer
    // will `root` walk a live-deal list?  (every round root but the first does once lists are wanted; the first only when its tiles had to be partitioned)
    bool listed_root(int root) { return want_lists && (root != first_root || parts_of(first_root).first > 1); }
    // deals below a chance node: global count when its child round is the sharded one
    uint32_t fan_of(int chance_id) const;
    bool boundary(int chance_id) const {
        return s->sharded && nodes[chance_id].kind == RS_NODE_PUBLIC_CHANCE && lane_round[nodes[chance_id].children[0]] == s->params.shard_round;
    }

    void annotate(int id, int d, int round);

    // ---- fused subtrees: every topmost chance-free subtree becomes ONE tree-specialised kernel ---------------------
    void mark_inside(int id);
    void fan_mode(int id);
    void mark_lane_round_inside(int root, int id);
    void mark_fused(int id);

    int resolve(int c) const;
    void mark_round_inside(int root, int id);
    void mark_round(int root);
    // `segments`: the utility row of a root below a LISTED parent is addressed by the parent's list position, one segment per cluster range of the parent (pos_rows)
    void layout_round(int root, size_t segments = 1);

    // an action node without valid actions (state.rs:125-157 can return none): worth 0, owns nothing, launches nothing
    bool dead_end(int id) const { return nodes[id].kind == RS_NODE_ACTION && nodes[id].n_children == 0; }

    bool chance_enum(const rs_tree_node &nd) const {
        return nd.kind == RS_NODE_PUBLIC_CHANCE && s->params.chance_mode == RS_CHANCE_ENUM;
    }

    // pass 1: decide which buffers exist (offsets only; the arena is allocated afterwards)
    void layout(int id);

    ChildSrc child_source(int c) const;

    void node_job(int id, NodeJob &job) const;

    double lanes(int id) const { return double(s->n_boards[lane_round[id]]) * s->n_clusters; }

    int build();

    // one subtree job of a tree-specialised kernel: `down` = the top-down half of a round subtree (deal batches), else the walk that updates the table
    int add_jit_job(int id, bool down, const std::vector<int> &sparse_slot, std::map<uint64_t, int> &by_fn);

    // ---- deal sweeps (rs_solver_create_deals): live-deal lists, the round subtrees' reach-down halves ------------------------------------------------
    std::vector<int> sparse_slot;   // per tree node: index of its compact job (the list of its live deals), -1 = it walks every lane
    int emit_deal_lists();
    int emit_round_walks();
    std::vector<char> sigma_node;   // per tree node: this sweep's shadow record of the node holds its strategy (an opponent node with a shadow; rs_solver.cpp ShadowJob.sigma)
    int emit_row_sums();
    int emit_apply();

    // pass 2 (after the arena exists): emit jobs and launches
    int emit();
};

}  // namespace rs
