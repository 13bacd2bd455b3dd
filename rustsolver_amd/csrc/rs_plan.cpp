// rs_plan.cpp -- the host-side traversal scheduler, part 1: from (tree, table, leaves, parameters) to a static launch plan per traverser.
// MCCFRTrainer (cfr.rs:146-297) for the batched lane model: the branchy public-tree walk happens ONCE, here on the host:
//   top-down, by tree depth   : opponent nodes write reach[child] = sigma[a]*reach (cfr.rs:585),
//                               ENUM chance nodes expand reach * 1/len to the child round's boards
//   bottom-up, by tree depth  : opponent nodes write util = sum sigma*u (cfr.rs:588), traverser nodes
//                               run the regret / strategy_sum update (cfr.rs:612-621 or :413-464),
//                               ENUM chance nodes sum their deals (cfr.rs:519)
// Nodes of one depth, kernel kind and action count share one launch (blockIdx.y = node); every chance-free subtree (and, with chance nodes, every round subtree) becomes
// one tree-specialised kernel (rs_jit.cpp).  Terminal children cost no launch and no buffer: their utility is a constant or a sign lookup folded into the consuming
// kernel.  Deal batches (lanes = deals, rs_solver_create_deals) add the sweep's shadow, live-deal lists, cluster ranges and work lists.  rs_solver.cpp replays the plan.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "rs_plan_builder.hpp"

using namespace rs;

namespace rs {

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

size_t PlanBuilder::alloc(int round, size_t segments) {
    const size_t off = arena;
    arena += round_up(s->pitch[round] * segments * sizeof(float), 256);
    return off + 1;
}

// deals below a chance node: global count when its child round is the sharded one
uint32_t PlanBuilder::fan_of(int chance_id) const {
    const int c = nodes[chance_id].children[0];
    if (s->sharded && lane_round[c] == s->params.shard_round) return s->params.shard_global_boards / s->n_boards[lane_round[chance_id]];
    return s->n_boards[lane_round[c]] / s->n_boards[lane_round[chance_id]];
}

void PlanBuilder::annotate(int id, int d, int round) {
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
void PlanBuilder::mark_inside(int id) {
    for (int k = 0; k < nodes[id].n_children; ++k) {
        inside[nodes[id].children[k]] = 1;
        mark_inside(nodes[id].children[k]);
    }
}

void PlanBuilder::fan_mode(int id) {   // may the kernel of root `id` take over work of the ENUM chance node above it?
    const int par = nodes[id].parent;
    if (par >= 0 && chance_enum(nodes[par]) && !boundary(par) && s->n_clusters % 4 == 0 && !s->deal_mode) {
        if (s->knobs.fan == kUnset || s->knobs.fan == 1) fan_root[id] = 1;
    }
}

void PlanBuilder::mark_lane_round_inside(int root, int id) {
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

void PlanBuilder::mark_fused(int id) {
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

int PlanBuilder::resolve(int c) const {
    while (nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE) c = nodes[c].children[0];
    return c;
}

// pass 1: decide which buffers exist (offsets only; the arena is allocated afterwards)
void PlanBuilder::layout(int id) {
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

ChildSrc PlanBuilder::child_source(int c) const {
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

void PlanBuilder::node_job(int id, NodeJob &job) const {
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

int PlanBuilder::build() {
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
        round_mode = s->deal_mode && s->params.fuse_subtrees && nodes[first_root].kind == RS_NODE_ACTION &&
                     nodes[first_root].n_children > 0;
        if (s->deal_mode && s->table->dtype != RS_I32 && !round_mode)
            return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: float tables run through the generated round subtrees only (the first node below the root must be an action node)");
        want_lists = s->deal_mode && s->params.opp_mode == RS_OPP_SAMPLE && s->table->dtype == RS_I32;   // f32 deal sweeps walk every deal (their delta rows are per deal)
        want_parts = round_mode && want_lists;
        // three streets, 4 M deals per batch: 13.49 -> 12.51 ms; 1 M: 5.98 -> 5.74; but 64 K: 2.14 -> 2.34 (latency-bound: the list adds a dependent load per entry),
        // so only batches beyond the small-batch switch (RS_JIT_SCAN_ALL = 1 / 0 forces either)
        scan_parent = round_mode && want_lists && s->deals.n_deals > kScanParentMin;   // with list-position rows it pays from 128 K deals per batch on (1.21 -> 1.07 ms; 64 K: 1.00 -> 1.02,
                                                                                       // lossless abstractions at 64 K 2.5 -> 2.9: gpurun_out/r04d/ab_scan_small.log)
        if (kn.scan_all != kUnset) scan_parent = round_mode && want_lists && kn.scan_all == 0;
        pos_rows = scan_parent;
        lds_limit = s->lds_limit;
        if (kn.lds_max != kUnset) lds_limit = std::min(lds_limit, kn.lds_max);
        lds_limit -= int(kWorklistLdsBytes);   // the work-list kernels keep their ticket in front of the tiles
    }
    if (round_mode) {
        mark_round(first_root);
        layout_round(first_root);
    } else {
        // lane sweeps: round subtrees above the last round (full-width cfr() with ENUM chance nodes; prune keeps the level plan's NaN bookkeeping)
        lane_rounds = !s->deal_mode && s->params.fuse_subtrees && s->params.chance_mode == RS_CHANCE_ENUM && s->params.opp_mode == RS_OPP_FULL;
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
int PlanBuilder::add_jit_job(int id, bool down, const std::vector<int> &sparse_slot, std::map<uint64_t, int> &by_fn) {
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
    const bool sparse = sparse_slot[id] >= 0;
    const Parts parts = sparse ? parts_of(id) : Parts{1u, 0u, 0u, 0u};
    if (parts.first > 1) lds_need = lds_need / std::max<size_t>(1, parts.pitch) * parts.second;   // tiles cover one range
    const bool seg = seg_root(id) && !down && t->dtype == RS_I32;
    const bool rows = s->deal_mode && rows_root(id) && !down;
    const bool use_lds = s->deal_mode && lds_need > 0 && lds_need <= size_t(lds_limit) && !down && !seg && !rows && t->dtype == RS_I32;   // f32 deal sweeps add nothing anywhere: no tiles
    if (s->deal_mode && sigma_node.empty() && !s->shadow_off_p[p].empty()) {
        sigma_node.assign(n, 0);
        for (size_t q = 0; q < n; ++q) {
            const rs_tree_node &qn = nodes[q];
            if (qn.kind != RS_NODE_ACTION || qn.n_children == 0 || qn.player == p) continue;
            const bool shadowed = s->shadow_off_p[p][size_t(qn.index)] != SIZE_MAX;
            const uint32_t half = qn.n_children <= 2 ? 2u : (qn.n_children <= 4 ? 4u : 8u);
            sigma_node[q] = shadowed && s->shadow_rec_p[p][size_t(qn.index)] == half;
        }
    }
    // staged rows: when every node of the round subtree has a shadow record this sweep, the list walkers read them from rows their waves stage in LDS (rs_device.hpp stage_rows)
    JitStage stage;
    const int32_t *stage_rows_of[2] = {nullptr, nullptr};
    if (s->deal_mode && sparse && !s->shadow_off_p[p].empty() && !s->knobs.no_stage) {
        stage.off.assign(n, 0);
        bool all = true;
        std::vector<int> stack{id};
        while (!stack.empty()) {
            const int q = stack.back();
            stack.pop_back();
            const rs_tree_node &qn = nodes[q];
            if (qn.kind == RS_NODE_ACTION && qn.n_children > 0) {
                const size_t ti = size_t(qn.index);
                const int32_t *base = s->shadow_off_p[p][ti] == SIZE_MAX ? nullptr : s->d_shadow + (s->shadow_off_p[p][ti] - s->shadow_rowoff_p[p][ti]);
                if (!base || s->shadow_stride_p[p][ti] % 4 || (stage_rows_of[qn.player] && stage_rows_of[qn.player] != base)) all = false;   // one row per player, or none
                else {
                    stage.ch[qn.player] = int(s->shadow_stride_p[p][ti] / 4);
                    stage.chp[qn.player] = stage.ch[qn.player];   // rows back to back in LDS: one address register for all of the wave's stores
                    stage.off[size_t(q)] = int(s->shadow_rowoff_p[p][ti]);
                    stage_rows_of[qn.player] = s->d_shadow + (s->shadow_off_p[p][ti] - s->shadow_rowoff_p[p][ti]);
                }
            }
            for (int k = 0; k < qn.n_children; ++k) {
                const int c = qn.children[k];
                const bool chance = nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE;
                if (!(round_mode && chance)) stack.push_back(c);
            }
        }
        if (!all) stage.ch[0] = stage.ch[1] = 0;
    }
    // the dense reach-down kernel of the first round of a big batch: staged too, from the row copy setup_table_shadow keeps for it (one deal per lane)
    bool dense_down_staged = false;
    if (s->deal_mode && down && !sparse && id == first_root && !s->down_off_p[p].empty() && !s->knobs.no_stage && s->knobs.lanes == kUnset) {
        stage = JitStage{};
        stage.off.assign(n, 0);
        stage_rows_of[0] = stage_rows_of[1] = nullptr;
        bool all = true, any = false;
        std::vector<int> stack{id};
        while (!stack.empty()) {
            const int q = stack.back();
            stack.pop_back();
            const rs_tree_node &qn = nodes[q];
            if (qn.kind == RS_NODE_ACTION && qn.n_children > 0) {
                const size_t ti = size_t(qn.index);
                const int32_t *base = s->down_off_p[p][ti] == SIZE_MAX ? nullptr : s->d_shadow + (s->down_off_p[p][ti] - s->down_rowoff_p[p][ti]);
                if (!base || (stage_rows_of[qn.player] && stage_rows_of[qn.player] != base)) all = false;
                else {
                    stage.ch[qn.player] = stage.chp[qn.player] = int(s->down_stride_p[p][ti] / 4);
                    stage.off[size_t(q)] = int(s->down_rowoff_p[p][ti]);
                    stage_rows_of[qn.player] = base;
                    any = true;
                }
            }
            for (int k = 0; k < qn.n_children; ++k) {
                const int c = qn.children[k];
                const bool chance = nodes[c].kind == RS_NODE_PRIVATE_CHANCE || nodes[c].kind == RS_NODE_PUBLIC_CHANCE;
                if (!(round_mode && chance)) stack.push_back(c);
            }
        }
        dense_down_staged = all && any;
        if (!dense_down_staged) {
            stage = JitStage{};
            stage_rows_of[0] = stage_rows_of[1] = nullptr;
        }
    }
    JitSubtree js;
    jit_emit_subtree(nodes, id, p, has_own, leaf_buf, leaf_flags, t->dtype, s->params.mode & RS_UPD_ARITH_MASK,
                     s->params.opp_mode == RS_OPP_SAMPLE, s->deal_mode, use_lds, sparse, down, (s->params.mode & RS_UPD_PRUNE) != 0,
                     dense_down_staged ? 1 : (use_lds && s->knobs.lanes == kUnset) ? ((!sparse && s->deals.n_deals > kSmallDealBatch) ? 2 : 1) : ((id == first_root || nodes[id].round_idx == nodes[first_root].round_idx) ? jit_lanes : jit_lanes_below),
                     round_mode ? &fused_root : (lane_rounds ? &next_root : nullptr), js, s->knobs,
                     int(fan_root[id]), sparse && s->d_attr[nodes[id].round_idx] != nullptr, pos_rows, true, s->ordered, seg, rows, sigma_node.empty() ? nullptr : &sigma_node, s->deal_mode && handoff_root(id), &stage);
    const bool xfan = fan_root[id] == 1;
    const int fan_par = fan_root[id] ? nodes[id].parent : -1;   // the ENUM chance node whose work this kernel takes over
    // the kernel itself is compiled (or fetched from the caches) after BOTH traversers' plans are complete: every distinct source no cache holds goes to a helper process
    // (rs_solver.cpp, jit_get_kernels: hipRTC serialises compiles inside a process; a three-street deal solver needs several dozen kernels of a second of hipRTC each)
    const uint64_t fn = jit_source_key(js.source);
    auto bi = by_fn.find(fn);
    if (bi == by_fn.end()) {
        bi = by_fn.emplace(fn, int(plan.jit.size())).first;
        plan.jit.emplace_back();
        plan.jit.back().source = js.source;
        plan.jit.back().entry = js.entry;
        plan.jit.back().src_off[0] = js.src_struct;
        plan.jit.back().src_off[1] = js.src_entry;
        plan.jit.back().src_off[2] = js.src_body;
        plan.jit.back().stride = js.args_size;
        plan.jit.back().threads = js.threads;
        plan.jit.back().worklist = js.worklist;
        plan.jit.back().seg = seg;
        plan.jit.back().rows = rows;
        plan.jit.back().staged = js.stage_lds_bytes != 0;
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
    put_ptr(js.off_out, uptr(id));
    put_ptr(js.off_seed, s->d_seed());
    put_f32(js.off_reach_const, reach[id].cst);
    put_f32(js.off_scale, s->params.scale);
    // xfan: n_clusters % 4 == 0, exactly lanes / 4 vectors, no padding lanes
    const uint32_t n_vec = xfan ? uint32_t(size_t(s->n_boards[lane_round[id]]) * s->n_clusters / 4) : uint32_t(s->pitch[lane_round[id]] / size_t(js.lanes));
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
            if (rows) {   // the node's delta rows [2A][batch pitch], indexed by this job's list position (traverser nodes only)
                const bool own = an.player == p && an.n_children > 0;
                put_ptr(js.off_dreg + 8 * k, own ? s->d_drows + plan.drow_off[size_t(an.index)] : nullptr);
                put_ptr(js.off_dssm + 8 * k, nullptr);
            } else if (t->dtype != RS_I32) {   // the node's per-deal delta rows [2A][pitch] (traverser nodes only)
                put_ptr(js.off_dreg + 8 * k, (an.player == p && an.n_children > 0) ? plan.d_frows + plan.frow_off[size_t(an.index)] : nullptr);
                put_ptr(js.off_dssm + 8 * k, nullptr);
            } else {
                put_ptr(js.off_dreg + 8 * k, (char *)t->d_dregrets + t->cell_off[an.index] * 4);
                put_ptr(js.off_dssm + 8 * k, (char *)t->d_dssum + t->cell_off[an.index] * 4);
            }
            const bool shadowed = !s->shadow_off_p[p].empty() && s->shadow_off_p[p][an.index] != SIZE_MAX;   // f32 tables have no shadow at all
            put_ptr(js.off_shd + 8 * k, shadowed ? s->d_shadow + s->shadow_off_p[p][an.index] : nullptr);
            put_u32(js.off_sstride + 4 * k, shadowed ? s->shadow_stride_p[p][an.index] : 0u);
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
            put_u32(js.off_bbit + 4 * k, (down && size_t(b) < mask_bit_of_root.size() && mask_bit_of_root[size_t(b)] >= 0) ? uint32_t(mask_bit_of_root[size_t(b)]) : 0u);
        }
        // the reach-down kernel of a root whose next-round roots are compacted by mask writes the mask row (and the reach of live entries alone)
        put_ptr(js.off_bmask, (down && size_t(id) < mask_of_root.size()) ? mask_of_root[size_t(id)] : nullptr);
        if (sparse) {   // the subtree walks only its live deals
            const CompactJob &cj = plan.compact_jobs[size_t(sparse_slot[id])];
            put_ptr(js.off_list, cj.list + size_t(part) * cj.list_stride);
            put_ptr(js.off_count, cj.count + size_t(part) * cj.count_stride);
            // position-indexed rows: the entries of every list but the first root's (all of whose deals are live, with the constant root reach) carry their reach
            put_ptr(js.off_rlist, (pos_rows && id != first_root) ? plan.d_rlists + (cj.list - plan.d_lists) + size_t(part) * cj.list_stride : nullptr);
            put_ptr(js.off_plist, (pos_rows && id != first_root) ? plan.d_plists + (cj.list - plan.d_lists) + size_t(part) * cj.list_stride : nullptr);
            put_ptr(js.off_klist, rows ? plan.d_klists + (cj.list - plan.d_lists) + size_t(part) * cj.list_stride : nullptr);
        }
        put_ptr(js.off_hrow, handoff_root(id) ? plan.d_hrows + hrow_off[size_t(id)] : nullptr);
        put_u32(js.off_hpitch, uint32_t(s->pitch[0] + kRowStagger));
        put_ptr(js.off_rowp, stage_rows_of[0]);
        put_ptr(js.off_rowp + 8, stage_rows_of[1]);
        if (rows) {   // what k_row_sums adds up after the walks: per traverser node and array the A rows beside the job's key row (row by row where A tiles do not fit together)
            const uint32_t bp = uint32_t(s->pitch[0] + kRowStagger);   // the delta rows' own pitch (JArgs.rp of this form)
            const CompactJob *cj = sparse ? &plan.compact_jobs[size_t(sparse_slot[id])] : nullptr;
            for (size_t k = 0; k < js.node_ids.size(); ++k) {
                const rs_tree_node &an = nodes[js.node_ids[k]];
                if (an.player != p || an.n_children == 0) continue;
                const uint32_t A = uint32_t(an.n_children), ncl = t->nodes[size_t(an.index)].n_clusters, tpn = uint32_t(t->pitch[size_t(an.index)]);
                for (int arr = 0; arr < 2; ++arr) {
                    RowSumJob rj{};
                    rj.key = cj ? plan.d_klists + (cj->list - plan.d_lists) : s->deals.d_cluster[r][p];
                    rj.count = cj ? cj->count : nullptr;
                    rj.n_const = s->deals.n_deals;
                    rj.pitch = bp;
                    rj.tpitch = tpn;
                    rj.n_clusters = ncl;
                    const int32_t *src = s->d_drows + plan.drow_off[size_t(an.index)] + size_t(arr) * A * bp;
                    int32_t *dst = static_cast<int32_t *>(arr == 0 ? t->d_dregrets : t->d_dssum) + t->cell_off[an.index];
                    if (ncl > kRowSumMaxCells) {   // no LDS tile holds a row: its deltas go straight into the table (k_row_apply)
                        rj.rows = src;
                        rj.dst = static_cast<int32_t *>(arr == 0 ? t->regrets_ptr(an.index) : t->ssum_ptr(an.index));
                        rj.n_rows = A;
                        rj.direct = 1;
                        if (!s->kept_node.empty() && s->kept_node[size_t(an.index)]) {   // the node's kept records take the same additions (rs_solver.cpp setup_table_shadow)
                            const uint32_t half = A <= 2 ? 2u : (A <= 4 ? 4u : 8u);
                            rj.mirror = s->d_shadow + s->shadow_off_p[p][size_t(an.index)] + (arr ? half : 0u);
                            rj.mstride = s->shadow_stride_p[p][size_t(an.index)];
                            rj.primary = s->d_kept_primary;
                        }
                        plan.row_jobs.push_back(rj);
                    } else if (A * ncl <= kRowSumMaxCells) {
                        rj.rows = src;
                        rj.dst = dst;
                        rj.n_rows = A;
                        plan.row_jobs.push_back(rj);
                    } else
                        for (uint32_t a = 0; a < A; ++a) {
                            rj.rows = src + size_t(a) * bp;
                            rj.dst = dst + size_t(a) * tpn;
                            rj.n_rows = 1;
                            plan.row_jobs.push_back(rj);
                        }
                }
            }
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
        put_u32(js.off_rp, rows ? uint32_t(s->pitch[0] + kRowStagger) : rp);
        put_ptr(js.off_prune, (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr);
        put_ptr(js.off_attr, s->ordered ? s->d_arec_p[p] : (sparse ? s->d_attr[nodes[id].round_idx] : nullptr));
        if (sparse && s->d_attr[nodes[id].round_idx]) s->attr_used |= 1u << nodes[id].round_idx;
        if (use_lds) {
            std::vector<std::pair<size_t, size_t>> tiles;   // (ints, k)
            for (size_t k = 0; k < js.node_ids.size(); ++k) {
                const rs_tree_node &an = nodes[js.node_ids[k]];
                if (an.player == p && an.n_children > 0) tiles.emplace_back(size_t(2) * an.n_children * rp, k);
            }
            std::sort(tiles.begin(), tiles.end());
            const size_t limit = size_t(lds_limit) / 4;
            size_t resident = 0, n_res = 0;
            while (n_res < tiles.size()) {
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
        if (js.stage_lds_bytes) JL.lds_bytes = std::max(JL.lds_bytes, js.stage_lds_bytes);
    }
    JL.n_jobs += 1;
    JL.max_n_vec = std::max(JL.max_n_vec, n_vec);
    if (xfan) JL.bytes += bytes + lanes(id) * (4.0 * js.leaf_terms.size() + 4.0) + lanes(fan_par) * (reach[id].ptr ? 4.0 : 0.0);
    else JL.bytes += (bytes + lanes(id) * (4.0 * js.leaf_terms.size() + (reach[id].ptr ? 4.0 : 0.0) + 4.0)) / parts.first;
    }   // parts
    return RS_OK;
}

// pass 2 (after the arena exists): emit jobs and launches
int PlanBuilder::emit() {
    const size_t n = nodes.size();
    const rs_table *t = s->table;
    if (s->deal_mode && t->dtype != RS_I32) {   // per-deal delta rows of this traverser's nodes: [2A][deal pitch] floats each
        plan.frow_off.assign(t->nodes.size(), SIZE_MAX);
        size_t floats = 0;
        for (size_t i = 0; i < t->nodes.size(); ++i) {
            const rs_node_desc &d = t->nodes[i];
            if (d.n_actions == 0 || d.player != p) continue;
            plan.frow_off[i] = floats;
            floats += size_t(2) * d.n_actions * s->pitch[0];
        }
        hipError_t ef = hipMalloc((void **)&plan.d_frows, std::max<size_t>(floats, 64) * sizeof(float));
        if (ef == hipSuccess) ef = hipMemsetAsync(plan.d_frows, 0, std::max<size_t>(floats, 64) * sizeof(float), t->stream);
        if (ef != hipSuccess) return hip_fail(ef, "rs_solver_create_deals: per-deal delta rows");
        plan.aux_bytes += floats * sizeof(float);
    }
    if (s->rows) {   // delta rows: one buffer for both traversers' sweeps, offsets per traverser node of an eligible round
        plan.drow_off.assign(t->nodes.size(), SIZE_MAX);
        size_t ints = 0;
        for (size_t i = 0; i < t->nodes.size(); ++i) {
            const rs_node_desc &d = t->nodes[i];
            if (d.n_actions == 0 || d.player != p || !rows_round_ok(s, p, d.round_idx)) continue;
            plan.drow_off[i] = ints;
            ints += size_t(2) * d.n_actions * (s->pitch[0] + kRowStagger);
        }
        if (!s->d_drows) {
            const size_t need = std::max<size_t>(std::max(drows_ints(s, 0), drows_ints(s, 1)), 64);
            hipError_t ed = hipMalloc((void **)&s->d_drows, need * sizeof(int32_t));
            if (ed != hipSuccess) return hip_fail(ed, "rs_solver_create_deals: delta rows");
            s->other_bytes += need * sizeof(int32_t);
        }
    }
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
            std::map<uint64_t, int> by_fn;
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
    if (int rc = emit_deal_lists()) return rc;
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
            std::map<uint64_t, int> by_fn;
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
    if (int rc = emit_round_walks()) return rc;
    if (int rc = emit_row_sums()) return rc;
    if (!s->sharded) plan.split = plan.launches.size();   // deal batches: phase 0 = the sweep, phase 1 = the apply below
    if (int rc = emit_apply()) return rc;
    // value returned at node 0
    const ChildSrc root = child_source(0);
    if (root.kind != CH_BUF) return fail(RS_ERR_INVALID, "rs_solver_create: the root has no action node below it");
    plan.root_util = root.buf;
    plan.root_lanes = s->pitch[0];
    return RS_OK;
}


PlanBuilder *plan_builder_new(rs_solver *s, int traverser) { return new (std::nothrow) PlanBuilder(s, traverser); }
int plan_builder_layout(PlanBuilder *b) { return b ? b->build() : fail(RS_ERR_OOM, "rs_solver_create: out of host memory"); }
int plan_builder_emit(PlanBuilder *b) { return b ? b->emit() : fail(RS_ERR_OOM, "rs_solver_create: out of host memory"); }
void plan_builder_free(PlanBuilder *b) { delete b; }

}  // namespace rs
