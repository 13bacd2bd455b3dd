// rs_plan_deals.cpp -- the host-side traversal scheduler, part 1b: what only DEAL sweeps need (rs_solver_create_deals: lanes are sampled deals, mccfr() as the reference
// runs it, cfr.rs:299-479).  A deal has one run-out, so the tree is cut at the chance nodes into ONE generated subtree per betting round and root, each with a reach-down
// kernel and a walk that updates the table; below a sampled opponent node most deals are off their path, so every round subtree walks a compacted list of its LIVE deals;
// where the LDS delta tiles of a subtree's traverser nodes do not fit one workgroup together the cluster axis is cut into ranges (one list and one kernel job per range);
// list-walking kernels with tiles pull (job, trip) items from a device-built work list; after the sweep the i32 deltas are applied over the traverser's own nodes.
#include <algorithm>
#include <cstring>
#include <map>

#include "rs_plan_builder.hpp"

using namespace rs;

namespace rs {

uint32_t shadow_row_layout(const std::vector<uint32_t> &n_actions, bool wide, std::vector<uint32_t> &rec, std::vector<uint32_t> &off) {
    const size_t k = n_actions.size();
    rec.assign(k, 0);
    off.assign(k, 0);
    uint32_t at = 0;
    // records by size, the largest first (sizes are powers of two: 2 .. 16 ints): every record starts at a multiple of its size -- 16-byte loads find 16-byte aligned records,
    // and a row can be cut into parts at any multiple of 16 ints that is not inside a record of 16 (rs_jit.cpp, the staged rows of the traverser go through LDS in parts)
    for (uint32_t size = 16; size >= 2; size /= 2)
        for (size_t i = 0; i < k; ++i) {
            const uint32_t half = n_actions[i] <= 2 ? 2 : (n_actions[i] <= 4 ? 4 : 8);   // rs_device.hpp shadow_half<A>()
            const uint32_t r = wide ? 2 * half : half;
            if (r != size) continue;
            rec[i] = r;
            off[i] = at;
            at += r;
        }
    return k == 1 ? at : uint32_t(round_up(size_t(at), 4));   // a node on its own (dense walks, rs_solver.cpp setup_table_shadow): records back to back
}

bool rows_round_ok(const rs_solver *s, int p, int round) {
    if (!s->rows || !s->deal_mode || s->table->dtype != RS_I32) return false;
    if (round == s->first_round) return false;   // the dense walk of the first round keeps its resident tiles (river game 0.66 against 1.69 ms per batch with rows): rows are for the list walkers
    if (s->ordered && round == s->order_round) return false;   // ordered sweeps: the last round's lists come in runs of equal traverser cluster, summed by wave segments (seg_add)
    const rs_table *t = s->table;
    uint32_t k = 0;
    bool tiled = false;
    for (size_t i = 0; i < t->nodes.size(); ++i)
        if (t->nodes[i].round_idx == round && t->nodes[i].player == p && t->nodes[i].n_actions > 0) {
            k = std::max(k, t->nodes[i].n_clusters);
            tiled = tiled || t->tiled(int(i));
        }
    return k > 0 && (k <= kRowSumMaxCells || (s->direct_rows && !tiled));   // one row's LDS tile must fit -- or the rows go straight into the table's (plain) rows
}

bool rows_round_direct(const rs_solver *s, int p, int round) {
    if (!s->direct_rows || !rows_round_ok(s, p, round)) return false;
    const rs_table *t = s->table;
    uint32_t k = 0;
    for (size_t i = 0; i < t->nodes.size(); ++i)
        if (t->nodes[i].round_idx == round && t->nodes[i].player == p && t->nodes[i].n_actions > 0) k = std::max(k, t->nodes[i].n_clusters);
    return k > kRowSumMaxCells;
}

size_t drows_ints(const rs_solver *s, int p) {
    const rs_table *t = s->table;
    size_t ints = 0;
    for (size_t i = 0; i < t->nodes.size(); ++i) {
        const rs_node_desc &d = t->nodes[i];
        if (d.n_actions > 0 && d.player == p && rows_round_ok(s, p, d.round_idx)) ints += size_t(2) * d.n_actions * (s->pitch[0] + kRowStagger);
    }
    return ints;
}

PlanBuilder::Parts PlanBuilder::parts_of(int root) const {
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
    if (!want_parts || sum_a == 0 || 2 * sum_a * pitch <= limit || seg_root(root) || rows_root(root)) return Parts{1u, pitch, n_cl, pitch};
    // Partitioning costs list indirection (gathers instead of row loads, a bucketing pass).  When most tiles would be resident anyway --
    // 1 081 clusters miss the budget by 1 % and keep 5 of 7 -- it loses (measured 1.33 against 0.84 ms per batch): only partition when
    // fewer than half of the tile bytes fit.
    if (limit * 2 >= 2 * sum_a * pitch) return Parts{1u, pitch, n_cl, pitch};
    const uint32_t r = uint32_t(limit / (2 * sum_a)) / 64u * 64u;
    if (r < 64u || (n_cl + r - 1) / r > 64u) return Parts{1u, pitch, n_cl, pitch};   // k_compact_live handles up to 64 ranges
    return Parts{(n_cl + r - 1) / r, r, n_cl, pitch};
}

void PlanBuilder::mark_round_inside(int root, int id) {
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

void PlanBuilder::mark_round(int root) {
    fused_root[root] = 1;
    mark_round_inside(root, root);
}

// `segments`: the utility row of a root below a LISTED parent is addressed by the parent's list position, one segment per cluster range of the parent (pos_rows)
void PlanBuilder::layout_round(int root, size_t segments) {
    util_off[root] = alloc(lane_round[root], segments);
    const size_t below = (pos_rows && listed_root(root)) ? size_t(parts_of(root).first) : size_t(1);
    for (int b : bnd[size_t(root)]) {
        nan_slot[size_t(b)] = n_nan++;
        layout_round(b, below);
    }
}

int PlanBuilder::emit_deal_lists() {
    const size_t n = nodes.size();
    const rs_table *t = s->table;
    // ---- sparse deal sweeps: which subtree roots get a compacted list of their live deals ---------------------------------
    // mccfr() follows ONE opponent action per node (cfr.rs:467-476): below a sampled node most deals are off their path (NaN reach).  A
    // subtree kernel that walks every deal would compute nothing for them; instead the live ones are compacted and only they are walked.
    sparse_slot.assign(n, -1);
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
            list_elems += size_t(pr.first) * s->pitch[lane_round[ids[k]]] + kRowStagger;
            n_counts += pr.first;
        }
        hipError_t ea = hipMalloc((void **)&plan.d_lists, list_elems * sizeof(uint32_t));
        if (ea == hipSuccess && pos_rows) ea = hipMalloc((void **)&plan.d_rlists, list_elems * sizeof(float));
        if (ea == hipSuccess && pos_rows) ea = hipMalloc((void **)&plan.d_plists, list_elems * sizeof(uint32_t));
        bool any_rows = false;
        for (int id : ids) any_rows = any_rows || rows_root(id);
        if (ea == hipSuccess && any_rows) ea = hipMalloc((void **)&plan.d_klists, list_elems * sizeof(uint32_t));
        plan.aux_bytes += list_elems * sizeof(uint32_t) * (1 + (pos_rows ? 2 : 0) + (any_rows ? 1 : 0));
        plan.n_count_words = n_counts * kCountStride;
        if (ea == hipSuccess) ea = hipMalloc((void **)&plan.d_counts, n_counts * kCountStride * sizeof(uint32_t));
        if (ea == hipSuccess) ea = hipMemsetAsync(plan.d_counts, 0, n_counts * kCountStride * sizeof(uint32_t), t->stream);
        for (const Launch &L0 : plan.launches) plan.counts_zeroed_by_shadow = plan.counts_zeroed_by_shadow || L0.kind == L_SHADOW;   // pushed before any compaction (rs_plan.cpp)
        plan.counts_zeroed_by_shadow = plan.counts_zeroed_by_shadow && plan.n_count_words < (size_t(1) << 32);
        if (ea == hipSuccess) ea = hipMalloc((void **)&plan.d_compact_jobs, n_sparse * sizeof(CompactJob));
        if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: live-deal lists");
        plan.compact_jobs.resize(n_sparse);
        plan.count_off.assign(n_sparse + 1, 0);
        size_t at = 0, cat = 0;
        for (size_t k = 0; k < n_sparse; ++k) {
            const int id = ids[k];
            sparse_slot[id] = int(k);
            plan.compact_round.push_back(int(nodes[id].round_idx));
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
                cj.key = static_cast<const uint32_t *>(s->d_arec_p[p]) + 2 * nodes[id].round_idx + p;
                cj.key_stride = 8;
            }
            plan.count_off[k] = cat;
            at += size_t(n_parts[k]) * s->pitch[lane_round[id]] + kRowStagger;   // the lists of different roots are walked at the same positions at the same time
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
        // Liveness masks.  Where the sibling scan compacts a parent's next-round roots (big batches), the parent's reach-down kernel writes one mask word per entry -- a bit
        // per root it hands a reach to -- and the reach rows of those roots alone; the scan then reads 4 bytes per entry instead of a float per (entry, root), nine in ten
        // of them NaN (a sampled opponent sends a deal down one branch), and the kernel stops writing them.
        mask_of_root.assign(nodes.size(), nullptr);
        mask_bit_of_root.assign(nodes.size(), -1);
        const bool siblings_on = s->knobs.no_siblings == kUnset ? s->deals.n_deals > kSiblingsMinDeals : s->knobs.no_siblings == 0;
        if (scan_parent && siblings_on && !parent_root_.empty()) {
            std::map<int, std::vector<size_t>> children;   // parent root -> its listed next-round roots (positions in ids)
            for (size_t k = 0; k < n_sparse; ++k) {
                const int par = parent_root_[size_t(ids[k])];
                if (par >= 0) children[par].push_back(k);
            }
            size_t words = 0;
            std::vector<std::pair<int, size_t>> rows;   // (parent, offset)
            for (auto &kv : children) {
                const int par = kv.first;
                bool ok = kv.second.size() <= 32 && kv.second.size() == bnd[size_t(par)].size();   // every next-round root of the parent is listed, and has a bit
                const bool par_listed = sparse_slot[size_t(par)] >= 0;
                if (par_listed) ok = ok && plan.compact_jobs[size_t(sparse_slot[size_t(par)])].n_parts == 1 && pos_rows;
                for (size_t k : kv.second) {
                    const CompactJob &cj = plan.compact_jobs[k];
                    ok = ok && cj.n_parts == 1 && !cj.key && cj.reach && (cj.src_list ? (cj.src_parts == 1 && cj.pos_rows) : !par_listed);
                }
                if (!ok) continue;
                rows.emplace_back(par, words);
                words += (par_listed ? size_t(plan.compact_jobs[size_t(sparse_slot[size_t(par)])].list_stride) : s->pitch[lane_round[size_t(par)]]) + kRowStagger;
            }
            if (words) {
                if (hipMalloc((void **)&plan.d_bmask, words * sizeof(uint32_t)) != hipSuccess) return fail(RS_ERR_OOM, "rs_solver_create: liveness masks of the round subtrees");
                if (hipMemsetAsync(plan.d_bmask, 0, words * sizeof(uint32_t), t->stream) != hipSuccess) return fail(RS_ERR_HIP, "rs_solver_create: liveness masks of the round subtrees");
                plan.aux_bytes += words * sizeof(uint32_t);
                for (auto &pr : rows) {
                    mask_of_root[size_t(pr.first)] = plan.d_bmask + pr.second;
                    int bit = 0;
                    for (size_t k : children[pr.first]) {
                        plan.compact_jobs[k].mask = plan.d_bmask + pr.second;
                        plan.compact_jobs[k].bit = uint32_t(bit);
                        mask_bit_of_root[size_t(ids[k])] = bit++;
                    }
                }
            }
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
        // siblings (the roots below one parent: same source list, or the whole batch) share one scan of their source; a launch with cluster ranges keeps one job per root
        // (worth it when a source holds many tiles: three streets, 5 000-bucket files, 4 M deals per batch 7.44 -> 6.94 ms, 1 M 3.04 -> 2.92; at 256 K deals and below the fewer,
        // longer-lived workgroups lose a few percent -- RS_JIT_NO_SIBLINGS = 1 / 0 forces either)
        bool plain = s->knobs.no_siblings == kUnset ? s->deals.n_deals > kSiblingsMinDeals : s->knobs.no_siblings == 0;
        for (int k = first; k < first + count && plain; ++k) {
            const CompactJob &cj = plan.compact_jobs[size_t(k)];
            plain = cj.n_parts == 1 && !cj.key && cj.reach && (cj.src_list ? cj.src_parts == 1 : true);
        }
        if (plain) {
            L.first_group = int(plan.compact_groups.size());
            for (int k = first; k < first + count;) {
                const CompactJob &a = plan.compact_jobs[size_t(k)];
                int e = k + 1;
                while (e < first + count && e - k < 16 && plan.compact_jobs[size_t(e)].src_list == a.src_list && plan.compact_jobs[size_t(e)].src_count == a.src_count &&
                       plan.compact_jobs[size_t(e)].n_lanes == a.n_lanes && plan.compact_jobs[size_t(e)].pos_rows == a.pos_rows && plan.compact_jobs[size_t(e)].mask == a.mask)
                    ++e;
                plan.compact_groups.push_back(CompactGroup{uint32_t(k), uint32_t(e - k)});
                k = e;
            }
            L.n_groups = int(plan.compact_groups.size()) - L.first_group;
        }
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
        // hand-off rows: every root with next-round roots below it (i.e. with a reach-down kernel), unless its lists are cut into cluster ranges
        if (s->params.opp_mode == RS_OPP_SAMPLE && t->dtype == RS_I32) {
            hrow_off.assign(n, SIZE_MAX);
            size_t floats = 0;
            const size_t hp = s->pitch[0] + kRowStagger;
            for (size_t r = 0; r < roots_of_round.size(); ++r)
                for (int root : roots_of_round[r]) {
                    if (bnd[size_t(root)].empty() || parts_of(root).first > 1) continue;
                    size_t opp = 0;   // opponent nodes of the round subtree: an upper bound on what the kernels hand over (at most 10 nodes + the word of drawn actions)
                    std::vector<int> stack{root};
                    while (!stack.empty()) {
                        const int q = stack.back();
                        stack.pop_back();
                        const rs_tree_node &qn = nodes[size_t(q)];
                        if (qn.kind == RS_NODE_ACTION && qn.player != p && qn.n_children > 0) ++opp;
                        for (int k = 0; k < qn.n_children; ++k) {
                            const int c = qn.children[k];
                            if (nodes[size_t(c)].kind != RS_NODE_PRIVATE_CHANCE && nodes[size_t(c)].kind != RS_NODE_PUBLIC_CHANCE) stack.push_back(c);
                        }
                    }
                    if (!opp) continue;
                    hrow_off[size_t(root)] = floats;
                    floats += (std::min<size_t>(opp, 10) + 1) * hp;
                }
            if (floats) {
                if (hipMalloc((void **)&plan.d_hrows, floats * sizeof(float)) != hipSuccess) return fail(RS_ERR_OOM, "rs_solver_create: hand-off rows of the round subtrees");
                plan.aux_bytes += floats * sizeof(float);
            } else hrow_off.clear();
        }
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
            std::map<uint64_t, int> by_fn;
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
    if (!plan.compact_groups.empty()) {
        hipError_t eg = hipMalloc((void **)&plan.d_compact_groups, plan.compact_groups.size() * sizeof(CompactGroup));
        if (eg == hipSuccess) eg = hipMemcpy(plan.d_compact_groups, plan.compact_groups.data(), plan.compact_groups.size() * sizeof(CompactGroup), hipMemcpyHostToDevice);
        if (eg != hipSuccess) return hip_fail(eg, "rs_solver_create: sibling groups of the live-deal lists");
        plan.aux_bytes += plan.compact_groups.size() * sizeof(CompactGroup);
    }
    return RS_OK;
}

int PlanBuilder::emit_round_walks() {   // ---- round subtrees, bottom-up: last round first; the subtrees of one round are independent of each other
    // delta rows: the rows a round's walks stored are summed while the NEXT launches of the walk run (the round above: gather-bound walks beside a streaming pass), so the
    // summing launch of round r joins the group of round r - 1; the first round's comes last, on its own
    auto push_row_sums = [&](size_t first, size_t end, int group) {
        if (end <= first) return;
        Launch L;
        L.kind = L_ROWSUM;
        L.group = group;
        L.first_job = int(first);
        L.n_jobs = int(end - first);
        L.n_actions = plan.row_jobs[first].direct ? 1 : 0;   // a round's jobs are all of one kind (a player's nodes of a round share their cluster count)
        double entries = 0.0;
        for (size_t k = first; k < end; ++k) {
            const RowSumJob &j = plan.row_jobs[k];
            entries += double(j.n_rows + 1) * (j.count ? double(s->deals.n_deals) / 8.0 : double(j.n_const));   // a list holds a share of the batch (an estimate for the profiling hooks)
            if (!j.direct) plan.row_max_cells = std::max(plan.row_max_cells, j.n_rows * j.n_clusters);
        }
        L.bytes = entries * 4.0;
        plan.launches.push_back(L);
    };
    if (round_mode) {
        size_t pending = plan.row_jobs.size();   // row jobs [pending, size) wait for their summing launch
        for (size_t r = roots_of_round.size(); r-- > 0;) {
            std::map<uint64_t, int> by_fn;
            const size_t before = plan.row_jobs.size();
            for (int root : roots_of_round[r]) {
                if (sparse_slot[size_t(root)] < 0) plan.dense_roots[nodes[size_t(root)].round_idx] += 1;
                if (int rc = add_jit_job(root, false, sparse_slot, by_fn)) return rc;
            }
            const int group = ++next_group;
            push_row_sums(pending, before, group);   // the round below left rows: sum them beside this round's walks
            pending = before;
            for (auto &kv : by_fn) {
                Launch L;
                L.kind = L_TREE;
                L.group = group;
                L.first_job = kv.second;
                L.bytes = plan.jit[kv.second].bytes;
                plan.launches.push_back(L);
            }
        }
        push_row_sums(pending, plan.row_jobs.size(), 0);
    }
    return RS_OK;
}

int PlanBuilder::emit_row_sums() {   // delta rows: the job descriptors of the summing launches emit_round_walks placed
    if (plan.row_jobs.empty()) return RS_OK;
    hipError_t e = hipMalloc((void **)&plan.d_row_jobs, plan.row_jobs.size() * sizeof(RowSumJob));
    if (e == hipSuccess) e = hipMemcpy(plan.d_row_jobs, plan.row_jobs.data(), plan.row_jobs.size() * sizeof(RowSumJob), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "rs_solver_create_deals: row-sum jobs");
    plan.aux_bytes += plan.row_jobs.size() * sizeof(RowSumJob);
    return RS_OK;
}

int PlanBuilder::emit_apply() {
    const rs_table *t = s->table;
    if (s->deal_mode && t->dtype != RS_I32) {   // the ordered apply (float tables, binary32 or binary16): member lists per round of the traverser's cluster ids, then one job per traverser node
        std::vector<ApplyF32Job> jobs;
        const uint32_t n = s->deals.n_deals;
        size_t scratch = 0;
        for (int r = 0; r < s->n_rounds; ++r) {
            uint32_t k = 0;
            for (size_t i = 0; i < t->nodes.size(); ++i)
                if (t->nodes[i].round_idx == r && t->nodes[i].player == p && t->nodes[i].n_actions > 0) k = std::max(k, t->nodes[i].n_clusters);
            if (!k) continue;
            if (size_t(k) * 4 > 64 * 1024) return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: f32 tables take at most 16 384 clusters per node (the member lists' LDS histogram)");
            hipError_t e = hipMalloc((void **)&plan.d_member_start[r], (size_t(k) + 1) * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&plan.d_members[r], std::max<size_t>(n, 1) * 4);
            if (e != hipSuccess) return hip_fail(e, "rs_solver_create_deals: member lists");
            plan.aux_bytes += (size_t(k) + 1) * 4 + size_t(n) * 4;
            scratch = std::max(scratch, (member_list_tiles(n) + 1) * size_t(k));
        }
        if (hipMalloc((void **)&plan.d_member_scratch, std::max<size_t>(scratch, 1) * 4) != hipSuccess) return fail(RS_ERR_OOM, "rs_solver_create_deals: member-list scratch");
        plan.aux_bytes += scratch * 4;
        for (size_t i = 0; i < t->nodes.size(); ++i) {
            const rs_node_desc &d = t->nodes[i];
            if (d.n_actions == 0 || d.player != p) continue;
            ApplyF32Job j{};
            j.reg = t->regrets_ptr(int(i));
            j.ssm = t->ssum_ptr(int(i));
            j.rows = plan.d_frows + plan.frow_off[i];
            j.start = plan.d_member_start[d.round_idx];
            j.members = plan.d_members[d.round_idx];
            j.n_actions = d.n_actions;
            j.tpitch = uint32_t(t->pitch[i]);
            j.n_clusters = d.n_clusters;
            jobs.push_back(j);
            plan.f32_max_clusters = std::max(plan.f32_max_clusters, d.n_clusters);
        }
        if (!jobs.empty()) {
            hipError_t e = hipMalloc((void **)&plan.d_f32_jobs, jobs.size() * sizeof(ApplyF32Job));
            if (e == hipSuccess) e = hipMemcpy(plan.d_f32_jobs, jobs.data(), jobs.size() * sizeof(ApplyF32Job), hipMemcpyHostToDevice);
            if (e != hipSuccess) return hip_fail(e, "rs_solver_create_deals: f32 apply jobs");
            plan.n_f32_jobs = int(jobs.size());
        }
        Launch L;
        L.kind = L_APPLY;
        L.bytes = double(plan.n_f32_jobs) * 0.0;
        plan.launches.push_back(L);
        return RS_OK;
    }
    if (s->deal_mode) {   // table += delta, delta = 0: over the traverser's own nodes (nobody else's deltas were written: the other half of the delta arrays stays unread)
        std::vector<ApplyJob> aj;
        double cells = 0.0;
        for (size_t i = 0; i < t->nodes.size(); ++i) {
            const rs_node_desc &d = t->nodes[i];
            if (d.n_actions == 0 || d.player != p) continue;
            if (rows_round_direct(s, p, d.round_idx) && s->rows) continue;   // its deltas never touch the delta tables
            const size_t nc = size_t(d.n_actions) * t->pitch[i];
            if ((t->cell_off[i] % kVec) || (nc % kVec)) { aj.clear(); plan.apply_whole = true; break; }   // never with 64-lane padded pitches; the whole-table form is the fallback
            aj.push_back(ApplyJob{t->cell_off[i] / kVec, nc / kVec});
            plan.apply_max_vec = std::max(plan.apply_max_vec, nc / kVec);
            cells += double(nc);
        }
        if (!aj.empty()) {
            hipError_t ea = hipMalloc((void **)&plan.d_apply_jobs, aj.size() * sizeof(ApplyJob));
            if (ea == hipSuccess) ea = hipMemcpy(plan.d_apply_jobs, aj.data(), aj.size() * sizeof(ApplyJob), hipMemcpyHostToDevice);
            if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: apply jobs");
            plan.n_apply_jobs = int(aj.size());
            std::vector<size_t> off(aj.size());   // data-parallel sweeps sum these cells over the ranks, packed back to back
            size_t at = 0;
            for (size_t k = 0; k < aj.size(); ++k) {
                off[k] = at;
                at += aj[k].n_vec;
            }
            plan.pack_vec = at;
            ea = hipMalloc((void **)&plan.d_pack_off, aj.size() * sizeof(size_t));
            if (ea == hipSuccess) ea = hipMemcpy(plan.d_pack_off, off.data(), aj.size() * sizeof(size_t), hipMemcpyHostToDevice);
            if (ea != hipSuccess) return hip_fail(ea, "rs_solver_create: apply jobs");
        }
        Launch L;
        L.kind = L_APPLY;
        L.bytes = (plan.n_apply_jobs ? cells : double(t->n_cells)) * 32.0;
        plan.launches.push_back(L);
    }
    return RS_OK;
}

}  // namespace rs
