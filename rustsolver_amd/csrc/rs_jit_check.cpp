// rs_jit_check.cpp -- diagnostics (rustsolver_amd_diag.h): generate and COMPILE, without a GPU, every tree-specialised kernel form a solver could pick for a tree.
// The CPU test-suite runs these so that an emitter change that breaks a form nobody measures any more is caught before it reaches a GPU.
#include <algorithm>
#include <map>
#include <string>

#include "rs_plan.hpp"

using namespace rs;

extern "C" {

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
            }
        }
        for (size_t i = 0; i < n; ++i) {
            const rs_tree_node &nd = nodes[i];
            if (nd.kind != RS_NODE_ACTION || !closed[i] || nd.n_children == 0) continue;
            if (nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_ACTION && closed[nd.parent]) continue;   // not topmost
            for (int fan = 0; fan < 2; ++fan) {   // below a public chance node also the form that takes over the node's expand step
                if (fan && !(nd.parent >= 0 && nodes[nd.parent].kind == RS_NODE_PUBLIC_CHANCE)) continue;
                JitSubtree js;
                jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, dtype, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, false, false, false, false,
                                 (mode & RS_UPD_PRUNE) != 0, 4, nullptr, js, knobs, fan);
                if (seen.count(js.source)) continue;
                seen[js.source] = 1;
            }
        }
    }
    if (int rc = jit_compile_many(seen, knobs.dump != 0)) return rc;   // every distinct source, by helper processes side by side (rs_jit_cache.cpp)
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
    const std::vector<char> sigma_all(n, 1);
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
                JitSubtree js;   // opponent nodes read from strategy records (what a solver with table shadows emits); the work-list and list-position forms below keep the regrets path
                jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                 (mode & RS_UPD_PRUNE) != 0, lanes, &root, js, knobs, 0, sparse, false, false, false, false, false, &sigma_all, true);   // ... and the reach-down kernel hands its draws to the walk   // sparse forms fetch packed per-deal records (the separate gathers remain as the
                                                                                               // fallback for solvers whose leaves do not share one buffer: compiled by the GPU tests)
                if (down && js.boundary_roots.empty()) continue;   // a last-round subtree hands no reach on
                if (!seen.count(js.source)) {
                    seen[js.source] = 1;
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
                    }
                }
                if (f6 == 0 || f6 == 4) {   // deal sweeps on RS_F32 tables: the dense reach-down half and the dense walk that stores its deltas per deal (no prune on floats)
                    JitSubtree jf;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_F32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, false, false, down, false, lanes,
                                     &root, jf, knobs, 0, false);
                    if (!(down && jf.boundary_roots.empty()) && !seen.count(jf.source)) {
                        seen[jf.source] = 1;
                    }
                    if (f6 == 4 && lanes == 1 && int(i) == first) {   // ... and on binary16 tables (the same walks: gather_tbl converts; one kernel is enough to build the path)
                        JitSubtree jh;
                        jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_F16, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, false, false, down, false, lanes,
                                         &root, jh, knobs);
                        if (!seen.count(jh.source)) seen[jh.source] = 1;
                    }
                }
                if (lanes == 1 && sparse && !lds) {   // staged rows: the list walkers of large batches (delta-rows walk, reach-down kernel) read their records from rows staged in LDS
                    JitStage st;
                    st.off.assign(n, 0);
                    for (int q = 0; q < 2; ++q) {   // the round subtree's nodes of player q, in index order: one row (rs_plan.hpp shadow_row_layout)
                        std::vector<int> ids;
                        std::vector<uint32_t> acts, rec, off;
                        std::vector<int> stack{int(i)};
                        while (!stack.empty()) {
                            const int x = stack.back();
                            stack.pop_back();
                            const rs_tree_node &xn = nodes[size_t(x)];
                            if (xn.kind != RS_NODE_ACTION || xn.n_children == 0) continue;
                            if (xn.player == q) ids.push_back(x);
                            for (int k = 0; k < xn.n_children; ++k)
                                if (nodes[size_t(xn.children[k])].kind == RS_NODE_ACTION || nodes[size_t(xn.children[k])].kind == RS_NODE_TERMINAL) stack.push_back(xn.children[k]);
                        }
                        std::sort(ids.begin(), ids.end());
                        for (int x : ids) acts.push_back(uint32_t(nodes[size_t(x)].n_children));
                        if (ids.empty()) continue;
                        st.ch[q] = int(shadow_row_layout(acts, q == p, rec, off) / 4);
                        st.chp[q] = st.ch[q];
                        for (size_t m = 0; m < ids.size(); ++m) st.off[size_t(ids[m])] = int(off[m]);
                    }
                    JitSubtree jt;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, false, true, down,
                                     (mode & RS_UPD_PRUNE) != 0, 1, &root, jt, knobs, 0, true, true, false, false, false, !down, &sigma_all, true, &st);
                    if (!(down && jt.boundary_roots.empty()) && jt.staged && !seen.count(jt.source)) seen[jt.source] = 1;
                }
                if (f6 == 5) {   // delta rows (rs_kernel_forms.delta_rows): the list walk stores its deltas by position
                    JitSubtree jr;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, false, sparse, false,
                                     (mode & RS_UPD_PRUNE) != 0, lanes, &root, jr, knobs, 0, sparse, sparse, false, false, false, true, &sigma_all, true);
                    if (!seen.count(jr.source)) {
                        seen[jr.source] = 1;
                    }
                }
                if (sparse && lds && !down) {   // the work-list form every list-walking kernel with LDS tiles is launched in
                    JitSubtree jw;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                     (mode & RS_UPD_PRUNE) != 0, lanes, &root, jw, knobs, 0, sparse, true, true);
                    if (!seen.count(jw.source)) {
                        seen[jw.source] = 1;
                    }
                }
                if (sparse && !js.boundary_roots.empty()) {   // the forms that address the rows shared with the next round by list position (large batches)
                    JitSubtree jp;
                    jit_emit_subtree(nodes, int(i), p, has_own, leaf_buf, leaf_flags, RS_I32, mode & RS_UPD_ARITH_MASK, opp_mode == RS_OPP_SAMPLE, true, lds, sparse, down,
                                     (mode & RS_UPD_PRUNE) != 0, lanes, &root, jp, knobs, 0, sparse, true);
                    if (!seen.count(jp.source)) {
                        seen[jp.source] = 1;
                    }
                }
            }
        }
    }
    if (int rc = jit_compile_many(seen, knobs.dump != 0)) return rc;   // every distinct source, by helper processes side by side (rs_jit_cache.cpp)
    if (n_kernels) *n_kernels = int(seen.size());
    return RS_OK;
}

}  // extern "C"
