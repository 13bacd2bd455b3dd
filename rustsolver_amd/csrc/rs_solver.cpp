// rs_solver.cpp -- the host-side traversal scheduler, part 2: creating a solver around a plan (rs_plan.cpp builds it) and replaying it -- rs_iterate, optionally as one
// hipGraph; sharded sweeps and data-parallel deal batches in two phases around their collective; rs_train's schedule (cfr.rs:188-297).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "rs_plan.hpp"

using namespace rs;

namespace {

WorklistDesc worklist_desc(const JitLaunch &JL) { return WorklistDesc{JL.d_blob, JL.d_wl, uint32_t(JL.stride), JL.off_count, uint32_t(JL.n_jobs), JL.deals_per_trip}; }

// wl_done: the work list of this launch was built by its group's shared k_worklist launch (run_range)
int run_launch(rs_solver *s, const Plan &plan, const Launch &L, hipStream_t tree_stream = nullptr, bool wl_done = false) {
    rs_table *t = s->table;
    if (!tree_stream) tree_stream = t->stream;
    static const int prof_kind[] = {RS_K_REACH, RS_K_REACH, RS_K_CHANCE, RS_K_UPDATE, RS_K_NODE_UTIL, RS_K_CHANCE, RS_K_TREE, RS_K_CHANCE, RS_K_DISCOUNT};
    if (L.kind == L_APPLY && t->dtype != RS_I32 && s->deal_mode) {   // every cell's deltas summed in deal order: the traverser's deals listed per cluster, round by round, then the node jobs
        const int pl = &plan == &s->plan[1] ? 1 : 0;
        prof_begin(t, RS_K_DISCOUNT, L.bytes);
        hipError_t ef = hipSuccess;
        for (int r = 0; r < s->n_rounds && ef == hipSuccess; ++r) {
            if (!plan.d_members[r]) continue;
            uint32_t k = 0;
            for (size_t i = 0; i < t->nodes.size(); ++i)
                if (t->nodes[i].round_idx == r && t->nodes[i].player == pl && t->nodes[i].n_actions > 0) k = std::max(k, t->nodes[i].n_clusters);
            ef = launch_member_lists(s->deals.d_cluster[r][pl], s->deals.n_deals, int(k), plan.d_member_scratch, plan.d_member_scratch + member_list_tiles(s->deals.n_deals) * size_t(k),
                                     plan.d_member_start[r], plan.d_members[r], t->stream);
        }
        if (ef == hipSuccess) ef = launch_apply_f32_rows(plan.d_f32_jobs, plan.n_f32_jobs, plan.f32_max_clusters, uint32_t(s->pitch[0]), t->dtype, (s->params.mode & RS_UPD_RMPLUS) != 0, t->stream);
        prof_end(t);
        RS_HIP(ef, "k_apply_f32_rows");
        return RS_OK;
    }
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
        // the counters were zeroed by the launch that opened the sweep (L_SHADOW below) when there is one
        hipError_t ec = plan.counts_zeroed_by_shadow ? hipSuccess : hipMemsetAsync(plan.d_counts + c_lo * kCountStride, 0, (c_hi - c_lo) * kCountStride * sizeof(uint32_t), t->stream);
        if (ec == hipSuccess)
            ec = L.n_groups ? launch_compact_siblings(plan.d_compact_jobs, plan.d_compact_groups + L.first_group, L.n_groups, plan.compact_max_lanes, t->stream)
                            : launch_compact_live(plan.d_compact_jobs + L.first_job, L.n_jobs, plan.compact_max_lanes, t->stream);
        prof_end(t);
        RS_HIP(ec, "k_compact_live");
        return RS_OK;
    }
    if (L.kind == L_ORDER) {
        if (s->order_ahead) return RS_OK;   // the caller sorted this batch's records itself (solver_order_on)
        prof_begin(t, RS_K_REACH, L.bytes);
        hipError_t eo = launch_order(s->order_job[&plan == &s->plan[1] ? 1 : 0], t->stream);
        prof_end(t);
        RS_HIP(eo, "k_order");
        return RS_OK;
    }
    if (L.kind == L_ROWSUM) {
        prof_begin(t, RS_K_DISCOUNT, L.bytes);
        const uint32_t chunk = s->knobs.rows_chunk != kUnset && s->knobs.rows_chunk > 0 ? uint32_t(s->knobs.rows_chunk) : kRowSumChunk;
        // direct rows under a communicator: the additions are written out as items for the ranks to exchange (solver_exchange_deltas applies everybody's)
        // small batches: a summing job's fixed cost -- zeroing and scanning its 64 KB tile, 574 jobs for the turn of the three-street tree: 18 us whatever the batch -- exceeds
        // one atomic per non-zero delta into the delta table (k_row_apply with the job's delta-table rows as its destination): same sums, integer adds
        const bool few = !L.n_actions && s->deals.n_deals <= kRowApplyMaxDeals;
        hipError_t er = few ? launch_row_apply(plan.d_row_jobs + L.first_job, L.n_jobs, s->deals.n_deals, tree_stream) : L.n_actions ? (s->comm ? launch_rows_to_items(plan.d_row_jobs, L.first_job, L.n_jobs, s->deals.n_deals, s->d_items, s->d_item_count, s->item_cap, tree_stream)
                                               : launch_row_apply(plan.d_row_jobs + L.first_job, L.n_jobs, s->deals.n_deals, tree_stream))
                                    : launch_row_sums(plan.d_row_jobs + L.first_job, L.n_jobs, s->deals.n_deals, chunk, plan.row_max_cells, tree_stream);
        prof_end(t);
        RS_HIP(er, "k_row_sums");
        return RS_OK;
    }
    if (L.kind == L_PACK) {
        RS_HIP(launch_pack_attr(s->d_pack_jobs, s->n_pack_jobs, s->deals.n_deals, t->stream), "k_pack_attr");
        return RS_OK;
    }
    if (L.kind == L_SHADOW) {
        prof_begin(t, RS_K_STRATEGY, L.bytes);
        const int which = &plan == &s->plan[1] ? 1 : 0;   // the shadow holds what THIS traverser's sweep reads: regrets everywhere, strategy sums at its own nodes
        hipError_t es = launch_build_shadow(s->d_shadow_jobs + size_t(which) * s->n_shadow_jobs, s->n_shadow_jobs, s->shadow_max_clusters, t->stream, L.n_jobs ? s->d_seed_state : nullptr,
                                            plan.counts_zeroed_by_shadow ? plan.d_counts : nullptr, uint32_t(plan.n_count_words));
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
        blocks = std::min<size_t>(std::max<size_t>(blocks, 1), (JL.lds_bytes && !JL.staged) ? 256 * 4 : 256 * 16);   // LDS tiles: fewer, longer-lived workgroups
        if (JL.persistent) blocks = std::min<size_t>(blocks, size_t(s->n_cus));   // 224 VGPRs: one workgroup per CU is all that fits; more would only flush more
        if ((JL.seg || JL.rows) && JL.n_jobs > 1) blocks = std::min<size_t>(blocks, std::max<size_t>(64, size_t(s->n_cus) * 32 / size_t(JL.n_jobs) * 4));   // list walkers: a list holds a share of the batch
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
        if (!JL.members.empty()) {   // one launch for the kernels of a whole group (merge_small_groups)
            // The merged kernel keeps the registers of its largest member (two workgroups per CU), and a list holds a fraction of the batch: a grid sized for the whole batch
            // per job is 98 % workgroups that find nothing and leave, thirty-odd rounds of them (64 K deals, 73 river lists: 96 us, a quarter of the sweep).  Sized for four
            // times the average list of a round in which every deal is walked twice; the grid-stride loop takes what is longer.  The lists are uneven (the check-through
            // lines hold most of the deals): between 4 K and 96 K deals the long ones then make several trips one after the other in a sweep that is a chain of latencies --
            // sixteen times the average there (8 K deals 0.38 -> 0.36 ms per batch, 16 K 0.45 -> 0.40, 32 K 0.47 -> 0.44, 64 K 0.58 -> 0.53; 4 K 0.31 -> 0.34 and 128 K and beyond 0-6 %
            // slower: left at four).
            // (A flat grid -- every workgroup finds its (job, trip) from a prefix of the counts -- balances exactly and costs each workgroup the prefix: no better.)
            if (s->knobs.max_blocks == kUnset) {
                const size_t mult = (s->deals.n_deals > 4096 && s->deals.n_deals <= 98304) ? 32 : 8;
                blocks = std::min<size_t>(blocks, std::max<size_t>(4, (size_t(s->deals.n_deals) * mult / size_t(JL.n_jobs) + size_t(JL.threads) - 1) / size_t(JL.threads)));
            }
            MergedArgs A = JL.margs;
            void *mparams[] = {&A, &flags};
            e = hipModuleLaunchKernel(JL.fn, (unsigned)blocks, (unsigned)JL.n_jobs, 1, (unsigned)JL.threads, 1, 1, (unsigned)JL.lds_bytes, tree_stream, mparams, nullptr);
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
static int run_plan_inner(rs_solver *s, int p, int phase);
int run_plan(rs_solver *s, int p, int phase = -1) {
    rs_table *t = s->table;
    if (s->n_kept_jobs && s->kept_epoch != t->epoch) {   // the table was written by somebody else since the kept records were last in step with it: transpose them again
        RS_HIP(launch_build_shadow(s->d_kept_jobs, s->n_kept_jobs, s->kept_max_clusters, t->stream, nullptr), "k_build_shadow (kept records)");
        s->kept_epoch = t->epoch;
    }
    const int rc = run_plan_inner(s, p, phase);
    const bool in_step = s->n_kept_jobs && s->kept_epoch == t->epoch && rc == RS_OK;
    ++t->epoch;                                           // a sweep writes the table; its own kept records took the same additions
    if (in_step) s->kept_epoch = t->epoch;
    return rc;
}
static int run_plan_inner(rs_solver *s, int p, int phase) {
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
    if (table && table->dtype != RS_I32) {   // the float form (binary32 or binary16 tables, f32 arithmetic): dense walks, no atomics, every cell's f32 deltas added in deal order
                                             // (bit-identical to the oracle's sequential loop), ONE rounding to the table's type and the RM+ floor on the write-back
        if (!params || !params->fuse_subtrees) return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: float tables need fuse_subtrees = 1 (the generated kernels)");
        if (params->mode & RS_UPD_PRUNE) return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: RS_UPD_PRUNE needs an RS_I32 table (cfr.rs:352 compares i32 regrets)");
        if (params->chance_mode != RS_CHANCE_PASS) return fail(RS_ERR_UNSUPPORTED, "rs_solver_create_deals: a deal has one run-out: use RS_CHANCE_PASS (cfr.rs:306-313)");
        return solver_create_impl(table, tree, deals, leaves_p0, leaves_p1, params, out);
    }
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

void rs::solver_table_discounted(rs_solver *s, float d, uint64_t epoch_before) {
    if (!s || !s->table || !s->n_kept_jobs || s->kept_epoch != epoch_before) return;   // nothing kept, or already out of step (rebuilt at the next sweep)
    rs_table *t = s->table;
    // the kept region is whole groups of 64 ints, padding zero: its two halves as the two arrays of one discount sweep
    if (launch_discount(s->d_shadow, s->d_shadow + s->kept_ints / 2, s->kept_ints / 2, d, RS_I32, t->stream) == hipSuccess) s->kept_epoch = t->epoch;
}

bool rs::solver_is_primary(const rs_solver *s) { return s && s->table && s->primary; }

// ---- ordered sweeps whose records are sorted AHEAD of the sweep, by the caller, on a stream of its choice (rs_trainer.cpp) -------------------------------------------------
bool rs::solver_order_ahead(rs_solver *s, bool on, int (*before_sweep)(void *ctx, int traverser), void *ctx) {
    if (!s || !s->ordered || s->comm || s->sharded || s->plan[0].graph_exec || s->plan[1].graph_exec) return false;
    s->order_ahead = on;
    s->before_sweep = on ? before_sweep : nullptr;
    s->before_sweep_ctx = on ? ctx : nullptr;
    return true;
}
// traverser p's records from the given per-deal arrays (the layout of rs_deal_batch: cluster[round][player], one leaf row, prune flags or null), on `stream`
int rs::solver_order_on(rs_solver *s, int p, hipStream_t stream, const uint32_t *const cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS], const float *leaf, const uint8_t *prune) {
    if (!s || !s->table || !s->ordered || p < 0 || p > 1) return fail(RS_ERR_INVALID, "solver_order_on: not an ordered deal solver");
    OrderJob oj = s->order_job[p];
    oj.key = cluster[s->order_round][p];
    for (int r = 0; r < s->n_rounds; ++r)
        for (int pl = 0; pl < 2; ++pl) oj.cid[2 * r + pl] = cluster[r][pl];
    oj.leaf = leaf;
    oj.prune = (s->params.mode & RS_UPD_PRUNE) ? prune : nullptr;
    RS_HIP(hipSetDevice(s->table->device), "hipSetDevice");
    RS_HIP(launch_order(oj, stream), "k_order (ahead of the sweep)");
    return RS_OK;
}

int rs::solver_kept_primary(rs_solver *s, bool on) {
    if (!s || !s->table || !s->n_kept_jobs || s->primary == on) return RS_OK;
    rs_table *t = s->table;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    if (on) {
        for (rs_solver *o : t->solvers)
            if (o != s && o->primary) return RS_OK;   // somebody else's records are the working copy: this loop runs with the table's rows (both written)
        if (s->kept_epoch != t->epoch) {
            RS_HIP(launch_build_shadow(s->d_kept_jobs, s->n_kept_jobs, s->kept_max_clusters, t->stream, nullptr), "k_build_shadow (kept records)");
            s->kept_epoch = t->epoch;
        }
        const uint32_t one = 1;
        RS_HIP(hipMemcpyAsync(s->d_kept_primary, &one, sizeof(one), hipMemcpyHostToDevice, t->stream), "kept records: working copy on");
        RS_HIP(hipStreamSynchronize(t->stream), "kept records: working copy on");   // `one` is a stack variable
        s->primary = true;
        return RS_OK;
    }
    // off: the table's rows of the kept nodes from the records, which stay in step with them
    RS_HIP(launch_unbuild_shadow(s->d_kept_jobs, s->n_kept_jobs, s->kept_max_clusters, t->stream), "k_unbuild_shadow");
    RS_HIP(hipMemsetAsync(s->d_kept_primary, 0, sizeof(uint32_t), t->stream), "kept records: working copy off");
    s->primary = false;
    s->kept_epoch = ++t->epoch;
    return RS_OK;
}

int rs::solver_discount_primary(rs_solver *s, float d) {
    rs_table *t = s->table;
    RS_HIP(launch_discount_jobs(s->d_disc_jobs, s->n_disc_jobs, s->disc_max_vec, d, t->dtype, t->stream), "k_discount_jobs");
    RS_HIP(launch_discount(s->d_shadow, s->d_shadow + s->kept_ints / 2, s->kept_ints / 2, d, RS_I32, t->stream), "k_discount (kept records)");
    s->kept_epoch = ++t->epoch;
    return RS_OK;
}

void rs::solver_release_device(rs_solver *s) {
    if (!s || !s->table) return;   // already detached (its table was destroyed first)
    rs_table *t = s->table;
    (void)hipSetDevice(t->device);
    (void)hipStreamSynchronize(t->stream);
    if (s->primary) {   // the table's rows back from the records BEFORE anything they live in is freed (k_unbuild_shadow reads d_shadow through d_kept_jobs)
        (void)solver_kept_primary(s, false);
        (void)hipStreamSynchronize(t->stream);
    }
    for (int p = 0; p < 2; ++p) {
        Plan &pl = s->plan[p];
        if (pl.graph_exec) (void)hipGraphExecDestroy(pl.graph_exec);
        if (pl.graph) (void)hipGraphDestroy(pl.graph);
        if (pl.d_jobs) (void)hipFree(pl.d_jobs);
        if (pl.d_chance_jobs) (void)hipFree(pl.d_chance_jobs);
        if (pl.d_reach_nan) (void)hipFree(pl.d_reach_nan);
        if (pl.d_bmask) (void)hipFree(pl.d_bmask);
        if (pl.d_lists) (void)hipFree(pl.d_lists);
        if (pl.d_rlists) (void)hipFree(pl.d_rlists);
        if (pl.d_plists) (void)hipFree(pl.d_plists);
        if (pl.d_klists) (void)hipFree(pl.d_klists);
        if (pl.d_hrows) (void)hipFree(pl.d_hrows);
        if (pl.d_compact_groups) (void)hipFree(pl.d_compact_groups);
        if (pl.d_row_jobs) (void)hipFree(pl.d_row_jobs);
        if (pl.d_apply_jobs) (void)hipFree(pl.d_apply_jobs);
        if (pl.d_pack_off) (void)hipFree(pl.d_pack_off);
        if (pl.d_frows) (void)hipFree(pl.d_frows);
        if (pl.d_f32_jobs) (void)hipFree(pl.d_f32_jobs);
        if (pl.d_member_scratch) (void)hipFree(pl.d_member_scratch);
        for (int r = 0; r < RS_MAX_ROUNDS; ++r) {
            if (pl.d_member_start[r]) (void)hipFree(pl.d_member_start[r]);
            if (pl.d_members[r]) (void)hipFree(pl.d_members[r]);
        }
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
    if (s->d_kept_jobs) (void)hipFree(s->d_kept_jobs);
    if (s->d_kept_primary) (void)hipFree(s->d_kept_primary);
    if (s->d_disc_jobs) (void)hipFree(s->d_disc_jobs);
    s->d_kept_jobs = nullptr;
    s->d_kept_primary = nullptr;
    s->d_disc_jobs = nullptr;
    s->n_kept_jobs = 0;
    for (int r = 0; r < RS_MAX_ROUNDS; ++r) {
        if (s->d_attr[r] && s->d_attr[r] != s->d_arec) (void)hipFree(s->d_attr[r]);
        s->d_attr[r] = nullptr;
    }
    for (int tp = 0; tp < 2; ++tp) {
        if (s->d_arec_p[tp]) (void)hipFree(s->d_arec_p[tp]);
        s->d_arec_p[tp] = nullptr;
    }
    if (s->d_drows) (void)hipFree(s->d_drows);
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
    if (s->d_packed) (void)hipFree(s->d_packed);
    if (s->d_items) (void)hipFree(s->d_items);
    if (s->d_items_all) (void)hipFree(s->d_items_all);
    if (s->d_item_count) (void)hipFree(s->d_item_count);
    s->d_packed = nullptr;
    s->d_items = s->d_items_all = s->d_item_count = nullptr;
    s->d_arena = nullptr;
    s->d_seed_state = nullptr;
    t->solvers.erase(std::remove(t->solvers.begin(), t->solvers.end(), s), t->solvers.end());
    s->table = nullptr;
}

// deal sweeps through generated kernels: the AoS shadow of the table the sweep gathers from, rebuilt at the start of every sweep (k_build_shadow)
static int setup_table_shadow(rs_solver *s) {
    rs_table *table = s->table;
    const rs_tree *tree = &s->tree;
    const size_t n = tree->nodes.size();
    hipError_t e = hipSuccess;
    (void)n;
    (void)e;
    if (s->deal_mode && s->params.fuse_subtrees && table->dtype == RS_I32) {   // AoS shadow of every table node for the deal kernels' gathers (rs_device.hpp gather_rec)
        std::vector<ShadowJob> jobs;
        size_t ints = 0;
        // A record is worth transposing when the sweep reads it: sampled sweeps reach a node of a later round with probability ~ 1 / (round subtrees of that round), so a
        // node gets a shadow only while n_deals / roots * 8 >= its cells -- 64 K deals against 180 234 river clusters (lossless abstraction, 2 GB table) spent 0.9 ms per
        // sweep transposing records nobody read.  Without one the kernels gather the table's own rows (rs_device.hpp gather_node).  RS_JIT_SHADOW_ALL keeps every shadow.
        std::vector<size_t> round_roots(size_t(s->n_rounds) + 1, 0);
        int first_round = -1;
        {
            const int first = [&] { int c = 0; while (tree->nodes[size_t(c)].kind == RS_NODE_PRIVATE_CHANCE || tree->nodes[size_t(c)].kind == RS_NODE_PUBLIC_CHANCE) c = tree->nodes[size_t(c)].children[0]; return c; }();
            if (tree->nodes[size_t(first)].kind == RS_NODE_ACTION) first_round = int(tree->nodes[size_t(first)].round_idx);
            if (tree->nodes[size_t(first)].kind == RS_NODE_ACTION) round_roots[size_t(std::min<int>(tree->nodes[size_t(first)].round_idx, s->n_rounds))] += 1;
            for (size_t i = 1; i < n; ++i)
                if (tree->nodes[i].kind == RS_NODE_PUBLIC_CHANCE && tree->nodes[i].n_children > 0) {
                    const rs_tree_node &c = tree->nodes[size_t(tree->nodes[i].children[0])];
                    if (c.kind == RS_NODE_ACTION) round_roots[size_t(std::min<int>(c.round_idx, s->n_rounds))] += 1;
                }
        }
        const bool shadow_all = s->knobs.shadow_all != 0 || s->params.opp_mode != RS_OPP_SAMPLE;
        // Rows.  A deal addresses every node of one player inside a round subtree with the SAME cluster id (get_cluster depends on round and player only, cfr.rs:361-365), so the
        // records of those nodes sit side by side: one ROW per cluster, per (round subtree, player).  A walk's gathers at its own nodes then land in the row's two or three cache
        // lines instead of one line per node (a river subtree: 7 own nodes, 192 bytes; 7 opponent nodes, 96 bytes).  comp_root: the topmost action node reached from a node
        // without crossing a chance node.  Only where the rows get STAGED (the list walkers of a sampled sweep's later rounds): a dense walk -- the first round, every deal
        // at every node at about the same time -- is better off with one array per node, whose few KB stay in L1 while all the CU's waves are at that node (the as-coded river
        // game, 4 M deals per batch: 0.64 ms with node arrays, 0.77 ms with rows).
        const bool staged_rows = s->params.opp_mode == RS_OPP_SAMPLE && !s->knobs.no_stage;
        std::vector<int> comp_root(n, -1), tree_of(table->nodes.size(), -1);
        for (size_t id = 0; id < n; ++id) {
            const rs_tree_node &nd = tree->nodes[id];
            if (nd.kind != RS_NODE_ACTION || nd.n_children <= 0 || nd.index < 0) continue;
            int q = int(id);
            while (tree->nodes[size_t(q)].parent >= 0 && tree->nodes[size_t(tree->nodes[size_t(q)].parent)].kind == RS_NODE_ACTION) q = tree->nodes[size_t(q)].parent;
            comp_root[id] = q;
            if (size_t(nd.index) < tree_of.size()) tree_of[size_t(nd.index)] = int(id);
        }
        // Kept records.  Where a round's delta rows go straight into the table (rows_round_direct: more clusters than a summing tile holds -- the lossless abstractions of
        // solve_three_street, 180 234 river clusters, 2 GB of table), transposing the table once per sweep costs more than the whole sweep, and without a shadow the walks
        // gather the table's own rows: ~105 four-byte gathers per walked deal and subtree, a sector each (1.2 of 1.7 ms per 64 K-deal batch).  Those nodes KEEP their records
        // between sweeps instead: k_row_apply adds every delta to the record as well as to the table row (integer additions: the two stay equal), rs_discount sweeps the
        // records too, and any other write to the table (rs_table.epoch) has them rebuilt at the start of the next sweep.  One set of WIDE records serves both traversers'
        // sweeps (the opponent's role reads the regrets half and matches them itself: a kept record cannot hold a strategy).
        s->kept_node.assign(table->nodes.size(), 0);
        std::vector<ShadowJob> kept_jobs;
        for (size_t i = 0; i < table->nodes.size(); ++i) {
            const rs_node_desc &d = table->nodes[i];
            s->kept_node[i] = !s->knobs.no_kept && d.n_actions > 0 && !shadow_all && !table->tiled(int(i)) && s->rows && tree_of[i] >= 0 && rows_round_direct(s, int(d.player), int(d.round_idx));
        }
        {   // the records are a second copy of those nodes (1.3x their table rows): only while a quarter of the free memory holds them
            size_t need = 0, free_b = 0, total_b = 0;
            for (size_t i = 0; i < table->nodes.size(); ++i)
                if (s->kept_node[i]) need += size_t(table->nodes[i].n_clusters) * 2 * (table->nodes[i].n_actions <= 2 ? 2 : (table->nodes[i].n_actions <= 4 ? 4 : 8)) * sizeof(int32_t);
            if (need && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b / 4)) s->kept_node.assign(table->nodes.size(), 0);
        }
        for (int tp = 0; tp < 2; ++tp) {   // traverser tp's sweep: 2 * half ints per record at its own nodes, half at the opponent's
            s->shadow_off_p[tp].assign(table->nodes.size(), SIZE_MAX);
            s->shadow_stride_p[tp].assign(table->nodes.size(), 0);
            s->shadow_rec_p[tp].assign(table->nodes.size(), 0);
            s->shadow_rowoff_p[tp].assign(table->nodes.size(), 0);
        }
        {
            std::map<std::pair<int, int>, std::vector<size_t>> groups;
            for (size_t i = 0; i < table->nodes.size(); ++i)
                if (s->kept_node[i]) groups[{staged_rows ? comp_root[size_t(tree_of[i])] : -1 - int(i), int(table->nodes[i].player)}].push_back(i);
            for (auto &kv : groups) {
                std::vector<size_t> &mem = kv.second;
                std::vector<uint32_t> acts, recs, offs;
                uint32_t n_cl = 0;
                for (size_t i : mem) {
                    acts.push_back(table->nodes[i].n_actions);
                    n_cl = std::max(n_cl, table->nodes[i].n_clusters);
                }
                const uint32_t row = shadow_row_layout(acts, true, recs, offs);
                if (size_t(n_cl) * row >= (size_t(1) << 32)) {   // 32-bit record addressing: these stay without a shadow
                    for (size_t i : mem) s->kept_node[i] = 0;
                    continue;
                }
                for (size_t m = 0; m < mem.size(); ++m) {
                    const size_t i = mem[m];
                    const rs_node_desc &d = table->nodes[i];
                    for (int tp = 0; tp < 2; ++tp) {
                        s->shadow_off_p[tp][i] = ints + offs[m];
                        s->shadow_stride_p[tp][i] = row;
                        s->shadow_rec_p[tp][i] = recs[m];
                        s->shadow_rowoff_p[tp][i] = offs[m];
                    }
                    ShadowJob j{};
                    j.regrets = static_cast<const int32_t *>(table->regrets_ptr(int(i)));
                    j.ssum = static_cast<const int32_t *>(table->ssum_ptr(int(i)));
                    j.pitch = uint32_t(table->pitch[i]);
                    j.n_clusters = d.n_clusters;
                    j.n_actions = d.n_actions;
                    j.half = d.n_actions <= 2 ? 2 : (d.n_actions <= 4 ? 4 : 8);
                    j.stride = recs[m];
                    j.row_stride = row;
                    j.sigma = 0;
                    j.dst = reinterpret_cast<int32_t *>(ints + offs[m]);   // an offset for now
                    kept_jobs.push_back(j);
                    s->kept_max_clusters = std::max(s->kept_max_clusters, d.n_clusters);
                }
                ints += round_up(size_t(n_cl) * row, 64);
            }
            s->kept_ints = ints;
            s->n_kept_jobs = int(kept_jobs.size());
        }
        for (int tp = 0; tp < 2; ++tp) {
            std::map<std::pair<int, int>, std::vector<size_t>> groups;   // (component root, player) -> its table nodes, in ActionNode.index order
            for (size_t i = 0; i < table->nodes.size(); ++i) {
                const rs_node_desc &d = table->nodes[i];
                if (d.n_actions == 0 || s->kept_node[i]) continue;
                const size_t roots = std::max<size_t>(1, round_roots[size_t(std::min<int>(d.round_idx, s->n_rounds))]);
                if (!shadow_all && !table->tiled(int(i)) && size_t(s->deals.n_deals) * 8 < size_t(d.n_clusters) * d.n_actions * roots) continue;   // no shadow: J.shd = nullptr
                groups[{tree_of[i] >= 0 && staged_rows && int(d.round_idx) != first_round ? comp_root[size_t(tree_of[i])] : -1 - int(i), int(d.player)}].push_back(i);
            }
            for (auto &kv : groups) {
                std::vector<size_t> &mem = kv.second;
                std::vector<uint32_t> acts, recs, offs;
                uint32_t n_cl = 0;
                for (size_t i : mem) {
                    acts.push_back(table->nodes[i].n_actions);
                    n_cl = std::max(n_cl, table->nodes[i].n_clusters);
                }
                const bool wide = kv.first.second == tp;
                if (size_t(n_cl) * shadow_row_layout(acts, true, recs, offs) >= (size_t(1) << 32)) continue;   // the kernels address a record with 32-bit arithmetic: no shadow (either sweep)
                const uint32_t row = shadow_row_layout(acts, wide, recs, offs);
                for (size_t m = 0; m < mem.size(); ++m) {
                    const size_t i = mem[m];
                    const rs_node_desc &d = table->nodes[i];
                    const uint32_t half = d.n_actions <= 2 ? 2 : (d.n_actions <= 4 ? 4 : 8);
                    s->shadow_off_p[tp][i] = ints + offs[m];
                    s->shadow_stride_p[tp][i] = row;
                    s->shadow_rec_p[tp][i] = recs[m];
                    s->shadow_rowoff_p[tp][i] = offs[m];
                    ShadowJob j{};
                    j.regrets = static_cast<const int32_t *>(table->regrets_ptr(int(i)));
                    j.ssum = static_cast<const int32_t *>(table->ssum_ptr(int(i)));
                    j.pitch = uint32_t(table->pitch[i]);
                    j.n_clusters = d.n_clusters;
                    j.n_actions = d.n_actions;
                    j.half = half;
                    j.stride = recs[m];
                    j.row_stride = row;
                    j.sigma = (recs[m] == half && d.player != tp) ? 1u : 0u;   // PlanBuilder::sigma_node says the same to the emitter
                    j.dst = reinterpret_cast<int32_t *>(ints + offs[m]);   // an offset for now: the buffer does not exist yet
                    jobs.push_back(j);
                    s->shadow_max_clusters = std::max(s->shadow_max_clusters, d.n_clusters);
                }
                ints += round_up(size_t(n_cl) * row, 64);
            }
            // the first round once more, as rows for the dense reach-down kernel (rs_plan.hpp down_off_p)
            s->down_off_p[tp].assign(table->nodes.size(), SIZE_MAX);
            s->down_stride_p[tp].assign(table->nodes.size(), 0);
            s->down_rowoff_p[tp].assign(table->nodes.size(), 0);
            if (staged_rows && first_round >= 0 && s->n_rounds > 1 && (s->deals.n_deals > kDownRowsMinDeals || s->knobs.ordered == 1)) {   // (RS_JIT_ORDERED = 1: the tests' way to
                                                                                                                                       // give a small batch the big batches' forms)
                std::map<std::pair<int, int>, std::vector<size_t>> first;
                for (size_t i = 0; i < table->nodes.size(); ++i) {
                    const rs_node_desc &d = table->nodes[i];
                    if (d.n_actions == 0 || int(d.round_idx) != first_round || tree_of[i] < 0 || s->shadow_off_p[tp][i] == SIZE_MAX || table->tiled(int(i))) continue;
                    first[{comp_root[size_t(tree_of[i])], int(d.player)}].push_back(i);
                }
                for (auto &kv : first) {
                    std::vector<size_t> &mem = kv.second;
                    std::vector<uint32_t> acts, recs, offs;
                    uint32_t n_cl = 0;
                    for (size_t i : mem) {
                        acts.push_back(table->nodes[i].n_actions);
                        n_cl = std::max(n_cl, table->nodes[i].n_clusters);
                    }
                    const uint32_t row = uint32_t(round_up(size_t(shadow_row_layout(acts, false, recs, offs)), 4));   // narrow records for both roles: a reach-down kernel reads regrets (its own
                                                                                                                        // nodes' explored[] bits) or strategies, never strategy sums
                    if (row > uint32_t(kStageMaxChunks) * 4 || size_t(n_cl) * row >= (size_t(1) << 32)) continue;
                    for (size_t m = 0; m < mem.size(); ++m) {
                        const size_t i = mem[m];
                        const rs_node_desc &d = table->nodes[i];
                        s->down_off_p[tp][i] = ints + offs[m];
                        s->down_stride_p[tp][i] = row;
                        s->down_rowoff_p[tp][i] = offs[m];
                        ShadowJob j{};
                        j.regrets = static_cast<const int32_t *>(table->regrets_ptr(int(i)));
                        j.ssum = static_cast<const int32_t *>(table->ssum_ptr(int(i)));
                        j.pitch = uint32_t(table->pitch[i]);
                        j.n_clusters = d.n_clusters;
                        j.n_actions = d.n_actions;
                        j.half = d.n_actions <= 2 ? 2 : (d.n_actions <= 4 ? 4 : 8);
                        j.stride = recs[m];
                        j.row_stride = row;
                        j.sigma = d.player != tp ? 1u : 0u;
                        j.dst = reinterpret_cast<int32_t *>(ints + offs[m]);
                        jobs.push_back(j);
                    }
                    ints += round_up(size_t(n_cl) * row, 64);
                }
            }
            if (tp == 0) s->n_shadow_jobs = int(jobs.size());
        }
        s->other_bytes += std::max<size_t>(ints * 4, 256) + std::max<size_t>(jobs.size() * sizeof(ShadowJob), 256) + kept_jobs.size() * sizeof(ShadowJob);
        e = hipMalloc((void **)&s->d_shadow, std::max<size_t>(ints * 4, 256));
        if (e == hipSuccess) e = hipMemsetAsync(s->d_shadow, 0, std::max<size_t>(ints * 4, 256), table->stream);
        for (ShadowJob &j : jobs) j.dst = s->d_shadow + reinterpret_cast<size_t>(j.dst);
        for (ShadowJob &j : kept_jobs) j.dst = s->d_shadow + reinterpret_cast<size_t>(j.dst);
        if (e == hipSuccess && !kept_jobs.empty()) e = hipMalloc((void **)&s->d_kept_jobs, kept_jobs.size() * sizeof(ShadowJob));
        if (e == hipSuccess && !kept_jobs.empty()) e = hipMemcpy(s->d_kept_jobs, kept_jobs.data(), kept_jobs.size() * sizeof(ShadowJob), hipMemcpyHostToDevice);
        s->kept_epoch = ~uint64_t(0);
        if (e == hipSuccess && !kept_jobs.empty()) {   // the flag k_row_apply reads, and the table without the kept nodes as stretches of consecutive nodes (solver_discount_primary)
            e = hipMalloc((void **)&s->d_kept_primary, 256);
            if (e == hipSuccess) e = hipMemsetAsync(s->d_kept_primary, 0, 256, table->stream);
            std::vector<DiscountJob> runs;
            const size_t es = 4;   // kept records exist on RS_I32 tables only
            for (size_t i = 0; i < table->nodes.size();) {
                if (s->kept_node[i] || table->nodes[i].n_actions == 0) {
                    ++i;
                    continue;
                }
                size_t j = i, cells = 0;
                while (j < table->nodes.size() && !s->kept_node[j]) {
                    cells += size_t(table->nodes[j].n_actions) * table->pitch[j];
                    ++j;
                }
                runs.push_back(DiscountJob{static_cast<char *>(table->d_regrets) + table->cell_off[i] * es, static_cast<char *>(table->d_ssum) + table->cell_off[i] * es, cells / kVec});
                s->disc_max_vec = std::max(s->disc_max_vec, cells / kVec);
                i = j;
            }
            s->n_disc_jobs = int(runs.size());
            if (e == hipSuccess && !runs.empty()) e = hipMalloc((void **)&s->d_disc_jobs, runs.size() * sizeof(DiscountJob));
            if (e == hipSuccess && !runs.empty()) e = hipMemcpy(s->d_disc_jobs, runs.data(), runs.size() * sizeof(DiscountJob), hipMemcpyHostToDevice);
        }
        if (jobs.size() != size_t(2) * size_t(s->n_shadow_jobs)) return fail(RS_ERR_INVALID, "rs_solver_create: the two sweeps' shadows hold different node sets");
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_shadow_jobs, std::max<size_t>(jobs.size() * sizeof(ShadowJob), 256));
        if (e == hipSuccess && !jobs.empty()) e = hipMemcpy(s->d_shadow_jobs, jobs.data(), jobs.size() * sizeof(ShadowJob), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            return hip_fail(e, "rs_solver_create: table shadow");
        }
    }
    return RS_OK;
}

// delta rows (rs_kernel_forms.delta_rows): the list walkers of a deal sweep store their deltas by list position and one streaming pass per round sums them
// (rs_plan_deals.cpp rows_round_ok says for which rounds).  Measured on one MI355X, three streets, 5 000-bucket files (profiles/r03_deals.md): 4 M deals per batch 8.34 -> 7.47 ms,
// 1 M 3.28 -> 2.99; at 256 K deals the three cards it was measured on disagree (0.95-1.1x): the engine's choice beyond 512 K deals per batch
static void choose_delta_rows(rs_solver *s) {
    rs_table *table = s->table;
    const rs_tree *tree = &s->tree;
    {
        const bool can = s->deal_mode && s->params.fuse_subtrees && table->dtype == RS_I32;
        {
            int c = 0;
            while (tree->nodes[size_t(c)].kind == RS_NODE_PRIVATE_CHANCE || tree->nodes[size_t(c)].kind == RS_NODE_PUBLIC_CHANCE) c = tree->nodes[size_t(c)].children[0];
            s->first_round = tree->nodes[size_t(c)].kind == RS_NODE_ACTION ? int(tree->nodes[size_t(c)].round_idx) : 0;
        }
        s->direct_rows = can && (s->knobs.direct_rows == kUnset ? true : s->knobs.direct_rows != 0);
        const bool engine = s->params.opp_mode == RS_OPP_SAMPLE && s->deals.n_deals > kRowsMinDeals;   // only sampled sweeps have list walkers
        s->rows = can && (s->knobs.rows == kUnset ? engine : s->knobs.rows != 0);
        if (s->rows && s->knobs.rows == kUnset) {   // the rows cost 2 x actions x 4 B per traverser node and deal (26 GB at 4 M deals on the 706-node tree): only while a quarter of the free memory holds them
            size_t free_b = 0, total_b = 0;
            const size_t need = std::max(drows_ints(s, 0), drows_ints(s, 1)) * sizeof(int32_t);
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b / 4) s->rows = false;
        }
    }
}

// sparse (live-deal list) sweeps: pack the per-deal inputs of every round when ALL showdown / all-in leaves of both traversers share one buffer (the trainer's
// d_sign; otherwise the kernels keep their separate gathers).  RS_JIT_NO_PACK turns it off (A/B knob)
static int setup_deal_records(rs_solver *s) {
    rs_table *table = s->table;
    const rs_tree *tree = &s->tree;
    const size_t n = tree->nodes.size();
    hipError_t e = hipSuccess;
    (void)n;
    (void)e;
    if (s->deal_mode && s->params.fuse_subtrees && s->params.opp_mode == RS_OPP_SAMPLE && table->dtype == RS_I32) {
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
            const bool round_mode = fr.kind == RS_NODE_ACTION && fr.n_children > 0;
            s->order_round = s->n_rounds - 1;
            uint32_t bins[2] = {0, 0};
            for (size_t i = 0; i < table->nodes.size(); ++i)
                if (table->nodes[i].round_idx == s->order_round && table->nodes[i].n_actions > 0) bins[table->nodes[i].player] = std::max(bins[table->nodes[i].player], table->nodes[i].n_clusters);
            const bool fits = bins[0] >= 1 && bins[1] >= 1 && bins[0] <= kOrderMaxBins && bins[1] <= kOrderMaxBins && s->deals.d_cluster[s->order_round][0] &&
                              s->deals.d_cluster[s->order_round][1] && s->deals.n_deals < (1u << 31) && table->dtype == RS_I32;
            // Measured (round 3, one MI355X, profiles/r03_deals_ab.md): the segment-summing river kernels are 1.3-1.4x faster than the tile kernels (three streets, 5 000-bucket files,
            // 4 M deals per batch: 20.7 + 16.1 ms against 29.6 + 19.7 ms over seven batches), but the sort and the records cost 0.15 ms per sweep: 8.36 against 8.56 ms per batch there,
            // 3.57 against 3.40 at 1 M deals, and on the river game 0.84 against 0.69 -- a wash at best, so the form is opt-in (rs_kernel_forms.deal_order = RS_FORM_ON)
            // Round 4 (profiles/r04_deals.md): with the list walkers staging their deals' rows in LDS the runs of an ordered sweep share the traverser's row, the last round sums
            // its deltas along the runs (no delta rows, no summing pass for it), and the form wins on three streets beyond half a million deals per batch: 6.5 -> 6.0 ms at 4 M
            const bool engine = s->params.opp_mode == RS_OPP_SAMPLE && s->n_rounds > 1 && s->deals.n_deals > kOrderMinDeals;
            s->ordered = round_mode && fits && (s->knobs.ordered == kUnset ? engine : s->knobs.ordered != 0);
            if (s->ordered) {
                const size_t pitch = round_up(s->deals.n_deals, kLanePad);
                const uint32_t n = s->deals.n_deals;
                const uint32_t n_chunks = uint32_t(std::min<size_t>(size_t(s->n_cus) * 2, (size_t(n) + kOrderThreads - 1) / kOrderThreads));
                const uint32_t chunk = uint32_t(round_up((size_t(n) + n_chunks - 1) / n_chunks, kOrderThreads));
                const uint32_t max_bins = std::max(bins[0], bins[1]);
                s->other_bytes += 2 * pitch * 32 + size_t(4) * max_bins * sizeof(uint32_t);
                for (int tp = 0; tp < 2 && e == hipSuccess; ++tp) {
                    e = hipMalloc(&s->d_arec_p[tp], pitch * 32);
                    if (e == hipSuccess) e = hipMemsetAsync(s->d_arec_p[tp], 0, pitch * 32, table->stream);
                }
                s->d_arec = s->d_arec_p[0];
                if (e == hipSuccess) e = hipMalloc((void **)&s->d_order_tot, size_t(4) * max_bins * sizeof(uint32_t));
                if (e != hipSuccess) {
                    return hip_fail(e, "rs_solver_create: ordered deal records");
                }
                for (int tp = 0; tp < 2; ++tp) {
                    OrderJob &oj = s->order_job[tp];
                    oj = OrderJob{};
                    oj.key = s->deals.d_cluster[s->order_round][tp];
                    for (int r = 0; r < s->n_rounds; ++r)
                        for (int pl = 0; pl < 2; ++pl) oj.cid[2 * r + pl] = s->deals.d_cluster[r][pl];
                    oj.leaf = leaf;
                    oj.prune = (s->params.mode & RS_UPD_PRUNE) ? s->deals.d_prune : nullptr;
                    oj.tot = s->d_order_tot + size_t(2) * max_bins * tp;
                    oj.cursor = oj.tot + bins[tp];
                    oj.arec = s->d_arec_p[tp];
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
                return hip_fail(e, "rs_solver_create: packed deal inputs");
            }
            s->n_pack_jobs = int(jobs.size());
        }
    }
    return RS_OK;
}

// Small deal batches: a sweep is a chain of ~40 dependent launches of 10-25 us each (one wave's walk), and every hand-over between the groups of independent round subtrees --
// six kernels of different shapes on three streams -- costs another ~10 us of cross-queue signalling (kernel trace of a 4 K-deal batch: 7 such gaps per sweep, a quarter of the
// batch).  Below kMergeMaxDeals the kernels of a group are compiled into ONE kernel whose entry point dispatches on blockIdx.y (jit_merge_sources) and the whole sweep runs on
// the table's stream: no fork, no join.  The members keep their argument blobs; nothing about a subtree's code changes, so neither do its bits.
constexpr uint32_t kMergeMaxDeals = 393216;   // 256 K deals: 1.20 against 1.28 ms per batch, 512 K: 1.73 either way (three streets, 5 000-bucket files)
static int merge_small_groups(rs_solver *s) {
    if (!s->deal_mode || s->deals.n_deals > kMergeMaxDeals || s->knobs.no_merge) return RS_OK;
    for (int p = 0; p < 2; ++p) {
        Plan &plan = s->plan[p];
        std::vector<Launch> out;
        size_t removed_before_split = 0;
        for (size_t i = 0; i < plan.launches.size();) {
            size_t j = i + 1;
            if (plan.launches[i].group > 0)
                while (j < plan.launches.size() && plan.launches[j].group == plan.launches[i].group) ++j;
            std::vector<size_t> trees;
            bool ok = true;
            for (size_t k = i; k < j; ++k) {
                const Launch &L = plan.launches[k];
                if (L.kind != L_TREE) continue;
                const JitLaunch &JL = plan.jit[size_t(L.first_job)];
                ok = ok && JL.threads == 256 && !JL.worklist && !JL.persistent && (JL.staged || JL.lds_bytes == 0) && JL.members.empty() && !JL.absorbed;
                trees.push_back(k);
            }
            std::string merged;
            if (ok && trees.size() >= 2 && trees.size() <= size_t(kMergeMax)) {
                std::vector<const std::string *> srcs;
                std::vector<const size_t *> offs;
                for (size_t k : trees) {
                    srcs.push_back(&plan.jit[size_t(plan.launches[k].first_job)].source);
                    offs.push_back(plan.jit[size_t(plan.launches[k].first_job)].src_off);
                }
                merged = jit_merge_sources(srcs, offs, "rs_tree_p" + std::to_string(p) + "_deals_merged" + std::to_string(trees.size()));
            }
            if (merged.empty()) {   // nothing to merge (one subtree kernel, or kernels with LDS tiles): the group's launches one after the other all the same -- at these sizes a
                                    // fork and a join cost more than two short kernels side by side save
                for (size_t k = i; k < j; ++k) {
                    out.push_back(plan.launches[k]);
                    if (j - i <= 2) out.back().group = 0;   // (the tile kernels of the smallest batches, one launch per shape, stay side by side)
                }
                i = j;
                continue;
            }
            JitLaunch M;
            M.source = std::move(merged);
            M.entry = "rs_tree_p" + std::to_string(p) + "_deals_merged" + std::to_string(trees.size());
            M.threads = 256;
            M.staged = true;   // its LDS is a staging area (or nothing): the grid of the forms without tiles
            for (size_t k : trees) {
                JitLaunch &JL = plan.jit[size_t(plan.launches[k].first_job)];
                M.members.push_back(plan.launches[k].first_job);
                M.lds_bytes = std::max(M.lds_bytes, JL.lds_bytes);
                M.max_n_vec = std::max(M.max_n_vec, JL.max_n_vec);
                M.n_jobs += JL.n_jobs;
                M.bytes += JL.bytes;
                M.seg = M.seg || JL.seg;
                M.rows = M.rows || JL.rows;
                JL.absorbed = true;
            }
            const int mi = int(plan.jit.size());
            plan.jit.push_back(std::move(M));
            bool placed = false;
            for (size_t k = i; k < j; ++k) {   // the group's other launches (the summing pass of the round below) stay, in order, on the same stream; the trees become one launch
                Launch L = plan.launches[k];
                L.group = 0;
                if (L.kind == L_TREE) {
                    if (placed) {
                        if (k < plan.split) ++removed_before_split;
                        continue;
                    }
                    L.first_job = mi;
                    L.bytes = plan.jit[size_t(mi)].bytes;
                    placed = true;
                }
                out.push_back(L);
            }
            i = j;
        }
        plan.launches.swap(out);
        plan.split -= std::min(plan.split, removed_before_split);
    }
    return RS_OK;
}

// Only the list-walking kernels read the packed records: a round whose subtrees all walk the whole batch (the first round; every round of a one-round game) needs none
static int trim_packed_records(rs_solver *s) {
    hipError_t e = hipSuccess;
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
                return hip_fail(e, "rs_solver_create: packed deal inputs");
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
    return RS_OK;
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
    // streams for the independent round subtrees of a deal sweep
    if (hipDeviceGetAttribute(&s->lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, table->device) != hipSuccess || s->lds_limit < 1024) s->lds_limit = 64 * 1024;
    if (hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, table->device) != hipSuccess || s->n_cus < 1) s->n_cus = 256;
    if (s->deal_mode && !s->knobs.no_overlap && s->params.fuse_subtrees) {
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
    choose_delta_rows(s);
    if (int rc2 = setup_deal_records(s)) {
        rs_solver_destroy(s);
        return rc2;
    }
    if (int rc2 = setup_table_shadow(s)) {   // after the two above: which nodes KEEP their records depends on where the delta rows go
        rs_solver_destroy(s);
        return rc2;
    }
    struct Builders {   // one plan builder per traverser; freed on every way out
        PlanBuilder *b[2] = {nullptr, nullptr};
        ~Builders() {
            plan_builder_free(b[0]);
            plan_builder_free(b[1]);
        }
    } pb;
    pb.b[0] = plan_builder_new(s, 0);
    pb.b[1] = plan_builder_new(s, 1);
    if ((rc = plan_builder_layout(pb.b[0])) != RS_OK || (rc = plan_builder_layout(pb.b[1])) != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    if (int rc2 = trim_packed_records(s)) {
        rs_solver_destroy(s);
        return rc2;
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
    if ((rc = plan_builder_emit(pb.b[0])) != RS_OK || (rc = plan_builder_emit(pb.b[1])) != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    if ((rc = merge_small_groups(s)) != RS_OK) {
        rs_solver_destroy(s);
        return rc;
    }
    {   // the generated kernels of both plans: compiled together (concurrently where no cache has them), then bound to their launches
        std::vector<JitRequest> reqs;
        for (int p = 0; p < 2; ++p)
            for (JitLaunch &JL : s->plan[p].jit)
                if (!JL.absorbed) reqs.push_back(JitRequest{&JL.source, &JL.entry, nullptr});
        if (!reqs.empty() && (rc = jit_get_kernels(reqs, table->device, s->knobs.dump != 0)) != RS_OK) {
            rs_solver_destroy(s);
            return rc;
        }
        size_t k = 0;
        for (int p = 0; p < 2; ++p)
            for (JitLaunch &JL : s->plan[p].jit) {
                if (!JL.absorbed) JL.fn = reqs[k++].fn;
                JL.source = std::string();
            }
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
            if (JL.blob.empty()) continue;   // a merged launch: its members hold the blobs
            if ((e = hipMalloc((void **)&JL.d_blob, JL.blob.size())) != hipSuccess ||
                (e = hipMemcpyAsync(JL.d_blob, JL.blob.data(), JL.blob.size(), hipMemcpyHostToDevice, table->stream)) != hipSuccess) {
                rc = hip_fail(e, "rs_solver_create: tree-kernel argument upload");
                rs_solver_destroy(s);
                return rc;
            }
        }
        for (JitLaunch &JL : pl.jit) {   // merged launches: where every member's jobs start on the grid's y axis, and its blob
            unsigned at = 0;
            for (size_t k = 0; k < JL.members.size(); ++k) {
                const JitLaunch &m = pl.jit[size_t(JL.members[k])];
                JL.margs.blob[k] = m.d_blob;
                JL.margs.first[k] = at;
                at += unsigned(m.n_jobs);
            }
            for (size_t k = JL.members.size(); k <= size_t(kMergeMax); ++k) JL.margs.first[k] = at;
            if (!JL.members.empty() && at <= unsigned(kMergeCountJobs) && pl.d_counts) {   // where every job's list counter lies (the merged kernel's early exit)
                unsigned row = 0;
                uint32_t per_block = 0;
                bool same = true;
                for (int mk : JL.members) {
                    const JitLaunch &m = pl.jit[size_t(mk)];
                    same = same && (per_block == 0 || per_block == m.deals_per_trip);
                    per_block = m.deals_per_trip;
                    for (int j = 0; j < m.n_jobs; ++j, ++row) {
                        const uint32_t *cnt = nullptr;
                        std::memcpy(&cnt, m.blob.data() + size_t(j) * m.stride + m.off_count, sizeof(cnt));
                        JL.margs.cidx[row] = (cnt && cnt >= pl.d_counts) ? uint32_t(cnt - pl.d_counts) : 0xffffffffu;
                    }
                }
                if (same && per_block) {
                    JL.margs.per_block = per_block;
                    JL.margs.counts = pl.d_counts;
                }
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
        RS_HIP(launch_unpermute_f32(s->plan[traverser].root_util, s->d_arec_p[traverser], d_root_util, s->deals.n_deals, s->table->stream), "rs_iterate: root util by deal id");
        return RS_OK;
    }
    if (d_root_util)
        RS_HIP(hipMemcpyAsync(d_root_util, s->plan[traverser].root_util, s->plan[traverser].root_lanes * sizeof(float),
                              hipMemcpyDeviceToDevice, s->table->stream),
               "rs_iterate: root util copy");
    return RS_OK;
}

// Data-parallel deal batches, between sweep and apply: every rank must end up adding the deltas of the UNION batch.
//  * rounds that sum through the delta tables (LDS tiles, k_row_sums): the traverser's delta cells -- the ranges the apply pass runs over, nobody else's hold anything --
//    packed back to back, ONE in-place ncclInt32 all-reduce, unpacked (round 4 reduced both whole delta arrays: twice the bytes, and every cell of the direct rounds on top);
//  * rounds whose delta rows go straight into the table (more clusters than a summing tile holds: direct rows): the walks' rows as 12-byte (job, row, cluster, delta) items,
//    counts and items all-gathered, every rank's items added to table and kept records (k_apply_items) -- a 2 GB lossless table exchanges what its 64 K deals touched, not
//    itself, and keeps direct rows and kept records under a communicator.
// Integer adds commute: N ranks x n deals = one GPU with N x n deals, bit for bit.  Two small device-to-host reads (the counts) per sweep.
static int solver_exchange_deltas(rs_solver *s, int p) {
    rs_table *t = s->table;
    Plan &plan = s->plan[p];
    const int world = comm_world(s->comm);
    s->dp_bytes_last = 0;
    if (plan.apply_whole || !plan.n_apply_jobs) {
        if (plan.apply_whole) {
            if (int rc = rs_comm_allreduce_deltas(s->comm, t)) return rc;
            s->dp_bytes_last += uint64_t(t->n_cells) * 8;
        }
    } else {
        RS_HIP(launch_pack_cells(t->d_dregrets, t->d_dssum, plan.d_apply_jobs, plan.d_pack_off, plan.n_apply_jobs, plan.apply_max_vec, s->d_packed, plan.pack_vec, false, t->stream), "k_pack_cells");
        if (int rc = comm_allreduce_i32(s->comm, t, s->d_packed, plan.pack_vec * 8)) return rc;
        RS_HIP(launch_pack_cells(t->d_dregrets, t->d_dssum, plan.d_apply_jobs, plan.d_pack_off, plan.n_apply_jobs, plan.apply_max_vec, s->d_packed, plan.pack_vec, true, t->stream), "k_pack_cells");
        s->dp_bytes_last += uint64_t(plan.pack_vec) * 32;
    }
    if (!s->d_items) return RS_OK;   // no round of this solver adds rows straight into the table
    bool any = false;
    for (const Launch &L : plan.launches) any = any || (L.kind == L_ROWSUM && L.n_actions);
    if (!any) return RS_OK;
    std::vector<uint32_t> counts(size_t(world) + 1);
    for (int attempt = 0;; ++attempt) {
        if (int rc = comm_allgather_u32(s->comm, t, s->d_item_count, s->d_item_count + 1, 1)) return rc;
        RS_HIP(hipMemcpyAsync(counts.data(), s->d_item_count, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream), "deal delta items: counts");
        RS_HIP(hipStreamSynchronize(t->stream), "deal delta items: counts");
        uint32_t most = 0;
        for (int r = 0; r < world; ++r) most = std::max(most, counts[size_t(r) + 1]);
        if (most <= s->item_cap) break;
        if (attempt) return fail(RS_ERR_HIP, "deal delta items: the count changed between two passes over the same rows");
        // some rank's rows hold more items than the buffers: every rank grows (the counts are everybody's) and writes its items again -- the rows are still there
        (void)hipFree(s->d_items);
        (void)hipFree(s->d_items_all);
        s->d_items = s->d_items_all = nullptr;
        const size_t sized = std::max<size_t>(size_t(world), s->items_world);
        s->other_bytes -= size_t(s->item_cap) * 12 * (sized + 1);
        s->item_cap = uint32_t(std::min<uint64_t>(uint64_t(most) + most / 8 + 1024, 0xfffffff0ull / 3));
        if (most > s->item_cap) return fail(RS_ERR_OOM, "deal delta items: more than 2^32 / 3 items in one sweep");
        RS_HIP(hipMalloc((void **)&s->d_items, size_t(s->item_cap) * 12), "deal delta items");
        RS_HIP(hipMalloc((void **)&s->d_items_all, size_t(s->item_cap) * 12 * sized), "deal delta items");
        s->other_bytes += size_t(s->item_cap) * 12 * (sized + 1);
        RS_HIP(hipMemsetAsync(s->d_item_count, 0, sizeof(uint32_t), t->stream), "deal delta items");
        for (const Launch &L : plan.launches)
            if (L.kind == L_ROWSUM && L.n_actions)
                RS_HIP(launch_rows_to_items(plan.d_row_jobs, L.first_job, L.n_jobs, s->deals.n_deals, s->d_items, s->d_item_count, s->item_cap, t->stream), "k_rows_to_items");
    }
    uint32_t most = 0;
    for (int r = 0; r < world; ++r) most = std::max(most, counts[size_t(r) + 1]);
    if (most == 0) return RS_OK;
    if (int rc = comm_allgather_u32(s->comm, t, s->d_items, s->d_items_all, size_t(most) * 3)) return rc;   // every rank sends `most` items' worth: the tail beyond its count is never read
    for (int r = 0; r < world; ++r)
        RS_HIP(launch_apply_items(plan.d_row_jobs, s->d_items_all + size_t(r) * most * 3, counts[size_t(r) + 1], t->stream), "k_apply_items");
    s->dp_bytes_last += uint64_t(most) * 12 * uint64_t(world) + 4 * uint64_t(world);
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
    if (s->deal_mode && s->comm && s->table->dtype != RS_I32)
        return fail(RS_ERR_UNSUPPORTED, "rs_iterate: data-parallel deal batches all-reduce i32 deltas; a deal solver on a float table runs on one GPU");
    if (s->order_ahead && s->comm) return fail(RS_ERR_UNSUPPORTED, "rs_iterate: this solver's deal records are sorted ahead by its trainer, which runs on one GPU");
    if (s->before_sweep)
        if (int rc = s->before_sweep(s->before_sweep_ctx, traverser)) return rc;
    if (s->deal_mode && s->comm) {   // data-parallel deal batches: sweep, exchange the deltas of the ranks, apply the union
        if (s->d_item_count) RS_HIP(hipMemsetAsync(s->d_item_count, 0, sizeof(uint32_t), s->table->stream), "deal delta items");
        if (int rc = run_plan(s, traverser, 0)) return rc;
        if (int rc = solver_exchange_deltas(s, traverser)) return rc;
        s->dp_bytes_total += s->dp_bytes_last;
        s->dp_sweeps += 1;
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
    if (!s->sharded && s->deal_mode && s->rows && s->direct_rows)   // phase 0 must leave the table untouched (the host sums the delta tables between the phases)
        for (int r = 0; r < s->n_rounds; ++r)
            if (rows_round_direct(s, traverser, r))
                return fail(RS_ERR_UNSUPPORTED, "rs_iterate_phase: this solver adds the delta rows of its large rounds straight into the table during the walks, so the delta tables "
                                                "a host would exchange between the phases are incomplete: create it with rs_kernel_forms.direct_rows = RS_FORM_OFF");
    RS_HIP(hipSetDevice(s->table->device), "hipSetDevice");
    if (s->before_sweep && phase == 0)
        if (int rc = s->before_sweep(s->before_sweep_ctx, traverser)) return rc;
    if (int rc = run_plan(s, traverser, phase)) return rc;
    return phase == 1 ? copy_root(s, traverser, d_root_util) : RS_OK;
}

int rs_solver_attach_comm(rs_solver *s, rs_comm *comm) {
    if (!s) return fail(RS_ERR_INVALID, "rs_solver_attach_comm: solver is NULL");
    if (comm && s->order_ahead) return fail(RS_ERR_UNSUPPORTED, "rs_solver_attach_comm: this solver's deal records are sorted ahead by its trainer (one GPU); attach the communicator through rs_deal_trainer_attach_comm before the first batch");
    if (comm && s->deal_mode && s->table && s->table->dtype == RS_I32) {   // the buffers of solver_exchange_deltas
        rs_table *t = s->table;
        RS_HIP(hipSetDevice(t->device), "hipSetDevice");
        const size_t pack_vec = std::max(s->plan[0].pack_vec, s->plan[1].pack_vec);
        if (pack_vec && !s->d_packed) {
            RS_HIP(hipMalloc((void **)&s->d_packed, pack_vec * 32), "rs_solver_attach_comm: packed delta cells");
            s->other_bytes += pack_vec * 32;
        }
        bool direct = false;
        for (int p = 0; p < 2; ++p)
            for (const Launch &L : s->plan[p].launches) direct = direct || (L.kind == L_ROWSUM && L.n_actions);
        const size_t world = size_t(comm_world(comm));
        if (direct && s->items_world && world > s->items_world) {   // a communicator of more ranks than the buffers were sized for: again, bigger
            RS_HIP(hipStreamSynchronize(t->stream), "rs_solver_attach_comm");
            (void)hipFree(s->d_item_count);
            (void)hipFree(s->d_items);
            (void)hipFree(s->d_items_all);
            s->other_bytes -= size_t(s->item_cap) * 12 * (size_t(s->items_world) + 1);
            s->d_item_count = s->d_items = s->d_items_all = nullptr;
        }
        if (direct && !s->d_item_count) RS_HIP(hipMalloc((void **)&s->d_item_count, (world + 1) * sizeof(uint32_t) + 256), "rs_solver_attach_comm: item counts");
        if (direct) s->items_world = uint32_t(std::max<size_t>(world, s->items_world));
        if (direct && !s->d_items) {
            s->item_cap = uint32_t(std::min<uint64_t>(uint64_t(s->deals.n_deals) * 16 + 4096, 0xfffffff0ull / 3));   // grown when a sweep writes more (solver_exchange_deltas)
            RS_HIP(hipMalloc((void **)&s->d_items, size_t(s->item_cap) * 12), "rs_solver_attach_comm: delta items");
            RS_HIP(hipMalloc((void **)&s->d_items_all, size_t(s->item_cap) * 12 * size_t(s->items_world)), "rs_solver_attach_comm: delta items");
            s->other_bytes += size_t(s->item_cap) * 12 * (size_t(s->items_world) + 1);
        }
    }
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

int rs_solver_training_loop(rs_solver *s, int on) {
    if (!s) return fail(RS_ERR_INVALID, "rs_solver_training_loop: solver is NULL");
    if (!s->table) return fail(RS_ERR_INVALID, "rs_solver_training_loop: the solver's table has been destroyed");
    return solver_kept_primary(s, on != 0);
}

int rs_train(rs_solver *s, uint64_t iterations, uint64_t discount_interval, uint64_t discount_cap) {
    if (!s) return fail(RS_ERR_INVALID, "rs_train: solver is NULL");
    if (!s->table) return fail(RS_ERR_INVALID, "rs_train: the solver's table has been destroyed");
    if (discount_interval == 0) return fail(RS_ERR_INVALID, "rs_train: discount_interval must be > 0");
    uint64_t t = 0, threshold = discount_interval;
    if (iterations >= kKeptPrimaryMinTrips)
        if (int rc = solver_kept_primary(s, true)) return rc;   // kept shadow records (if any) are the working copy until the loop is over
    int rc = RS_OK;
    while (t < iterations && rc == RS_OK) {        // cfr.rs:207
        for (int player = 0; player < 2 && rc == RS_OK; ++player)  // cfr.rs:216-224
            rc = rs_iterate(s, player, nullptr);
        if (rc != RS_OK) break;
        t += 1;                                     // cfr.rs:226
        if (t > discount_cap) continue;             // cfr.rs:240-242
        if (t > threshold) {                        // cfr.rs:243
            rc = rs_discount(s->table, rs_discount_factor(t, discount_interval));  // cfr.rs:248-261
            threshold = t + discount_interval;      // cfr.rs:262
        }
    }
    const int rc_off = solver_kept_primary(s, false);
    return rc != RS_OK ? rc : rc_off;
}

size_t rs_solver_workspace_bytes(const rs_solver *s) { return s ? s->arena_bytes + s->plan[0].aux_bytes + s->plan[1].aux_bytes + s->other_bytes : 0; }

int rs_solver_exchange_bytes(const rs_solver *s, uint64_t *bytes, uint64_t *sweeps) {
    if (!s || !bytes || !sweeps) return fail(RS_ERR_INVALID, "rs_solver_exchange_bytes: NULL argument");
    *bytes = s->dp_bytes_total;
    *sweeps = s->dp_sweeps;
    return RS_OK;
}
int rs_solver_forms(const rs_solver *s) { return s ? ((s->ordered ? 1 : 0) | (s->rows ? 2 : 0)) : RS_ERR_INVALID; }
int rs_solver_walk_counts(rs_solver *s, int traverser, uint64_t *out) {
    if (!s || !s->table || !out || traverser < 0 || traverser > 1) return fail(RS_ERR_INVALID, "rs_solver_walk_counts: bad argument");
    if (!s->deal_mode) return fail(RS_ERR_INVALID, "rs_solver_walk_counts: not a deal-batch solver");
    const Plan &plan = s->plan[traverser];
    for (int r = 0; r < RS_MAX_ROUNDS; ++r) out[r] = uint64_t(plan.dense_roots[r]) * s->deals.n_deals;
    if (plan.n_count_words) {
        std::vector<uint32_t> host(plan.n_count_words);
        RS_HIP(hipSetDevice(s->table->device), "hipSetDevice");
        RS_HIP(hipStreamSynchronize(s->table->stream), "rs_solver_walk_counts: sync");
        RS_HIP(hipMemcpy(host.data(), plan.d_counts, plan.n_count_words * sizeof(uint32_t), hipMemcpyDeviceToHost), "rs_solver_walk_counts: counters");
        for (size_t k = 0; k < plan.compact_round.size(); ++k)
            for (size_t c = plan.count_off[k]; c < plan.count_off[k + 1]; ++c) out[plan.compact_round[k]] += host[c * kCountStride];
    }
    return RS_OK;
}

int rs_solver_n_launches(const rs_solver *s, int traverser) {
    if (!s || traverser < 0 || traverser > 1) return RS_ERR_INVALID;
    return int(s->plan[traverser].launches.size());
}


}  // extern "C"
