/*
 * rs_oracle_mt.c -- threaded driver around the oracle's per-lane traversal, used ONLY as the timed
 * CPU baseline in bench.py (`cpu_baseline`) and by tests that need a faster checker.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see rs_oracle.h).
 *
 * The reference runs N_THREADS = 8 Hogwild workers (cfr.rs:195-229).  Lanes never share table cells
 * in the lane model, so splitting the root lanes across threads is race-free and gives results
 * identical to the single-threaded sweep.
 */
#include "rs_oracle.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct {
    const orc_ctx *ctx;
    int player;
    size_t lo, hi;
    float *root_util;
} mt_job;

static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    orc_iterate_range(j->ctx, j->player, j->lo, j->hi, j->root_util);
    return NULL;
}

/* one traverser sweep over all root lanes with `n_threads` workers */
void orc_iterate_mt(const orc_ctx *ctx, int player, float *root_util, int n_threads) {
    size_t n_lanes = (size_t)ctx->n_boards[0] * ctx->n_clusters;
    pthread_t *th;
    mt_job *jobs;
    int i;
    if (n_threads < 1) n_threads = 1;
    if (n_threads == 1) {
        orc_iterate_range(ctx, player, 0, n_lanes, root_util);
        return;
    }
    th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    jobs = (mt_job *)malloc((size_t)n_threads * sizeof(mt_job));
    for (i = 0; i < n_threads; i++) {
        jobs[i].ctx = ctx;
        jobs[i].player = player;
        jobs[i].lo = n_lanes * (size_t)i / (size_t)n_threads;
        jobs[i].hi = n_lanes * (size_t)(i + 1) / (size_t)n_threads;
        jobs[i].root_util = root_util;
        pthread_create(&th[i], NULL, mt_worker, &jobs[i]);
    }
    for (i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}

/* `iterations` full iterations (both traversers), no discount: the timed CPU-baseline loop */
void orc_run_iterations_mt(const orc_ctx *ctx, size_t iterations, int n_threads) {
    size_t t;
    int player;
    for (t = 0; t < iterations; t++)
        for (player = 0; player < 2; player++) orc_iterate_mt(ctx, player, NULL, n_threads);
}

/* ---- bulk accessors for the test harness: SoA [A][n_lanes] <-> the boxed reference layout ------ */
void orc_table_set_node_i32(orc_table *tb, int index, const int32_t *regrets, const int32_t *ssum) {
    size_t n = tb->row_len[index], k;
    for (k = 0; k < n; k++) {
        orc_infoset *is = &tb->rows[index][k];
        int i;
        for (i = 0; i < is->n_actions; i++) {
            is->regrets[i] = regrets[(size_t)i * n + k];
            is->strategy_sum[i] = ssum[(size_t)i * n + k];
        }
    }
}
void orc_table_get_node_i32(const orc_table *tb, int index, int32_t *regrets, int32_t *ssum) {
    size_t n = tb->row_len[index], k;
    for (k = 0; k < n; k++) {
        const orc_infoset *is = &tb->rows[index][k];
        int i;
        for (i = 0; i < is->n_actions; i++) {
            regrets[(size_t)i * n + k] = is->regrets[i];
            ssum[(size_t)i * n + k] = is->strategy_sum[i];
        }
    }
}
void orc_table_set_node_f32(orc_table *tb, int index, const float *regrets, const float *ssum) {
    size_t n = tb->row_len[index], k;
    for (k = 0; k < n; k++) {
        orc_infoset *is = &tb->rows[index][k];
        int i;
        for (i = 0; i < is->n_actions; i++) {
            is->fregrets[i] = regrets[(size_t)i * n + k];
            is->fstrategy_sum[i] = ssum[(size_t)i * n + k];
        }
    }
}
void orc_table_get_node_f32(const orc_table *tb, int index, float *regrets, float *ssum) {
    size_t n = tb->row_len[index], k;
    for (k = 0; k < n; k++) {
        const orc_infoset *is = &tb->rows[index][k];
        int i;
        for (i = 0; i < is->n_actions; i++) {
            regrets[(size_t)i * n + k] = is->fregrets[i];
            ssum[(size_t)i * n + k] = is->fstrategy_sum[i];
        }
    }
}


/* create_infosets (infoset.rs:8-49) with the reference's own sizes: card_abs[round_idx].get_size(player) */
static int create_sizes_rec(const orc_tree *t, const uint32_t sizes[ORC_MAX_ROUNDS][2], orc_table *tb, int node_id) {
    const orc_node *nd = &t->nodes[node_id];
    int i;
    if (nd->kind == ORC_ACTION) {
        size_t n = sizes[nd->round_idx][nd->player], k;
        tb->rows[nd->index] = (orc_infoset *)calloc(n ? n : 1, sizeof(orc_infoset));
        tb->row_len[nd->index] = n;
        for (k = 0; k < n; k++) {
            orc_infoset *is = &tb->rows[nd->index][k];
            is->n_actions = nd->n_children;
            is->regrets = (int32_t *)calloc((size_t)nd->n_children, sizeof(int32_t));
            is->strategy_sum = (int32_t *)calloc((size_t)nd->n_children, sizeof(int32_t));
            if (!is->regrets || !is->strategy_sum) return -1;
        }
    }
    for (i = 0; i < nd->n_children; i++)
        if (create_sizes_rec(t, sizes, tb, nd->children[i]) != 0) return -1;
    return 0;
}
int orc_table_create_sizes(const orc_tree *t, const uint32_t sizes[ORC_MAX_ROUNDS][2], int dtype, orc_table *out) {
    out->n_rows = t->n_action_nodes;
    out->dtype = dtype;
    out->rows = (orc_infoset **)calloc((size_t)out->n_rows, sizeof(orc_infoset *));
    out->row_len = (size_t *)calloc((size_t)out->n_rows, sizeof(size_t));
    if (!out->rows || !out->row_len || dtype != ORC_T_I32) return -1;
    return create_sizes_rec(t, sizes, out, 0);
}
