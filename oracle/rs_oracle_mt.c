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
            if (tb->dtype == ORC_T_I32) {
                is->regrets = (int32_t *)calloc((size_t)nd->n_children, sizeof(int32_t));
                is->strategy_sum = (int32_t *)calloc((size_t)nd->n_children, sizeof(int32_t));
                if (!is->regrets || !is->strategy_sum) return -1;
            } else {   /* f32 deal tables (extension) */
                is->fregrets = (float *)calloc((size_t)nd->n_children, sizeof(float));
                is->fstrategy_sum = (float *)calloc((size_t)nd->n_children, sizeof(float));
                if (!is->fregrets || !is->fstrategy_sum) return -1;
            }
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
    if (!out->rows || !out->row_len || (dtype != ORC_T_I32 && dtype != ORC_T_F32 && dtype != ORC_T_F16)) return -1;
    return create_sizes_rec(t, sizes, out, 0);
}


/* ---- threaded deal sweeps for the timed CPU baseline of the deal-batch leg -----------------------------------------
 * Deals are split across threads; each thread accumulates into the shared delta table with relaxed atomic adds
 * (integer adds commute, so the result equals the single-threaded sweep), then one thread applies the deltas. */
typedef struct {
    const orc_deal_ctx *dc;
    int player;
    size_t lo, hi;
} deal_job;

static float traverse_deal_mt(const orc_deal_ctx *dc, int node_id, int player, size_t deal, float cfr_reach);

static void *deal_worker(void *arg) {
    deal_job *j = (deal_job *)arg;
    size_t d;
    for (d = j->lo; d < j->hi; d++) (void)traverse_deal_mt(j->dc, 0, j->player, d, 1.0f);
    return NULL;
}

/* same recursion as orc_traverse_deal (rs_oracle.c), reference-style per-visit allocations, atomic delta adds */
static float traverse_deal_mt(const orc_deal_ctx *dc, int node_id, int player, size_t deal, float cfr_reach) {
    const orc_ctx *ctx = dc->ctx;
    const orc_node *nd = &ctx->tree->nodes[node_id];
    if (nd->kind == ORC_PRIVATE_CHANCE || nd->kind == ORC_PUBLIC_CHANCE)
        return traverse_deal_mt(dc, nd->children[0], player, deal, cfr_reach);
    if (nd->kind == ORC_TERMINAL) {
        const orc_leaf *lf = &ctx->leaves[node_id];
        float s, v = (float)nd->value;
        if (nd->ttype == ORC_UNCONTESTED) return (player == nd->last_to_act) ? -1.0f * v : 1.0f * v;
        s = lf->buf[deal];
        if (lf->kind == ORC_LEAF_UTIL) return s;
        if (s == 0.0f) return 0.0f;
        return ((player == 0) ? (s > 0.0f) : (s < 0.0f)) ? 1.0f * v : -1.0f * v;
    }
    {
        int n = nd->n_children, i;
        size_t cluster_idx = dc->cidx[nd->round_idx][nd->player][deal];
        const orc_infoset *infoset = &ctx->table->rows[nd->index][cluster_idx];
        orc_infoset *dinfo = &dc->delta->rows[nd->index][cluster_idx];
        float *utils = (float *)calloc((size_t)n, sizeof(float));      /* vec![0f32; n_actions], cfr.rs:372 */
        float *strategy = (float *)calloc((size_t)n, sizeof(float));   /* infoset.rs:85 */
        float util = 0.0f;
        orc_get_strategy(infoset->regrets, n, strategy);
        if (nd->player == player) {
            int32_t r[ORC_MAX_ACTIONS], s[ORC_MAX_ACTIONS];
            for (i = 0; i < n; i++) utils[i] = traverse_deal_mt(dc, nd->children[i], player, deal, cfr_reach);
            for (i = 0; i < n; i++) { r[i] = infoset->regrets[i]; s[i] = infoset->strategy_sum[i]; }
            util = orc_update_infoset(r, s, n, utils, cfr_reach, ctx->scale, ctx->mode, 0);
            for (i = 0; i < n; i++) {
                __atomic_fetch_add(&dinfo->regrets[i], (int32_t)((uint32_t)r[i] - (uint32_t)infoset->regrets[i]), __ATOMIC_RELAXED);
                __atomic_fetch_add(&dinfo->strategy_sum[i], (int32_t)((uint32_t)s[i] - (uint32_t)infoset->strategy_sum[i]), __ATOMIC_RELAXED);
            }
        } else if (ctx->opp_mode == ORC_OPP_SAMPLE) {
            int a = orc_weighted_index(strategy, n, orc_sample_bits(ctx->sample_seed, (uint32_t)nd->index, dc->lane_base + deal));
            util = traverse_deal_mt(dc, nd->children[a], player, deal, cfr_reach * strategy[a]);
        } else {
            for (i = 0; i < n; i++) {
                utils[i] = traverse_deal_mt(dc, nd->children[i], player, deal, strategy[i] * cfr_reach);
                util += utils[i] * strategy[i];
            }
        }
        free(utils);
        free(strategy);
        return util;
    }
}

/* `sweeps` iterations (both players each) over all deals with n_threads workers; returns nothing, times are the caller's */
void orc_run_deal_sweeps_mt(orc_deal_ctx *dc, orc_ctx *ctx, size_t sweeps, int n_threads) {
    size_t t;
    int player, i;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    deal_job *jobs = (deal_job *)malloc((size_t)n_threads * sizeof(deal_job));
    for (t = 0; t < sweeps; t++)
        for (player = 0; player < 2; player++) {
            size_t row, j;
            int k;
            ctx->sample_seed = orc_sweep_seed(12345, 2 * t + (size_t)player);
            for (i = 0; i < n_threads; i++) {
                jobs[i].dc = dc;
                jobs[i].player = player;
                jobs[i].lo = dc->n_deals * (size_t)i / (size_t)n_threads;
                jobs[i].hi = dc->n_deals * (size_t)(i + 1) / (size_t)n_threads;
                pthread_create(&th[i], NULL, deal_worker, &jobs[i]);
            }
            for (i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
            for (row = 0; row < (size_t)ctx->table->n_rows; row++)
                for (j = 0; j < ctx->table->row_len[row]; j++) {
                    orc_infoset *is = &ctx->table->rows[row][j], *di = &dc->delta->rows[row][j];
                    for (k = 0; k < is->n_actions; k++) {
                        is->regrets[k] = (int32_t)((uint32_t)is->regrets[k] + (uint32_t)di->regrets[k]);
                        is->strategy_sum[k] = (int32_t)((uint32_t)is->strategy_sum[k] + (uint32_t)di->strategy_sum[k]);
                        di->regrets[k] = di->strategy_sum[k] = 0;
                    }
                }
        }
    free(th);
    free(jobs);
}


/* ---- tuned CPU layout for the non-strawman baseline (SURVEY.md 8(d) "cpu_soa"): every info set of a node lives in ONE
 * contiguous pool (regrets then strategy_sum, [lane][A]), so neighbouring lanes share cache lines and nothing is boxed.
 * The orc_infoset pointers simply point into the pools, so the same orc_traverse runs on it (with ref_alloc = 0). */
int orc_table_create_flat(const orc_tree *t, const uint32_t *n_boards, uint32_t n_clusters, orc_table *out) {
    int i;
    out->n_rows = t->n_action_nodes;
    out->dtype = ORC_T_I32;
    out->rows = (orc_infoset **)calloc((size_t)out->n_rows, sizeof(orc_infoset *));
    out->row_len = (size_t *)calloc((size_t)out->n_rows, sizeof(size_t));
    if (!out->rows || !out->row_len) return -1;
    for (i = 0; i < t->n_nodes; i++) {
        const orc_node *nd = &t->nodes[i];
        size_t n, k;
        int32_t *pool;
        if (nd->kind != ORC_ACTION) continue;
        n = (size_t)n_boards[nd->round_idx] * n_clusters;
        out->rows[nd->index] = (orc_infoset *)malloc(n * sizeof(orc_infoset));
        pool = (int32_t *)calloc(n * (size_t)nd->n_children * 2, sizeof(int32_t));   /* leaked with the table: baseline only */
        if (!out->rows[nd->index] || !pool) return -1;
        out->row_len[nd->index] = 0;   /* orc_table_free must not free the pooled slices */
        for (k = 0; k < n; k++) {
            orc_infoset *is = &out->rows[nd->index][k];
            is->n_actions = nd->n_children;
            is->regrets = pool + k * (size_t)nd->n_children * 2;
            is->strategy_sum = is->regrets + nd->n_children;
            is->fregrets = is->fstrategy_sum = NULL;
        }
    }
    return 0;
}
void orc_table_fill_flat(orc_table *tb, const orc_tree *t, const uint32_t *n_boards, uint32_t n_clusters, uint64_t seed) {
    int i;
    uint64_t x = seed;
    for (i = 0; i < t->n_nodes; i++) {
        const orc_node *nd = &t->nodes[i];
        size_t n, k;
        int a;
        if (nd->kind != ORC_ACTION) continue;
        n = (size_t)n_boards[nd->round_idx] * n_clusters;
        for (k = 0; k < n; k++)
            for (a = 0; a < nd->n_children; a++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                tb->rows[nd->index][k].regrets[a] = (int32_t)((x >> 33) % 2000001u) - 1000000;
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                tb->rows[nd->index][k].strategy_sum[a] = (int32_t)((x >> 33) % 1000001u);
            }
    }
}


/* ---- showdown oracle (checker for rs_showdown_sign): the value of 7 cards is the best of its C(7,5) = 21 five-card
 * hands, each ranked by the textbook rules.  Deliberately brute force and unrelated to the product's bit tricks.
 * card = 4*rank + suit (cfr.rs:592). */
static uint32_t rank5(const uint8_t *c) {
    int r[5], i, j, flush = 1, straight = 0, high;
    int cnt[13] = {0};
    for (i = 0; i < 5; i++) {
        r[i] = c[i] >> 2;
        cnt[r[i]]++;
        if ((c[i] & 3) != (c[0] & 3)) flush = 0;
    }
    for (i = 0; i < 5; i++)   /* sort ranks descending */
        for (j = i + 1; j < 5; j++)
            if (r[j] > r[i]) { int t = r[i]; r[i] = r[j]; r[j] = t; }
    high = r[0];
    if (r[0] - r[4] == 4 && r[0] != r[1] && r[1] != r[2] && r[2] != r[3] && r[3] != r[4]) straight = 1;
    if (r[0] == 12 && r[1] == 3 && r[2] == 2 && r[3] == 1 && r[4] == 0) { straight = 1; high = 3; } /* wheel */
    {
        /* order ranks by (count desc, rank desc) */
        int ord[5], n = 0, q, k;
        for (k = 4; k >= 1; k--)
            for (q = 12; q >= 0; q--)
                if (cnt[q] == k) ord[n++] = q;
        if (straight && flush) return (8u << 20) | (uint32_t)high;
        if (cnt[ord[0]] == 4) return (7u << 20) | ((uint32_t)ord[0] << 4) | (uint32_t)ord[1];
        if (cnt[ord[0]] == 3 && cnt[ord[1]] == 2) return (6u << 20) | ((uint32_t)ord[0] << 4) | (uint32_t)ord[1];
        if (flush) return (5u << 20) | ((uint32_t)r[0] << 16) | ((uint32_t)r[1] << 12) | ((uint32_t)r[2] << 8) | ((uint32_t)r[3] << 4) | (uint32_t)r[4];
        if (straight) return (4u << 20) | (uint32_t)high;
        if (cnt[ord[0]] == 3) return (3u << 20) | ((uint32_t)ord[0] << 8) | ((uint32_t)ord[1] << 4) | (uint32_t)ord[2];
        if (cnt[ord[0]] == 2 && cnt[ord[1]] == 2) return (2u << 20) | ((uint32_t)ord[0] << 8) | ((uint32_t)ord[1] << 4) | (uint32_t)ord[2];
        if (cnt[ord[0]] == 2) return (1u << 20) | ((uint32_t)ord[0] << 12) | ((uint32_t)ord[1] << 8) | ((uint32_t)ord[2] << 4) | (uint32_t)ord[3];
        return ((uint32_t)r[0] << 16) | ((uint32_t)r[1] << 12) | ((uint32_t)r[2] << 8) | ((uint32_t)r[3] << 4) | (uint32_t)r[4];
    }
}
uint32_t orc_evaluate7(const uint8_t *c7) {
    uint32_t best = 0;
    int a, b, i, n;
    for (a = 0; a < 7; a++)
        for (b = a + 1; b < 7; b++) {
            uint8_t h[5];
            uint32_t v;
            for (i = 0, n = 0; i < 7; i++)
                if (i != a && i != b) h[n++] = c7[i];
            v = rank5(h);
            if (v > best) best = v;
        }
    return best;
}
/* cards[9][n]: board x5, player-0 hole x2, player-1 hole x2 -> sign(score0 - score1), cfr.rs:324-333 */
void orc_showdown_sign(const uint8_t *cards, size_t n, float *sign) {
    size_t l;
    for (l = 0; l < n; l++) {
        uint8_t h0[7], h1[7];
        uint32_t s0, s1;
        int i;
        for (i = 0; i < 5; i++) h0[i + 2] = h1[i + 2] = cards[(size_t)i * n + l];
        h0[0] = cards[5 * n + l]; h0[1] = cards[6 * n + l];
        h1[0] = cards[7 * n + l]; h1[1] = cards[8 * n + l];
        s0 = orc_evaluate7(h0);
        s1 = orc_evaluate7(h1);
        sign[l] = s0 == s1 ? 0.0f : (s0 > s1 ? 1.0f : -1.0f);
    }
}


/* ---- threaded train() from CARDS for the timed CPU baseline of the device-trainer leg --------------------------------------------
 * Per sweep every worker first deals its share of the batch -- generate_hand (cfr.rs:100-143), get_cluster for every (round, player)
 * (canonical index, optional bucket file, dense id; the reference repeats this at every action node visit, cfr.rs:357-365, here it is
 * done ONCE per deal, which favours the CPU), the showdown sign once per deal (the reference evaluates both hands at every showdown
 * it reaches, cfr.rs:323-347) -- then traverses it for both players exactly like orc_run_deal_sweeps_mt. */
#include "hand_index.h"

typedef struct {
    const orc_hand_indexer *ix[ORC_MAX_ROUNDS];       /* per tree round: init(2, [2, 3 + street]) */
    const uint32_t *cluster_arr[ORC_MAX_ROUNDS];      /* bucket file or NULL */
    const uint64_t *keys[ORC_MAX_ROUNDS][2];          /* sorted buckets of cluster_map[player] ... */
    const uint32_t *ids[ORC_MAX_ROUNDS][2];           /* ... and their dense ids */
    size_t n_keys[ORC_MAX_ROUNDS][2];
    int n_rounds, first_street;
    uint64_t board_mask, seed;
    const uint8_t *hands[2];
    uint32_t n_hands[2];
    uint32_t *cidx[ORC_MAX_ROUNDS][2];                /* written per sweep; the orc_deal_ctx points at the same arrays */
    float *sign;                                      /* likewise, shared by every showdown leaf */
} orc_cards_ctx;

typedef struct {
    const orc_cards_ctx *cc;
    size_t lo, hi, first_deal;
    int failed;
} cards_job;

void orc_showdown_sign(const uint8_t *cards, size_t n, float *out);

static void *cards_worker(void *arg) {
    cards_job *j = (cards_job *)arg;
    const orc_cards_ctx *cc = j->cc;
    size_t d;
    for (d = j->lo; d < j->hi; d++) {
        uint8_t c9[9], hand[7];
        int r, p, i;
        if (orc_generate_hand(cc->seed, j->first_deal + d, cc->board_mask, cc->hands[0], cc->n_hands[0], cc->hands[1], cc->n_hands[1], c9) != 0) {
            j->failed = 1;
            continue;
        }
        for (r = 0; r < cc->n_rounds; r++)
            for (p = 0; p < 2; p++) {
                const int nb = 3 + cc->first_street + r;
                uint64_t bucket;
                size_t lo = 0, hi = cc->n_keys[r][p];
                hand[0] = c9[5 + 2 * p];
                hand[1] = c9[6 + 2 * p];
                for (i = 0; i < nb; i++) hand[2 + i] = c9[i];
                bucket = orc_hand_index_last(cc->ix[r], hand);
                if (cc->cluster_arr[r]) bucket = cc->cluster_arr[r][bucket];
                while (lo < hi) {   /* cluster_map[player].get(&bucket) */
                    const size_t mid = (lo + hi) / 2;
                    if (cc->keys[r][p][mid] < bucket) lo = mid + 1;
                    else hi = mid;
                }
                if (lo >= cc->n_keys[r][p] || cc->keys[r][p][lo] != bucket) j->failed = 1;   /* Rust: unwrap on None */
                else cc->cidx[r][p][d] = cc->ids[r][p][lo];
            }
        {   /* orc_showdown_sign reads [9][n] column-major; one deal = n 1 */
            orc_showdown_sign(c9, 1, &cc->sign[d]);
        }
    }
    return NULL;
}

int orc_run_train_cards_mt(orc_deal_ctx *dc, orc_ctx *ctx, const orc_cards_ctx *cc, size_t sweeps, int n_threads) {
    size_t t;
    int i, failed = 0;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    cards_job *jobs = (cards_job *)malloc((size_t)n_threads * sizeof(cards_job));
    for (t = 0; t < sweeps; t++) {
        for (i = 0; i < n_threads; i++) {
            jobs[i].cc = cc;
            jobs[i].lo = dc->n_deals * (size_t)i / (size_t)n_threads;
            jobs[i].hi = dc->n_deals * (size_t)(i + 1) / (size_t)n_threads;
            jobs[i].first_deal = t * dc->n_deals;
            jobs[i].failed = 0;
            pthread_create(&th[i], NULL, cards_worker, &jobs[i]);
        }
        for (i = 0; i < n_threads; i++) {
            pthread_join(th[i], NULL);
            failed |= jobs[i].failed;
        }
        orc_run_deal_sweeps_mt(dc, ctx, 1, n_threads);
    }
    free(th);
    free(jobs);
    return failed ? -1 : 0;
}
