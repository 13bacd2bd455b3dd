/*
 * cpu_soa.c -- the NON-strawman CPU baseline of bench.py (`cpu_soa`): the same lane-model CFR sweep as the
 * GPU engine on the GPU's own SoA layout (per action node regrets[A][pitch], strategy_sum[A][pitch], lane axis fastest), walked
 * block by block -- a block of SOA_BL consecutive lanes goes through the whole public tree with every per-lane quantity held as a
 * short vector, so every inner loop runs over contiguous lanes and gcc vectorises it (-O3 -march=native) -- by persistent threads,
 * one contiguous lane range each, first-touched by its owner.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY, like the rest of oracle/: it is the thing bench.py's `cpu_baseline` leg times beside the GPU,
 * never part of the product.  PARITY UNPINNED by the reference (it has no test for this path); bench.py asserts that
 * this file and the literal per-lane restatement (rs_oracle.c: orc_traverse) leave bit-identical tables on the same seeded inputs
 * before it times anything.
 *
 * Semantics followed (paths relative to the reference root):
 *   regret matching          src/solver/infoset.rs:83-102
 *   cfr() action block       src/solver/cfr.rs:559-625   (opponent: reach * strategy[i] :585, util += utils[i]*strategy[i] :588)
 *   mccfr() clamp update     src/solver/cfr.rs:413-464   (i64 add, clamp to i32, scale 100)
 *   cfr() wrapping update    src/solver/cfr.rs:612-621   (saturating `as i32`, wrapping +=, scale 10000)
 *   terminals                src/solver/cfr.rs:314-348
 * Full-width opponents, no pruning, pass-through chance nodes, i32 tables: the configuration bench.py's headline runs (config 2).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

#define SOA_BL 256          /* lanes per block: 1 KiB per f32 row, the whole walk stays in L1/L2 */
#define SOA_MAX_DEPTH 24

typedef struct {
    const orc_tree *tree;
    int32_t **reg;          /* [action node index] -> [A][pitch] */
    int32_t **ssm;
    size_t pitch;           /* lanes rounded up to SOA_BL */
    size_t n_lanes;
    const float *sign;      /* [pitch]: > 0 player 0 wins a showdown, < 0 player 1, 0 tie */
    float scale;
    int mode;               /* ORC_UPD_CLAMP_I64 / ORC_UPD_WRAP_I32 */
} soa_ctx;

/* (scale * reach * x) as i64, added to an i32 and clamped to the i32 range: once |delta| >= 2^32 the clamp decides, so the delta is
 * limited to +-2^32 first (exact in f32), which also makes the conversion defined in C; NaN -> 0 like Rust's `as i64` */
static inline int32_t add_clamp(int32_t old, float delta) {
    float d = delta != delta ? 0.0f : delta;
    d = d > 4294967296.0f ? 4294967296.0f : d;
    d = d < -4294967296.0f ? -4294967296.0f : d;
    int64_t v = (int64_t)old + (int64_t)d;
    v = v > INT32_MAX ? INT32_MAX : v;
    v = v < INT32_MIN ? INT32_MIN : v;
    return (int32_t)v;
}
/* Rust `x as i32` (truncate, saturate, NaN -> 0) then wrapping add */
static inline int32_t add_wrap(int32_t old, float delta) {
    float d = delta != delta ? 0.0f : delta;
    int32_t di = d >= 2147483648.0f ? INT32_MAX : d <= -2147483648.0f ? INT32_MIN : (int32_t)d;
    return (int32_t)((uint32_t)old + (uint32_t)di);
}

/* util[0..SOA_BL) of `node` for traverser p over lanes [l0, l0 + SOA_BL); reach == NULL means 1.0 for every lane */
static void walk(const soa_ctx *c, int node, int p, const float *restrict reach, float *restrict util, size_t l0, int depth) {
    const orc_node *nd = &c->tree->nodes[node];
    if (nd->kind == ORC_PRIVATE_CHANCE || nd->kind == ORC_PUBLIC_CHANCE) {   /* cfr.rs:306-313 */
        walk(c, nd->children[0], p, reach, util, l0, depth);
        return;
    }
    if (nd->kind == ORC_TERMINAL) {
        const float pot = (float)nd->value;
        if (nd->ttype == ORC_UNCONTESTED) {                                   /* cfr.rs:316-322 */
            const float v = (p == nd->last_to_act) ? -1.0f * pot : 1.0f * pot;
            for (int i = 0; i < SOA_BL; i++) util[i] = v;
        } else {                                                              /* cfr.rs:323-347 */
            const float *s = c->sign + l0;
            const float win0 = (p == 0) ? pot : -pot;
            for (int i = 0; i < SOA_BL; i++) util[i] = s[i] == 0.0f ? 0.0f : (s[i] > 0.0f ? win0 : -win0);
        }
        return;
    }
    const int A = nd->n_children;
    float sigma[ORC_MAX_ACTIONS][SOA_BL], u[ORC_MAX_ACTIONS][SOA_BL], norm[SOA_BL];
    int32_t *restrict R = c->reg[nd->index] + l0, *restrict S = c->ssm[nd->index] + l0;
    const size_t P = c->pitch;
    /* get_strategy, infoset.rs:83-102 */
    for (int i = 0; i < SOA_BL; i++) norm[i] = 0.0f;
    for (int a = 0; a < A; a++)
        for (int i = 0; i < SOA_BL; i++) {
            const int32_t r = R[a * P + i];
            norm[i] += r > 0 ? (float)r : 0.0f;
        }
    const float uni = 1.0f / (float)A;
    for (int a = 0; a < A; a++)
        for (int i = 0; i < SOA_BL; i++) {
            const int32_t r = R[a * P + i];
            sigma[a][i] = norm[i] > 0.0f ? (r > 0 ? (float)r / norm[i] : 0.0f) : uni;
        }
    if (nd->player == p) {
        for (int a = 0; a < A; a++) walk(c, nd->children[a], p, reach, u[a], l0, depth + 1);   /* cfr.rs:578-581 */
        for (int i = 0; i < SOA_BL; i++) util[i] = 0.0f;
        for (int a = 0; a < A; a++)
            for (int i = 0; i < SOA_BL; i++) util[i] += u[a][i] * sigma[a][i];                 /* cfr.rs:588 */
        const float scale = c->scale;
        if (c->mode == ORC_UPD_CLAMP_I64) {
            for (int a = 0; a < A; a++)
                for (int i = 0; i < SOA_BL; i++) {
                    const float sr = scale * (reach ? reach[i] : 1.0f);
                    R[a * P + i] = add_clamp(R[a * P + i], sr * (u[a][i] - util[i]));
                    S[a * P + i] = add_clamp(S[a * P + i], sr * sigma[a][i]);
                }
        } else {
            for (int a = 0; a < A; a++)
                for (int i = 0; i < SOA_BL; i++) {
                    const float sr = scale * (reach ? reach[i] : 1.0f);
                    R[a * P + i] = add_wrap(R[a * P + i], sr * (u[a][i] - util[i]));
                    S[a * P + i] = add_wrap(S[a * P + i], sr * sigma[a][i]);
                }
        }
    } else {
        float child_reach[SOA_BL];
        for (int i = 0; i < SOA_BL; i++) util[i] = 0.0f;
        for (int a = 0; a < A; a++) {
            for (int i = 0; i < SOA_BL; i++) child_reach[i] = sigma[a][i] * (reach ? reach[i] : 1.0f);   /* cfr.rs:585 */
            walk(c, nd->children[a], p, child_reach, u[a], l0, depth + 1);
            for (int i = 0; i < SOA_BL; i++) util[i] += u[a][i] * sigma[a][i];
        }
    }
}

typedef struct {
    soa_ctx *c;
    size_t blk_lo, blk_hi;      /* this thread's blocks */
    size_t iterations;
    uint64_t seed;
    int op;                     /* 0 = fill (first touch), 1 = iterate */
    int64_t rlo, rhi, slo, shi;
} soa_job;

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static void *soa_worker(void *arg) {
    soa_job *j = (soa_job *)arg;
    soa_ctx *c = j->c;
    if (j->op == 0) {   /* every thread fills (= first-touches) its own lane range of every row */
        const uint64_t rspan = (uint64_t)(j->rhi - j->rlo) + 1, sspan = (uint64_t)(j->shi - j->slo) + 1;
        for (int n = 0; n < c->tree->n_nodes; n++) {
            const orc_node *nd = &c->tree->nodes[n];
            if (nd->kind != ORC_ACTION) continue;
            for (int a = 0; a < nd->n_children; a++)
                for (size_t l = j->blk_lo * SOA_BL; l < j->blk_hi * SOA_BL; l++) {
                    const uint64_t key = ((uint64_t)nd->index << 40) ^ ((uint64_t)a << 32) ^ (uint64_t)l;
                    const int live = l < c->n_lanes;
                    c->reg[nd->index][(size_t)a * c->pitch + l] = live ? (int32_t)(j->rlo + (int64_t)(splitmix64(j->seed ^ key * 0x9E3779B97F4A7C15ull) % rspan)) : 0;
                    c->ssm[nd->index][(size_t)a * c->pitch + l] = live ? (int32_t)(j->slo + (int64_t)(splitmix64((j->seed ^ 0x5353554Dull) ^ key * 0x9E3779B97F4A7C15ull) % sspan)) : 0;
                }
        }
        return NULL;
    }
    float util[SOA_BL];
    for (size_t it = 0; it < j->iterations; it++)   /* lanes never share cells, so a thread may run ahead of the others: no barrier per iteration */
        for (int p = 0; p < 2; p++)
            for (size_t b = j->blk_lo; b < j->blk_hi; b++) walk(c, 0, p, NULL, util, b * SOA_BL, 0);
    return NULL;
}

static void run_threads(soa_ctx *c, soa_job proto, int n_threads) {
    const size_t n_blk = c->pitch / SOA_BL;
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_blk) n_threads = (int)n_blk;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    soa_job *jobs = (soa_job *)malloc((size_t)n_threads * sizeof(soa_job));
    for (int i = 0; i < n_threads; i++) {
        jobs[i] = proto;
        jobs[i].c = c;
        jobs[i].blk_lo = n_blk * (size_t)i / (size_t)n_threads;
        jobs[i].blk_hi = n_blk * (size_t)(i + 1) / (size_t)n_threads;
        pthread_create(&th[i], NULL, soa_worker, &jobs[i]);
    }
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}

/* ---- C entry points (ctypes: oracle/orc.py SoaSolver) ------------------------------------------------------------------------------ */
soa_ctx *soa_create(const orc_tree *tree, size_t n_lanes, const float *sign, float scale, int mode) {
    soa_ctx *c = (soa_ctx *)calloc(1, sizeof(soa_ctx));
    if (!c) return NULL;
    c->tree = tree;
    c->n_lanes = n_lanes;
    c->pitch = (n_lanes + SOA_BL - 1) / SOA_BL * SOA_BL;
    c->scale = scale;
    c->mode = mode;
    c->reg = (int32_t **)calloc((size_t)tree->n_action_nodes, sizeof(int32_t *));
    c->ssm = (int32_t **)calloc((size_t)tree->n_action_nodes, sizeof(int32_t *));
    float *sg = NULL;
    if (posix_memalign((void **)&sg, 64, c->pitch * sizeof(float))) return NULL;
    memset(sg, 0, c->pitch * sizeof(float));
    memcpy(sg, sign, n_lanes * sizeof(float));
    c->sign = sg;
    for (int n = 0; n < tree->n_nodes; n++) {
        const orc_node *nd = &tree->nodes[n];
        if (nd->kind != ORC_ACTION) continue;
        /* malloc'ed, NOT zeroed here: the pages are first touched by the thread that owns the lanes (soa_fill / soa_set_node) */
        if (posix_memalign((void **)&c->reg[nd->index], 64, (size_t)nd->n_children * c->pitch * 4) ||
            posix_memalign((void **)&c->ssm[nd->index], 64, (size_t)nd->n_children * c->pitch * 4))
            return NULL;
    }
    return c;
}
void soa_destroy(soa_ctx *c) {
    if (!c) return;
    for (int i = 0; i < c->tree->n_action_nodes; i++) {
        free(c->reg[i]);
        free(c->ssm[i]);
    }
    free(c->reg);
    free(c->ssm);
    free((void *)c->sign);
    free(c);
}
/* seeded fill by the owning threads (NUMA first touch) */
void soa_fill(soa_ctx *c, uint64_t seed, int64_t rlo, int64_t rhi, int64_t slo, int64_t shi, int n_threads) {
    soa_job j;
    memset(&j, 0, sizeof(j));
    j.op = 0;
    j.seed = seed;
    j.rlo = rlo; j.rhi = rhi; j.slo = slo; j.shi = shi;
    run_threads(c, j, n_threads);
}
void soa_get_node(const soa_ctx *c, int index, int n_actions, int32_t *regrets, int32_t *ssum) {   /* [A][n_lanes], unpadded */
    for (int a = 0; a < n_actions; a++) {
        memcpy(regrets + (size_t)a * c->n_lanes, c->reg[index] + (size_t)a * c->pitch, c->n_lanes * 4);
        memcpy(ssum + (size_t)a * c->n_lanes, c->ssm[index] + (size_t)a * c->pitch, c->n_lanes * 4);
    }
}
void soa_run(soa_ctx *c, size_t iterations, int n_threads) {
    soa_job j;
    memset(&j, 0, sizeof(j));
    j.op = 1;
    j.iterations = iterations;
    run_threads(c, j, n_threads);
}
size_t soa_table_bytes(const soa_ctx *c) {
    size_t cells = 0;
    for (int n = 0; n < c->tree->n_nodes; n++)
        if (c->tree->nodes[n].kind == ORC_ACTION) cells += (size_t)c->tree->nodes[n].n_children * c->pitch;
    return cells * 8;
}

/* ====================================================================================================================================
 * The same for the enumerating cfr() over a MULTI-ROUND tree (BASELINE configs[2]: 706 action nodes, boards 1 / t / r per round, ENUM chance nodes,
 * src/solver/cfr.rs:502-522): `soae_*`.  The sweep decomposes by cluster -- cfr() enumerates boards, never clusters -- so the unit of work is a block of
 * SOAE_BL consecutive clusters walked through the WHOLE tree, boards enumerated in deal order at the public chance nodes exactly as orc_traverse does
 * (reach * (1.0 / len) down, util = util + u up).  Layout: per action node [cluster block][A][boards of the node's round][SOAE_BL] -- everything a unit
 * touches is contiguous, first-touched by the thread that owns the unit.  Blocks are narrow (20 clusters) so that a 5 000-cluster table yields 250 units for
 * the host's threads, dealt out round-robin and kept for every iteration; below the last chance level the sibling boards of a unit are contiguous and are
 * walked together, so the inner loops there -- 99 % of the work -- run over up to 1 280 lanes.
 * bench.py asserts identity with the per-lane restatement (orc_traverse, ORC_CHANCE_ENUM) on a small table before timing.
 * ==================================================================================================================================== */
#define SOAE_BL 20   /* 5 000 clusters = 250 units: one per thread on a 256-thread host */

typedef struct {
    const orc_tree *tree;
    int n_rounds;
    uint32_t boards[ORC_MAX_ROUNDS];
    size_t n_clusters, n_blocks;
    int32_t **reg, **ssm;          /* [action node index] -> [block][A][boards][SOAE_BL] */
    float *sign[ORC_MAX_ROUNDS];   /* [block][boards][SOAE_BL] */
    float scale;
    int mode;
} soae_ctx;

/* util[0 .. nb * SOAE_BL) of `node` for traverser p over boards [b, b + nb) of round r, cluster block blk: the lanes of consecutive boards are contiguous in this
 * layout, so the subtrees below the LAST chance level are walked for up to SOAE_NB sibling boards at once (inner loops of up to 1 024 lanes); above it nb == 1 */
#define SOAE_NB 64
#define SOAE_LMAX (SOAE_NB * SOAE_BL)
static void ewalk(const soae_ctx *c, int node, int p, int r, size_t b, size_t nb, size_t blk, const float *restrict reach, float *restrict util) {
    const orc_node *nd = &c->tree->nodes[node];
    const size_t L = nb * SOAE_BL;
    if (nd->kind == ORC_PRIVATE_CHANCE) {
        ewalk(c, nd->children[0], p, r, b, nb, blk, reach, util);
        return;
    }
    if (nd->kind == ORC_PUBLIC_CHANCE) {   /* cfr.rs:502-522; only ever entered with nb == 1 */
        const uint32_t fan = c->boards[r + 1] / c->boards[r];
        const float inv = 1.0f / (float)fan;
        float child_reach[SOAE_LMAX], u[SOAE_LMAX];
        for (int i = 0; i < SOAE_BL; i++) util[i] = 0.0f;
        const int last = (r + 2 == c->n_rounds);   /* the child round has no chance node below: its boards can be walked together */
        const size_t step = last ? SOAE_NB : 1;
        for (size_t d0 = 0; d0 < fan; d0 += step) {
            const size_t n = (fan - d0) < step ? (fan - d0) : step;
            for (size_t k = 0; k < n; k++)
                for (int i = 0; i < SOAE_BL; i++) child_reach[k * SOAE_BL + i] = (reach ? reach[i] : 1.0f) * inv;
            ewalk(c, nd->children[0], p, r + 1, b * fan + d0, n, blk, child_reach, u);
            for (size_t k = 0; k < n; k++)   /* util = util + u, deal by deal in index order (cfr.rs:519) */
                for (int i = 0; i < SOAE_BL; i++) util[i] = util[i] + u[k * SOAE_BL + i];
        }
        return;
    }
    if (nd->kind == ORC_TERMINAL) {
        const float pot = (float)nd->value;
        if (nd->ttype == ORC_UNCONTESTED) {
            const float v = (p == nd->last_to_act) ? -1.0f * pot : 1.0f * pot;
            for (size_t i = 0; i < L; i++) util[i] = v;
        } else {
            const float *s = c->sign[r] + (blk * c->boards[r] + b) * SOAE_BL;
            const float win0 = (p == 0) ? pot : -pot;
            for (size_t i = 0; i < L; i++) util[i] = s[i] == 0.0f ? 0.0f : (s[i] > 0.0f ? win0 : -win0);
        }
        return;
    }
    const int A = nd->n_children;
    const size_t B = c->boards[r], P = B * SOAE_BL;   /* P: elements between two actions' rows of this block */
    float sigma[ORC_MAX_ACTIONS][SOAE_LMAX], u[ORC_MAX_ACTIONS][SOAE_LMAX], norm[SOAE_LMAX];
    int32_t *restrict R = c->reg[nd->index] + (blk * (size_t)A * B + b) * SOAE_BL, *restrict S = c->ssm[nd->index] + (blk * (size_t)A * B + b) * SOAE_BL;
    for (size_t i = 0; i < L; i++) norm[i] = 0.0f;
    for (int a = 0; a < A; a++)
        for (size_t i = 0; i < L; i++) {
            const int32_t x = R[a * P + i];
            norm[i] += x > 0 ? (float)x : 0.0f;
        }
    const float uni = 1.0f / (float)A;
    for (int a = 0; a < A; a++)
        for (size_t i = 0; i < L; i++) {
            const int32_t x = R[a * P + i];
            sigma[a][i] = norm[i] > 0.0f ? (x > 0 ? (float)x / norm[i] : 0.0f) : uni;
        }
    if (nd->player == p) {
        for (int a = 0; a < A; a++) ewalk(c, nd->children[a], p, r, b, nb, blk, reach, u[a]);
        for (size_t i = 0; i < L; i++) util[i] = 0.0f;
        for (int a = 0; a < A; a++)
            for (size_t i = 0; i < L; i++) util[i] += u[a][i] * sigma[a][i];
        const float scale = c->scale;
        if (c->mode == ORC_UPD_CLAMP_I64) {
            for (int a = 0; a < A; a++)
                for (size_t i = 0; i < L; i++) {
                    const float sr = scale * (reach ? reach[i] : 1.0f);
                    R[a * P + i] = add_clamp(R[a * P + i], sr * (u[a][i] - util[i]));
                    S[a * P + i] = add_clamp(S[a * P + i], sr * sigma[a][i]);
                }
        } else {
            for (int a = 0; a < A; a++)
                for (size_t i = 0; i < L; i++) {
                    const float sr = scale * (reach ? reach[i] : 1.0f);
                    R[a * P + i] = add_wrap(R[a * P + i], sr * (u[a][i] - util[i]));
                    S[a * P + i] = add_wrap(S[a * P + i], sr * sigma[a][i]);
                }
        }
    } else {
        float child_reach[SOAE_LMAX];
        for (size_t i = 0; i < L; i++) util[i] = 0.0f;
        for (int a = 0; a < A; a++) {
            for (size_t i = 0; i < L; i++) child_reach[i] = sigma[a][i] * (reach ? reach[i] : 1.0f);
            ewalk(c, nd->children[a], p, r, b, nb, blk, child_reach, u[a]);
            for (size_t i = 0; i < L; i++) util[i] += u[a][i] * sigma[a][i];
        }
    }
}

typedef struct {
    soae_ctx *c;
    int tid, n_threads;
    size_t iterations;
    uint64_t seed;
    int op;   /* 0 = fill (first touch), 1 = iterate */
    int64_t rlo, rhi, slo, shi;
} soae_job;

static void *soae_worker(void *arg) {
    soae_job *j = (soae_job *)arg;
    soae_ctx *c = j->c;
    if (j->op == 0) {
        const uint64_t rspan = (uint64_t)(j->rhi - j->rlo) + 1, sspan = (uint64_t)(j->shi - j->slo) + 1;
        for (size_t blk = (size_t)j->tid; blk < c->n_blocks; blk += (size_t)j->n_threads)
            for (int n = 0; n < c->tree->n_nodes; n++) {
                const orc_node *nd = &c->tree->nodes[n];
                if (nd->kind != ORC_ACTION) continue;
                const size_t B = c->boards[nd->round_idx];
                for (int a = 0; a < nd->n_children; a++)
                    for (size_t b = 0; b < B; b++)
                        for (int i = 0; i < SOAE_BL; i++) {
                            const size_t cl = blk * SOAE_BL + (size_t)i, lane = b * c->n_clusters + cl;
                            const uint64_t key = ((uint64_t)nd->index << 40) ^ ((uint64_t)a << 36) ^ (uint64_t)lane;
                            const int live = cl < c->n_clusters;
                            const size_t at = ((blk * (size_t)nd->n_children + (size_t)a) * B + b) * SOAE_BL + (size_t)i;
                            c->reg[nd->index][at] = live ? (int32_t)(j->rlo + (int64_t)(splitmix64(j->seed ^ key * 0x9E3779B97F4A7C15ull) % rspan)) : 0;
                            c->ssm[nd->index][at] = live ? (int32_t)(j->slo + (int64_t)(splitmix64((j->seed ^ 0x5353554Dull) ^ key * 0x9E3779B97F4A7C15ull) % sspan)) : 0;
                        }
            }
        return NULL;
    }
    float util[SOAE_LMAX];
    for (size_t it = 0; it < j->iterations; it++)
        for (int p = 0; p < 2; p++)
            for (size_t blk = (size_t)j->tid; blk < c->n_blocks; blk += (size_t)j->n_threads) ewalk(c, 0, p, 0, 0, 1, blk, NULL, util);
    return NULL;
}

static int soae_run_threads(soae_ctx *c, soae_job proto, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > c->n_blocks) n_threads = (int)c->n_blocks;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    soae_job *jobs = (soae_job *)malloc((size_t)n_threads * sizeof(soae_job));
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, (size_t)64 << 20);   /* the walk keeps ~100 KB of per-lane vectors per tree level on the stack */
    for (int i = 0; i < n_threads; i++) {
        jobs[i] = proto;
        jobs[i].c = c;
        jobs[i].tid = i;
        jobs[i].n_threads = n_threads;
        pthread_create(&th[i], &attr, soae_worker, &jobs[i]);
    }
    pthread_attr_destroy(&attr);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
    return n_threads;
}

/* sign[r]: [boards[r]][n_clusters] per round (the per-lane oracle's lane order), may be NULL for a round without showdown terminals */
soae_ctx *soae_create(const orc_tree *tree, int n_rounds, const uint32_t *boards, size_t n_clusters, const float *const *sign, float scale, int mode) {
    soae_ctx *c = (soae_ctx *)calloc(1, sizeof(soae_ctx));
    if (!c || n_rounds < 1 || n_rounds > ORC_MAX_ROUNDS) return NULL;
    c->tree = tree;
    c->n_rounds = n_rounds;
    c->n_clusters = n_clusters;
    c->n_blocks = (n_clusters + SOAE_BL - 1) / SOAE_BL;
    c->scale = scale;
    c->mode = mode;
    for (int r = 0; r < n_rounds; r++) {
        c->boards[r] = boards[r];
        if (r > 0 && (boards[r] % boards[r - 1]) != 0) return NULL;
        if (posix_memalign((void **)&c->sign[r], 64, c->n_blocks * boards[r] * SOAE_BL * sizeof(float))) return NULL;
        memset(c->sign[r], 0, c->n_blocks * boards[r] * SOAE_BL * sizeof(float));
        if (sign && sign[r])
            for (size_t b = 0; b < boards[r]; b++)
                for (size_t cl = 0; cl < n_clusters; cl++)
                    c->sign[r][((cl / SOAE_BL) * boards[r] + b) * SOAE_BL + cl % SOAE_BL] = sign[r][b * n_clusters + cl];
    }
    c->reg = (int32_t **)calloc((size_t)tree->n_action_nodes, sizeof(int32_t *));
    c->ssm = (int32_t **)calloc((size_t)tree->n_action_nodes, sizeof(int32_t *));
    for (int n = 0; n < tree->n_nodes; n++) {
        const orc_node *nd = &tree->nodes[n];
        if (nd->kind != ORC_ACTION) continue;
        if (nd->round_idx >= n_rounds) return NULL;
        const size_t bytes = c->n_blocks * (size_t)nd->n_children * boards[nd->round_idx] * SOAE_BL * 4;
        if (posix_memalign((void **)&c->reg[nd->index], 64, bytes ? bytes : 64) || posix_memalign((void **)&c->ssm[nd->index], 64, bytes ? bytes : 64)) return NULL;
    }
    return c;
}
void soae_destroy(soae_ctx *c) {
    if (!c) return;
    for (int i = 0; i < c->tree->n_action_nodes; i++) {
        free(c->reg[i]);
        free(c->ssm[i]);
    }
    for (int r = 0; r < c->n_rounds; r++) free(c->sign[r]);
    free(c->reg);
    free(c->ssm);
    free(c);
}
int soae_fill(soae_ctx *c, uint64_t seed, int64_t rlo, int64_t rhi, int64_t slo, int64_t shi, int n_threads) {
    soae_job j;
    memset(&j, 0, sizeof(j));
    j.op = 0;
    j.seed = seed;
    j.rlo = rlo; j.rhi = rhi; j.slo = slo; j.shi = shi;
    return soae_run_threads(c, j, n_threads);
}
/* [A][boards * n_clusters] in the per-lane oracle's lane order (lane = board * n_clusters + cluster) */
void soae_get_node(const soae_ctx *c, int index, int n_actions, int round, int32_t *regrets, int32_t *ssum) {
    const size_t B = c->boards[round], L = B * c->n_clusters;
    for (int a = 0; a < n_actions; a++)
        for (size_t b = 0; b < B; b++)
            for (size_t cl = 0; cl < c->n_clusters; cl++) {
                const size_t at = (((cl / SOAE_BL) * (size_t)n_actions + (size_t)a) * B + b) * SOAE_BL + cl % SOAE_BL;
                regrets[(size_t)a * L + b * c->n_clusters + cl] = c->reg[index][at];
                ssum[(size_t)a * L + b * c->n_clusters + cl] = c->ssm[index][at];
            }
}
int soae_run(soae_ctx *c, size_t iterations, int n_threads) {   /* returns the threads actually used (at most one per cluster block) */
    soae_job j;
    memset(&j, 0, sizeof(j));
    j.op = 1;
    j.iterations = iterations;
    return soae_run_threads(c, j, n_threads);
}
size_t soae_table_bytes(const soae_ctx *c) {
    size_t cells = 0;
    for (int n = 0; n < c->tree->n_nodes; n++)
        if (c->tree->nodes[n].kind == ORC_ACTION) cells += (size_t)c->tree->nodes[n].n_children * c->boards[c->tree->nodes[n].round_idx] * c->n_blocks * SOAE_BL;
    return cells * 8;
}
