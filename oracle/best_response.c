/*
 * best_response.c -- CPU restatement of MCCFRTrainer::calc_br (cfr.rs:629-745), the reader of the average strategy that
 * train() runs at every discount tick (cfr.rs:244-246), and of the real best response it stands in for.
 *
 * TEST INFRASTRUCTURE ONLY (see rs_oracle.h): loaded by tests/, smoke() and bench.py's cpu_baseline leg, never by the product.
 * PARITY UNPINNED: the reference holds no test or fixture for calc_br and cannot be built here; pinned by hand-derived known
 * answers (tests/golden/known_answers_br.json) and an independent Python restatement (oracle/np_restate.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

/* get_final_strategy of one info set whatever the table's element type (infoset.rs:104-123) */
static void final_strategy_of(const orc_table *tb, int row, size_t lane, float *out) {
    const orc_infoset *is = &tb->rows[row][lane];
    if (tb->dtype == ORC_T_I32) orc_get_final_strategy(is->strategy_sum, is->n_actions, out);
    else orc_get_strategy_f32(is->fstrategy_sum, is->n_actions, out);   /* same formula over floats (extension dtypes) */
}

/* ---- calc_br AS CODED ------------------------------------------------------------------------------------------------
 * `op` is vec![vec![1.0; 1]; 2] (cfr.rs:631): ONE "hand" per player, so of the n_buckets final strategies that
 * abstract_br_infoset collects (cfr.rs:669-672) only probabilites[0] -- bucket 0 -- is ever used (cfr.rs:677-679), and the
 * terminal "payoffs" are op[opp][0] * (+-)pot * (1 / op[opp][0]) (cfr.rs:703-741): a placeholder, not an exploitability.
 * Restated with the vectors kept (H = 1) so that every f32 operation happens in the reference's order. */
#define BR_H 1
typedef struct { float v[2][BR_H]; } br_vec;       /* Vec<Vec<f32>>: [player][hand] */
typedef struct { float v[2]; } br_res;             /* the [player][0] the callers read */

static br_res abstract_br(const orc_tree *tree, const orc_table *tb, int node_id, br_vec op);

/* cfr.rs:695-745 */
static br_res abstract_br_terminal(const orc_node *tn, br_vec op) {
    br_res res = {{0.0f, 0.0f}};
    const float money_f = (float)tn->value;                          /* cfr.rs:701 */
    float opp_ges_p[2] = {0.0f, 0.0f};
    int p, g;
    for (p = 0; p < 2; p++) {
        const int opp = 1 - p;
        for (g = 0; g < BR_H; g++) {
            float payoff;
            if (tn->ttype == ORC_UNCONTESTED)                        /* cfr.rs:712: (op * sign) * money */
                payoff = op.v[opp][g] * (p == (int)tn->last_to_act ? -1.0f : 1.0f) * money_f;
            else                                                     /* cfr.rs:726: SHOWDOWN and ALLIN alike, +pot for both players */
                payoff = op.v[opp][g] * money_f;
            res.v[p] += payoff;
            opp_ges_p[p] += op.v[opp][g];
        }
        res.v[p] *= 1.0f / opp_ges_p[p];                             /* cfr.rs:716 / :730: 0 reach gives 0 * inf = NaN, as there */
    }
    return res;
}

/* cfr.rs:658-693 */
static br_res abstract_br_infoset(const orc_tree *tree, const orc_table *tb, const orc_node *an, br_vec op) {
    float prob[BR_H][ORC_MAX_ACTIONS];
    br_res payoffs[ORC_MAX_ACTIONS], res = {{0.0f, 0.0f}};
    const int player = an->player, opp = 1 - an->player;
    int a, h, max_index = 0;
    float max_val;
    memset(payoffs, 0, sizeof payoffs);   /* an action node always has children (payoffs[0] would panic in Rust otherwise) */
    for (h = 0; h < BR_H; h++) final_strategy_of(tb, an->index, (size_t)h, prob[h]);   /* cfr.rs:669-672, the buckets that are read */
    for (a = 0; a < an->n_children; a++) {
        br_vec newop = op;                                                               /* cfr.rs:676 */
        for (h = 0; h < BR_H; h++) newop.v[player][h] *= prob[h][a];                     /* cfr.rs:677-679 */
        payoffs[a] = abstract_br(tree, tb, an->children[a], newop);
    }
    max_val = payoffs[0].v[player];                                                      /* cfr.rs:684-691: strict <, first maximum */
    for (a = 1; a < an->n_children; a++)
        if (max_val < payoffs[a].v[player]) {
            max_val = payoffs[a].v[player];
            max_index = a;
        }
    res.v[player] = max_val;
    res.v[opp] = payoffs[max_index].v[opp];
    return res;
}

/* cfr.rs:640-656 */
static br_res abstract_br(const orc_tree *tree, const orc_table *tb, int node_id, br_vec op) {
    const orc_node *n = &tree->nodes[node_id];
    switch (n->kind) {
    case ORC_TERMINAL: return abstract_br_terminal(n, op);
    case ORC_PUBLIC_CHANCE:
    case ORC_PRIVATE_CHANCE: return abstract_br(tree, tb, n->children[0], op);
    default: return abstract_br_infoset(tree, tb, n, op);
    }
}

/* cfr.rs:629-638 */
void orc_calc_br(const orc_tree *tree, const orc_table *tb, float out[2]) {
    br_vec op;
    br_res r;
    int p, h;
    for (p = 0; p < 2; p++)
        for (h = 0; h < BR_H; h++) op.v[p][h] = 1.0f;
    r = abstract_br(tree, tb, 0, op);
    out[0] = r.v[0];
    out[1] = r.v[1];
}

/* ---- the real best response (NOT in the reference, whose calc_br is the placeholder above) ----------
 * Value per deal, in the trainer's leaf utilities (cfr.rs:314-348), of player p against the opponent's average strategy when p
 * plays mode 0: a best response inside the abstraction (one action per cluster), mode 1: its own average strategy.  Vector form over
 * both ranges of a single-round tree on a full board; deals weigh as generate_hand draws them on a full board (cfr.rs:124-137):
 * P(h0, h1) = [no shared card] / (N0 * N1(h0)).  f64, every sum in ascending hand / action order (the product does the same). */
uint32_t orc_evaluate7(const uint8_t *c7);

typedef struct {
    const orc_tree *tree;
    const orc_table *tb;
    int mode, p;
    size_t n[2];
    const uint32_t *cid[2];
    uint64_t *mask[2];
    uint32_t *score[2];
    double *pw[2];
} br_ctx;

static void br_walk(const br_ctx *c, int node_id, const double *q, double *v) {
    const orc_node *n = &c->tree->nodes[node_id];
    const int p = c->p, o = 1 - c->p;
    const size_t np = c->n[p], no = c->n[o];
    size_t h, g;
    int a;
    if (n->kind == ORC_TERMINAL) {
        const double pot = (double)(float)n->value;
        for (h = 0; h < np; h++) {
            double acc = 0.0;
            for (g = 0; g < no; g++) {
                double u;
                if (c->mask[p][h] & c->mask[o][g]) continue;
                if (n->ttype == ORC_UNCONTESTED) u = (p == (int)n->last_to_act) ? -pot : pot;          /* cfr.rs:316-322 */
                else u = c->score[p][h] > c->score[o][g] ? pot : (c->score[p][h] < c->score[o][g] ? -pot : 0.0);   /* cfr.rs:323-347 */
                acc += q[g] * u;
            }
            v[h] = c->pw[p][h] * acc;
        }
        return;
    }
    if (n->kind != ORC_ACTION) {
        br_walk(c, n->children[0], q, v);
        return;
    }
    {
        const int A = n->n_children;
        double *vch = (double *)malloc((size_t)A * np * sizeof(double));
        float sig[ORC_MAX_ACTIONS];
        if ((int)n->player == p) {
            for (a = 0; a < A; a++) br_walk(c, n->children[a], q, vch + (size_t)a * np);
            if (c->mode == 0) {
                /* per cluster: sum the hands' values per action (ascending hand order), first maximum */
                size_t n_clusters = c->tb->row_len[n->index], k;
                double *s = (double *)calloc(n_clusters * (size_t)A, sizeof(double));
                for (a = 0; a < A; a++)
                    for (h = 0; h < np; h++) s[(size_t)c->cid[p][h] * A + a] += vch[(size_t)a * np + h];
                for (h = 0; h < np; h++) {
                    int best = 0;
                    k = c->cid[p][h];
                    for (a = 1; a < A; a++)
                        if (s[k * A + best] < s[k * A + a]) best = a;
                    v[h] = vch[(size_t)best * np + h];
                }
                free(s);
            } else {
                for (h = 0; h < np; h++) {
                    double acc = 0.0;
                    final_strategy_of(c->tb, n->index, c->cid[p][h], sig);
                    for (a = 0; a < A; a++) acc += (double)sig[a] * vch[(size_t)a * np + h];
                    v[h] = acc;
                }
            }
        } else {
            double *qch = (double *)malloc((size_t)A * no * sizeof(double));
            for (g = 0; g < no; g++) {
                final_strategy_of(c->tb, n->index, c->cid[o][g], sig);
                for (a = 0; a < A; a++) qch[(size_t)a * no + g] = q[g] * (double)sig[a];
            }
            for (a = 0; a < A; a++) br_walk(c, n->children[a], qch + (size_t)a * no, vch + (size_t)a * np);
            for (h = 0; h < np; h++) {
                double acc = 0.0;
                for (a = 0; a < A; a++) acc += vch[(size_t)a * np + h];
                v[h] = acc;
            }
            free(qch);
        }
        free(vch);
    }
}

int orc_best_response(const orc_tree *tree, const orc_table *tb, const uint8_t *board, const uint8_t *hands_p0, size_t n0, const uint32_t *cid_p0,
                      const uint8_t *hands_p1, size_t n1, const uint32_t *cid_p1, int mode, double *out) {
    br_ctx c;
    const uint8_t *hands[2];
    double *w0, *ones, *v;
    size_t h, g;
    int p, i;
    memset(&c, 0, sizeof c);
    c.tree = tree; c.tb = tb; c.mode = mode;
    c.n[0] = n0; c.n[1] = n1; c.cid[0] = cid_p0; c.cid[1] = cid_p1;
    hands[0] = hands_p0; hands[1] = hands_p1;
    for (p = 0; p < 2; p++) {
        c.mask[p] = (uint64_t *)malloc(c.n[p] * sizeof(uint64_t));
        c.score[p] = (uint32_t *)malloc(c.n[p] * sizeof(uint32_t));
        for (h = 0; h < c.n[p]; h++) {
            uint8_t c7[7];
            c7[0] = hands[p][2 * h]; c7[1] = hands[p][2 * h + 1];
            for (i = 0; i < 5; i++) c7[2 + i] = board[i];
            c.mask[p][h] = (1ull << c7[0]) | (1ull << c7[1]);
            c.score[p][h] = orc_evaluate7(c7);
        }
    }
    w0 = (double *)malloc(n0 * sizeof(double));
    ones = (double *)malloc(n1 * sizeof(double));
    for (h = 0; h < n0; h++) {
        size_t cnt = 0;
        for (g = 0; g < n1; g++) cnt += (c.mask[0][h] & c.mask[1][g]) == 0;
        w0[h] = cnt ? 1.0 / ((double)n0 * (double)cnt) : 0.0;
    }
    for (g = 0; g < n1; g++) ones[g] = 1.0;
    c.pw[0] = w0; c.pw[1] = ones;
    v = (double *)malloc((n0 > n1 ? n0 : n1) * sizeof(double));
    for (p = 0; p < 2; p++) {
        double total = 0.0;
        c.p = p;
        br_walk(&c, 0, p == 0 ? ones : w0, v);   /* the opponent's initial reach carries its share of the deal probability */
        for (h = 0; h < c.n[p]; h++) total += v[h];
        out[p] = total;
    }
    free(v); free(w0); free(ones);
    for (p = 0; p < 2; p++) { free(c.mask[p]); free(c.score[p]); }
    return 0;
}

/* ---- best response over MULTI-ROUND trees (the reader train() would need at cfr.rs:244-246 for a flop- or turn-start game) -------------
 * The same vector walk as orc_best_response with one change of index: a LANE is (run-out b, hand h) instead of a hand.  generate_hand (cfr.rs:100-143) completes the
 * board to five cards BEFORE the hands are drawn -- K = 5 - n_board0 new cards, uniform over ORDERED sequences without replacement (the first one is the turn of a
 * flop start), so P(B) = 1 / (D (D-1) ..) with D = 52 - n_board0 -- then player 0's combo uniformly among its range's combos that avoid the FULL board, then
 * player 1's among those that avoid board and player 0:  P(B, h0, h1) = P(B) [h0, h1, B disjoint] / (N0(B) N1(B, h0)).  Every showdown -- also an all-in on the flop --
 * compares the seven-card hands on the full board (cfr.rs:323-347: hand.get_hand uses all of hand.board), and chance nodes pass through (one run-out per deal,
 * cfr.rs:306-313).  So every node carries vectors over ALL NB x n lanes; what changes from round to round is only the cluster id a lane is looked up under:
 * cid[r][p][prefix_r(b) * n_p + h], prefix_r(b) = b / (completions left after r new cards) = the index of the first r new cards, run-outs enumerated with the first new
 * card most significant, cards ascending among those still in the deck.
 * mode 0: at each of p's nodes every cluster takes the action with the largest SUM of its lanes' counterfactual values (all run-outs, all hands: a best response inside
 * the abstraction when the abstraction has perfect recall; in general the value of a valid pure strategy of the abstracted game, i.e. a lower bound on it);
 * mode 1: p plays its own average strategy.  f64, every sum in ascending lane / action order.  PARITY UNPINNED (the reference has nothing of the kind). */
typedef struct {
    const orc_tree *tree;
    const orc_table *tb;
    int mode, p, n_rounds;
    size_t n[2], NB;
    size_t per_prefix[ORC_MAX_ROUNDS];          /* run-outs per distinct board prefix of round r */
    const uint32_t *cid[ORC_MAX_ROUNDS][2];
    const uint64_t *hmask[2];
    const uint64_t *bmask;                      /* [NB] the new cards of a run-out */
    uint32_t *score[2];                         /* [NB * n_p] */
    double *pw[2];                              /* [NB * n_p] traverser weights */
    /* sorted mode (RS_BR_SORTED, rs_br.hip): per traverser side pl, per run-out: the opponent's valid hands sorted by (score, index), the holders of every card in that
     * order, and where every traverser hand stands in both */
    int sorted;
    const uint8_t *hands[2];
    uint16_t *ord[2], *nv[2], *cl[2], *nl[2], *nle[2];
    uint8_t *cc[2], *kl[2], *kle[2];
    int16_t *same[2];
} brr_ctx;

#define BRR_HOLDERS 51
typedef struct { uint32_t score; uint16_t hand; } brr_key;
static int brr_key_cmp(const void *a, const void *b) {
    const brr_key *x = (const brr_key *)a, *y = (const brr_key *)b;
    if (x->score != y->score) return x->score < y->score ? -1 : 1;
    return x->hand < y->hand ? -1 : (x->hand > y->hand ? 1 : 0);
}
static void brr_build_index(brr_ctx *c, int pl) {
    const int o = 1 - pl;
    const size_t np = c->n[pl], no = c->n[o], NB = c->NB;
    size_t b, h, g;
    brr_key *keys = (brr_key *)malloc((no ? no : 1) * sizeof(brr_key));
    uint32_t *hs = (uint32_t *)malloc(52 * BRR_HOLDERS * sizeof(uint32_t));
    c->ord[pl] = (uint16_t *)calloc(NB * no + 1, 2); c->nv[pl] = (uint16_t *)calloc(NB + 1, 2);
    c->cl[pl] = (uint16_t *)calloc(NB * 52 * BRR_HOLDERS, 2); c->cc[pl] = (uint8_t *)calloc(NB * 52, 1);
    c->nl[pl] = (uint16_t *)calloc(NB * np + 1, 2); c->nle[pl] = (uint16_t *)calloc(NB * np + 1, 2);
    c->kl[pl] = (uint8_t *)calloc(NB * np * 2 + 1, 1); c->kle[pl] = (uint8_t *)calloc(NB * np * 2 + 1, 1);
    c->same[pl] = (int16_t *)malloc((np ? np : 1) * sizeof(int16_t));
    for (h = 0; h < np; h++) {
        c->same[pl][h] = -1;
        for (g = 0; g < no; g++)
            if (c->hmask[pl][h] == c->hmask[o][g]) c->same[pl][h] = (int16_t)g;
    }
    for (b = 0; b < NB; b++) {
        size_t nv = 0, i;
        int cd;
        for (g = 0; g < no; g++)
            if (!(c->hmask[o][g] & c->bmask[b])) { keys[nv].score = c->score[o][b * no + g]; keys[nv].hand = (uint16_t)g; nv++; }
        qsort(keys, nv, sizeof(brr_key), brr_key_cmp);
        c->nv[pl][b] = (uint16_t)nv;
        for (i = 0; i < nv; i++) c->ord[pl][b * no + i] = keys[i].hand;
        for (cd = 0; cd < 52; cd++) {
            size_t cnt = 0;
            for (i = 0; i < nv; i++) {
                const size_t gg = keys[i].hand;
                if (c->hands[o][2 * gg] == cd || c->hands[o][2 * gg + 1] == cd) {
                    c->cl[pl][(b * 52 + (size_t)cd) * BRR_HOLDERS + cnt] = (uint16_t)gg;
                    hs[cd * BRR_HOLDERS + cnt] = keys[i].score;
                    cnt++;
                }
            }
            c->cc[pl][b * 52 + (size_t)cd] = (uint8_t)cnt;
        }
        for (h = 0; h < np; h++) {
            const size_t lane = b * np + h;
            uint32_t sp;
            size_t a = 0, e = 0;
            int t;
            if (c->hmask[pl][h] & c->bmask[b]) continue;
            sp = c->score[pl][lane];
            for (i = 0; i < nv; i++) { a += keys[i].score < sp; e += keys[i].score <= sp; }
            c->nl[pl][lane] = (uint16_t)a;
            c->nle[pl][lane] = (uint16_t)e;
            for (t = 0; t < 2; t++) {
                const int cd2 = c->hands[pl][2 * h + t];
                size_t j, aa = 0, ee = 0;
                for (j = 0; j < c->cc[pl][b * 52 + (size_t)cd2]; j++) { aa += hs[cd2 * BRR_HOLDERS + j] < sp; ee += hs[cd2 * BRR_HOLDERS + j] <= sp; }
                c->kl[pl][2 * lane + t] = (uint8_t)aa;
                c->kle[pl][2 * lane + t] = (uint8_t)ee;
            }
        }
    }
    free(keys); free(hs);
}
/* a terminal by rank order, every sum in the order rs_br.hip k_br_terminal_sorted uses: P in 64 chunks of ceil(nv / 64) sorted positions (chunk sums from 0.0, offsets
 * sequential over the chunks, P[i] = offset + the sum inside the chunk up to i), Pc sequential per card */
static void brr_terminal_sorted(const brr_ctx *c, const orc_node *n, const double *q, double *v) {
    const int p = c->p, o = 1 - c->p;
    const size_t np = c->n[p], no = c->n[o], NB = c->NB;
    const double pot = (double)(float)n->value;
    const int unc = n->ttype == ORC_UNCONTESTED;
    const double value = unc ? ((p == (int)n->last_to_act) ? -pot : pot) : pot;
    double *P = (double *)malloc((no + 1) * sizeof(double)), *Pc = (double *)malloc(52 * BRR_HOLDERS * sizeof(double)), O[65];
    size_t b, h;
    for (b = 0; b < NB; b++) {
        const size_t nv = c->nv[p][b], len = (nv + 63) / 64;
        const double *qb = q + b * no;
        const uint16_t *ord = c->ord[p] + b * no;
        size_t k, i;
        int cd;
        double run = 0.0, T;
        for (k = 0; k < 64; k++) {
            const size_t i0 = k * len < nv ? k * len : nv, i1 = i0 + len < nv ? i0 + len : nv;
            double sum = 0.0;
            for (i = i0; i < i1; i++) sum += qb[ord[i]];
            O[k] = sum;
        }
        for (k = 0; k < 64; k++) { const double cs = O[k]; O[k] = run; run += cs; }
        O[64] = run;
        for (k = 0; k < 64; k++) {
            const size_t i0 = k * len < nv ? k * len : nv, i1 = i0 + len < nv ? i0 + len : nv;
            double local = 0.0;
            for (i = i0; i < i1; i++) { local += qb[ord[i]]; P[i] = O[k] + local; }
        }
        for (cd = 0; cd < 52; cd++) {
            const size_t cnt = c->cc[p][b * 52 + (size_t)cd];
            double r2 = 0.0;
            for (i = 0; i < cnt; i++) { r2 += qb[c->cl[p][(b * 52 + (size_t)cd) * BRR_HOLDERS + i]]; Pc[cd * BRR_HOLDERS + i] = r2; }
        }
        T = O[64];
        for (h = 0; h < np; h++) {
            const size_t lane = b * np + h;
            size_t c0, c1, n0, n1;
            double T0, T1, acc;
            if (c->hmask[p][h] & c->bmask[b]) { v[lane] = 0.0; continue; }
            c0 = c->hands[p][2 * h]; c1 = c->hands[p][2 * h + 1];
            n0 = c->cc[p][b * 52 + c0]; n1 = c->cc[p][b * 52 + c1];
            T0 = n0 ? Pc[c0 * BRR_HOLDERS + n0 - 1] : 0.0; T1 = n1 ? Pc[c1 * BRR_HOLDERS + n1 - 1] : 0.0;
            if (unc) {
                const int sm = c->same[p][h];
                acc = value * (((T - T0) - T1) + (sm >= 0 ? qb[sm] : 0.0));
            } else {
                const size_t nl = c->nl[p][lane], nle = c->nle[p][lane];
                const size_t a0 = c->kl[p][2 * lane], a1 = c->kl[p][2 * lane + 1], e0 = c->kle[p][2 * lane], e1 = c->kle[p][2 * lane + 1];
                const double L = nl ? P[nl - 1] : 0.0, LE = nle ? P[nle - 1] : 0.0;
                const double L0 = a0 ? Pc[c0 * BRR_HOLDERS + a0 - 1] : 0.0, L1 = a1 ? Pc[c1 * BRR_HOLDERS + a1 - 1] : 0.0;
                const double E0 = e0 ? Pc[c0 * BRR_HOLDERS + e0 - 1] : 0.0, E1 = e1 ? Pc[c1 * BRR_HOLDERS + e1 - 1] : 0.0;
                const double win = (L - L0) - L1;
                const double lose = ((T - LE) - (T0 - E0)) - (T1 - E1);
                acc = value * (win - lose);
            }
            v[lane] = c->pw[p][lane] * acc;
        }
    }
    free(P); free(Pc);
}

static size_t brr_cluster(const brr_ctx *c, int r, int pl, size_t b, size_t h) { return c->cid[r][pl][(b / c->per_prefix[r]) * c->n[pl] + h]; }

static void brr_walk(const brr_ctx *c, int node_id, const double *q, double *v) {
    const orc_node *n = &c->tree->nodes[node_id];
    const int p = c->p, o = 1 - c->p;
    const size_t np = c->n[p], no = c->n[o], NB = c->NB;
    size_t b, h, g;
    int a;
    if (n->kind == ORC_TERMINAL && c->sorted) {
        brr_terminal_sorted(c, n, q, v);
        return;
    }
    if (n->kind == ORC_TERMINAL) {
        const double pot = (double)(float)n->value;
        for (b = 0; b < NB; b++)
            for (h = 0; h < np; h++) {
                double acc = 0.0;
                const uint64_t mine = c->hmask[p][h];
                if (mine & c->bmask[b]) { v[b * np + h] = 0.0; continue; }      /* the hand uses a card of this run-out: no such deal */
                for (g = 0; g < no; g++) {
                    double u;
                    const uint64_t theirs = c->hmask[o][g];
                    if ((theirs & c->bmask[b]) || (theirs & mine)) continue;
                    if (n->ttype == ORC_UNCONTESTED) u = (p == (int)n->last_to_act) ? -pot : pot;                                       /* cfr.rs:316-322 */
                    else u = c->score[p][b * np + h] > c->score[o][b * no + g] ? pot : (c->score[p][b * np + h] < c->score[o][b * no + g] ? -pot : 0.0);
                    acc += q[b * no + g] * u;
                }
                v[b * np + h] = c->pw[p][b * np + h] * acc;
            }
        return;
    }
    if (n->kind != ORC_ACTION) {
        brr_walk(c, n->children[0], q, v);
        return;
    }
    {
        const int A = n->n_children, r = n->round_idx;
        double *vch = (double *)malloc((size_t)A * NB * np * sizeof(double));
        float sig[ORC_MAX_ACTIONS];
        if ((int)n->player == p) {
            for (a = 0; a < A; a++) brr_walk(c, n->children[a], q, vch + (size_t)a * NB * np);
            if (c->mode == 0) {
                const size_t n_clusters = c->tb->row_len[n->index];
                double *s = (double *)calloc(n_clusters * (size_t)A, sizeof(double));
                for (a = 0; a < A; a++)
                    for (b = 0; b < NB; b++)
                        for (h = 0; h < np; h++) s[brr_cluster(c, r, p, b, h) * A + a] += vch[(size_t)a * NB * np + b * np + h];
                for (b = 0; b < NB; b++)
                    for (h = 0; h < np; h++) {
                        const size_t k = brr_cluster(c, r, p, b, h);
                        int best = 0;
                        for (a = 1; a < A; a++)
                            if (s[k * A + best] < s[k * A + a]) best = a;
                        v[b * np + h] = vch[(size_t)best * NB * np + b * np + h];
                    }
                free(s);
            } else {
                for (b = 0; b < NB; b++)
                    for (h = 0; h < np; h++) {
                        double acc = 0.0;
                        final_strategy_of(c->tb, n->index, brr_cluster(c, r, p, b, h), sig);
                        for (a = 0; a < A; a++) acc += (double)sig[a] * vch[(size_t)a * NB * np + b * np + h];
                        v[b * np + h] = acc;
                    }
            }
        } else {
            double *qch = (double *)malloc((size_t)A * NB * no * sizeof(double));
            for (b = 0; b < NB; b++)
                for (g = 0; g < no; g++) {
                    final_strategy_of(c->tb, n->index, brr_cluster(c, r, o, b, g), sig);
                    for (a = 0; a < A; a++) qch[(size_t)a * NB * no + b * no + g] = q[b * no + g] * (double)sig[a];
                }
            for (a = 0; a < A; a++) brr_walk(c, n->children[a], qch + (size_t)a * NB * no, vch + (size_t)a * NB * np);
            for (b = 0; b < NB; b++)
                for (h = 0; h < np; h++) {
                    double acc = 0.0;
                    for (a = 0; a < A; a++) acc += vch[(size_t)a * NB * np + b * np + h];
                    v[b * np + h] = acc;
                }
            free(qch);
        }
        free(vch);
    }
}

/* the run-outs of a board with n_board0 cards: out_cards[NB][5] (the initial cards first, in the order given), returns NB */
size_t orc_br_runouts(const uint8_t *board0, int n_board0, uint8_t *out_cards) {
    uint8_t deck[52];
    int D = 0, i, j, k, c;
    const int K = 5 - n_board0;
    size_t nb = 0;
    for (c = 0; c < 52; c++) {
        int used = 0;
        for (i = 0; i < n_board0; i++) used |= board0[i] == c;
        if (!used) deck[D++] = (uint8_t)c;
    }
    if (K == 0) {
        if (out_cards) memcpy(out_cards, board0, 5);
        return 1;
    }
    for (i = 0; i < D; i++) {
        if (K == 1) {
            if (out_cards) { memcpy(out_cards + nb * 5, board0, (size_t)n_board0); out_cards[nb * 5 + 4] = deck[i]; }
            nb++;
            continue;
        }
        for (j = 0; j < D; j++) {
            if (j == i) continue;
            if (out_cards) {
                for (k = 0; k < n_board0; k++) out_cards[nb * 5 + k] = board0[k];
                out_cards[nb * 5 + 3] = deck[i];
                out_cards[nb * 5 + 4] = deck[j];
            }
            nb++;
        }
    }
    return nb;
}

int orc_best_response_rounds(const orc_tree *tree, const orc_table *tb, const uint8_t *board0, int n_board0, const uint8_t *hands_p0, size_t n0,
                             const uint8_t *hands_p1, size_t n1, const uint32_t *const *cid /* [n_rounds * 2]: [r * 2 + p] -> [prefixes_r][n_p] */, int n_rounds,
                             int mode, double *out) {
    brr_ctx c;
    const uint8_t *hands[2];
    const int K = 5 - n_board0, D = 52 - n_board0;
    uint8_t *cards;
    uint64_t *bmask, *hm[2];
    double *w0, *ones, *q0, *v;
    size_t NB, b, h, g;
    int p, r, i;
    if (n_board0 < 3 || n_board0 > 5 || n_rounds < 1 || n_rounds > K + 1) return -1;
    memset(&c, 0, sizeof c);
    NB = orc_br_runouts(board0, n_board0, NULL);
    cards = (uint8_t *)malloc(NB * 5);
    orc_br_runouts(board0, n_board0, cards);
    c.tree = tree; c.tb = tb; c.mode = mode & 0xff; c.sorted = (mode & 0x100) != 0; c.n_rounds = n_rounds; c.NB = NB;
    c.n[0] = n0; c.n[1] = n1;
    hands[0] = hands_p0; hands[1] = hands_p1;
    for (r = 0; r < n_rounds; r++) {   /* completions left after r new cards: P(D - r, K - r) */
        size_t left = 1;
        for (i = r; i < K; i++) left *= (size_t)(D - i);
        c.per_prefix[r] = left;
        c.cid[r][0] = cid[r * 2];
        c.cid[r][1] = cid[r * 2 + 1];
    }
    bmask = (uint64_t *)malloc(NB * sizeof(uint64_t));
    for (b = 0; b < NB; b++) {
        bmask[b] = 0;
        for (i = n_board0; i < 5; i++) bmask[b] |= 1ull << cards[b * 5 + i];
    }
    c.bmask = bmask;
    for (p = 0; p < 2; p++) {
        hm[p] = (uint64_t *)malloc(c.n[p] * sizeof(uint64_t));
        c.score[p] = (uint32_t *)malloc(NB * c.n[p] * sizeof(uint32_t));
        for (h = 0; h < c.n[p]; h++) hm[p][h] = (1ull << hands[p][2 * h]) | (1ull << hands[p][2 * h + 1]);
        for (b = 0; b < NB; b++)
            for (h = 0; h < c.n[p]; h++) {
                uint8_t c7[7];
                c7[0] = hands[p][2 * h]; c7[1] = hands[p][2 * h + 1];
                for (i = 0; i < 5; i++) c7[2 + i] = cards[b * 5 + i];
                c.score[p][b * c.n[p] + h] = (hm[p][h] & bmask[b]) ? 0u : orc_evaluate7(c7);
            }
        c.hmask[p] = hm[p];
    }
    /* the deal distribution: weight of (b, h0), everything player 1 contributes is 1 */
    w0 = (double *)malloc(NB * n0 * sizeof(double));
    ones = (double *)malloc(NB * n1 * sizeof(double));
    {
        double pb = 1.0;
        for (i = 0; i < K; i++) pb /= (double)(D - i);
        for (b = 0; b < NB; b++) {
            size_t cnt0 = 0;
            for (h = 0; h < n0; h++) cnt0 += (hm[0][h] & bmask[b]) == 0;
            for (h = 0; h < n0; h++) {
                size_t cnt1 = 0;
                if (hm[0][h] & bmask[b]) { w0[b * n0 + h] = 0.0; continue; }
                for (g = 0; g < n1; g++) cnt1 += ((hm[1][g] & bmask[b]) == 0) && ((hm[1][g] & hm[0][h]) == 0);
                w0[b * n0 + h] = cnt1 ? pb / ((double)cnt0 * (double)cnt1) : 0.0;
            }
            for (g = 0; g < n1; g++) ones[b * n1 + g] = 1.0;
        }
    }
    c.pw[0] = w0; c.pw[1] = ones;
    c.hands[0] = hands_p0; c.hands[1] = hands_p1;
    if (c.sorted)
        for (p = 0; p < 2; p++) brr_build_index(&c, p);
    v = (double *)malloc(NB * (n0 > n1 ? n0 : n1) * sizeof(double));
    for (p = 0; p < 2; p++) {
        double total = 0.0;
        c.p = p;
        q0 = p == 0 ? ones : w0;
        brr_walk(&c, 0, q0, v);
        for (b = 0; b < NB * c.n[p]; b++) total += v[b];
        out[p] = total;
    }
    free(v); free(w0); free(ones); free(bmask); free(cards);
    for (p = 0; p < 2; p++) { free(hm[p]); free(c.score[p]); }
    if (c.sorted)
        for (p = 0; p < 2; p++) { free(c.ord[p]); free(c.nv[p]); free(c.cl[p]); free(c.cc[p]); free(c.nl[p]); free(c.nle[p]); free(c.kl[p]); free(c.kle[p]); free(c.same[p]); }
    return 0;
}
