/*
 * rs_oracle.c -- CPU restatement of RustSolver's regret/strategy-update hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see rs_oracle.h for what that means here).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "rs_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * Rust numeric casts
 * ==================================================================================== */

/* Rust `f32 as i64`: truncate toward zero, saturate, NaN -> 0 (Rust >= 1.45).
 * Used at cfr.rs:424, :433, :446, :455. */
int64_t orc_f32_as_i64(float x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0f) return INT64_MAX;
    if (x <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)x;
}

/* Rust `f32 as i32`: same rules at 32 bits.  Used at cfr.rs:256-257, :617, :619. */
int32_t orc_f32_as_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

/* `i64::from(r) + (x as i64)` (cfr.rs:445, :454): only an infinite delta (the cast saturates to i64::MIN / MAX) can overflow it; a release build
 * wraps (overflow checks off), a debug build panics.  Restated as the wrap, in unsigned arithmetic so that it is defined C. */
static int64_t wrapping_add_i64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static int32_t wrapping_add_i32(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

/* ======================================================================================
 * options.rs / state.rs / tree_builder.rs  -- the [action_node] axis of the table
 * ==================================================================================== */

/* options.rs:52-81 as shipped: board "4d5dAs3cKs" (5 cards -> River), pot 35, stacks 500/500,
 * one street with bets {0.5, 1.0} and raise {3.0}. */
void orc_options_default_river(orc_options *o) {
    memset(o, 0, sizeof(*o));
    o->stack_sizes[0] = 500;
    o->stack_sizes[1] = 500;
    o->starting_pot = 35;
    o->n_board_cards = 5;
    o->n_rounds = 1;
    o->n_bet_sizes[0] = 2;
    o->bet_sizes[0][0] = 0.5;
    o->bet_sizes[0][1] = 1.0;
    o->n_raise_sizes[0] = 1;
    o->raise_sizes[0][0] = 3.0;
}

/* options.rs:68-77 with the commented vectors enabled (three streets from a 3-card board),
 * minus the 2.0 river bet, i.e. {0.5,1.0} / {3.0} on every street (SURVEY.md section 8 shapes). */
void orc_options_three_street(orc_options *o) {
    int r;
    orc_options_default_river(o);
    o->n_board_cards = 3;
    o->n_rounds = 3;
    for (r = 1; r < 3; r++) {
        o->n_bet_sizes[r] = 2;
        o->bet_sizes[r][0] = 0.5;
        o->bet_sizes[r][1] = 1.0;
        o->n_raise_sizes[r] = 1;
        o->raise_sizes[r][0] = 3.0;
    }
}

/* state.rs:26-41, :43-51 */
typedef struct {
    uint32_t stack, wager;
    int has_folded;
} player_state;
typedef struct {
    player_state players[ORC_MAX_PLAYERS];
    uint32_t pot;
    uint8_t raise_count;
    uint8_t current;
    int round;
    int bets_settled;
} game_state;

/* state.rs:53-72 (`impl From<&Options> for GameState`) */
static int state_from_options(const orc_options *o, game_state *s) {
    int p;
    for (p = 0; p < 2; p++) {
        s->players[p].stack = o->stack_sizes[p];
        s->players[p].wager = 0;
        s->players[p].has_folded = 0;
    }
    switch (o->n_board_cards) { /* state.rs:60-65 */
    case 3: s->round = ORC_FLOP; break;
    case 4: s->round = ORC_TURN; break;
    case 5: s->round = ORC_RIVER; break;
    default: return -1; /* panic!("invalid board mask") */
    }
    s->current = 0;
    s->bets_settled = 0;
    s->pot = o->starting_pot;
    s->raise_count = 0;
    return 0;
}

static int is_uncontested(const game_state *s) { /* state.rs:87-94 */
    return s->players[0].has_folded || s->players[1].has_folded;
}
static int is_allin(const game_state *s) { /* state.rs:100-107 */
    return s->players[0].stack == 0 || s->players[1].stack == 0;
}
static int is_terminal(const game_state *s) { /* state.rs:95-99 */
    return s->round == ORC_RIVER || is_allin(s) || is_uncontested(s);
}

/* state.rs:108-124 */
static int to_next_street(const game_state *s, game_state *n) {
    *n = *s;
    n->bets_settled = 0;
    n->current = 0;
    n->players[0].wager = 0;
    n->players[1].wager = 0;
    if (s->round == ORC_FLOP) n->round = ORC_TURN;
    else if (s->round == ORC_TURN) n->round = ORC_RIVER;
    else return -1; /* panic!("Should not get here") */
    return 0;
}

/* state.rs:125-157: order is Check, Call, Fold, Bet.., Raise.. */
static int valid_actions(const game_state *s, const orc_options *o, int round_idx, int *kinds, double *amts) {
    const player_state *cur = &s->players[s->current];
    const player_state *oth = &s->players[1 - s->current];
    int n = 0, i;
    if (oth->wager == 0) { kinds[n] = ORC_ACT_CHECK; amts[n++] = 0; }
    if (oth->wager > cur->wager) { kinds[n] = ORC_ACT_CALL; amts[n++] = 0; }
    if (oth->wager > cur->wager) { kinds[n] = ORC_ACT_FOLD; amts[n++] = 0; }
    if (oth->wager == 0) {
        for (i = 0; i < o->n_bet_sizes[round_idx]; i++) {
            double bet_size = o->bet_sizes[round_idx][i];
            double chips = bet_size * (double)s->pot;
            kinds[n] = ORC_ACT_BET; amts[n++] = bet_size;
            if (chips > ORC_ALLIN_THRESHOLD * (double)cur->stack) break;
        }
    }
    if (s->raise_count < ORC_MAX_RAISES && !is_allin(s) && oth->wager > cur->wager) {
        for (i = 0; i < o->n_raise_sizes[round_idx]; i++) {
            double raise_size = o->raise_sizes[round_idx][i];
            double chips = raise_size * (double)oth->wager;
            kinds[n] = ORC_ACT_RAISE; amts[n++] = raise_size;
            if (chips > ORC_ALLIN_THRESHOLD * (double)cur->stack) break;
        }
    }
    return n;
}

/* Rust `f64 as u32` (saturating, NaN -> 0) -- state.rs:162,:163,:172,:173 */
static uint32_t f64_as_u32(double x) {
    if (x != x) return 0;
    if (x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}

/* state.rs:158-212 */
static void apply_action(const game_state *s, int kind, double amt, game_state *n) {
    player_state *cur, *oth;
    *n = *s;
    cur = &n->players[n->current];
    oth = &n->players[1 - n->current];
    switch (kind) {
    case ORC_ACT_BET: {
        uint32_t chips = f64_as_u32((double)n->pot * amt);
        if (chips > f64_as_u32((double)cur->stack * ORC_ALLIN_THRESHOLD)) chips = cur->stack;
        cur->stack -= chips;
        cur->wager = chips;
        n->pot += chips;
        n->current = (uint8_t)(1 - n->current);
        break;
    }
    case ORC_ACT_RAISE: {
        uint32_t chips = f64_as_u32((double)oth->wager * amt);
        if (chips > f64_as_u32((double)cur->stack * ORC_ALLIN_THRESHOLD)) chips = cur->stack;
        cur->stack -= chips;
        cur->wager += chips;
        n->raise_count += 1;
        n->pot += chips;
        n->current = (uint8_t)(1 - n->current);
        break;
    }
    case ORC_ACT_CALL: {
        uint32_t wager_diff = s->players[1 - s->current].wager - cur->wager;
        if (cur->stack >= wager_diff) {
            n->pot += wager_diff;
            cur->stack -= wager_diff;
        } else {
            n->pot += cur->stack;
            cur->stack = 0;
        }
        n->bets_settled = 1;
        break;
    }
    case ORC_ACT_CHECK:
        if (n->current == ORC_MAX_PLAYERS - 1) n->bets_settled = 1;
        n->current = (uint8_t)(1 - n->current);
        break;
    case ORC_ACT_FOLD: {
        uint32_t wager_diff = s->players[1 - s->current].wager - cur->wager;
        cur->has_folded = 1;
        n->pot -= wager_diff;
        n->bets_settled = 1;
        break;
    }
    }
}

/* tree.rs:48-53 `create_node` */
static int tree_create_node(orc_tree *t, int parent, int kind) {
    orc_node *nd;
    if (t->n_nodes == t->cap) {
        int ncap = t->cap ? t->cap * 2 : 64;
        orc_node *nn = (orc_node *)realloc(t->nodes, (size_t)ncap * sizeof(orc_node));
        if (!nn) return -1;
        t->nodes = nn;
        t->cap = ncap;
    }
    nd = &t->nodes[t->n_nodes];
    memset(nd, 0, sizeof(*nd));
    nd->kind = kind;
    nd->parent = parent;
    nd->index = -1;
    return t->n_nodes++;
}
static void tree_add_child(orc_tree *t, int node, int child) { /* tree.rs:37-39 */
    orc_node *nd = &t->nodes[node];
    nd->children[nd->n_children++] = child;
}

static int build_action_nodes(orc_tree *t, const orc_options *o, int parent, int round_idx, const game_state *s);

/* tree_builder.rs:116-133 */
static int build_terminal(orc_tree *t, int parent, const game_state *s) {
    int id = tree_create_node(t, parent, ORC_TERMINAL);
    orc_node *nd = &t->nodes[id];
    nd->value = s->pot;
    nd->ttype = ORC_SHOWDOWN;
    nd->last_to_act = s->current;
    nd->round = s->round;
    if (is_allin(s) && s->round != ORC_RIVER) nd->ttype = ORC_ALLIN;
    if (is_uncontested(s)) nd->ttype = ORC_UNCONTESTED;
    return id;
}

/* tree_builder.rs:134-143 */
static int build_public_chance(orc_tree *t, const orc_options *o, int parent, int round_idx, const game_state *s) {
    int id = tree_create_node(t, parent, ORC_PUBLIC_CHANCE);
    int child;
    t->nodes[id].round = s->round;
    child = build_action_nodes(t, o, id, round_idx + 1, s);
    tree_add_child(t, id, child);
    return id;
}

/* tree_builder.rs:91-115 */
static void build_action(orc_tree *t, const orc_options *o, int node, int round_idx, const game_state *s, int kind,
                         double amt) {
    game_state next, street;
    int child;
    apply_action(s, kind, amt, &next);
    if (next.bets_settled) {
        if (is_terminal(&next)) {
            child = build_terminal(t, node, &next);
        } else {
            to_next_street(&next, &street);
            child = build_public_chance(t, o, node, round_idx, &street);
        }
    } else {
        child = build_action_nodes(t, o, node, round_idx, &next);
    }
    tree_add_child(t, node, child);
    {
        orc_node *nd = &t->nodes[node];
        nd->action_kind[nd->n_children - 1] = kind; /* an.actions.push(action), tree_builder.rs:109-114 */
        nd->action_amt[nd->n_children - 1] = amt;
    }
}

/* tree_builder.rs:67-90 */
static int build_action_nodes(orc_tree *t, const orc_options *o, int parent, int round_idx, const game_state *s) {
    int kinds[ORC_MAX_ACTIONS + 4];
    double amts[ORC_MAX_ACTIONS + 4];
    int n, i;
    int id = tree_create_node(t, parent, ORC_ACTION);
    t->nodes[id].player = s->current;
    t->nodes[id].index = t->n_action_nodes;
    t->nodes[id].round_idx = (uint8_t)round_idx;
    t->n_action_nodes += 1;
    n = valid_actions(s, o, round_idx, kinds, amts);
    for (i = 0; i < n; i++) build_action(t, o, id, round_idx, s, kinds[i], amts[i]);
    return id;
}

/* tree_builder.rs:9-14 + :60-66 */
int orc_tree_build(const orc_options *o, orc_tree *out) {
    game_state s;
    int root, child;
    memset(out, 0, sizeof(*out));
    if (state_from_options(o, &s) != 0) return -1;
    root = tree_create_node(out, -1, ORC_PRIVATE_CHANCE);
    child = build_action_nodes(out, o, root, 0, &s);
    tree_add_child(out, root, child);
    return 0;
}

void orc_tree_free(orc_tree *t) {
    free(t->nodes);
    memset(t, 0, sizeof(*t));
}

/* ======================================================================================
 * infoset.rs
 * ==================================================================================== */

/* Infoset::init (infoset.rs:76-81): two separate zeroed heap slices */
static int infoset_init(orc_infoset *is, int n_actions, int dtype) {
    is->n_actions = n_actions;
    is->regrets = is->strategy_sum = NULL;
    is->fregrets = is->fstrategy_sum = NULL;
    if (dtype == ORC_T_I32) {
        is->regrets = (int32_t *)calloc((size_t)n_actions, sizeof(int32_t));
        is->strategy_sum = (int32_t *)calloc((size_t)n_actions, sizeof(int32_t));
        return (is->regrets && is->strategy_sum) ? 0 : -1;
    }
    is->fregrets = (float *)calloc((size_t)n_actions, sizeof(float));
    is->fstrategy_sum = (float *)calloc((size_t)n_actions, sizeof(float));
    return (is->fregrets && is->fstrategy_sum) ? 0 : -1;
}

/* create_infosets_rec (infoset.rs:20-49) */
static int create_infosets_rec(const orc_tree *t, const uint32_t *n_boards, uint32_t n_clusters, orc_table *tb,
                               int node_id) {
    const orc_node *nd = &t->nodes[node_id];
    int i;
    switch (nd->kind) {
    case ORC_ACTION: {
        size_t cluster_size = (size_t)n_boards[nd->round_idx] * n_clusters; /* get_size(an.player), infoset.rs:28-32 */
        int n_actions = nd->n_children;
        size_t k;
        tb->rows[nd->index] = (orc_infoset *)malloc(cluster_size * sizeof(orc_infoset));
        if (!tb->rows[nd->index]) return -1;
        tb->row_len[nd->index] = cluster_size;
        for (k = 0; k < cluster_size; k++)
            if (infoset_init(&tb->rows[nd->index][k], n_actions, tb->dtype) != 0) return -1;
        for (i = 0; i < n_actions; i++)
            if (create_infosets_rec(t, n_boards, n_clusters, tb, nd->children[i]) != 0) return -1;
        return 0;
    }
    case ORC_PRIVATE_CHANCE:
    case ORC_PUBLIC_CHANCE:
        return create_infosets_rec(t, n_boards, n_clusters, tb, nd->children[0]);
    default:
        return 0;
    }
}

/* create_infosets (infoset.rs:8-18) */
int orc_table_create(const orc_tree *t, const uint32_t *n_boards, uint32_t n_clusters, int dtype, orc_table *out) {
    out->n_rows = t->n_action_nodes;
    out->dtype = dtype;
    out->rows = (orc_infoset **)calloc((size_t)out->n_rows, sizeof(orc_infoset *));
    out->row_len = (size_t *)calloc((size_t)out->n_rows, sizeof(size_t));
    if (!out->rows || !out->row_len) return -1;
    return create_infosets_rec(t, n_boards, n_clusters, out, 0);
}

void orc_table_free(orc_table *tb) {
    int i;
    size_t k;
    if (!tb->rows) return;
    for (i = 0; i < tb->n_rows; i++) {
        if (!tb->rows[i]) continue;
        for (k = 0; k < tb->row_len[i]; k++) {
            free(tb->rows[i][k].regrets);
            free(tb->rows[i][k].strategy_sum);
            free(tb->rows[i][k].fregrets);
            free(tb->rows[i][k].fstrategy_sum);
        }
        free(tb->rows[i]);
    }
    free(tb->rows);
    free(tb->row_len);
    memset(tb, 0, sizeof(*tb));
}

/* Infoset::get_strategy (infoset.rs:83-102) */
void orc_get_strategy(const int32_t *regrets, int n, float *out) {
    float norm_sum = 0.0f;
    int i;
    for (i = 0; i < n; i++) out[i] = 0.0f;
    for (i = 0; i < n; i++)
        if (regrets[i] > 0) norm_sum += (float)regrets[i];
    for (i = 0; i < n; i++) {
        if (norm_sum > 0.0f) {
            if (regrets[i] > 0) out[i] = (float)regrets[i] / norm_sum;
        } else {
            out[i] = 1.0f / (float)n;
        }
    }
}

/* Infoset::get_final_strategy (infoset.rs:104-123): same formula over strategy_sum */
void orc_get_final_strategy(const int32_t *ssum, int n, float *out) {
    orc_get_strategy(ssum, n, out);
}

/* ======================================================================================
 * cfr.rs update blocks
 * ==================================================================================== */

static int64_t clamp_i64_to_i32(int64_t v) { /* cfr.rs:447-451 */
    if (v > (int64_t)INT32_MAX) return INT32_MAX;
    if (v < (int64_t)INT32_MIN) return INT32_MIN;
    return v;
}

float orc_update_infoset(int32_t *regrets, int32_t *ssum, int n, const float *utils, float cfr_reach, float scale,
                         int mode, int prune) {
    float strategy[ORC_MAX_ACTIONS];
    int explored[ORC_MAX_ACTIONS];
    float util = 0.0f;
    int i;
    orc_get_strategy(regrets, n, strategy); /* cfr.rs:376 / :574 */
    for (i = 0; i < n; i++) {               /* cfr.rs:378-393 / :576-589 */
        explored[i] = 1;
        if (prune && !(regrets[i] > ORC_PRUNE_THRESHOLD)) { /* cfr.rs:380 */
            explored[i] = 0;
            continue;
        }
        util += utils[i] * strategy[i];
    }
    if (mode == ORC_UPD_CLAMP_I64) {
        /* cfr.rs:418-464.  The strategy used at :433/:455 is the one computed before the loop. */
        for (i = 0; i < n; i++) {
            int64_t new_regret, new_ssum;
            if (!explored[i]) continue;
            new_regret = wrapping_add_i64((int64_t)regrets[i], orc_f32_as_i64(scale * cfr_reach * (utils[i] - util)));
            regrets[i] = (int32_t)clamp_i64_to_i32(new_regret);
            new_ssum = wrapping_add_i64((int64_t)ssum[i], orc_f32_as_i64(scale * cfr_reach * strategy[i]));
            ssum[i] = (int32_t)clamp_i64_to_i32(new_ssum);
        }
    } else {
        /* cfr.rs:612-621.  get_strategy() is re-run at :613 BEFORE the loop, so it sees the
         * pre-update regrets and equals `strategy` (no races in a lane-synchronous sweep). */
        for (i = 0; i < n; i++) {
            if (!explored[i]) continue;
            regrets[i] = wrapping_add_i32(regrets[i], orc_f32_as_i32(scale * cfr_reach * (utils[i] - util)));
            ssum[i] = wrapping_add_i32(ssum[i], orc_f32_as_i32(scale * cfr_reach * strategy[i]));
        }
    }
    return util;
}

float orc_node_util(const int32_t *regrets, int n, const float *utils) {
    float strategy[ORC_MAX_ACTIONS];
    float util = 0.0f;
    int i;
    orc_get_strategy(regrets, n, strategy);
    for (i = 0; i < n; i++) util += utils[i] * strategy[i]; /* cfr.rs:588 */
    return util;
}

/* cfr.rs:248-249 */
float orc_discount_factor(size_t tc, size_t interval) {
    float p = (float)(tc / interval);
    return p / (p + 1.0f);
}

/* cfr.rs:254-259 */
void orc_discount_infoset(int32_t *regrets, int32_t *ssum, int n, float d) {
    int k;
    for (k = 0; k < n; k++) {
        regrets[k] = orc_f32_as_i32((float)regrets[k] * d);
        ssum[k] = orc_f32_as_i32((float)ssum[k] * d);
    }
}

/* cfr.rs:250-261 */
void orc_discount_table(orc_table *tb, float d) {
    int i;
    size_t j;
    for (i = 0; i < tb->n_rows; i++)
        for (j = 0; j < tb->row_len[i]; j++) {
            orc_infoset *is = &tb->rows[i][j];
            if (tb->dtype == ORC_T_I32) orc_discount_infoset(is->regrets, is->strategy_sum, is->n_actions, d);
            else orc_discount_f32(is->fregrets, is->fstrategy_sum, is->n_actions, d,
                                  tb->dtype == ORC_T_F16 ? ORC_F_F16 : ORC_F_F32);
        }
}

/* ======================================================================================
 * extension modes (NOT reference semantics; checked GPU-vs-this only)
 * ==================================================================================== */

/* binary32 -> binary16 -> binary32, round to nearest even, overflow to inf, subnormals kept */
float orc_round_f16(float x) {
    union { float f; uint32_t u; } in, out;
    uint32_t sign, a;
    uint16_t h;
    in.f = x;
    sign = in.u & 0x80000000u;
    a = in.u ^ sign;
    if (a >= 0x47800000u) { /* >= 65536, inf or NaN */
        h = (a > 0x7f800000u) ? 0x7e00 : 0x7c00;
    } else if (a < 0x38800000u) { /* below the smallest normal half: 2^-14 */
        union { float f; uint32_t u; } magic, v;
        magic.u = ((127 - 15) + (23 - 10) + 1) << 23;
        v.u = a;
        v.f += magic.f;
        h = (uint16_t)(v.u - magic.u);
    } else {
        uint32_t mant_odd = (a >> 13) & 1u;
        a += ((uint32_t)(15 - 127) << 23) + 0xfffu;
        a += mant_odd;
        h = (uint16_t)(a >> 13);
    }
    /* back to f32 */
    {
        uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, r;
        if (e == 0x1f) r = 0x7f800000u | (m << 13);
        else if (e != 0) r = ((e + 112) << 23) | (m << 13);
        else if (m == 0) r = 0;
        else {
            int sh = 0;
            while (!(m & 0x400)) { m <<= 1; sh++; }
            m &= 0x3ff;
            r = ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
        out.u = r | sign;
    }
    return out.f;
}

static float store_round(float v, int storage) { return storage == ORC_F_F16 ? orc_round_f16(v) : v; }

void orc_get_strategy_f32(const float *regrets, int n, float *out) {
    float norm_sum = 0.0f;
    int i;
    for (i = 0; i < n; i++) out[i] = 0.0f;
    for (i = 0; i < n; i++)
        if (regrets[i] > 0.0f) norm_sum += regrets[i];
    for (i = 0; i < n; i++) {
        if (norm_sum > 0.0f) {
            if (regrets[i] > 0.0f) out[i] = regrets[i] / norm_sum;
        } else {
            out[i] = 1.0f / (float)n;
        }
    }
}

float orc_update_infoset_f32(float *regrets, float *ssum, int n, const float *utils, float cfr_reach, float scale,
                             int rmplus, int storage) {
    float strategy[ORC_MAX_ACTIONS];
    float util = 0.0f;
    int i;
    orc_get_strategy_f32(regrets, n, strategy);
    for (i = 0; i < n; i++) util += utils[i] * strategy[i];
    for (i = 0; i < n; i++) {
        float r = regrets[i] + scale * cfr_reach * (utils[i] - util);
        float s = ssum[i] + scale * cfr_reach * strategy[i];
        if (rmplus && !(r > 0.0f)) r = 0.0f;
        regrets[i] = store_round(r, storage);
        ssum[i] = store_round(s, storage);
    }
    return util;
}

float orc_node_util_f32(const float *regrets, int n, const float *utils) {
    float strategy[ORC_MAX_ACTIONS];
    float util = 0.0f;
    int i;
    orc_get_strategy_f32(regrets, n, strategy);
    for (i = 0; i < n; i++) util += utils[i] * strategy[i];
    return util;
}

void orc_discount_f32(float *regrets, float *ssum, int n, float d, int storage) {
    int k;
    for (k = 0; k < n; k++) {
        regrets[k] = store_round(regrets[k] * d, storage);
        ssum[k] = store_round(ssum[k] * d, storage);
    }
}

/* i32 tables, clamp arithmetic of cfr.rs:445-461, with the regret floored at 0 on write (RM+) */
float orc_update_infoset_rmplus(int32_t *regrets, int32_t *ssum, int n, const float *utils, float cfr_reach,
                                float scale) {
    float strategy[ORC_MAX_ACTIONS];
    float util = 0.0f;
    int i;
    orc_get_strategy(regrets, n, strategy);
    for (i = 0; i < n; i++) util += utils[i] * strategy[i];
    for (i = 0; i < n; i++) {
        int64_t nr = wrapping_add_i64((int64_t)regrets[i], orc_f32_as_i64(scale * cfr_reach * (utils[i] - util)));
        int64_t ns = wrapping_add_i64((int64_t)ssum[i], orc_f32_as_i64(scale * cfr_reach * strategy[i]));
        nr = clamp_i64_to_i32(nr);
        if (nr < 0) nr = 0;
        regrets[i] = (int32_t)nr;
        ssum[i] = (int32_t)clamp_i64_to_i32(ns);
    }
    return util;
}

/* ======================================================================================
 * lane traversal: cfr.rs:481-627 (and the terminal block :314-348) run once per lane
 * ==================================================================================== */

/* ---- opponent sampling (restated rand 0.7 WeightedIndex over supplied bits, see rs_oracle.h) -------------- */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint32_t orc_sample_bits(uint64_t seed, uint32_t node_index, uint64_t lane) {
    /* the 32-bit counter hash of rs_device.hpp sample_bits (lowbias32 finisher over seed, node and lane words; the upper seed half is added to the lane before its
     * multiply, so that which (node, lane) pairs collide depends on the sweep seed) */
    uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
    uint32_t n_mix = (node_index + 1u) * 0xC2B2AE35u;
    uint32_t l_mix = (((uint32_t)lane + s_hi) * 0x9E3779B9u) ^ ((uint32_t)(lane >> 32) * 0x27D4EB2Fu);
    uint32_t x = s_lo ^ n_mix ^ l_mix;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

uint64_t orc_sweep_seed(uint64_t base_seed, uint64_t call_index) {
    return orc_splitmix64(base_seed + call_index * 0x632BE59BD9B4E019ull);
}

int orc_weighted_index(const float *weights, int n, uint32_t bits) {
    float cumulative[ORC_MAX_ACTIONS];
    if (n <= 0) abort();   /* WeightedIndex::new(&[]) is Err(NoItem); the reference unwrap()s it (cfr.rs:471) */
    float total_weight = weights[0];          /* WeightedIndex::new: first weight seeds the total */
    float u01, chosen;
    int i, idx = 0;
    for (i = 1; i < n; i++) {
        cumulative[i - 1] = total_weight;     /* weights.push(total_weight.clone()) */
        total_weight += weights[i];
    }
    u01 = (float)(bits >> 9) * 1.1920928955078125e-07f;   /* (value1_2 - 1.0) with 23 random mantissa bits */
    chosen = u01 * total_weight + 0.0f;                     /* value0_1 * scale + low */
    for (i = 0; i < n - 1; i++)               /* binary_search_by(|w| if *w <= chosen {Less} else {Greater}).unwrap_err() */
        if (cumulative[i] <= chosen) idx = i + 1;
    return idx;
}

static uint32_t child_round_idx(const orc_tree *t, const orc_node *chance) {
    return t->nodes[chance->children[0]].round_idx;
}

/* Terminal arm: cfr.rs:314-348 (mccfr) == cfr.rs:523-557 (cfr) */
static float terminal_value(const orc_ctx *ctx, int node_id, const orc_node *tn, int player, size_t lane) {
    const orc_leaf *lf = &ctx->leaves[node_id];
    if (tn->ttype == ORC_UNCONTESTED) { /* cfr.rs:316-322 */
        if (player == tn->last_to_act) return -1.0f * (float)tn->value;
        return 1.0f * (float)tn->value;
    }
    if (lf->kind == ORC_LEAF_UTIL) return lf->buf[lane];
    { /* SHOWDOWN cfr.rs:323-334 and ALLIN :335-347: compare evaluator scores; buf holds sign(score0-score1) */
        float s = lf->buf[lane];
        int p_wins;
        if (s == 0.0f) return 0.0f;
        p_wins = (player == 0) ? (s > 0.0f) : (s < 0.0f);
        if (p_wins) return 1.0f * (float)tn->value;
        return -1.0f * (float)tn->value;
    }
}

float orc_traverse(const orc_ctx *ctx, int node_id, int player, uint32_t b, uint32_t c, float cfr_reach) {
    const orc_tree *t = ctx->tree;
    const orc_node *nd = &t->nodes[node_id];
    switch (nd->kind) {
    case ORC_PRIVATE_CHANCE:
        /* cfr.rs:310-313.  (cfr.rs:486-501 enumerates hole-card combos instead: in the lane model the
         * combos ARE the lanes, so the root passes straight through.) */
        return orc_traverse(ctx, nd->children[0], player, b, c, cfr_reach);
    case ORC_PUBLIC_CHANCE: {
        uint32_t r_child = child_round_idx(t, nd);
        uint32_t fan = ctx->n_boards[r_child] / ctx->n_boards[r_child - 1];
        if (ctx->chance_mode == ORC_CHANCE_PASS) /* cfr.rs:306-309: one sampled deal == the lane's own board */
            return orc_traverse(ctx, nd->children[0], player, b, c, cfr_reach);
        { /* cfr.rs:502-522: possible_deals.len() == fan; deals taken in index order */
            float child_cfr_reach = cfr_reach * (1.0f / (float)fan);
            float util = 0.0f;
            uint32_t d;
            for (d = 0; d < fan; d++) {
                float u = orc_traverse(ctx, nd->children[0], player, b * fan + d, c, child_cfr_reach);
                util = util + u; /* util.store(util.load() + u), cfr.rs:519 */
            }
            return util;
        }
    }
    case ORC_TERMINAL: {
        size_t lane = (size_t)b * ctx->n_clusters + c;
        return terminal_value(ctx, node_id, nd, player, lane);
    }
    default: break;
    }
    { /* Action arm: cfr.rs:559-625 */
        int n_actions = nd->n_children;
        size_t cluster_idx = (size_t)b * ctx->n_clusters + c; /* get_cluster(), cfr.rs:564-568 (lane model) */
        orc_infoset *infoset = &ctx->table->rows[nd->index][cluster_idx]; /* cfr.rs:573 */
        float utils_stack[ORC_MAX_ACTIONS], strategy_stack[ORC_MAX_ACTIONS];
        float *utils = utils_stack, *strategy = strategy_stack;
        float util = 0.0f;
        int is_int = (ctx->table->dtype == ORC_T_I32);
        int i;
        if (ctx->ref_alloc) { /* vec![0f32; n] at cfr.rs:572 and infoset.rs:85 */
            utils = (float *)calloc((size_t)n_actions, sizeof(float));
            strategy = (float *)calloc((size_t)n_actions, sizeof(float));
        }
        if (is_int) orc_get_strategy(infoset->regrets, n_actions, strategy); /* cfr.rs:574 */
        else orc_get_strategy_f32(infoset->fregrets, n_actions, strategy);

        if (nd->player == player) {
            for (i = 0; i < n_actions; i++) {
                /* pruning exists only in mccfr (cfr.rs:379-386); cfr() ignores its `prune` argument */
                if (ctx->prune && is_int && !(infoset->regrets[i] > ORC_PRUNE_THRESHOLD)) {
                    utils[i] = 0.0f;
                    continue;
                }
                utils[i] = orc_traverse(ctx, nd->children[i], player, b, c, cfr_reach); /* cfr.rs:578-581 */
            }
            if (is_int) {
                if (ctx->rmplus) util = orc_update_infoset_rmplus(infoset->regrets, infoset->strategy_sum, n_actions,
                                                                  utils, cfr_reach, ctx->scale);
                else util = orc_update_infoset(infoset->regrets, infoset->strategy_sum, n_actions, utils, cfr_reach,
                                               ctx->scale, ctx->mode, ctx->prune);
            } else {
                util = orc_update_infoset_f32(infoset->fregrets, infoset->fstrategy_sum, n_actions, utils, cfr_reach,
                                              ctx->scale, ctx->rmplus,
                                              ctx->table->dtype == ORC_T_F16 ? ORC_F_F16 : ORC_F_F32);
            }
        } else if (ctx->opp_mode == ORC_OPP_SAMPLE) {
            /* cfr.rs:467-476: sample one action from the strategy, recurse with cfr_reach * strategy[a_idx] */
            int a_idx = orc_weighted_index(strategy, n_actions, orc_sample_bits(ctx->sample_seed, (uint32_t)nd->index, cluster_idx));
            util = orc_traverse(ctx, nd->children[a_idx], player, b, c, cfr_reach * strategy[a_idx]);
        } else {
            for (i = 0; i < n_actions; i++) {
                utils[i] = orc_traverse(ctx, nd->children[i], player, b, c, strategy[i] * cfr_reach); /* cfr.rs:583-586 */
                util += utils[i] * strategy[i];                                                      /* cfr.rs:588 */
            }
        }
        if (ctx->ref_alloc) {
            free(utils);
            free(strategy);
        }
        return util;
    }
}

/* ======================================================================================
 * deal batches: cfr.rs:299-479 / :481-627 with real get_cluster() addressing, batch-synchronous
 * ==================================================================================== */
float orc_traverse_deal(const orc_deal_ctx *dc, int node_id, int player, size_t deal, float cfr_reach) {
    const orc_ctx *ctx = dc->ctx;
    const orc_tree *t = ctx->tree;
    const orc_node *nd = &t->nodes[node_id];
    if (nd->kind == ORC_PRIVATE_CHANCE || nd->kind == ORC_PUBLIC_CHANCE) /* cfr.rs:306-313: one run-out per deal */
        return orc_traverse_deal(dc, nd->children[0], player, deal, cfr_reach);
    if (nd->kind == ORC_TERMINAL) return terminal_value(ctx, node_id, nd, player, deal);
    {
        int n_actions = nd->n_children, i;
        /* let cluster_idx = card_abs[round_idx].get_cluster(&hand.board, an.player), cfr.rs:361-365 */
        size_t cluster_idx = dc->cidx[nd->round_idx][nd->player][deal];
        const orc_infoset *infoset = &ctx->table->rows[nd->index][cluster_idx];
        orc_infoset *dinfo = &dc->delta->rows[nd->index][cluster_idx];
        float utils[ORC_MAX_ACTIONS], strategy[ORC_MAX_ACTIONS], util = 0.0f;
        const int prune = ctx->prune && (!dc->prune_deal || dc->prune_deal[deal]);   /* the `prune` argument of mccfr(), cfr.rs:219-221 */
        if (ctx->table->dtype != ORC_T_I32) {
            /* float tables (extension; binary32, or binary16 storage held here as floats that are exactly representable as halves): every deal reads the table as of sweep
             * start; its visit contributes dr = (scale*reach)*(u-util), ds = (scale*reach)*sigma in f32, which are added to the cell's f32 delta IN DEAL ORDER (this loop
             * runs deals 0, 1, 2, ..), each delta starting the sweep at 0.0; orc_iterate_deals then adds the deltas to the table -- ONE rounding to the storage type per cell
             * and sweep, and the RM+ floor (regrets only) on that write.  No prune (float tables have none anywhere) */
            orc_get_strategy_f32(infoset->fregrets, n_actions, strategy);
            if (nd->player == player) {
                float k;
                for (i = 0; i < n_actions; i++) utils[i] = orc_traverse_deal(dc, nd->children[i], player, deal, cfr_reach);
                for (i = 0; i < n_actions; i++) util += utils[i] * strategy[i];
                k = ctx->scale * cfr_reach;
                for (i = 0; i < n_actions; i++) {
                    dinfo->fregrets[i] = dinfo->fregrets[i] + k * (utils[i] - util);
                    dinfo->fstrategy_sum[i] = dinfo->fstrategy_sum[i] + k * strategy[i];
                }
            } else if (ctx->opp_mode == ORC_OPP_SAMPLE) {
                int a_idx = orc_weighted_index(strategy, n_actions, orc_sample_bits(ctx->sample_seed, (uint32_t)nd->index, dc->lane_base + deal));
                util = orc_traverse_deal(dc, nd->children[a_idx], player, deal, cfr_reach * strategy[a_idx]);
            } else {
                for (i = 0; i < n_actions; i++) {
                    utils[i] = orc_traverse_deal(dc, nd->children[i], player, deal, strategy[i] * cfr_reach);
                    util += utils[i] * strategy[i];
                }
            }
            return util;
        }
        orc_get_strategy(infoset->regrets, n_actions, strategy);
        if (nd->player == player) {
            int32_t r[ORC_MAX_ACTIONS], s[ORC_MAX_ACTIONS];
            for (i = 0; i < n_actions; i++) {
                if (prune && !(infoset->regrets[i] > ORC_PRUNE_THRESHOLD)) {
                    utils[i] = 0.0f;
                    continue;
                }
                utils[i] = orc_traverse_deal(dc, nd->children[i], player, deal, cfr_reach);
            }
            memcpy(r, infoset->regrets, (size_t)n_actions * sizeof(int32_t));
            memcpy(s, infoset->strategy_sum, (size_t)n_actions * sizeof(int32_t));
            if (ctx->rmplus) util = orc_update_infoset_rmplus(r, s, n_actions, utils, cfr_reach, ctx->scale);
            else util = orc_update_infoset(r, s, n_actions, utils, cfr_reach, ctx->scale, ctx->mode, prune);
            for (i = 0; i < n_actions; i++) { /* delta against the snapshot value, accumulated with wrapping adds */
                dinfo->regrets[i] = wrapping_add_i32(dinfo->regrets[i], (int32_t)((uint32_t)r[i] - (uint32_t)infoset->regrets[i]));
                dinfo->strategy_sum[i] =
                    wrapping_add_i32(dinfo->strategy_sum[i], (int32_t)((uint32_t)s[i] - (uint32_t)infoset->strategy_sum[i]));
            }
        } else if (ctx->opp_mode == ORC_OPP_SAMPLE) { /* cfr.rs:467-476; the hash lane is the deal */
            int a_idx = orc_weighted_index(strategy, n_actions, orc_sample_bits(ctx->sample_seed, (uint32_t)nd->index, dc->lane_base + deal));
            util = orc_traverse_deal(dc, nd->children[a_idx], player, deal, cfr_reach * strategy[a_idx]);
        } else {
            for (i = 0; i < n_actions; i++) {
                utils[i] = orc_traverse_deal(dc, nd->children[i], player, deal, strategy[i] * cfr_reach);
                util += utils[i] * strategy[i];
            }
        }
        return util;
    }
}

void orc_iterate_deals(const orc_deal_ctx *dc, int player, float *root_util) {
    size_t d, j;
    int i, k;
    for (d = 0; d < dc->n_deals; d++) {
        float u = orc_traverse_deal(dc, 0, player, d, 1.0f);
        if (root_util) root_util[d] = u;
    }
    for (i = 0; i < dc->ctx->table->n_rows; i++) {
        int row_player = -1, q;   /* float tables: the write-back (and its RM+ floor) is the TRAVERSER's -- nobody else's rows took a delta in this sweep */
        for (q = 0; q < dc->ctx->tree->n_nodes; q++)
            if (dc->ctx->tree->nodes[q].kind == ORC_ACTION && dc->ctx->tree->nodes[q].index == i) row_player = dc->ctx->tree->nodes[q].player;
        for (j = 0; j < dc->ctx->table->row_len[i]; j++) {
            orc_infoset *is = &dc->ctx->table->rows[i][j], *di = &dc->delta->rows[i][j];
            if (dc->ctx->table->dtype != ORC_T_I32) {
                if (row_player != player) continue;
                const int storage = dc->ctx->table->dtype == ORC_T_F16 ? ORC_F_F16 : ORC_F_F32;
                for (k = 0; k < is->n_actions; k++) {
                    float r = is->fregrets[k] + di->fregrets[k];
                    if (dc->ctx->rmplus && !(r > 0.0f)) r = 0.0f;
                    is->fregrets[k] = store_round(r, storage);
                    is->fstrategy_sum[k] = store_round(is->fstrategy_sum[k] + di->fstrategy_sum[k], storage);
                    di->fregrets[k] = 0.0f;
                    di->fstrategy_sum[k] = 0.0f;
                }
                continue;
            }
            for (k = 0; k < is->n_actions; k++) {
                is->regrets[k] = wrapping_add_i32(is->regrets[k], di->regrets[k]);
                is->strategy_sum[k] = wrapping_add_i32(is->strategy_sum[k], di->strategy_sum[k]);
                di->regrets[k] = 0;
                di->strategy_sum[k] = 0;
            }
        }
    }
}

void orc_iterate_range(const orc_ctx *ctx, int player, size_t lane_lo, size_t lane_hi, float *root_util) {
    size_t lane;
    for (lane = lane_lo; lane < lane_hi; lane++) {
        uint32_t b = (uint32_t)(lane / ctx->n_clusters), c = (uint32_t)(lane % ctx->n_clusters);
        float u = orc_traverse(ctx, 0, player, b, c, 1.0f); /* cfr.rs:217 / :222: node 0, reach 1f32 */
        if (root_util) root_util[lane] = u;
    }
}

void orc_iterate(const orc_ctx *ctx, int player, float *root_util) {
    orc_iterate_range(ctx, player, 0, (size_t)ctx->n_boards[0] * ctx->n_clusters, root_util);
}

/* cfr.rs:188-265, made deterministic: the 8 Hogwild workers become one synchronous sweep over all
 * lanes per iteration, and the polling discount thread becomes a check after every `t += 1`. */
void orc_train(const orc_ctx *ctx, size_t iterations, size_t discount_interval, size_t discount_cap) {
    size_t t = 0, threshold = discount_interval;
    while (t < iterations) {               /* cfr.rs:207 */
        int player;
        for (player = 0; player < 2; player++) orc_iterate(ctx, player, NULL); /* cfr.rs:216-224 */
        t += 1;                            /* cfr.rs:226 */
        if (t > discount_cap) continue;    /* cfr.rs:240-242 (the discount thread exits) */
        if (t > threshold) {               /* cfr.rs:243 */
            float d = orc_discount_factor(t, discount_interval);
            orc_discount_table(ctx->table, d);
            threshold = t + discount_interval; /* cfr.rs:262 */
        }
    }
}
