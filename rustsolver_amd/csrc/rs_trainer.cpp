// rs_trainer.cpp -- MCCFRTrainer::init + train (cfr.rs:159-297) with everything between "pick a deal" and "write the
// regrets" on the GPU: generate_hand (cfr.rs:100-143), get_cluster for every round and player (cfr.rs:357-365), the showdown
// comparison (cfr.rs:323-347) and the sampled mccfr sweep (cfr.rs:299-479) run as one stream-ordered chain per batch of deals.
// The reference's unit of work -- one deal traversed once per player (cfr.rs:209-226) -- is one lane of a batch; where its eight
// threads race on shared info sets (cfr.rs:414) a batch is synchronous.
// Host code only: it drives the C ABI of this library.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rs_internal.hpp"

using namespace rs;

struct rs_deal_trainer {
    rs_table *table = nullptr;     // trainer.infosets
    rs_solver *solver = nullptr;
    rs_tree *tree = nullptr;       // private copy of the caller's tree (trainer.game_tree)
    rs_card_abs *abs[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};   // borrowed (trainer.card_abs)
    int n_rounds = 0;
    rs_deal_trainer_params params{};
    uint32_t n_hands[2] = {0, 0};
    uint8_t *d_hands[2] = {nullptr, nullptr};   // trainer.hand_ranges
    uint8_t *d_cards = nullptr;                 // [9][pitch]
    uint32_t *d_cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS] = {};
    float *d_sign = nullptr;
    uint32_t *d_err = nullptr;
    uint64_t t = 0;                // iterations done (deals), the shared counter of cfr.rs:200
    uint64_t threshold = 0;        // next discount tick (cfr.rs:203)
    uint64_t batches = 0;
    std::vector<uint8_t> h_hands[2];   // host copy of the ranges (rs_deal_trainer_best_response)
    std::vector<uint32_t> br_cluster[RS_MAX_ROUNDS][2];   // cluster ids of every (board prefix, hand) of every round: they never change, computed at the first best response
    bool br_cluster_ready = false;                        // ... and valid only once every (round, player) has been filled
    rs::BrRun *br_prepared[2] = {nullptr, nullptr};       // [pair loop, rank-order showdowns]: the game-only half of rs_best_response_rounds, kept between calls
    // train()'s prune schedule (cfr.rs:213-221): with a finite prune_threshold the solver runs in RS_UPD_PRUNE mode from the start and every
    // traverser visit honours the deal's flag byte -- all zero (= unpruned, bit for bit) until a batch reaches beyond the threshold
    uint8_t *d_prune = nullptr;        // [pitch] flags of the live batch
    bool live_prune = false;           // the live batch has deals beyond the threshold
    uint64_t live_first = 0, staged_first = 0;   // global number of deal 0 of the live / staged batch
    int tick_br = 0;                   // calc_br at every discount tick (cfr.rs:244-246)
    float last_br[2] = {0.0f, 0.0f};
    uint64_t last_br_t = 0;
    bool have_br = false;
    uint32_t world = 1, rank = 0;  // data-parallel training: this rank's share of every global batch
    // The NEXT batch is dealt (sample -> clusters -> showdown) into staging buffers on a second stream while the current one is swept -- the
    // sweep kernels leave most wave slots of a CU idle -- and swapped in with device-to-device copies.  Same deal numbers, same results.
    hipStream_t deal_stream = nullptr;
    hipEvent_t ev_dealt = nullptr, ev_taken = nullptr;
    uint8_t *s_cards = nullptr;
    uint32_t *s_cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS] = {};
    float *s_sign = nullptr;
    bool staged = false;           // the staging buffers hold the next batch (or will, once ev_dealt fires)
    bool taken_recorded = false;
    // Ordered sweeps walk 32-byte per-deal records sorted by the traverser's last-round cluster (rs_solver.cpp).  The records depend on the deals alone, so with a batch dealt
    // ahead they are sorted ahead as well, on the dealing stream: traverser p's records of batch k + 1 as soon as sweep p of batch k has let go of its own (ev_free), beside the
    // sweeps that follow -- two sorts (0.43 + 0.06 ms per 4 M-deal batch) and the hand-over copies (0.09 ms; nothing in such a sweep reads the live arrays, they are filled on the
    // dealing stream for the accessors) leave the batch's critical path.  The solver asks before every sweep (before_sweep) whether its records are the live batch's.
    bool ahead = false;
    uint8_t *s_prune = nullptr;        // prune flags of the staged batch
    hipEvent_t ev_free[2] = {nullptr, nullptr}, ev_sorted[2] = {nullptr, nullptr}, ev_main = nullptr;
    uint64_t arec_first[2] = {~uint64_t(0), ~uint64_t(0)};   // deal 0 of the batch whose records traverser p's buffer holds (or will, once ev_sorted[p] fires)
    bool wait_sorted[2] = {false, false};                    // the table's stream has not waited for ev_sorted[p] yet
};

static int before_sweep(void *ctx, int p);

extern "C" {

void rs_deal_trainer_destroy(rs_deal_trainer *tr) {
    if (!tr) return;
    for (int k = 0; k < 2; ++k)
        if (tr->br_prepared[k]) rs::br_free(tr->br_prepared[k]);
    if (tr->deal_stream) (void)hipStreamSynchronize(tr->deal_stream);   // a sort dealt ahead may still be writing the solver's records
    if (tr->solver) rs_solver_destroy(tr->solver);
    if (tr->table) {
        if (tr->d_prune) rs_dfree(tr->table, tr->d_prune);
        for (int p = 0; p < 2; ++p)
            if (tr->d_hands[p]) rs_dfree(tr->table, tr->d_hands[p]);
        if (tr->d_cards) rs_dfree(tr->table, tr->d_cards);
        if (tr->d_sign) rs_dfree(tr->table, tr->d_sign);
        if (tr->d_err) rs_dfree(tr->table, tr->d_err);
        if (tr->deal_stream) (void)hipStreamSynchronize(tr->deal_stream);
        if (tr->s_cards) rs_dfree(tr->table, tr->s_cards);
        if (tr->s_sign) rs_dfree(tr->table, tr->s_sign);
        if (tr->s_prune) rs_dfree(tr->table, tr->s_prune);
        for (int p = 0; p < 2; ++p) {
            if (tr->ev_free[p]) (void)hipEventDestroy(tr->ev_free[p]);
            if (tr->ev_sorted[p]) (void)hipEventDestroy(tr->ev_sorted[p]);
        }
        if (tr->ev_main) (void)hipEventDestroy(tr->ev_main);
        for (int r = 0; r < RS_MAX_ROUNDS; ++r)
            for (int p = 0; p < RS_MAX_PLAYERS; ++p)
                if (tr->s_cluster[r][p]) rs_dfree(tr->table, tr->s_cluster[r][p]);
        if (tr->deal_stream) (void)hipStreamDestroy(tr->deal_stream);
        if (tr->ev_dealt) (void)hipEventDestroy(tr->ev_dealt);
        if (tr->ev_taken) (void)hipEventDestroy(tr->ev_taken);
        for (int r = 0; r < RS_MAX_ROUNDS; ++r)
            for (int p = 0; p < RS_MAX_PLAYERS; ++p)
                if (tr->d_cluster[r][p]) rs_dfree(tr->table, tr->d_cluster[r][p]);
        rs_table_destroy(tr->table);
    }
    if (tr->tree) rs_tree_destroy(tr->tree);
    delete tr;
}

int rs_deal_trainer_create(const rs_tree *tree, rs_card_abs *const *card_abs, int n_rounds, const uint8_t *hands_p0, size_t n_hands_p0,
                           const uint8_t *hands_p1, size_t n_hands_p1, const rs_deal_trainer_params *params, int device, rs_deal_trainer **out) {
    if (!tree || !card_abs || !hands_p0 || !hands_p1 || !params || !out) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: NULL argument");
    if (n_rounds < 1 || n_rounds > RS_MAX_ROUNDS) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: 1..3 rounds");
    if (params->deals_per_batch == 0) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: deals_per_batch must be > 0");
    if (n_hands_p0 == 0 || n_hands_p1 == 0 || n_hands_p0 > 0xffffffffull || n_hands_p1 > 0xffffffffull)
        return fail(RS_ERR_INVALID, "rs_deal_trainer_create: empty hand range (Rust: choose().unwrap() on None, cfr.rs:129)");
    const int n_board = __builtin_popcountll(params->board_mask);
    if (n_board < 3 || n_board > 5 || params->board_mask >> 52) return fail(RS_ERR_INVALID, "invalid board mask");   // options.rs:41
    const int first_round = n_board - 3;   // BettingRound of tree round_idx 0 (options.rs:36-42)
    if (first_round + n_rounds > 3) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: more tree rounds than streets left after the board");
    uint32_t n_clusters[RS_MAX_ROUNDS][RS_MAX_PLAYERS] = {};
    uint32_t n_boards[RS_MAX_ROUNDS] = {1, 1, 1};
    for (int r = 0; r < n_rounds; ++r) {
        if (!card_abs[r]) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: card_abs[" + std::to_string(r) + "] is NULL");
        if (rs_card_abs_round(card_abs[r]) != first_round + r)
            return fail(RS_ERR_INVALID, "rs_deal_trainer_create: card_abs[" + std::to_string(r) + "] is not the abstraction of betting round " +
                                            std::to_string(first_round + r));
        for (int p = 0; p < 2; ++p) {
            n_clusters[r][p] = uint32_t(rs_card_abs_size(card_abs[r], p));   // get_size (infoset.rs:28-32)
            if (n_clusters[r][p] == 0) return fail(RS_ERR_INVALID, "rs_deal_trainer_create: an abstraction without clusters");
        }
    }
    rs_deal_trainer *tr = new (std::nothrow) rs_deal_trainer();
    if (!tr) return fail(RS_ERR_OOM, "rs_deal_trainer_create: out of memory");
    tr->params = *params;
    tr->world = params->world ? params->world : 1;
    tr->rank = params->rank;
    if (tr->rank >= tr->world) {
        delete tr;
        return fail(RS_ERR_INVALID, "rs_deal_trainer_create: rank must be below world");
    }
    if (uint64_t(tr->world) * params->deals_per_batch > 0xffffffffull) {
        delete tr;
        return fail(RS_ERR_INVALID, "rs_deal_trainer_create: world * deals_per_batch must fit 32 bits");
    }
    tr->n_rounds = n_rounds;
    tr->threshold = params->discount_interval;
    for (int r = 0; r < n_rounds; ++r) tr->abs[r] = card_abs[r];
    int rc = RS_OK;
    {   // private copy of the tree: the caller may drop theirs
        const int n = rs_tree_n_nodes(tree);
        std::vector<rs_tree_node> nodes(size_t(n > 0 ? n : 0));
        for (int i = 0; rc == RS_OK && i < n; ++i) rc = rs_tree_get_node(tree, i, &nodes[size_t(i)]);
        if (rc == RS_OK) rc = rs_tree_from_nodes(nodes.data(), n, &tr->tree);
        for (int i = 0; rc == RS_OK && i < n; ++i)
            if (nodes[size_t(i)].kind == RS_NODE_ACTION && nodes[size_t(i)].round_idx >= n_rounds)
                rc = fail(RS_ERR_INVALID, "rs_deal_trainer_create: the tree has more rounds than abstractions were passed");
    }
    const int dtype = params->table_dtype;
    if (rc == RS_OK && dtype != RS_I32 && dtype != RS_F32 && dtype != RS_F16) rc = fail(RS_ERR_INVALID, "rs_deal_trainer_create: table_dtype is RS_I32, RS_F32 or RS_F16");
    if (rc == RS_OK && dtype != RS_I32 && (params->prune_threshold != UINT64_MAX || tr->world > 1))
        rc = fail(RS_ERR_UNSUPPORTED, "rs_deal_trainer_create: float tables (extension) take no pruning (prune_threshold = UINT64_MAX: cfr.rs:352 compares i32 regrets) and run on one GPU");
    if (rc == RS_OK) rc = rs_create_infosets(tr->tree, n_clusters, n_boards, dtype, device, &tr->table);   // cfr.rs:176
    const size_t pitch = round_up(params->deals_per_batch, kLanePad);
    const size_t n_hands[2] = {n_hands_p0, n_hands_p1};
    const uint8_t *hands[2] = {hands_p0, hands_p1};
    for (int p = 0; rc == RS_OK && p < 2; ++p) {
        tr->n_hands[p] = uint32_t(n_hands[p]);
        tr->h_hands[p].assign(hands[p], hands[p] + 2 * n_hands[p]);
        for (size_t h = 0; rc == RS_OK && h < n_hands[p]; ++h) {
            const uint8_t a = hands[p][2 * h], b = hands[p][2 * h + 1];
            if (a >= 52 || b >= 52 || a == b) rc = fail(RS_ERR_INVALID, "rs_deal_trainer_create: bad hole cards in a range");
            else if ((1ull << a | 1ull << b) & params->board_mask)
                rc = fail(RS_ERR_INVALID, "rs_deal_trainer_create: a range combo uses a board card (remove_invalid_combos first, cfr.rs:163)");
        }
        if (rc == RS_OK) rc = rs_dmalloc(tr->table, n_hands[p] * 2, reinterpret_cast<void **>(&tr->d_hands[p]));
        if (rc == RS_OK) rc = rs_h2d(tr->table, tr->d_hands[p], hands[p], n_hands[p] * 2);
    }
    if (rc == RS_OK) rc = rs_dmalloc(tr->table, 9 * pitch, reinterpret_cast<void **>(&tr->d_cards));
    if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->d_cards, 0, 9 * pitch);
    if (rc == RS_OK) rc = rs_dmalloc(tr->table, pitch * sizeof(float), reinterpret_cast<void **>(&tr->d_sign));
    if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->d_sign, 0, pitch * sizeof(float));
    if (rc == RS_OK) rc = rs_dmalloc(tr->table, 256, reinterpret_cast<void **>(&tr->d_err));
    if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->d_err, 0, 256);
    // staging for the batch dealt ahead; small batches are bound by the NUMBER of launches on the table's stream (about 5 us each), and swapping a
    // staged batch in costs more of them (four copies + the flags) than dealing in place (two kernels): no staging up to 256 K deals
    // (round 5: from 64 K deals -- a small batch's sweep now runs on ONE stream, merged launches, so the dealing stream has a hardware queue of its own, and with the records
    // sorted ahead nothing is copied on the table's stream: 64 K deals 0.58 -> 0.56 ms, 128 K 0.81 -> 0.76, 256 K 1.27 -> 1.21; 16 K and 4 K lose 0.01-0.03)
    // A one-round game's sweeps are not ordered: its staged batch is swapped in with copies on the table's stream (four launches against two for dealing in place), which only
    // pays beyond 256 K deals (the river game as coded at 64 K deals: 0.10 against 0.12 s for 1 024 batches).
    const bool prefetch = params->prefetch == RS_FORM_ON || (params->prefetch != RS_FORM_OFF && params->deals_per_batch >= (n_rounds > 1 ? (1u << 16) : (1u << 18) + 1u));
    if (rc == RS_OK && prefetch) {
        rc = rs_dmalloc(tr->table, 9 * pitch, reinterpret_cast<void **>(&tr->s_cards));
        if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->s_cards, 0, 9 * pitch);
        if (rc == RS_OK) rc = rs_dmalloc(tr->table, pitch * sizeof(float), reinterpret_cast<void **>(&tr->s_sign));
        if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->s_sign, 0, pitch * sizeof(float));
        for (int r = 0; rc == RS_OK && r < n_rounds; ++r)
            for (int p = 0; rc == RS_OK && p < 2; ++p) {
                rc = rs_dmalloc(tr->table, pitch * sizeof(uint32_t), reinterpret_cast<void **>(&tr->s_cluster[r][p]));
                if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->s_cluster[r][p], 0, pitch * sizeof(uint32_t));
            }
        if (rc == RS_OK) {
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&tr->deal_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&tr->ev_dealt, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&tr->ev_taken, hipEventDisableTiming);
            if (e != hipSuccess) rc = hip_fail(e, "rs_deal_trainer_create: dealing stream");
        }
        if (rc == RS_OK) rc = rs_dmalloc(tr->table, pitch, reinterpret_cast<void **>(&tr->s_prune));
        if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->s_prune, 0, pitch);
        if (rc == RS_OK) {
            hipError_t e = hipSuccess;
            for (int p = 0; p < 2 && e == hipSuccess; ++p) {
                e = hipEventCreateWithFlags(&tr->ev_free[p], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&tr->ev_sorted[p], hipEventDisableTiming);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&tr->ev_main, hipEventDisableTiming);
            if (e != hipSuccess) rc = hip_fail(e, "rs_deal_trainer_create: dealing stream events");
        }
        if (rc == RS_OK) rc = rs_sync(tr->table);   // the memsets above ran on the table's stream
    }
    if (rc == RS_OK) rc = rs_dmalloc(tr->table, pitch, reinterpret_cast<void **>(&tr->d_prune));
    if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->d_prune, 0, pitch);
    rs_deal_batch batch{};
    batch.n_deals = params->deals_per_batch;
    batch.d_prune = tr->d_prune;
    for (int r = 0; rc == RS_OK && r < n_rounds; ++r)
        for (int p = 0; rc == RS_OK && p < 2; ++p) {
            rc = rs_dmalloc(tr->table, pitch * sizeof(uint32_t), reinterpret_cast<void **>(&tr->d_cluster[r][p]));
            if (rc == RS_OK) rc = rs_dmemset(tr->table, tr->d_cluster[r][p], 0, pitch * sizeof(uint32_t));
            batch.d_cluster[r][p] = tr->d_cluster[r][p];
        }
    if (rc == RS_OK) {
        const int n = rs_tree_n_nodes(tr->tree);
        std::vector<rs_leaf_desc> leaves(size_t(n), rs_leaf_desc{RS_LEAF_UNCONTESTED, nullptr});
        for (int i = 0; i < n; ++i) {
            rs_tree_node nd;
            rs_tree_get_node(tr->tree, i, &nd);
            if (nd.kind == RS_NODE_TERMINAL && nd.ttype != RS_TERM_UNCONTESTED) leaves[size_t(i)] = rs_leaf_desc{RS_LEAF_SIGN, tr->d_sign};
        }
        rs_solver_params sp = params->solver;
        sp.chance_mode = RS_CHANCE_PASS;   // one run-out per deal: the board is dealt up front (cfr.rs:115-122, :306-313)
        sp.deal_offset = tr->rank * params->deals_per_batch;
        if (params->prune_threshold != UINT64_MAX) sp.mode |= RS_UPD_PRUNE;   // cfr.rs:352, :379-386, :419-441, per deal through batch.d_prune
        rc = rs_solver_create_deals(tr->table, tr->tree, &batch, leaves.data(), leaves.data(), &sp, &tr->solver);
        if (rc == RS_OK && tr->deal_stream && tr->world == 1) tr->ahead = solver_order_ahead(tr->solver, true, before_sweep, tr);   // false: not an ordered solver
        if (rc == RS_OK && tr->deal_stream && !tr->ahead && params->prefetch != RS_FORM_ON && params->deals_per_batch <= (1u << 18)) {
            // dealt ahead from 64 K deals on BECAUSE the records can be sorted ahead with it; sweeps that are not ordered (more last-round clusters than the sort has bins: the
            // lossless abstractions; a communicator) would swap every staged batch in with eight copies on the table's stream: up to 256 K deals they deal in place
            (void)hipStreamDestroy(tr->deal_stream);
            tr->deal_stream = nullptr;
        }
    }
    if (rc != RS_OK) {
        rs_deal_trainer_destroy(tr);
        return rc;
    }
    *out = tr;
    return RS_OK;
}

rs_table *rs_deal_trainer_table(rs_deal_trainer *tr) { return tr ? tr->table : nullptr; }
rs_solver *rs_deal_trainer_solver(rs_deal_trainer *tr) { return tr ? tr->solver : nullptr; }
uint64_t rs_deal_trainer_iterations(const rs_deal_trainer *tr) { return tr ? tr->t : 0; }
const uint8_t *rs_deal_trainer_cards(const rs_deal_trainer *tr) { return tr ? tr->d_cards : nullptr; }
const float *rs_deal_trainer_signs(const rs_deal_trainer *tr) { return tr ? tr->d_sign : nullptr; }
const uint8_t *rs_deal_trainer_prune_flags(const rs_deal_trainer *tr) { return tr ? tr->d_prune : nullptr; }
const uint32_t *rs_deal_trainer_clusters(const rs_deal_trainer *tr, int round_idx, int player) {
    return tr && round_idx >= 0 && round_idx < tr->n_rounds && (player == 0 || player == 1) ? tr->d_cluster[round_idx][player] : nullptr;
}

// ---- records sorted ahead (tr->ahead) ----------------------------------------------------------------------------------------------------------------------------------
// traverser p's records of the STAGED batch, on the dealing stream, once everything the table's stream has been given so far is done with the buffer
static int sort_staged(rs_deal_trainer *tr, int p) {
    hipStream_t main = (hipStream_t)rs_stream(tr->table);
    hipError_t e = hipSetDevice(rs_table_device(tr->table));
    if (e == hipSuccess) e = hipEventRecord(tr->ev_free[p], main);
    if (e == hipSuccess) e = hipStreamWaitEvent(tr->deal_stream, tr->ev_free[p], 0);
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: records sorted ahead");
    if (int rc = solver_order_on(tr->solver, p, tr->deal_stream, tr->s_cluster, tr->s_sign, tr->s_prune)) return rc;
    e = hipEventRecord(tr->ev_sorted[p], tr->deal_stream);
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: records sorted ahead");
    tr->arec_first[p] = tr->staged_first;
    tr->wait_sorted[p] = true;
    return RS_OK;
}
// in front of every sweep of the trainer's solver (rs_iterate, rs_iterate_phase): traverser p's records must be the LIVE batch's and complete
static int before_sweep(void *ctx, int p) {
    rs_deal_trainer *tr = static_cast<rs_deal_trainer *>(ctx);
    if (!tr->ahead) return RS_OK;
    hipStream_t main = (hipStream_t)rs_stream(tr->table);
    hipError_t e = hipSetDevice(rs_table_device(tr->table));
    if (e == hipSuccess && tr->wait_sorted[p]) {
        e = hipStreamWaitEvent(main, tr->ev_sorted[p], 0);
        tr->wait_sorted[p] = false;
    }
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: records sorted ahead");
    if (tr->arec_first[p] == tr->live_first) return RS_OK;
    // another batch's (the one dealt ahead, sorted ahead too, and then a sweep of the live batch asked for once more; or none yet): sort the live batch's here
    if (tr->taken_recorded) e = hipStreamWaitEvent(main, tr->ev_taken, 0);   // the live arrays are filled on the dealing stream
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: records sorted ahead");
    if (int rc = solver_order_on(tr->solver, p, main, tr->d_cluster, tr->d_sign, tr->d_prune)) return rc;
    tr->arec_first[p] = tr->live_first;
    return RS_OK;
}

// sample -> clusters -> showdown of batch number `tr->batches` into (cards, cluster, sign) on `stream`
static int deal_into(rs_deal_trainer *tr, hipStream_t stream, uint8_t *cards, uint32_t *cluster[RS_MAX_ROUNDS][RS_MAX_PLAYERS], float *sign, uint64_t *first) {
    const uint32_t n = tr->params.deals_per_batch;
    const uint64_t first_deal = (tr->batches * tr->world + tr->rank) * uint64_t(n);   // global batch b = deals [b*world*n, (b+1)*world*n)
    *first = first_deal;
    // one launch deals the cards, compares the two hands (cfr.rs:323-333) and -- when dealing straight into the live buffers -- draws the prune flags
    const bool live = cards == tr->d_cards;
    if (int rc = deals_sample_on(tr->table, stream, tr->params.seed, first_deal, tr->params.board_mask, tr->d_hands[0], tr->n_hands[0], tr->d_hands[1],
                                 tr->n_hands[1], n, cards, tr->d_err, sign, live ? tr->d_prune : (tr->ahead ? tr->s_prune : nullptr), tr->params.prune_threshold))
        return rc;
    for (int r = 0; r < tr->n_rounds; ++r)
        if (int rc = card_abs_clusters_on(tr->abs[r], tr->table, stream, cards, n, cluster[r][0], cluster[r][1])) return rc;
    tr->batches += 1;
    return RS_OK;
}

// deal the batch after the one in the live buffers into the staging buffers, beside whatever the table's stream is doing
static int prefetch(rs_deal_trainer *tr) {
    if (!tr->deal_stream || tr->staged) return RS_OK;
    hipError_t e = hipSetDevice(rs_table_device(tr->table));
    if (e == hipSuccess && tr->taken_recorded) e = hipStreamWaitEvent(tr->deal_stream, tr->ev_taken, 0);   // the previous staged batch has been copied out
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: prefetch");
    if (int rc = deal_into(tr, tr->deal_stream, tr->s_cards, tr->s_cluster, tr->s_sign, &tr->staged_first)) return rc;
    e = hipEventRecord(tr->ev_dealt, tr->deal_stream);
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer: prefetch");
    tr->staged = true;
    return RS_OK;
}

// the live batch's prune flags (cfr.rs:213-221), on the table's stream; a batch dealt straight into the live buffers already has them
static int flag_live_batch(rs_deal_trainer *tr, bool have_flags) {
    const uint32_t n = tr->params.deals_per_batch;
    tr->live_prune = tr->params.prune_threshold != UINT64_MAX && tr->live_first + n - 1 > tr->params.prune_threshold;
    if (!tr->live_prune || have_flags) return RS_OK;
    return deal_prune_flags_on(tr->table, (hipStream_t)rs_stream(tr->table), tr->params.seed, tr->live_first, tr->params.prune_threshold, n, tr->d_prune);
}

// deal the next batch and derive everything the sweep reads from the cards (no table access)
static int deal_batch(rs_deal_trainer *tr, bool wait_live, bool fill_live);
int rs_deal_trainer_deal(rs_deal_trainer *tr) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_deal: trainer is NULL");
    return deal_batch(tr, true, true);
}
// wait_live: the table's stream waits until the live arrays hold the batch (a caller may read them next); rs_deal_trainer_train, whose sweeps read the sorted records alone, does
// that once, before it returns.  fill_live: with the records sorted ahead the live arrays (cards, cluster ids, signs, flags: 36 bytes per deal) exist for the accessors alone --
// a training loop fills them for its LAST batch only (the batches in between are never looked at: 0.27 ms of copies per 4 M-deal batch beside the sweeps)
static int deal_batch(rs_deal_trainer *tr, bool wait_live, bool fill_live) {
    if (!tr->staged) {
        if (int rc = deal_into(tr, (hipStream_t)rs_stream(tr->table), tr->d_cards, tr->d_cluster, tr->d_sign, &tr->live_first)) return rc;
        return flag_live_batch(tr, true);
    }
    // the batch was dealt ahead: swap it in
    hipStream_t main = (hipStream_t)rs_stream(tr->table);
    const size_t pitch = round_up(tr->params.deals_per_batch, kLanePad);
    hipError_t e = hipSetDevice(rs_table_device(tr->table));
    if (tr->ahead) {   // the sweeps read the sorted records alone: sort what rs_deal_trainer_train has not sorted already, fill the live arrays (for the accessors) on the dealing stream
        for (int p = 0; p < 2; ++p)
            if (tr->arec_first[p] != tr->staged_first)
                if (int rc = sort_staged(tr, p)) return rc;
        hipStream_t ds = tr->deal_stream;
        if (fill_live) {
            if (e == hipSuccess) e = hipEventRecord(tr->ev_main, main);   // whatever still reads the live arrays on the table's stream (a sort of the batch before)
            if (e == hipSuccess) e = hipStreamWaitEvent(ds, tr->ev_main, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(tr->d_cards, tr->s_cards, 9 * pitch, hipMemcpyDeviceToDevice, ds);
            if (e == hipSuccess) e = hipMemcpyAsync(tr->d_sign, tr->s_sign, pitch * sizeof(float), hipMemcpyDeviceToDevice, ds);
            if (e == hipSuccess) e = hipMemcpyAsync(tr->d_prune, tr->s_prune, pitch, hipMemcpyDeviceToDevice, ds);
            for (int r = 0; e == hipSuccess && r < tr->n_rounds; ++r)
                for (int p = 0; e == hipSuccess && p < 2; ++p)
                    e = hipMemcpyAsync(tr->d_cluster[r][p], tr->s_cluster[r][p], pitch * sizeof(uint32_t), hipMemcpyDeviceToDevice, ds);
        }
        if (e == hipSuccess) e = hipEventRecord(tr->ev_taken, ds);
        if (e == hipSuccess && wait_live) e = hipStreamWaitEvent(main, tr->ev_taken, 0);
        if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer_deal: swap");
        tr->taken_recorded = true;
        tr->staged = false;
        tr->live_first = tr->staged_first;
        tr->live_prune = tr->params.prune_threshold != UINT64_MAX && tr->live_first + tr->params.deals_per_batch - 1 > tr->params.prune_threshold;
        return RS_OK;
    }
    if (e == hipSuccess) e = hipStreamWaitEvent(main, tr->ev_dealt, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(tr->d_cards, tr->s_cards, 9 * pitch, hipMemcpyDeviceToDevice, main);
    if (e == hipSuccess) e = hipMemcpyAsync(tr->d_sign, tr->s_sign, pitch * sizeof(float), hipMemcpyDeviceToDevice, main);
    for (int r = 0; e == hipSuccess && r < tr->n_rounds; ++r)
        for (int p = 0; e == hipSuccess && p < 2; ++p)
            e = hipMemcpyAsync(tr->d_cluster[r][p], tr->s_cluster[r][p], pitch * sizeof(uint32_t), hipMemcpyDeviceToDevice, main);
    if (e == hipSuccess) e = hipEventRecord(tr->ev_taken, main);
    if (e != hipSuccess) return hip_fail(e, "rs_deal_trainer_deal: swap");
    tr->taken_recorded = true;
    tr->staged = false;
    tr->live_first = tr->staged_first;
    return flag_live_batch(tr, false);
}

// the end of a batch: the shared iteration counter and the discount check of cfr.rs:240-262 (t counts deals over ALL ranks)
int rs_deal_trainer_finish_batch(rs_deal_trainer *tr) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_finish_batch: trainer is NULL");
    tr->t += uint64_t(tr->world) * tr->params.deals_per_batch;   // cfr.rs:226, once per deal
    if (tr->params.discount_interval == 0 || tr->t > tr->params.discount_cap) return RS_OK;   // cfr.rs:240-242
    if (tr->t > tr->threshold) {                                  // cfr.rs:243
        if (tr->tick_br) {                                        // cfr.rs:244-246: calc_br on the table as it is before the sweep
            const bool was_primary = solver_is_primary(tr->solver);   // calc_br reads the table's rows: bring them up to date first
            if (was_primary)
                if (int rc = solver_kept_primary(tr->solver, false)) return rc;
            if (int rc = rs_calc_br(tr->table, tr->tree, tr->last_br)) return rc;
            if (was_primary)
                if (int rc = solver_kept_primary(tr->solver, true)) return rc;
            tr->last_br_t = tr->t;
            tr->have_br = true;
        }
        if (int rc = rs_discount(tr->table, rs_discount_factor(tr->t, tr->params.discount_interval))) return rc;   // cfr.rs:248-261
        tr->threshold = tr->t + tr->params.discount_interval;     // cfr.rs:262
    }
    return RS_OK;
}

int rs_deal_trainer_set_tick_br(rs_deal_trainer *tr, int enable) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_set_tick_br: trainer is NULL");
    tr->tick_br = enable;
    return RS_OK;
}
int rs_deal_trainer_last_br(const rs_deal_trainer *tr, float *out, uint64_t *iterations) {
    if (!tr || !out) return fail(RS_ERR_INVALID, "rs_deal_trainer_last_br: NULL argument");
    if (!tr->have_br) return fail(RS_ERR_INVALID, "rs_deal_trainer_last_br: no discount tick has run calc_br yet");
    out[0] = tr->last_br[0];
    out[1] = tr->last_br[1];
    if (iterations) *iterations = tr->last_br_t;
    return RS_OK;
}
int rs_deal_trainer_calc_br(rs_deal_trainer *tr, float *out) {
    if (!tr || !out) return fail(RS_ERR_INVALID, "rs_deal_trainer_calc_br: NULL argument");
    return rs_calc_br(tr->table, tr->tree, out);
}
// the trainer's own game: board = the cards of board_mask in ascending order (cfr.rs:108-115), cluster ids through get_cluster
// (hole cards first, then the board: cfr.rs:357-365)
int rs_deal_trainer_best_response(rs_deal_trainer *tr, int mode, double *out) {
    if (!tr || !out) return fail(RS_ERR_INVALID, "rs_deal_trainer_best_response: NULL argument");
    const int n_board0 = __builtin_popcountll(tr->params.board_mask);
    if (n_board0 < 3 || n_board0 > 5 || tr->n_rounds > 6 - n_board0) return fail(RS_ERR_UNSUPPORTED, "rs_deal_trainer_best_response: a board of 3..5 cards and at most one betting round per street");
    uint8_t board[5];
    int nb = 0;
    for (int c = 0; c < 52; ++c)
        if (tr->params.board_mask >> c & 1) board[nb++] = uint8_t(c);
    // the run-outs in the lane order of rs_best_response_rounds; round r looks a lane up under the cluster of (hole cards, initial board + its first r new cards)
    const size_t NB = rs_br_runouts(board, n_board0, nullptr);
    std::vector<uint8_t> runouts(NB * 5);
    rs_br_runouts(board, n_board0, runouts.data());
    const int K = 5 - n_board0, D = 52 - n_board0;
    // the ids are a function of ranges, board and abstractions, all fixed for the trainer's life: computed once.  They are built into locals and handed to the trainer only
    // when EVERY (round, player) succeeded -- a failing get_cluster must not leave a half-filled cache behind that the next call would take for the real thing
    const uint32_t *ptrs[RS_MAX_ROUNDS * 2] = {};
    std::vector<uint32_t> cluster[RS_MAX_ROUNDS][2];
    for (int r = 0; r < tr->n_rounds && !tr->br_cluster_ready; ++r) {
        size_t per_prefix = 1;
        for (int i = r; i < K; ++i) per_prefix *= size_t(D - i);
        const size_t n_prefix = NB / per_prefix, nc = size_t(2 + n_board0 + r);
        for (int p = 0; p < 2; ++p) {
            const size_t n = tr->n_hands[p];
            cluster[r][p].assign(n_prefix * n, 0);
            std::vector<uint8_t> cards;
            std::vector<size_t> where;
            for (size_t pf = 0; pf < n_prefix; ++pf) {
                const uint8_t *bc = &runouts[pf * per_prefix * 5];   // the first run-out with this prefix
                for (size_t h = 0; h < n; ++h) {
                    const uint8_t c0 = tr->h_hands[p][2 * h], c1 = tr->h_hands[p][2 * h + 1];
                    bool blocked = false;
                    for (int i = n_board0; i < n_board0 + r; ++i) blocked = blocked || bc[i] == c0 || bc[i] == c1;
                    if (blocked) continue;   // no such deal: the lane carries no weight
                    cards.push_back(c0);
                    cards.push_back(c1);
                    cards.insert(cards.end(), bc, bc + n_board0 + r);
                    where.push_back(pf * n + h);
                }
            }
            std::vector<uint32_t> ids(where.size());
            if (int rc = rs_card_abs_get_cluster(tr->abs[r], cards.data(), where.size(), p, ids.data())) return rc;
            for (size_t i = 0; i < where.size(); ++i) cluster[r][p][where[i]] = ids[i];
            (void)nc;
        }
    }
    if (!tr->br_cluster_ready) {
        for (int r = 0; r < tr->n_rounds; ++r)
            for (int p = 0; p < 2; ++p) tr->br_cluster[r][p] = std::move(cluster[r][p]);
        tr->br_cluster_ready = true;
    }
    for (int r = 0; r < tr->n_rounds; ++r)
        for (int p = 0; p < 2; ++p) ptrs[r * 2 + p] = tr->br_cluster[r][p].data();
    const int which = (mode & RS_BR_SORTED) ? 1 : 0;
    if (!tr->br_prepared[which])
        if (int rc = rs::br_prepare(tr->table, tr->tree, board, n_board0, tr->h_hands[0].data(), tr->n_hands[0], tr->h_hands[1].data(), tr->n_hands[1], ptrs, tr->n_rounds,
                                    which == 1, &tr->br_prepared[which]))
            return rc;
    return rs::br_execute(tr->br_prepared[which], mode & ~RS_BR_SORTED, out);
}

// what the best-response objects a trainer keeps between calls hold on the device (the game-only index of both showdown modes + the walks' workspaces), and a way to give
// the workspaces back (the next call allocates them again)
size_t rs_deal_trainer_br_bytes(const rs_deal_trainer *tr) {
    size_t b = 0;
    if (tr)
        for (int k = 0; k < 2; ++k) b += rs::br_held_bytes(tr->br_prepared[k]);
    return b;
}
int rs_deal_trainer_br_release(rs_deal_trainer *tr) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_br_release: trainer is NULL");
    for (int k = 0; k < 2; ++k) rs::br_release_workspace(tr->br_prepared[k]);
    return RS_OK;
}
int rs_deal_trainer_br_launches(const rs_deal_trainer *tr, int sorted) {   // launches of the last best-response call (level plan), -1: depth-first walk or no call yet
    return tr ? rs::br_last_launches(tr->br_prepared[sorted ? 1 : 0]) : -1;
}

int rs_deal_trainer_attach_comm(rs_deal_trainer *tr, rs_comm *comm) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_attach_comm: trainer is NULL");
    if (comm && tr->ahead) {   // sweeps under a communicator run phase by phase and sort their records themselves
        if (!solver_order_ahead(tr->solver, false, nullptr, nullptr))
            return fail(RS_ERR_UNSUPPORTED, "rs_deal_trainer_attach_comm: attach the communicator before the trainer's first batch");
        tr->ahead = false;
    }
    return rs_solver_attach_comm(tr->solver, comm);
}

// train(): every batch is deals_per_batch iterations of cfr.rs:207-226 on this rank (world times as many over all ranks)
int rs_deal_trainer_train(rs_deal_trainer *tr, uint64_t n_batches) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_train: trainer is NULL");
    // kept shadow records (rs_solver.cpp setup_table_shadow) are the working copy until the loop is over -- when the loop is long enough to pay for writing the table's rows back
    // at its end (a pass over those nodes: 2 ms for the 2 GB of solve_three_street); a caller that trains a batch at a time keeps table and records both up to date
    if (n_batches >= kKeptPrimaryMinTrips)
        if (int rc = solver_kept_primary(tr->solver, true)) return rc;
    int rc = RS_OK;
    for (uint64_t b = 0; b < n_batches && rc == RS_OK; ++b) {
        rc = deal_batch(tr, false, b + 1 == n_batches);
        if (rc == RS_OK) rc = prefetch(tr);   // deal the next batch beside this one's sweeps -- the one after the last as well: it waits in the staging buffers for the next call (a
                                              // batch is a function of the seed and its number, so nothing observable moves; a caller that trains a few batches per call no
                                              // longer pays 0.9 ms of un-overlapped dealing at 4 M deals in front of every call)
        for (int player = 0; player < 2 && rc == RS_OK; ++player) {   // cfr.rs:216-224; with a communicator: sweep, all-reduce the deltas, apply
            rc = rs_iterate(tr->solver, player, nullptr);
            if (rc == RS_OK && tr->ahead && tr->staged) rc = sort_staged(tr, player);   // this traverser's records of the NEXT batch, beside the sweeps that follow
        }
        if (rc == RS_OK) rc = rs_deal_trainer_finish_batch(tr);
    }
    if (tr->ahead && tr->taken_recorded && hipStreamWaitEvent((hipStream_t)rs_stream(tr->table), tr->ev_taken, 0) != hipSuccess)   // the live arrays, for whoever reads them next
        rc = rc != RS_OK ? rc : fail(RS_ERR_HIP, "rs_deal_trainer_train: dealing stream");
    const int rc_off = solver_kept_primary(tr->solver, false);   // the table's rows back from the records
    return rc != RS_OK ? rc : rc_off;
}

// synchronises; fails if any deal since the last call could not be sampled or addressed (the reference would spin or panic)
int rs_deal_trainer_status(rs_deal_trainer *tr) {
    if (!tr) return fail(RS_ERR_INVALID, "rs_deal_trainer_status: trainer is NULL");
    uint32_t err = 0;
    if (tr->deal_stream && hipStreamSynchronize(tr->deal_stream) != hipSuccess) return fail(RS_ERR_HIP, "rs_deal_trainer_status: dealing stream");
    if (int rc = rs_d2h(tr->table, &err, tr->d_err, sizeof(err))) return rc;
    if (err) {
        rs_dmemset(tr->table, tr->d_err, 0, sizeof(err));
        return fail(RS_ERR_INVALID, "generate_hand: no combo of a range fits the board and the other hand (the reference loops forever, cfr.rs:127-137)");
    }
    for (int r = 0; r < tr->n_rounds; ++r)
        if (int rc = rs_card_abs_status(tr->abs[r], tr->table)) return rc;
    return RS_OK;
}

}  // extern "C"
