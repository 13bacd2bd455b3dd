/* examples/solver_main.c -- the reference's `solver` binary (src/solver/main.rs:29-36) on top of the C ABI, in plain C:
 *     let options = options::default_flop();  let mut trainer = MCCFRTrainer::init(options);  trainer.train(10_000_000);
 * Everything between "pick a deal" and "write the regrets" runs on the GPU (rs_deal_trainer).
 *
 *   gcc -std=c99 -Iinclude examples/solver_main.c -Lrustsolver_amd -lrustsolver_amd -Wl,-rpath,$PWD/rustsolver_amd -o _ab/solver_main
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rustsolver_amd.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != RS_OK) {                                                    \
            fprintf(stderr, "%s: error %d: %s\n", #call, rc_, rs_last_error()); \
            return 1;   /* the reference panics on every error path */         \
        }                                                                      \
    } while (0)

static int card(char rank, char suit) {   /* 4 * rank + suit, rank 0..12 = 2..A (cfr.rs:592) */
    const char *ranks = "23456789TJQKA", *suits = "shdc";
    int r = 0, s = 0;
    while (ranks[r] != rank) ++r;
    while (suits[s] != suit) ++s;
    return 4 * r + s;
}

int main(int argc, char **argv) {
    const unsigned long long iterations = argc > 1 ? strtoull(argv[1], NULL, 10) : 10000000ull;   /* main.rs:33 */
    const unsigned deals_per_batch = 1u << 16;   /* 64 K deals per batch: the time-to-quality sweet spot of this game */
    /* options::default_flop() (options.rs:52-81): board 4d5dAs3cKs, random ranges, pot 35, stacks 500 */
    const char *board = "4d5dAs3cKs";
    uint64_t board_mask = 0;
    for (int i = 0; i < 5; ++i) board_mask |= 1ull << card(board[2 * i], board[2 * i + 1]);
    uint8_t hands[1326][2];
    size_t n_hands = 0;
    for (int a = 0; a < 52; ++a)   /* HandRange "random" + remove_invalid_combos (cfr.rs:161-163) */
        for (int b = 0; b < a; ++b)
            if (!((board_mask >> a) & 1) && !((board_mask >> b) & 1)) {
                hands[n_hands][0] = (uint8_t)a;
                hands[n_hands][1] = (uint8_t)b;
                ++n_hands;
            }
    rs_options options;
    rs_tree *tree = NULL;
    rs_card_abs *river = NULL;
    rs_deal_trainer *trainer = NULL;
    CHECK(rs_options_default(&options));
    CHECK(rs_tree_build(&options, &tree));                                            /* build_game_tree (cfr.rs:165) */
    CHECK(rs_card_abs_create(2, &hands[0][0], n_hands, &hands[0][0], n_hands, board_mask, NULL, 0, &river));   /* ISOMORPHIC river (cfr.rs:171) */
    printf("%zu combos per range, %zu / %zu river clusters, %d action nodes\n", n_hands, rs_card_abs_size(river, 0), rs_card_abs_size(river, 1),
           rs_tree_n_action_nodes(tree));
    rs_deal_trainer_params params;
    memset(&params, 0, sizeof(params));   /* every field this program does not name -- the kernel-form choices (rs_kernel_forms) among them -- means "the engine's own choice" */
    params.board_mask = board_mask;
    params.deals_per_batch = deals_per_batch;
    params.seed = (uint64_t)time(NULL);
    params.discount_interval = 100000;   /* cfr.rs:193 */
    params.discount_cap = 20000000;      /* cfr.rs:194 */
    params.solver.scale = 100.0f;        /* cfr.rs:424 */
    params.solver.mode = RS_UPD_CLAMP_I64;
    params.solver.chance_mode = RS_CHANCE_PASS;
    params.solver.use_graph = 0;         /* at 64 K deals per batch plain launches are faster than graph replays (0.079 against 0.094 ms per batch) */
    params.solver.fuse_subtrees = rs_jit_available();
    params.solver.opp_mode = RS_OPP_SAMPLE;
    params.solver.sample_seed = params.seed;
    params.solver.shard_world = params.solver.shard_rank = params.solver.shard_round = 0;
    params.solver.shard_global_boards = 0;
    params.solver.deal_offset = 0;
    params.world = 1;
    params.rank = 0;
    params.prune_threshold = 10000000;   /* cfr.rs:190 */
    rs_card_abs *abstractions[1] = {river};
    CHECK(rs_deal_trainer_create(tree, abstractions, 1, &hands[0][0], n_hands, &hands[0][0], n_hands, &params, 0, &trainer));
    CHECK(rs_deal_trainer_set_tick_br(trainer, 1));   /* "calc br" at every discount tick (cfr.rs:244-246) */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    CHECK(rs_deal_trainer_train(trainer, (iterations + deals_per_batch - 1) / deals_per_batch));   /* trainer.train(10_000_000) */
    CHECK(rs_deal_trainer_status(trainer));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("%llu iterations in %.3f ms (%.3g iterations/s)\n", (unsigned long long)rs_deal_trainer_iterations(trainer), s * 1e3,
           (double)rs_deal_trainer_iterations(trainer) / s);
    float br[2];
    uint64_t br_t = 0;
    if (rs_deal_trainer_last_br(trainer, br, &br_t) == RS_OK) printf("calc br (as coded, last tick at %llu)\n%g %g\n", (unsigned long long)br_t, br[0], br[1]);
    double brv[2], ev[2];
    CHECK(rs_deal_trainer_best_response(trainer, RS_BR_MAX, brv));
    CHECK(rs_deal_trainer_best_response(trainer, RS_BR_AVERAGE, ev));
    printf("average strategy: value %.4f / %.4f per deal, best responses %.4f / %.4f, exploitability %.4f\n", ev[0], ev[1], brv[0], brv[1],
           (brv[0] + brv[1]) / 2.0);
    float sigma[RS_MAX_ACTIONS];
    CHECK(rs_get_final_strategy(rs_deal_trainer_table(trainer), 0, 0, 0, sigma));   /* Infoset::get_final_strategy of the root, cluster 0 */
    printf("root, cluster 0: average strategy %.3f %.3f %.3f\n", sigma[0], sigma[1], sigma[2]);
    rs_deal_trainer_destroy(trainer);
    rs_card_abs_destroy(river);
    rs_tree_destroy(tree);
    return 0;
}
