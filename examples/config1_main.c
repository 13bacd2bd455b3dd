/* examples/config1_main.c -- BASELINE configs[0] in plain C: a 169-bucket, two-action tree adopted from the host's own node records (rs_tree_from_nodes: what a Rust
 * host does with the Tree<GameTreeNode> it already built, tree_builder.rs:9), a zero-filled table from create_infosets' sizes, three cfr() iterations through
 * rs_iterate, and the info set of bucket 90 read back (get-infoset + Infoset::get_strategy, infoset.rs:83-102).  The reference has no preflop round
 * (state.rs:8, :59-64): this is plumbing, the 169 is hand_indexer_s::init(1, [2]).size(0) (gen_abstraction/ehs.rs:30).
 *
 *   gcc -std=c99 -Iinclude examples/config1_main.c -Lrustsolver_amd -lrustsolver_amd -Wl,-rpath,$PWD/rustsolver_amd -o _ab/config1_main
 */
#include <stdio.h>
#include <string.h>

#include "rustsolver_amd.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != RS_OK) {                                                    \
            fprintf(stderr, "%s: error %d: %s\n", #call, rc_, rs_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

enum { N_BUCKETS = 169, POT = 3 };

int main(void) {
    rs_tree_node nodes[6];
    memset(nodes, 0, sizeof(nodes));
    /* 0: private chance -> 1: player 0 {fold -> 2, continue -> 3}; 3: player 1 {fold -> 4, call -> 5 showdown} */
    nodes[0].kind = RS_NODE_PRIVATE_CHANCE; nodes[0].parent = -1; nodes[0].n_children = 1; nodes[0].children[0] = 1;
    nodes[1].kind = RS_NODE_ACTION; nodes[1].parent = 0; nodes[1].n_children = 2; nodes[1].children[0] = 2; nodes[1].children[1] = 3; nodes[1].index = 0; nodes[1].player = 0;
    nodes[2].kind = RS_NODE_TERMINAL; nodes[2].parent = 1; nodes[2].value = POT; nodes[2].ttype = RS_TERM_UNCONTESTED; nodes[2].last_to_act = 0;
    nodes[3].kind = RS_NODE_ACTION; nodes[3].parent = 1; nodes[3].n_children = 2; nodes[3].children[0] = 4; nodes[3].children[1] = 5; nodes[3].index = 1; nodes[3].player = 1;
    nodes[4].kind = RS_NODE_TERMINAL; nodes[4].parent = 3; nodes[4].value = 2 * POT; nodes[4].ttype = RS_TERM_UNCONTESTED; nodes[4].last_to_act = 1;
    nodes[5].kind = RS_NODE_TERMINAL; nodes[5].parent = 3; nodes[5].value = 2 * POT; nodes[5].ttype = RS_TERM_SHOWDOWN; nodes[5].last_to_act = 1;
    rs_tree *tree = NULL;
    rs_table *table = NULL;
    rs_solver *solver = NULL;
    CHECK(rs_tree_from_nodes(nodes, 6, &tree));
    const uint32_t n_clusters[RS_MAX_ROUNDS][RS_MAX_PLAYERS] = {{N_BUCKETS, N_BUCKETS}, {0, 0}, {0, 0}};
    const uint32_t n_boards[RS_MAX_ROUNDS] = {1, 0, 0};
    CHECK(rs_create_infosets(tree, n_clusters, n_boards, RS_I32, 0, &table));   /* create_infosets(n_actions, tree, card_abs), infoset.rs:8 */
    /* the showdown: bucket c of player 0 beats bucket c of player 1 when c is even (one float per lane: sign of score[0] - score[1], cfr.rs:323-347) */
    const size_t pitch = rs_table_lane_pitch(table, 0);
    float sign[256];
    for (size_t c = 0; c < pitch && c < 256; ++c) sign[c] = (c % 3 == 0) ? 0.0f : ((c & 1) ? -1.0f : 1.0f);
    void *d_sign = NULL;
    CHECK(rs_dmalloc(table, pitch * sizeof(float), &d_sign));
    CHECK(rs_h2d(table, d_sign, sign, pitch * sizeof(float)));
    rs_leaf_desc leaves[6];
    memset(leaves, 0, sizeof(leaves));
    leaves[5].kind = RS_LEAF_SIGN;
    leaves[5].d_buf = (const float *)d_sign;
    rs_solver_params params;
    memset(&params, 0, sizeof(params));   /* unnamed fields = the engine's own choices */
    params.scale = 100.0f;                /* cfr.rs:424 */
    params.mode = RS_UPD_CLAMP_I64;
    params.chance_mode = RS_CHANCE_PASS;
    params.opp_mode = RS_OPP_FULL;
    params.fuse_subtrees = rs_jit_available();
    CHECK(rs_solver_create(table, tree, leaves, leaves, &params, &solver));
    for (int it = 0; it < 3; ++it)
        for (int player = 0; player < 2; ++player) CHECK(rs_iterate(solver, player, NULL));
    int32_t regrets[RS_MAX_ACTIONS], ssum[RS_MAX_ACTIONS];
    float sigma[RS_MAX_ACTIONS];
    CHECK(rs_get_infoset(table, 0, 0, 90, regrets, ssum));       /* &self.infosets[0][90] */
    CHECK(rs_get_strategy(table, 0, 0, 90, sigma));
    printf("node 0, bucket 90 after 3 iterations: regrets %d %d, strategy_sum %d %d, strategy %.4f %.4f\n", regrets[0], regrets[1], ssum[0], ssum[1], sigma[0], sigma[1]);
    rs_solver_destroy(solver);
    CHECK(rs_dfree(table, d_sign));
    rs_table_destroy(table);
    rs_tree_destroy(tree);
    return 0;
}
