/*
 * rustsolver_amd_diag.h -- diagnostics of librustsolver_amd.so: what bench.py, the tests and the profiling tools call, NOT part of the
 * drop-in surface that replaces the reference's items (that is rustsolver_amd.h).  Same conventions (status codes, d_* = device pointers,
 * everything on the table's stream).
 */
#ifndef RUSTSOLVER_AMD_DIAG_H
#define RUSTSOLVER_AMD_DIAG_H

#include "rustsolver_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs ---------------------------------------------------------------------------------------------------------- */
/* Synthetic fill on the device (bench / tests): cell value = lo + hash(seed, array, node, action,
 * lane) mod (hi - lo + 1), a pure function mirrored in rustsolver_amd/synth.py. */
int rs_table_fill_random(rs_table *table, uint64_t seed, int64_t regret_lo, int64_t regret_hi, int64_t ssum_lo,
                         int64_t ssum_hi);
/* dst[i] = lo + (hi - lo) * u(seed, i), u in [0,1) from the same hash; i < n */
int rs_fill_uniform_f32(rs_table *table, float *d_dst, size_t n, uint64_t seed, float lo, float hi);
/* the same with element i hashed as index_offset + i: a rank's slice of a per-lane row gets the values the whole row would hold there */
int rs_fill_uniform_f32_at(rs_table *table, float *d_dst, size_t n, uint64_t seed, float lo, float hi, uint64_t index_offset);
/* Fill and checksum keyed by the LOGICAL cell (action node, action, global lane) rather than the element index, for tests that emulate the ranks of a board-sharded sweep
 * one after another: lane_off[round_idx] = global index of the table's first lane on that round (0 for an unsharded table or a replicated round).  The fill writes
 * lo + hash % span (i32 and binary16 tables); the checksum adds, per round, an order-independent sum over the real cells to out[2 * round_idx] (regrets) and
 * out[2 * round_idx + 1] (strategy sums): the sums of the ranks' slices of a round equal the unsharded table's, whatever the pitches and tilings are */
int rs_table_fill_random_logical(rs_table *table, uint64_t seed, int64_t regret_lo, int64_t regret_hi, int64_t ssum_lo, int64_t ssum_hi, const uint64_t *lane_off /* [RS_MAX_ROUNDS] */);
int rs_table_checksum_logical(rs_table *table, const uint64_t *lane_off /* [RS_MAX_ROUNDS] */, uint64_t *out /* [RS_MAX_ROUNDS][2] */);
/* SURVEY.md 8(d) "value-range note": one regret cell in `one_in` (hashed) is overwritten with a value of magnitude 2 100 000 000 .. 2 147 000 000 (either sign), so that the
 * saturating adds of the clamp update (cfr.rs:445-461) are exercised; i32 tables */
int rs_table_plant_saturating(rs_table *table, uint64_t seed, uint32_t one_in);
/* one value in `one_in` of a per-lane f32 row becomes +-magnitude: as an RS_LEAF_UTIL row this drives (scale * reach) * (u - util) beyond 2^31, i.e. onto the exact i64 branch
 * of the clamp update (rs_device.hpp visit_i32) */
int rs_plant_outliers_f32(rs_table *table, float *d_dst, size_t n, uint64_t seed, uint32_t one_in, float magnitude);

/* ---- deal sweeps: how many (deal, round subtree) walks did the LAST sweep of `traverser` make?  out[r] = live-list entries summed over the round subtrees of betting
 * round r (a round whose subtrees walk the whole batch counts n_deals per subtree).  Synchronises.  The unit the deal kernels' costs are quoted per. */
int rs_solver_walk_counts(rs_solver *solver, int traverser, uint64_t *out /* [RS_MAX_ROUNDS] */);
/* ---- data-parallel deal sweeps: the bytes THIS rank has handed to the collectives since the solver was created (the packed delta cells of every all-reduce + every rank's
 * items of every all-gather) and the sweeps that exchanged anything.  bench.py --gpus N --dp-deals 1 reports bytes per batch from it. */
int rs_solver_exchange_bytes(const rs_solver *solver, uint64_t *bytes, uint64_t *sweeps);

/* ---- checks of the generated (hipRTC) kernels without a GPU ---------------------------------------------------------------------- */
/* generate + compile (no GPU needed) the tree-specialised kernels of every chance-free subtree, both traversers */
int rs_jit_check_tree(const rs_tree *tree, int dtype, int mode, int opp_mode, int *n_kernels);
/* the same for deal batches (rs_solver_create_deals): the kernels of every round subtree -- reach-down half and table-updating walk, dense and
 * over a live-deal list, with and without LDS tiles */
int rs_jit_check_tree_deals(const rs_tree *tree, int mode, int opp_mode, int *n_kernels);

/* ---- self-test of the short exact division regret matching uses on i32 tables (rs_device.hpp div_exact_pos) against the compiler's f32 division, on the device:
 * n hashed (regret, sum) pairs; *mismatches must be 0.  first_bad (may be NULL): [2] = the first differing pair. */
int rs_selftest_division(rs_table *table, size_t n, uint64_t seed, uint64_t *mismatches, float *first_bad);

/* ---- table checksum ----------------------------------------------------------------------------------------------------------------
 * out[0] / out[1] = sum over the REAL cells i of regrets / strategy_sum (pitch-padding lanes excluded) of splitmix64(i ^ bits(cell_i) * 0x9E3779B97F4A7C15)
 * mod 2^64: order-independent, so two tables of the same shape and layout hold the same bits iff (with overwhelming probability) the sums agree.
 * Synchronises.  Used to compare whole 135 GB tables (fused against level plan) without moving them to the host. */
int rs_table_checksum(rs_table *table, uint64_t *out /*[2]*/);

/* ---- profiling (bench.py roofline leg) ------------------------------------------------------------ */
enum { RS_K_UPDATE = 0, RS_K_NODE_UTIL = 1, RS_K_REACH = 2, RS_K_CHANCE = 3, RS_K_DISCOUNT = 4, RS_K_STRATEGY = 5,
       RS_K_TREE = 6, RS_K_COUNT = 7 };
typedef struct rs_profile {
    uint64_t launches[RS_K_COUNT];
    double ms[RS_K_COUNT];              /* sum of HIP-event durations on the table's stream */
    double algo_bytes[RS_K_COUNT];      /* sum of algorithmic bytes of those launches */
} rs_profile;
/* the rate (GB/s, read + write) a plain float4 copy of `bytes` reaches on this card, best of three grid sizes: the practical streaming ceiling
 * beside the 8 TB/s specification.  Allocates 2 x bytes for the duration of the call; synchronises. */
int rs_stream_probe(rs_table *table, size_t bytes, int reps, double *gbps);
int rs_profile_enable(rs_table *table, int on); /* on: bracket every launch with hipEvents (adds host work) */
int rs_profile_read(rs_table *table, rs_profile *out); /* synchronises, then accumulates pending events */
int rs_profile_reset(rs_table *table);
/* step boundaries: rs_profile_mark records one event on the table's stream (asynchronous, independent of rs_profile_enable); rs_profile_marks synchronises,
 * writes the min(cap, n) durations in ms between consecutive marks, sets *n_out = n = marks - 1 and forgets the marks */
int rs_profile_mark(rs_table *table);
int rs_profile_marks(rs_table *table, float *ms_out, size_t cap, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* RUSTSOLVER_AMD_DIAG_H */
