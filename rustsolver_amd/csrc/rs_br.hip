// rs_br.hip -- the readers of the AVERAGE strategy.
//
//  * rs_calc_br: MCCFRTrainer::calc_br exactly as coded (cfr.rs:629-745), the two numbers train() prints at every discount
//    tick (cfr.rs:244-246).  Its `op` vectors have length 1, so of the n_buckets final strategies abstract_br_infoset collects
//    (cfr.rs:669-672) only bucket 0 of every action node is used: ONE kernel gathers get_final_strategy() of lane 0 of every node
//    (the table stays in HBM, n_nodes x 8 floats come back), and the walk itself -- a handful of f32 operations per tree node -- runs
//    on the host in the reference's operation order.
//  * rs_best_response: what that placeholder stands in for: the value of a best response to the opponent's average strategy in the
//    abstracted game (and, mode RS_BR_AVERAGE, the value of the average strategy profile itself), in vector form over the two hand
//    ranges of a single-round game on a full board -- the configuration the reference ships (options::default_flop()).  Reach and
//    value vectors live on the device as f64; every sum runs in a fixed order, so the CPU oracle reproduces the result bit for bit.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <memory>
#include <vector>

#include "rs_internal.hpp"
#include "rs_device.hpp"
#include "rs_eval.hpp"

#pragma clang fp contract(off)

using namespace rs;

#define RS_HIP(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

namespace rs {

constexpr int kBrBlock = 256;

template <int DT> struct Elem;
template <> struct Elem<kDT_I32> {
    using val = int;
    static __device__ __forceinline__ val at(const void *base, size_t i) { return ((const int *)base)[i]; }
};
template <> struct Elem<kDT_F32> {
    using val = float;
    static __device__ __forceinline__ val at(const void *base, size_t i) { return ((const float *)base)[i]; }
};
template <> struct Elem<kDT_F16> {
    using val = float;
    static __device__ __forceinline__ val at(const void *base, size_t i) { return (float)((const _Float16 *)base)[i]; }
};

// Infoset::get_final_strategy (infoset.rs:104-123) of ONE lane of a node's strategy_sum rows [A][pitch]
template <int DT>
__device__ __forceinline__ void final_sigma(const void *ssum, size_t cell_off, uint32_t pitch, uint32_t n_actions, uint32_t lane,
                                            float (&sig)[RS_MAX_ACTIONS]) {
    using V = typename Elem<DT>::val;
    V r[RS_MAX_ACTIONS];
    float norm = 0.0f;
    for (uint32_t a = 0; a < n_actions; a++) {
        r[a] = Elem<DT>::at(ssum, cell_off + (size_t)a * pitch + lane);
        if (r[a] > (V)0) norm += (float)r[a];
    }
    const float uni = 1.0f / (float)n_actions;
    for (uint32_t a = 0; a < n_actions; a++) sig[a] = (norm > 0.0f) ? ((r[a] > (V)0) ? (float)r[a] / norm : 0.0f) : uni;
}

struct BrNodeRow {   // one action node's strategy_sum block
    uint64_t cell_off;
    uint32_t pitch, n_actions;
};

// calc_br's reader: probabilites[0] of every action node (cfr.rs:669-672 with :677-679)
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_bucket0_final_strategy(const void *__restrict__ ssum, const BrNodeRow *__restrict__ rows, uint32_t n_nodes,
                                                                      float *__restrict__ out /*[n_nodes][RS_MAX_ACTIONS]*/) {
    const uint32_t n = blockIdx.x * kBrBlock + threadIdx.x;
    if (n >= n_nodes) return;
    const BrNodeRow row = rows[n];
    float sig[RS_MAX_ACTIONS];
    final_sigma<DT>(ssum, row.cell_off, row.pitch, row.n_actions, 0u, sig);
    for (uint32_t a = 0; a < RS_MAX_ACTIONS; a++) out[(size_t)n * RS_MAX_ACTIONS + a] = a < row.n_actions ? sig[a] : 0.0f;
}

// ---- best response, vector form ------------------------------------------------------------------------------------------
// score[h] of hole cards + the five board cards (the evaluator the showdown-sign kernel uses; only compared, cfr.rs:326-333)
__global__ __launch_bounds__(kBrBlock) void k_hand_scores(const uint8_t *__restrict__ hands /*[n][2]*/, uint32_t n, uint32_t b0, uint32_t b1, uint32_t b2,
                                                          uint32_t b3, uint32_t b4, uint32_t *__restrict__ score) {
    const uint32_t h = blockIdx.x * kBrBlock + threadIdx.x;
    if (h >= n) return;
    uint32_t m[4] = {0, 0, 0, 0};
    add_card(m, b0); add_card(m, b1); add_card(m, b2); add_card(m, b3); add_card(m, b4);
    add_card(m, hands[2 * h]);
    add_card(m, hands[2 * h + 1]);
    score[h] = evaluate_suits(m);
}

// opponent node: q_a[h] = q[h] * sigma_bar(cluster(h), a) for every action (cfr.rs:585's reach product, over the whole range at once)
template <int DT>
__device__ __forceinline__ void br_opp_reach_body(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ cid, uint32_t n,
                                                           uint32_t n_pad, const double *__restrict__ q, double *__restrict__ q_out /*[A][n_pad]*/) {
    const uint32_t h = blockIdx.x * kBrBlock + threadIdx.x;
    if (h >= n) return;
    float sig[RS_MAX_ACTIONS];
    final_sigma<DT>(ssum, row.cell_off, row.pitch, row.n_actions, cid[h], sig);
    const double qh = q[h];
    for (uint32_t a = 0; a < row.n_actions; a++) q_out[(size_t)a * n_pad + h] = qh * (double)sig[a];
}

// terminal: v[hp] = pw[hp] * sum over the opponent's hands ho that share no card with hp, ascending, of q[ho] * u(hp, ho);
// u as the trainer's leaves (cfr.rs:314-348): UNCONTESTED -pot for the folder / +pot for the other, SHOWDOWN and ALLIN +-pot by score, 0 on a tie
__global__ __launch_bounds__(kBrBlock) void k_br_terminal(const uint64_t *__restrict__ mask_p, const uint32_t *__restrict__ score_p, const double *__restrict__ pw,
                                                          uint32_t n_p, const uint64_t *__restrict__ mask_o, const uint32_t *__restrict__ score_o,
                                                          const double *__restrict__ q, uint32_t n_o, int uncontested, double value, double *__restrict__ v) {
    const uint32_t hp = blockIdx.x * kBrBlock + threadIdx.x;
    if (hp >= n_p) return;
    const uint64_t mp = mask_p[hp];
    const uint32_t sp = score_p[hp];
    double acc = 0.0;
    for (uint32_t ho = 0; ho < n_o; ho++) {
        if (mp & mask_o[ho]) continue;
        const uint32_t so = score_o[ho];
        const double u = uncontested ? value : (sp > so ? value : (sp < so ? -value : 0.0));
        acc += q[ho] * u;
    }
    v[hp] = pw[hp] * acc;
}

// opponent node, on the way up: v[h] = v_0[h] + v_1[h] + ... in action order
__device__ __forceinline__ void br_sum_body(const double *__restrict__ vch /*[A][n_pad]*/, uint32_t n_actions, uint32_t n, uint32_t n_pad,
                                                     double *__restrict__ v) {
    const uint32_t h = blockIdx.x * kBrBlock + threadIdx.x;
    if (h >= n) return;
    double acc = 0.0;
    for (uint32_t a = 0; a < n_actions; a++) acc += vch[(size_t)a * n_pad + h];
    v[h] = acc;
}

// own node: one thread per info set (cluster).  RS_BR_MAX: the action with the largest value summed over the cluster's hands
// (ascending hand order; first maximum, strict < as cfr.rs:684-690) is played by every hand of the cluster; RS_BR_AVERAGE: sigma_bar.
template <int DT>
__device__ __forceinline__ void br_own_body(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ start /*[n_clusters + 2]*/,
                                                     const uint32_t *__restrict__ order, uint32_t n_clusters, uint32_t n_pad, const double *__restrict__ vch,
                                                     int mode, double *__restrict__ v) {
    // lanes that are no deal at all (the hand uses a card of the run-out) are listed after the last cluster: worth 0, summed nowhere (x + 0.0 == x)
    for (uint32_t i = start[n_clusters] + blockIdx.x * kBrBlock + threadIdx.x; i < start[n_clusters + 1]; i += gridDim.x * kBrBlock) v[order[i]] = 0.0;
    const uint32_t c = blockIdx.x * kBrBlock + threadIdx.x;
    if (c >= n_clusters) return;
    const uint32_t lo = start[c], hi = start[c + 1];
    if (lo == hi) return;
    if (mode == RS_BR_MAX && hi - lo <= 8) {   // a river info set holds a handful of lanes: its lane ids once, then every (action, lane) value in flight together -- two round
                                               // trips instead of two per value; the additions in list order as below
        uint32_t idx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) idx[k] = lo + k < hi ? order[lo + k] : order[lo];
        double s[RS_MAX_ACTIONS];
        uint32_t best = 0;
        for (uint32_t a = 0; a < row.n_actions; a++) {
            const double *va = vch + (size_t)a * n_pad;
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = va[idx[k]];
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (lo + k < hi) acc += t[k];
            s[a] = acc;
            if (a > 0 && s[best] < s[a]) best = a;
        }
        const double *vb = vch + (size_t)best * n_pad;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (lo + k < hi) v[idx[k]] = vb[idx[k]];
        return;
    }
    if (mode == RS_BR_MAX) {
        double s[RS_MAX_ACTIONS];
        for (uint32_t a = 0; a < row.n_actions; a++) {
            const double *va = vch + (size_t)a * n_pad;
            double acc = 0.0;
            uint32_t i = lo;
            for (; i + 8 <= hi; i += 8) {   // eight gathers in flight, the additions still one after the other in list order
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; k++) t[k] = va[order[i + k]];
#pragma unroll
                for (int k = 0; k < 8; k++) acc += t[k];
            }
            for (; i < hi; i++) acc += va[order[i]];
            s[a] = acc;
        }
        uint32_t best = 0;
        for (uint32_t a = 1; a < row.n_actions; a++)
            if (s[best] < s[a]) best = a;
        for (uint32_t i = lo; i < hi; i++) v[order[i]] = vch[(size_t)best * n_pad + order[i]];
    } else {
        float sig[RS_MAX_ACTIONS];
        final_sigma<DT>(ssum, row.cell_off, row.pitch, row.n_actions, c, sig);
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t h = order[i];
            double acc = 0.0;
            for (uint32_t a = 0; a < row.n_actions; a++) acc += (double)sig[a] * vch[(size_t)a * n_pad + h];
            v[h] = acc;
        }
    }
}

static unsigned grid1(uint32_t n) { return (n + kBrBlock - 1) / kBrBlock ? (n + kBrBlock - 1) / kBrBlock : 1; }

#define RS_BR_DT(dtype_, CALL)                 \
    do {                                       \
        if ((dtype_) == RS_I32) { CALL(kDT_I32); } \
        else if ((dtype_) == RS_F32) { CALL(kDT_F32); } \
        else { CALL(kDT_F16); }                \
    } while (0)

// ---- host side of calc_br: cfr.rs:640-745 over the gathered bucket-0 strategies -------------------------------------------------
struct Pay { float v[2]; };   // res[player][0]
struct AsCoded {
    const std::vector<rs_tree_node> &nodes;
    const std::vector<float> &prob;   // [index][RS_MAX_ACTIONS]
    Pay terminal(const rs_tree_node &tn, const float (&op)[2]) const {   // cfr.rs:695-745
        Pay res{{0.0f, 0.0f}};
        const float money_f = float(tn.value);
        for (int p = 0; p < 2; ++p) {
            const int opp = 1 - p;
            float opp_ges = 0.0f;
            float payoff;
            if (tn.ttype == RS_TERM_UNCONTESTED) payoff = op[opp] * (p == int(tn.last_to_act) ? -1.0f : 1.0f) * money_f;   // cfr.rs:712
            else payoff = op[opp] * money_f;                                                                                // cfr.rs:726
            res.v[p] += payoff;
            opp_ges += op[opp];
            res.v[p] *= 1.0f / opp_ges;                                                                                     // cfr.rs:716 / :730
        }
        return res;
    }
    Pay walk(int id, const float (&op)[2]) const {
        const rs_tree_node &n = nodes[size_t(id)];
        if (n.kind == RS_NODE_TERMINAL) return terminal(n, op);
        if (n.kind != RS_NODE_ACTION) return walk(n.children[0], op);   // cfr.rs:646-651: both chance kinds go to child 0
        Pay pay[RS_MAX_ACTIONS];
        const int player = n.player, opp = 1 - n.player;
        for (int a = 0; a < n.n_children; ++a) {
            float newop[2] = {op[0], op[1]};
            newop[player] *= prob[size_t(n.index) * RS_MAX_ACTIONS + size_t(a)];   // cfr.rs:677-679
            pay[a] = walk(n.children[a], newop);
        }
        float max_val = pay[0].v[player];
        int max_index = 0;
        for (int a = 1; a < n.n_children; ++a)
            if (max_val < pay[a].v[player]) {   // cfr.rs:686
                max_val = pay[a].v[player];
                max_index = a;
            }
        Pay res{{0.0f, 0.0f}};
        res.v[player] = max_val;
        res.v[opp] = pay[max_index].v[opp];
        return res;
    }
};

static int check_tree_against_table(const rs_table *t, const rs_tree *tree, const char *fn) {
    for (const rs_tree_node &n : tree->nodes) {
        if (n.kind != RS_NODE_ACTION) continue;
        if (n.index < 0 || size_t(n.index) >= t->nodes.size()) return fail(RS_ERR_INVALID, std::string(fn) + ": the tree has an action node the table lacks");
        const rs_node_desc &nd = t->nodes[size_t(n.index)];
        if (int(nd.n_actions) != n.n_children || nd.n_actions == 0 || nd.n_actions > RS_MAX_ACTIONS)
            return fail(RS_ERR_INVALID, std::string(fn) + ": action counts of tree and table differ at node " + std::to_string(n.index));
        if (nd.n_clusters == 0) return fail(RS_ERR_OOB, std::string(fn) + ": node " + std::to_string(n.index) + " has no bucket 0 (Rust: index out of bounds)");
    }
    return RS_OK;
}

static BrNodeRow row_of(const rs_table *t, int index) {
    // rows are tile[] elements apart inside the first tile of a (possibly tiled) node block: all that bucket 0 / a one-board table ever touches
    return BrNodeRow{uint64_t(t->cell_off[size_t(index)]), uint32_t(t->tile[size_t(index)]), t->nodes[size_t(index)].n_actions};
}

}  // namespace rs

extern "C" {

int rs_calc_br(rs_table *t, const rs_tree *tree, float *out) {
    if (!t || !tree || !out) return fail(RS_ERR_INVALID, "rs_calc_br: NULL argument");
    if (tree->nodes.empty()) return fail(RS_ERR_INVALID, "rs_calc_br: empty tree");
    if (int rc = check_tree_against_table(t, tree, "rs_calc_br")) return rc;
    if (int rc = table_settle(t)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const uint32_t n_nodes = uint32_t(t->nodes.size());
    std::vector<BrNodeRow> rows(n_nodes);
    for (uint32_t n = 0; n < n_nodes; ++n) rows[n] = row_of(t, int(n));
    std::vector<float> prob(size_t(n_nodes) * RS_MAX_ACTIONS);
    if (n_nodes) {
        BrNodeRow *d_rows = nullptr;
        float *d_prob = nullptr;
        RS_HIP(hipMalloc(&d_rows, rows.size() * sizeof(BrNodeRow)), "hipMalloc(calc_br rows)");
        hipError_t e = hipMalloc(&d_prob, prob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(BrNodeRow), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
#define RS_B0(DT_) hipLaunchKernelGGL((k_bucket0_final_strategy<DT_>), dim3(grid1(n_nodes)), dim3(kBrBlock), 0, t->stream, t->d_ssum, d_rows, n_nodes, d_prob)
            RS_BR_DT(t->dtype, RS_B0);
#undef RS_B0
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(prob.data(), d_prob, prob.size() * sizeof(float), hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        (void)hipFree(d_rows);
        if (d_prob) (void)hipFree(d_prob);
        if (e != hipSuccess) return hip_fail(e, "rs_calc_br");
    }
    const AsCoded walk{tree->nodes, prob};
    const float op[2] = {1.0f, 1.0f};   // cfr.rs:631
    const Pay r = walk.walk(0, op);
    out[0] = r.v[0];                    // cfr.rs:633-636
    out[1] = r.v[1];
    return RS_OK;
}

}  // extern "C"

// ---- best response proper -----------------------------------------------------------------------------------------------
// Lanes are (run-out b, hand h), lane = b * n_hands + h: generate_hand (cfr.rs:100-143) completes the board to five cards before it draws the hands, every showdown
// compares seven-card hands on the FULL board and chance nodes pass through (cfr.rs:306-313), so every node carries vectors over all NB * n lanes; the betting round
// of a node only selects which cluster id a lane is looked up under.  A single-round game on a full board is NB = 1.
namespace rs {

// seven-card scores of every lane (0 where the hand uses a card of the run-out: no such deal)
__global__ __launch_bounds__(kBrBlock) void k_lane_scores(const uint8_t *__restrict__ hands /*[n][2]*/, uint32_t n, const uint8_t *__restrict__ boards /*[NB][5]*/, uint32_t n_board0,
                                                          uint32_t NB, uint32_t *__restrict__ score) {
    const size_t lane = (size_t)blockIdx.x * kBrBlock + threadIdx.x;
    if (lane >= (size_t)NB * n) return;
    const uint32_t b = (uint32_t)(lane / n), h = (uint32_t)(lane - (size_t)b * n);
    const uint32_t c0 = hands[2 * h], c1 = hands[2 * h + 1];
    uint32_t m[4] = {0, 0, 0, 0};
    bool blocked = false;
    for (uint32_t i = 0; i < 5; i++) {
        const uint32_t c = boards[b * 5 + i];
        add_card(m, c);
        if (i >= n_board0 && (c == c0 || c == c1)) blocked = true;
    }
    add_card(m, c0);
    add_card(m, c1);
    score[lane] = blocked ? 0u : evaluate_suits(m);
}

// the deal distribution: w0[b, h0] = P(B) / (N0(B) * N1(B, h0)) (0 for a blocked lane or when no h1 fits)
__global__ __launch_bounds__(kBrBlock) void k_br_weights(const uint64_t *__restrict__ mask0, uint32_t n0, const uint64_t *__restrict__ mask1, uint32_t n1,
                                                         const uint64_t *__restrict__ bmask, const uint32_t *__restrict__ cnt0, uint32_t NB, double pb,
                                                         double *__restrict__ w0) {
    const size_t lane = (size_t)blockIdx.x * kBrBlock + threadIdx.x;
    if (lane >= (size_t)NB * n0) return;
    const uint32_t b = (uint32_t)(lane / n0), h = (uint32_t)(lane - (size_t)b * n0);
    const uint64_t mine = mask0[h], bm = bmask[b];
    if (mine & bm) {
        w0[lane] = 0.0;
        return;
    }
    uint32_t cnt1 = 0;
    for (uint32_t g = 0; g < n1; g++) cnt1 += ((mask1[g] & bm) == 0 && (mask1[g] & mine) == 0) ? 1u : 0u;
    w0[lane] = cnt1 ? pb / ((double)cnt0[b] * (double)cnt1) : 0.0;
}

// terminal: v[b, hp] = pw[b, hp] * sum over the opponent's hands ho of the SAME run-out that share no card with hp or the run-out, ascending, of q[b, ho] * u;
// u as the trainer's leaves (cfr.rs:314-348).  One workgroup = up to 256 hands of one run-out; the opponent's side of that run-out is staged in LDS.
__device__ __forceinline__ void br_terminal_boards_body(const uint64_t *__restrict__ mask_p, const uint32_t *__restrict__ score_p, const double *__restrict__ pw,
                                                                 uint32_t n_p, const uint64_t *__restrict__ mask_o, const uint32_t *__restrict__ score_o,
                                                                 const double *__restrict__ q, uint32_t n_o, const uint64_t *__restrict__ bmask, int uncontested,
                                                                 double value, double *__restrict__ v) {
    extern __shared__ unsigned char br_lds[];
    double *lq = (double *)br_lds;
    uint64_t *lm = (uint64_t *)(lq + n_o);
    uint32_t *ls = (uint32_t *)(lm + n_o);
    const uint32_t b = blockIdx.y;
    const uint64_t bm = bmask[b];
    for (uint32_t g = threadIdx.x; g < n_o; g += kBrBlock) {
        lq[g] = q[(size_t)b * n_o + g];
        lm[g] = mask_o[g];
        ls[g] = score_o[(size_t)b * n_o + g];
    }
    __syncthreads();
    const uint32_t hp = blockIdx.x * kBrBlock + threadIdx.x;
    if (hp >= n_p) return;
    const size_t lane = (size_t)b * n_p + hp;
    const uint64_t mp = mask_p[hp];
    if (mp & bm) {
        v[lane] = 0.0;
        return;
    }
    const uint32_t sp = score_p[lane];
    const uint64_t avoid = mp | bm;
    double acc = 0.0;
    for (uint32_t ho = 0; ho < n_o; ho++) {
        if (lm[ho] & avoid) continue;
        const uint32_t so = ls[ho];
        const double u = uncontested ? value : (sp > so ? value : (sp < so ? -value : 0.0));
        acc += lq[ho] * u;
    }
    v[lane] = pw[lane] * acc;
}

// ---- showdowns by rank order (RS_BR_SORTED): O(n log n) per run-out instead of the O(n^2) pair loop of cfr.rs:323-347 ------------------------------------------------
// For a showdown leaf, a traverser hand h collects value * (sum of q over the opponent hands it beats - sum over those that beat it), over the opponent hands that share
// no card with h or the run-out.  With the opponent's hands of a run-out SORTED by (score, index) -- the scores do not depend on the tree node, so the order and every
// position below are computed once per call (k_br_index) -- that is a difference of prefix sums P over the sorted order, corrected for the at most 2 x 51 opponent hands
// that hold one of h's cards through per-card prefix sums Pc:
//     win  = (P[nl - 1] - Pc[c0][kl0 - 1]) - Pc[c1][kl1 - 1]                     nl  = opponent hands with a smaller score, kl_t = those among the holders of h's card t
//     lose = ((T - P[nle - 1]) - (Tc[c0] - Pc[c0][kle0 - 1])) - (Tc[c1] - Pc[c1][kle1 - 1])      nle, kle_t = ... with a smaller or equal score; T, Tc = totals
//     acc  = value * (win - lose)
// (a hand holding BOTH cards is the same hand: equal score, in neither sum).  An uncontested leaf is value * (((T - Tc[c0]) - Tc[c1]) + q[same hand]).
// Every sum has ONE fixed order, which the oracle (orc_best_response_rounds, sorted mode) follows to the bit: P in 64 chunks of ceil(n / 64) consecutive sorted positions --
// chunk sums sequential from 0.0, chunk offsets sequential over the chunks, P[i] = offset + the sequential sum inside the chunk up to i -- and Pc sequential per card.
constexpr int kBrCardHolders = 51, kBrSortN = 2048;   // a card is held by at most 51 two-card hands; 1 326 hands at most, padded to a power of two for the sort
struct BrIndex {   // of one side as the OPPONENT (sorted hands, holders of every card) and of the other side's hands as the traverser (positions), all per run-out
    uint16_t *ord = nullptr;       // [NB][n_o] opponent hand at sorted position i (i < nv[b])
    uint16_t *nv = nullptr;        // [NB] opponent hands that avoid the run-out
    uint16_t *cl = nullptr;        // [NB][52][51] holders of card c in sorted order
    uint8_t *cc = nullptr;         // [NB][52] their number
    uint16_t *nl = nullptr, *nle = nullptr;   // [NB][n_p]
    uint8_t *kl = nullptr, *kle = nullptr;    // [NB][n_p][2]
    int16_t *same = nullptr;       // [n_p] the opponent's hand with the same two cards, -1 = none
};
__global__ __launch_bounds__(kBrBlock) void k_br_index(const uint8_t *__restrict__ hands_p, uint32_t n_p, const uint64_t *__restrict__ mask_p, const uint32_t *__restrict__ score_p,
                                                       const uint8_t *__restrict__ hands_o, uint32_t n_o, const uint64_t *__restrict__ mask_o,
                                                       const uint32_t *__restrict__ score_o, const uint64_t *__restrict__ bmask, BrIndex ix) {
    __shared__ unsigned long long key[kBrSortN];                 // (score << 16) | hand, ~0 = padding
    __shared__ uint16_t lcl[52][kBrCardHolders];
    __shared__ uint32_t lcs[52][kBrCardHolders];                 // the holders' scores
    __shared__ uint32_t lcc[52];
    __shared__ uint32_t lnv;
    const uint32_t b = blockIdx.x;
    const uint64_t bm = bmask[b];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kBrSortN; i += kBrBlock)
        key[i] = (i < n_o && !(mask_o[i] & bm)) ? ((unsigned long long)score_o[(size_t)b * n_o + i] << 16 | i) : ~0ull;
    if (threadIdx.x == 0) lnv = 0;
    __syncthreads();
    for (uint32_t k = 2; k <= (uint32_t)kBrSortN; k <<= 1)       // bitonic sort, ascending
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < (uint32_t)kBrSortN; i += kBrBlock) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long x = key[i], y = key[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        key[i] = y;
                        key[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n_o; i += kBrBlock) {
        if (key[i] != ~0ull) {
            ix.ord[(size_t)b * n_o + i] = (uint16_t)(key[i] & 0xffffu);
            if (i + 1 == n_o || key[i + 1] == ~0ull) lnv = i + 1;
        } else ix.ord[(size_t)b * n_o + i] = 0;
    }
    __syncthreads();
    const uint32_t nv = lnv;
    if (threadIdx.x == 0) ix.nv[b] = (uint16_t)nv;
    if (threadIdx.x < 52) {                                      // the holders of card c, in sorted order
        const uint32_t c = threadIdx.x;
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < nv; ++i) {
            const uint32_t g = (uint32_t)(key[i] & 0xffffu);
            if (hands_o[2 * g] == c || hands_o[2 * g + 1] == c) {
                if (cnt < (uint32_t)kBrCardHolders) {
                    lcl[c][cnt] = (uint16_t)g;
                    lcs[c][cnt] = (uint32_t)(key[i] >> 16);
                    ix.cl[((size_t)b * 52 + c) * kBrCardHolders + cnt] = (uint16_t)g;
                }
                ++cnt;
            }
        }
        lcc[c] = min(cnt, (uint32_t)kBrCardHolders);
        ix.cc[(size_t)b * 52 + c] = (uint8_t)lcc[c];
    }
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < n_p; h += kBrBlock) {     // where every traverser hand stands in the opponent's order
        const size_t lane = (size_t)b * n_p + h;
        if (mask_p[h] & bm) {
            ix.nl[lane] = ix.nle[lane] = 0;
            ix.kl[2 * lane] = ix.kl[2 * lane + 1] = ix.kle[2 * lane] = ix.kle[2 * lane + 1] = 0;
            continue;
        }
        const unsigned long long sp = score_p[lane];
        uint32_t lo = 0, hi = nv;                                // first position with score >= sp
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((key[mid] >> 16) < sp) lo = mid + 1; else hi = mid;
        }
        ix.nl[lane] = (uint16_t)lo;
        hi = nv;                                                 // first position with score > sp
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((key[mid] >> 16) <= sp) lo = mid + 1; else hi = mid;
        }
        ix.nle[lane] = (uint16_t)lo;
        for (int t = 0; t < 2; ++t) {
            const uint32_t c = hands_p[2 * h + t];
            uint32_t a = 0, e = 0;
            for (uint32_t j = 0; j < lcc[c]; ++j) {
                a += lcs[c][j] < sp ? 1u : 0u;
                e += lcs[c][j] <= sp ? 1u : 0u;
            }
            ix.kl[2 * lane + t] = (uint8_t)a;
            ix.kle[2 * lane + t] = (uint8_t)e;
        }
    }
}
// one workgroup per run-out: wave 0 builds P (chunked, see above) and the per-card prefixes in LDS, then all threads evaluate the traverser's hands
__device__ __forceinline__ void br_terminal_sorted_body(const uint8_t *__restrict__ hands_p, const uint64_t *__restrict__ mask_p, const double *__restrict__ pw,
                                                                 uint32_t n_p, const double *__restrict__ q, uint32_t n_o, const uint64_t *__restrict__ bmask, BrIndex ix,
                                                                 int uncontested, double value, double *__restrict__ v) {
    extern __shared__ unsigned char br_lds[];
    double *P = (double *)br_lds;                                // [n_o]
    double *Pc = P + n_o;                                        // [52][51]
    double *O = Pc + 52 * kBrCardHolders;                        // [65] chunk offsets, O[64] = total
    double *Q = O + 65;                                          // [n_o] the opponent's reach of this run-out: the scans below walk it in rank and card order
    const uint32_t b = blockIdx.x;
    const uint32_t nv = ix.nv[b];
    const double *__restrict__ qg = q + (size_t)b * n_o;
    for (uint32_t i = threadIdx.x; i < n_o; i += kBrBlock) Q[i] = qg[i];   // one coalesced pass; the dependent gathers then stay inside the LDS
    __syncthreads();
    const double *qb = Q;
    const uint16_t *__restrict__ ord = ix.ord + (size_t)b * n_o;
    const uint32_t len = (nv + 63u) / 64u;
    if (threadIdx.x < 64) {
        const uint32_t k = threadIdx.x, i0 = min(nv, k * len), i1 = min(nv, i0 + len);
        double sum = 0.0;
        for (uint32_t i = i0; i < i1; ++i) sum += qb[ord[i]];
        O[k] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                      // offsets: sequential over the chunks
        double run = 0.0;
        for (uint32_t k = 0; k < 64; ++k) {
            const double c = O[k];
            O[k] = run;
            run += c;
        }
        O[64] = run;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t k = threadIdx.x, i0 = min(nv, k * len), i1 = min(nv, i0 + len);
        double local = 0.0;
        for (uint32_t i = i0; i < i1; ++i) {
            local += qb[ord[i]];
            P[i] = O[k] + local;
        }
    } else if (threadIdx.x < 64 + 52) {
        const uint32_t c = threadIdx.x - 64, cnt = ix.cc[(size_t)b * 52 + c];
        const uint16_t *__restrict__ cl = ix.cl + ((size_t)b * 52 + c) * kBrCardHolders;
        double run = 0.0;
        for (uint32_t j = 0; j < cnt; ++j) {
            run += qb[cl[j]];
            Pc[c * kBrCardHolders + j] = run;
        }
    }
    __syncthreads();
    const double T = O[64];
    const uint64_t bm = bmask[b];
    for (uint32_t h = threadIdx.x; h < n_p; h += kBrBlock) {
        const size_t lane = (size_t)b * n_p + h;
        if (mask_p[h] & bm) {
            v[lane] = 0.0;
            continue;
        }
        const uint32_t c0 = hands_p[2 * h], c1 = hands_p[2 * h + 1];
        const uint32_t n0 = ix.cc[(size_t)b * 52 + c0], n1 = ix.cc[(size_t)b * 52 + c1];
        const double T0 = n0 ? Pc[c0 * kBrCardHolders + n0 - 1] : 0.0, T1 = n1 ? Pc[c1 * kBrCardHolders + n1 - 1] : 0.0;
        double acc;
        if (uncontested) {
            const int sm = ix.same[h];
            acc = value * (((T - T0) - T1) + (sm >= 0 ? qb[sm] : 0.0));
        } else {
            const uint32_t nl = ix.nl[lane], nle = ix.nle[lane];
            const uint32_t a0 = ix.kl[2 * lane], a1 = ix.kl[2 * lane + 1], e0 = ix.kle[2 * lane], e1 = ix.kle[2 * lane + 1];
            const double L = nl ? P[nl - 1] : 0.0, LE = nle ? P[nle - 1] : 0.0;
            const double L0 = a0 ? Pc[c0 * kBrCardHolders + a0 - 1] : 0.0, L1 = a1 ? Pc[c1 * kBrCardHolders + a1 - 1] : 0.0;
            const double E0 = e0 ? Pc[c0 * kBrCardHolders + e0 - 1] : 0.0, E1 = e1 ? Pc[c1 * kBrCardHolders + e1 - 1] : 0.0;
            const double win = (L - L0) - L1;
            const double lose = ((T - LE) - (T0 - E0)) - (T1 - E1);
            acc = value * (win - lose);
        }
        v[lane] = pw[lane] * acc;
    }
}

// one JOB per node of the level plan (all nodes of one tree depth and kind in one launch)
struct BrJob {
    BrNodeRow row;                 // action nodes: the node's strategy_sum block
    const uint32_t *cid;           // opponent node: the opponent's lane -> cluster of the node's round
    const uint32_t *start, *order; // own node: the lanes of every info set of the node's round
    const double *q;               // opponent node: the reach that comes in; terminal: the opponent's reach
    double *q_out;                 // opponent node: [A][n_pad] reach of its children
    const double *vch;             // action node: [A][n_pad] values of its children
    double *v;                     // where the node's own value goes (a slot of its parent's vch, or the root vector)
    uint32_t n_clusters, n_children;
    int uncontested, pad_;
    double value;
    const double *sum_src[RS_MAX_ACTIONS];   // own node by groups: child a is an opponent's node whose value is the sum of ITS children's rows (sum_n[a] of them, n_pad apart,
    uint32_t sum_n[RS_MAX_ACTIONS];          // from sum_src[a]) -- added up while the rows are staged instead of written by k_br_sum_jobs and read again; 0: child a's row is vch + a * n_pad
    const uint32_t *tord, *perm;   // own node by columns (k_br_own_cols_jobs): [kmax][n_sets] the k-th lane of the i-th info set, and which cluster that info set is
    uint32_t n_sets, kmax;
};
// The level plan's form (round 5): ONE workgroup per run-out takes every leaf of the tree in turn.  What the leaves share -- the run-out's rank order, the holders of every
// card, where each of the traverser's hands stands in that order, its weight -- is read ONCE (25 KB per run-out and side; the one-leaf kernel above read it again at each of the
// 1 073 leaves of the three-street tree: 45 of its 111 MB per launch), the order and the card lists into LDS, the per-hand positions into registers (at most kBrHandsPerThread
// hands per thread); per leaf the workgroup streams the opponent's reach of its run-out in and its hands' values out.  The sums are br_terminal_sorted_body's, order and all.
constexpr int kBrHandsPerThread = 6;   // 1 326 two-card hands at most, 256 threads
constexpr int kBrChunkMax = 21;        // ... in 64 chunks of the rank order
// (the kernel is compiled for four range sizes -- <HPT, CHUNK> = <1, 4> up to 256 combos, <2, 8> up to 512, <4, 16> up to 1 024, <6, 21> beyond: its scans are unrolled to those
// bounds, and at <6, 21> a 200-combo range paid for 1 326 -- until then small ranges kept a workgroup per (run-out, leaf), 2.5 M workgroups per traverser)
static_assert(kBrCardHolders == 3 * 17, "the card scans of k_br_terminal_sorted_loop run in three blocks of 17");
template <int HPT, int CHUNK, bool REG_IDX>
__global__ __launch_bounds__(kBrBlock) void k_br_terminal_sorted_loop(const uint8_t *__restrict__ hands_p, const uint64_t *__restrict__ mask_p, const double *__restrict__ pw,
                                                                      uint32_t n_p, uint32_t n_o, const uint64_t *__restrict__ bmask, BrIndex ix,
                                                                      const BrJob *__restrict__ jobs, uint32_t n_jobs) {
    extern __shared__ unsigned char br_lds[];
    double *P = (double *)br_lds;                                // [n_o]
    double *Pc = P + n_o;                                        // [52][51]
    double *O = Pc + 52 * kBrCardHolders;                        // [65]
    double *Q = O + 65;                                          // [n_o]
    uint16_t *Lord = (uint16_t *)(Q + n_o);                      // [n_o rounded up to 4]
    uint16_t *Lcl = Lord + ((n_o + 3u) & ~3u);                   // [52][51]
    uint8_t *Lcc = (uint8_t *)(Lcl + 52 * kBrCardHolders);       // [52]
    const uint32_t b = blockIdx.x;
    const uint32_t nv = ix.nv[b];
    const uint64_t bm = bmask[b];
    for (uint32_t i = threadIdx.x; i < n_o; i += kBrBlock) Lord[i] = ix.ord[(size_t)b * n_o + i];
    for (uint32_t i = threadIdx.x; i < 52u * kBrCardHolders; i += kBrBlock) Lcl[i] = ix.cl[(size_t)b * 52 * kBrCardHolders + i];
    if (threadIdx.x < 52) Lcc[threadIdx.x] = ix.cc[(size_t)b * 52 + threadIdx.x];
    // this thread's hands: h = threadIdx.x + k * 256
    uint32_t hw0[HPT], hw1[HPT];     // hw0 = nl | nle << 16; hw1 = a0 | a1 << 8 | e0 << 16 | e1 << 24
    uint32_t hc[HPT];                              // c0 | c1 << 8 | live << 16 | (the opponent's hand of the same two cards + 1, or 0) << 17
    double hp[HPT];
#pragma unroll
    for (int k = 0; k < HPT; ++k) {
        const uint32_t h = threadIdx.x + (uint32_t)k * kBrBlock;
        hw0[k] = hw1[k] = hc[k] = 0;
        hp[k] = 0.0;
        if (h < n_p) {
            const size_t lane = (size_t)b * n_p + h;
            const bool live = !(mask_p[h] & bm);
            hc[k] = (uint32_t)hands_p[2 * h] | (uint32_t)hands_p[2 * h + 1] << 8 | (live ? 1u << 16 : 0u);
            if (live) {
                hw0[k] = (uint32_t)ix.nl[lane] | (uint32_t)ix.nle[lane] << 16;
                hw1[k] = (uint32_t)ix.kl[2 * lane] | (uint32_t)ix.kl[2 * lane + 1] << 8 | (uint32_t)ix.kle[2 * lane] << 16 | (uint32_t)ix.kle[2 * lane + 1] << 24;
                hc[k] |= (uint32_t)(ix.same[h] + 1) << 17;
                hp[k] = pw[lane];
            }
        }
    }
    const uint32_t len = (nv + 63u) / 64u;
    __syncthreads();
    // What the scans below walk is the same for every leaf, so it lives in registers: wave 0's lane k owns chunk k of the rank order (at most CHUNK positions), the first
    // 52 lanes of wave 1 a card's holders each.  Per leaf every value is then fetched from LDS in one go and added up in registers, in the body's order: a chunk's running sums,
    // the chunk offsets (one after the other over the lanes: readlane), P = offset + running sum; a card's running sums.  (The first version walked ord -> Q -> add -> store as
    // four dependent LDS operations per step, 51 steps for a card: 11.5 us per leaf and workgroup.)
    const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
    const uint32_t i0 = min(nv, wl * len), cnt0 = wave == 0 ? min(nv, i0 + len) - i0 : 0u;
    const uint32_t cardc = wl < 52u ? wl : 0u, cntc = (wave == 1 && wl < 52u) ? Lcc[cardc] : 0u;
    // REG_IDX (full ranges): the positions the scans read live in registers -- 72 of them, two workgroups per CU; from LDS at every leaf the kernel fits three workgroups
    // per CU at 168 registers and is slower (0.099 against 0.096 s per call).  The small-range forms (64-126 registers) read them from LDS: 200 combos 0.029 -> 0.026 s.
    uint32_t ordr[REG_IDX ? CHUNK : 1], clr[REG_IDX ? kBrCardHolders : 1];
    if (REG_IDX) {
#pragma unroll
        for (int k = 0; k < (REG_IDX ? CHUNK : 1); ++k) ordr[k] = (uint32_t)k < cnt0 ? Lord[i0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < (REG_IDX ? kBrCardHolders : 1); ++k) clr[k] = (uint32_t)k < cntc ? Lcl[cardc * kBrCardHolders + k] : 0u;
    }
    uint32_t most_holders = 0;   // of any card in this run-out: a small range's card scans end after the first block of 17
    for (uint32_t c = 0; c < 52u; ++c) most_holders = max(most_holders, (uint32_t)Lcc[c]);
    // the opponent's reach of the NEXT leaf is fetched while this one is worked on: with two workgroups per CU nothing else hides a leaf's round trip to memory
    double qn[HPT];
    if (blockIdx.y < n_jobs) {
        const double *__restrict__ q0 = jobs[blockIdx.y].q + (size_t)b * n_o;
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const uint32_t i = threadIdx.x + (uint32_t)k * kBrBlock;
            qn[k] = i < n_o ? q0[i] : 0.0;
        }
    }
    for (uint32_t j = blockIdx.y; j < n_jobs; j += gridDim.y) {
        double *__restrict__ v = jobs[j].v;
        const int uncontested = jobs[j].uncontested;
        const double value = jobs[j].value;
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const uint32_t i = threadIdx.x + (uint32_t)k * kBrBlock;
            if (i < n_o) Q[i] = qn[k];
        }
        if (j + gridDim.y < n_jobs) {
            const double *__restrict__ q1 = jobs[j + gridDim.y].q + (size_t)b * n_o;
#pragma unroll
            for (int k = 0; k < HPT; ++k) {
                const uint32_t i = threadIdx.x + (uint32_t)k * kBrBlock;
                qn[k] = i < n_o ? q1[i] : 0.0;
            }
        }
        __syncthreads();
        if (wave == 0) {
            double t[CHUNK];
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) t[k] = Q[REG_IDX ? ordr[REG_IDX ? k : 0] : ((uint32_t)k < cnt0 ? Lord[i0 + k] : 0u)];
            double run = 0.0;
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                if ((uint32_t)k < cnt0) run += t[k];
                t[k] = run;                                      // the chunk's running sum up to and including position k
            }
            const int s_lo = __double2loint(run), s_hi = __double2hiint(run);
            double off = 0.0, tot = 0.0;
            for (uint32_t k = 0; k < 64; ++k) {                  // offsets: sequential over the chunks
                const double c = __hiloint2double(__builtin_amdgcn_readlane(s_hi, (int)k), __builtin_amdgcn_readlane(s_lo, (int)k));
                if (wl == k) off = tot;
                tot += c;
            }
            if (!uncontested) {                                  // an uncontested leaf reads the totals only
#pragma unroll
                for (int k = 0; k < CHUNK; ++k)
                    if ((uint32_t)k < cnt0) P[i0 + k] = off + t[k];
            }
            if (wl == 0) O[64] = tot;
        } else if (wave == 1 && wl < 52u) {
            double run = 0.0;
#pragma unroll
            for (int k0 = 0; k0 < kBrCardHolders; k0 += 17) {
                if ((uint32_t)k0 >= most_holders) break;
                double t[17];
#pragma unroll
                for (int k = 0; k < 17; ++k) t[k] = Q[REG_IDX ? clr[REG_IDX ? k0 + k : 0] : ((uint32_t)(k0 + k) < cntc ? Lcl[cardc * kBrCardHolders + k0 + k] : 0u)];
#pragma unroll
                for (int k = 0; k < 17; ++k)
                    if ((uint32_t)(k0 + k) < cntc) {
                        run += t[k];
                        Pc[cardc * kBrCardHolders + k0 + k] = run;
                    }
            }
        }
        __syncthreads();
        const double T = O[64];
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const uint32_t h = threadIdx.x + (uint32_t)k * kBrBlock;
            if (h >= n_p) continue;
            const size_t lane = (size_t)b * n_p + h;
            if (!((hc[k] >> 16) & 1u)) {
                v[lane] = 0.0;
                continue;
            }
            const uint32_t c0 = hc[k] & 0xffu, c1 = (hc[k] >> 8) & 0xffu;
            const uint32_t n0 = Lcc[c0], n1 = Lcc[c1];
            const double T0 = n0 ? Pc[c0 * kBrCardHolders + n0 - 1] : 0.0, T1 = n1 ? Pc[c1 * kBrCardHolders + n1 - 1] : 0.0;
            double acc;
            if (uncontested) {
                acc = value * (((T - T0) - T1) + ((hc[k] >> 17) ? Q[(hc[k] >> 17) - 1u] : 0.0));
            } else {
                const uint32_t nl = hw0[k] & 0xffffu, nle = hw0[k] >> 16;
                const uint32_t a0 = hw1[k] & 0xffu, a1 = (hw1[k] >> 8) & 0xffu, e0 = (hw1[k] >> 16) & 0xffu, e1 = hw1[k] >> 24;
                const double L = nl ? P[nl - 1] : 0.0, LE = nle ? P[nle - 1] : 0.0;
                const double L0 = a0 ? Pc[c0 * kBrCardHolders + a0 - 1] : 0.0, L1 = a1 ? Pc[c1 * kBrCardHolders + a1 - 1] : 0.0;
                const double E0 = e0 ? Pc[c0 * kBrCardHolders + e0 - 1] : 0.0, E1 = e1 ? Pc[c1 * kBrCardHolders + e1 - 1] : 0.0;
                const double win = (L - L0) - L1;
                const double lose = ((T - LE) - (T0 - E0)) - (T1 - E1);
                acc = value * (win - lose);
            }
            v[lane] = hp[k] * acc;
        }
        __syncthreads();                                         // Q, P, Pc, O are the next leaf's
    }
}

// the same with one WAVE per info set, for rounds whose info sets hold many lanes (a flop info set of the 1 176-combo game: 3 800 of the 2.8 M lanes -- one thread per info set
// left three workgroups walking 3 800 dependent gathers each, 13 ms per node).  The lanes of the wave fetch 64 list entries at a time; the additions still run one after the
// other in list order (every lane adds the 64 values in the same order, read from its neighbours' registers), so the sums keep the oracle's bits.
template <int DT>
__device__ __forceinline__ void br_own_wave_body(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ start /*[n_clusters + 2]*/,
                                                          const uint32_t *__restrict__ order, uint32_t n_clusters, uint32_t n_pad, const double *__restrict__ vch,
                                                          int mode, double *__restrict__ v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * kBrBlock + threadIdx.x) >> 6, n_waves = gridDim.x * (kBrBlock >> 6);
    for (uint32_t i = start[n_clusters] + blockIdx.x * kBrBlock + threadIdx.x; i < start[n_clusters + 1]; i += gridDim.x * kBrBlock) v[order[i]] = 0.0;
    for (uint32_t c = wave; c < n_clusters; c += n_waves) {
        const uint32_t lo = start[c], hi = start[c + 1];
        if (lo == hi) continue;
        if (mode == RS_BR_MAX) {
            double s[RS_MAX_ACTIONS];
            for (uint32_t a = 0; a < row.n_actions; a++) {
                const double *va = vch + (size_t)a * n_pad;
                double acc = 0.0;
                for (uint32_t i = lo; i < hi; i += 64) {
                    const uint32_t cnt = min(64u, hi - i);
                    const double t = lane < cnt ? va[order[i + lane]] : 0.0;
                    const int t_lo = __double2loint(t), t_hi = __double2hiint(t);
                    for (uint32_t k = 0; k < cnt; ++k)   // cnt is wave-uniform
                        acc += __hiloint2double(__builtin_amdgcn_readlane(t_hi, (int)k), __builtin_amdgcn_readlane(t_lo, (int)k));
                }
                s[a] = acc;
            }
            uint32_t best = 0;
            for (uint32_t a = 1; a < row.n_actions; a++)
                if (s[best] < s[a]) best = a;
            for (uint32_t i = lo + lane; i < hi; i += 64) v[order[i]] = vch[(size_t)best * n_pad + order[i]];
        } else {
            float sig[RS_MAX_ACTIONS];
            final_sigma<DT>(ssum, row.cell_off, row.pitch, row.n_actions, c, sig);
            for (uint32_t i = lo + lane; i < hi; i += 64) {
                const uint32_t h = order[i];
                double acc = 0.0;
                for (uint32_t a = 0; a < row.n_actions; a++) acc += (double)sig[a] * vch[(size_t)a * n_pad + h];
                v[h] = acc;
            }
        }
    }
}


// ---- own nodes by groups of run-outs (level plan, RS_BR_MAX) ------------------------------------------------------------------------------------------------------
// A lossless river abstraction puts four lanes in an info set, each in a run-out of its own (the two orders of turn and river card, the swap of the two suits the flop does
// not hold): a thread per info set (br_own_body) fetches every 8-byte child value as a cache line of its own and moves nine times its algorithmic bytes (profiles/r05_br.md:
// 461 GB per call at the HBM roofline).  But the info sets of such a round stay within FEW run-outs: the run-outs fall into components (run-outs that share an info set) of
// two, four or eight run-outs there.  br_prepare packs components into groups of `cap` run-outs; a workgroup takes one group: the group's rows of a child's values come in
// whole (9 KB of consecutive doubles per run-out) into LDS, the info sets add up from LDS in br_own_body's order (same sums, same bits) and note the action that leads; each
// thread keeps the leader's value of the lanes it fetched and writes them out as whole rows.  Each child row is read once, the node's row written once.
struct BrGroups {
    const uint32_t *runouts;   // [n_groups][cap]: the group's run-outs (0xffffffff: none)
    const uint32_t *cstart;    // [n_groups + 1]: the group's info sets, as positions in lstart
    const uint32_t *lstart;    // [info sets + 1]: an info set's lanes in llane
    const uint16_t *llane;     // slot * n_hands + hand, in the order of BrSide::d_order (ascending lane)
    const uint16_t *lane_set;  // [n_groups][cap * n_hands]: the lane's info set within its group (0xffff: none -- the hand holds a card of the run-out, or no run-out in the slot)
    const uint32_t *set_cluster;   // [info sets]: the cluster id (k_br_opp_reach_grouped_jobs looks the strategy-sum rows up once per info set)
    uint32_t n_groups, cap, n_hands, max_sets;
};
constexpr uint32_t kBrGroupLanes = 5400;   // lanes a group is packed up to (cap = kBrGroupLanes / n_hands run-outs, at least the largest component): four run-outs of 1 176-1 326 hands
constexpr int kBrGroupBlock = 1024;    // one workgroup per CU at eight run-outs of 1 176 hands (75 KB of staged values + 26 KB of sums and leaders): sixteen waves of it
constexpr int kBrGroupPerThread = 11;  // cap * n_hands <= 11 264 values per row set (eight run-outs of 1 326 hands)
__global__ __launch_bounds__(kBrGroupBlock) void k_br_own_grouped_jobs(const BrJob *__restrict__ jobs, BrGroups g, uint32_t n_pad) {
    extern __shared__ double br_group_lds[];
    const BrJob j = jobs[blockIdx.y];
    const uint32_t grp = blockIdx.x, tid = threadIdx.x, GH = g.cap * g.n_hands;
    double *stage = br_group_lds, *bestv = stage + GH;
    uint32_t *besta = reinterpret_cast<uint32_t *>(bestv + g.max_sets);
    uint32_t goff[kBrGroupPerThread];   // the thread's values of a row set: global lane, or none
    uint32_t setof[kBrGroupPerThread];
    double outv[kBrGroupPerThread], regs[kBrGroupPerThread];
#pragma unroll
    for (int k = 0; k < kBrGroupPerThread; k++) {
        const uint32_t e = tid + uint32_t(k) * kBrGroupBlock;
        goff[k] = 0xffffffffu;
        setof[k] = 0xffffu;
        outv[k] = 0.0;   // a lane in no info set (its hand holds a card of the run-out) is worth 0
        if (e < GH) {
            const uint32_t slot = e / g.n_hands, ro = g.runouts[(size_t)grp * g.cap + slot];
            if (ro != 0xffffffffu) {
                goff[k] = ro * g.n_hands + (e - slot * g.n_hands);
                setof[k] = g.lane_set[(size_t)grp * GH + e];
            }
        }
    }
    const uint32_t c_lo = g.cstart[grp], c_hi = g.cstart[grp + 1];
    // child a's values of this thread's lanes: the child's row, or -- the child an opponent's node -- the sum of its children's rows in action order (br_sum_body's additions)
    const BrJob *__restrict__ jp = jobs + blockIdx.y;   // (sum_src / sum_n are indexed by the action: read where they lie, not from the by-value copy)
    auto fetch = [&](uint32_t a) {
        const uint32_t ns = jp->sum_n[a];
        if (ns == 0) {
            const double *va = j.vch + (size_t)a * n_pad;
#pragma unroll
            for (int k = 0; k < kBrGroupPerThread; k++) regs[k] = goff[k] != 0xffffffffu ? va[goff[k]] : 0.0;
            return;
        }
#pragma unroll
        for (int k = 0; k < kBrGroupPerThread; k++) regs[k] = 0.0;
        const double *src = jp->sum_src[a];
        for (uint32_t c = 0; c < ns; c++) {
            const double *vc = src + (size_t)c * n_pad;
#pragma unroll
            for (int k = 0; k < kBrGroupPerThread; k++) regs[k] += goff[k] != 0xffffffffu ? vc[goff[k]] : 0.0;
        }
    };
    fetch(0);
    for (uint32_t a = 0; a < j.n_children; a++) {
#pragma unroll
        for (int k = 0; k < kBrGroupPerThread; k++) {
            const uint32_t e = tid + uint32_t(k) * kBrGroupBlock;
            if (e < GH) stage[e] = regs[k];
        }
        __syncthreads();
        if (a + 1 < j.n_children) fetch(a + 1);   // the next child's rows are on their way while this one's info sets add up
        for (uint32_t c = c_lo + tid; c < c_hi; c += kBrGroupBlock) {
            const uint32_t lo = g.lstart[c], hi = g.lstart[c + 1];
            double acc = 0.0;
            for (uint32_t i = lo; i < hi; i++) acc += stage[g.llane[i]];   // list order: br_own_body's sum
            if (a == 0 || bestv[c - c_lo] < acc) {                          // first maximum, strict < (cfr.rs:684-690)
                bestv[c - c_lo] = acc;
                besta[c - c_lo] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBrGroupPerThread; k++)
            if (setof[k] != 0xffffu && besta[setof[k]] == a) outv[k] = stage[tid + uint32_t(k) * kBrGroupBlock];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kBrGroupPerThread; k++)
        if (goff[k] != 0xffffffffu) j.v[goff[k]] = outv[k];
}

// The opponent's reach through the same groups: sigma_bar of an info set is computed ONCE (a lane-parallel kernel gathers the strategy-sum rows for each of its 4.3 lanes)
// into LDS; the group's lanes then stream through: q in, A products out, whole rows.  Lanes in no info set take cluster 0's strategy, as br_opp_reach_body has it (their
// reach is never looked at: the leaves skip hands that hold a card of the run-out).
template <int DT>
__global__ __launch_bounds__(kBrGroupBlock) void k_br_opp_reach_grouped_jobs(const void *__restrict__ ssum, const BrJob *__restrict__ jobs, BrGroups g, uint32_t n_pad) {
    extern __shared__ double br_group_lds[];
    float *sg = reinterpret_cast<float *>(br_group_lds);   // [A][max_sets + 1]
    const BrJob j = jobs[blockIdx.y];
    // workgroups go round the eight XCDs in turn: XCD x takes the groups [x n/8, (x+1) n/8) in order, so that neighbouring groups -- whose info sets are neighbours in the
    // strategy-sum rows (br_prepare packs the components in the order of their cluster ids) -- meet in one L2 at about the same time
    const uint32_t per_xcd = (g.n_groups + 7u) / 8u, grp = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (grp >= g.n_groups || (blockIdx.x >> 3) >= per_xcd) return;
    const uint32_t tid = threadIdx.x, GH = g.cap * g.n_hands, A = j.row.n_actions, pitch = g.max_sets + 1;
    const uint32_t c_lo = g.cstart[grp], n_sets = g.cstart[grp + 1] - c_lo;
    for (uint32_t k = tid; k <= n_sets; k += kBrGroupBlock) {
        float sig[RS_MAX_ACTIONS];
        final_sigma<DT>(ssum, j.row.cell_off, j.row.pitch, A, k < n_sets ? g.set_cluster[c_lo + k] : 0u, sig);
        const uint32_t at = k < n_sets ? k : g.max_sets;
        for (uint32_t a = 0; a < A; a++) sg[a * pitch + at] = sig[a];
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t e = tid; e < GH; e += kBrGroupBlock) {
        const uint32_t slot = e / g.n_hands, ro = g.runouts[(size_t)grp * g.cap + slot];
        if (ro == 0xffffffffu) continue;
        const uint32_t h = ro * g.n_hands + (e - slot * g.n_hands), set = g.lane_set[(size_t)grp * GH + e];
        const uint32_t at = set != 0xffffu ? set : g.max_sets;
        const double qh = j.q[h];
        for (uint32_t a = 0; a < A; a++) j.q_out[(size_t)a * n_pad + h] = qh * (double)sg[a * pitch + at];
    }
}

// ---- own nodes by columns (level plan) ---------------------------------------------------------------------------------------------------------------------------
// A lossless turn abstraction puts ~92 lanes in an info set: one hand under one turn card, every river card (and the suit-swapped twin).  A wave per info set
// (br_own_wave_body) reads 64 lanes of 64 run-outs per step: every 8-byte value a cache line of its own, eight times the algorithmic bytes (profiles/r05_br.md).  Here a THREAD
// takes an info set, and the info sets are ordered by their first lane: neighbouring threads hold neighbouring hands under the same turn card, so the k-th lanes of a wave's
// info sets lie side by side in one run-out's row (where a hand holds a river card its list is one entry ahead of its neighbours': two rows).  The lane lists come
// transposed ([k][info set]) so that the indices load coalesced too.  The sums run down each list in order: br_own_body's, and the wave kernel's, bits.
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_own_cols_jobs(const void *__restrict__ ssum, const BrJob *__restrict__ jobs, uint32_t n_pad, int mode) {
    const BrJob j = jobs[blockIdx.y];
    for (uint32_t i = j.start[j.n_clusters] + blockIdx.x * kBrBlock + threadIdx.x; i < j.start[j.n_clusters + 1]; i += gridDim.x * kBrBlock) j.v[j.order[i]] = 0.0;
    const uint32_t i = blockIdx.x * kBrBlock + threadIdx.x;
    if (i >= j.n_sets) return;
    const uint32_t *__restrict__ col = j.tord + i;
    const uint32_t A = j.row.n_actions;
    if (mode == RS_BR_MAX) {
        double s_best = 0.0;
        uint32_t best = 0;
        for (uint32_t a = 0; a < A; a++) {
            const double *va = j.vch + (size_t)a * n_pad;
            double acc = 0.0;
            for (uint32_t k = 0; k < j.kmax; k += 8) {   // eight gathers in flight (sixteen: no faster), the additions one after the other in list order
                uint32_t idx[8];
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; q++) idx[q] = k + q < j.kmax ? col[(size_t)(k + q) * j.n_sets] : 0xffffffffu;
#pragma unroll
                for (int q = 0; q < 8; q++) t[q] = idx[q] != 0xffffffffu ? va[idx[q]] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (idx[q] != 0xffffffffu) acc += t[q];
            }
            if (a == 0 || s_best < acc) s_best = acc, best = a;   // first maximum, strict < (cfr.rs:684-690)
        }
        const double *vb = j.vch + (size_t)best * n_pad;
        for (uint32_t k = 0; k < j.kmax; k += 8) {   // the winner's values: eight lanes' indices, then their values, then the stores
            uint32_t idx[8];
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) idx[q] = k + q < j.kmax ? col[(size_t)(k + q) * j.n_sets] : 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = idx[q] != 0xffffffffu ? vb[idx[q]] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (idx[q] != 0xffffffffu) j.v[idx[q]] = t[q];
        }
    } else {
        float sig[RS_MAX_ACTIONS];
        final_sigma<DT>(ssum, j.row.cell_off, j.row.pitch, A, j.perm[i], sig);
        for (uint32_t k = 0; k < j.kmax; k++) {
            const uint32_t h = col[(size_t)k * j.n_sets];
            if (h == 0xffffffffu) continue;
            double acc = 0.0;
            for (uint32_t a = 0; a < A; a++) acc += (double)sig[a] * j.vch[(size_t)a * n_pad + h];
            j.v[h] = acc;
        }
    }
}

// ---- the kernels: one node per launch (the depth-first walk), or one JOB per node and grid row (the level plan: all nodes of one tree depth and kind in one launch) -------------
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_opp_reach(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ cid, uint32_t n, uint32_t n_pad,
                                                           const double *__restrict__ q, double *__restrict__ q_out) {
    br_opp_reach_body<DT>(ssum, row, cid, n, n_pad, q, q_out);
}
// (level plan) The strategy-sum rows are gathered by cluster id, and a lossless abstraction numbers its clusters hand by hand: workgroups go round the eight XCDs in turn, so
// workgroup x takes hands of slice x % 8 only (every run-out of them) -- an XCD's L2 then sees an eighth (with the suit-swapped twins: a quarter) of the node's rows instead of
// all of them (the river's 7.7 MB per node against 4 MB of L2: 250 MB fetched per node, profiles/r05_br.md).  gridDim.x = 8 * ceil(run-outs * ceil(n_hands / 8) / kBrBlock).
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_opp_reach_jobs(const void *__restrict__ ssum, const BrJob *__restrict__ jobs, uint32_t n, uint32_t n_pad, uint32_t n_hands) {
    const BrJob j = jobs[blockIdx.y];
    const uint32_t xcd = blockIdx.x & 7u, w = blockIdx.x >> 3, per = (n_hands + 7u) / 8u, lo = xcd * per;
    if (lo >= n_hands) return;
    const uint32_t hs = min(per, n_hands - lo), idx = w * kBrBlock + threadIdx.x, b = idx / hs;
    const uint32_t h = b * n_hands + lo + (idx - b * hs);
    if (h >= n || b >= n / n_hands) return;
    float sig[RS_MAX_ACTIONS];
    final_sigma<DT>(ssum, j.row.cell_off, j.row.pitch, j.row.n_actions, j.cid[h], sig);
    const double qh = j.q[h];
    for (uint32_t a = 0; a < j.row.n_actions; a++) j.q_out[(size_t)a * n_pad + h] = qh * (double)sig[a];
}
__global__ __launch_bounds__(kBrBlock) void k_br_sum(const double *__restrict__ vch, uint32_t n_actions, uint32_t n, uint32_t n_pad, double *__restrict__ v) {
    br_sum_body(vch, n_actions, n, n_pad, v);
}
__global__ __launch_bounds__(kBrBlock) void k_br_sum_jobs(const BrJob *__restrict__ jobs, uint32_t n, uint32_t n_pad) {
    const BrJob j = jobs[blockIdx.y];
    br_sum_body(j.vch, j.n_children, n, n_pad, j.v);
}
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_own(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ start, const uint32_t *__restrict__ order,
                                                     uint32_t n_clusters, uint32_t n_pad, const double *__restrict__ vch, int mode, double *__restrict__ v) {
    br_own_body<DT>(ssum, row, start, order, n_clusters, n_pad, vch, mode, v);
}
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_own_jobs(const void *__restrict__ ssum, const BrJob *__restrict__ jobs, uint32_t n_pad, int mode) {
    const BrJob j = jobs[blockIdx.y];
    br_own_body<DT>(ssum, j.row, j.start, j.order, j.n_clusters, n_pad, j.vch, mode, j.v);
}
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_own_wave(const void *__restrict__ ssum, BrNodeRow row, const uint32_t *__restrict__ start, const uint32_t *__restrict__ order,
                                                          uint32_t n_clusters, uint32_t n_pad, const double *__restrict__ vch, int mode, double *__restrict__ v) {
    br_own_wave_body<DT>(ssum, row, start, order, n_clusters, n_pad, vch, mode, v);
}
template <int DT>
__global__ __launch_bounds__(kBrBlock) void k_br_own_wave_jobs(const void *__restrict__ ssum, const BrJob *__restrict__ jobs, uint32_t n_pad, int mode) {
    const BrJob j = jobs[blockIdx.y];
    br_own_wave_body<DT>(ssum, j.row, j.start, j.order, j.n_clusters, n_pad, j.vch, mode, j.v);
}
__global__ __launch_bounds__(kBrBlock) void k_br_terminal_sorted(const uint8_t *__restrict__ hands_p, const uint64_t *__restrict__ mask_p, const double *__restrict__ pw,
                                                                 uint32_t n_p, const double *__restrict__ q, uint32_t n_o, const uint64_t *__restrict__ bmask, BrIndex ix,
                                                                 int uncontested, double value, double *__restrict__ v) {
    br_terminal_sorted_body(hands_p, mask_p, pw, n_p, q, n_o, bmask, ix, uncontested, value, v);
}
__global__ __launch_bounds__(kBrBlock) void k_br_terminal_sorted_jobs(const uint8_t *__restrict__ hands_p, const uint64_t *__restrict__ mask_p, const double *__restrict__ pw,
                                                                      uint32_t n_p, uint32_t n_o, const uint64_t *__restrict__ bmask, BrIndex ix,
                                                                      const BrJob *__restrict__ jobs) {
    const BrJob j = jobs[blockIdx.y];
    br_terminal_sorted_body(hands_p, mask_p, pw, n_p, j.q, n_o, bmask, ix, j.uncontested, j.value, j.v);
}
__global__ __launch_bounds__(kBrBlock) void k_br_terminal_boards(const uint64_t *__restrict__ mask_p, const uint32_t *__restrict__ score_p, const double *__restrict__ pw,
                                                                 uint32_t n_p, const uint64_t *__restrict__ mask_o, const uint32_t *__restrict__ score_o,
                                                                 const double *__restrict__ q, uint32_t n_o, const uint64_t *__restrict__ bmask, int uncontested,
                                                                 double value, double *__restrict__ v) {
    br_terminal_boards_body(mask_p, score_p, pw, n_p, mask_o, score_o, q, n_o, bmask, uncontested, value, v);
}
__global__ __launch_bounds__(kBrBlock) void k_br_terminal_boards_jobs(const uint64_t *__restrict__ mask_p, const uint32_t *__restrict__ score_p, const double *__restrict__ pw,
                                                                      uint32_t n_p, const uint64_t *__restrict__ mask_o, const uint32_t *__restrict__ score_o, uint32_t n_o,
                                                                      const uint64_t *__restrict__ bmask, const BrJob *__restrict__ jobs) {
    const BrJob j = jobs[blockIdx.z];
    br_terminal_boards_body(mask_p, score_p, pw, n_p, mask_o, score_o, j.q, n_o, bmask, j.uncontested, j.value, j.v);
}

constexpr size_t kBrLevelPlanBytes = size_t(96) << 30;   // the level plan's workspace is kept with the object: not beyond 96 GB (a third of the card; 16 GB until round 5, when
                                                         // the plan bought launches only -- since the leaf loop it halves the terminals' traffic, and full 1 176-combo ranges need 59 GB)

struct BrSide {
    uint32_t n_hands = 0;
    uint32_t n = 0, n_pad = 0;     // lanes = NB * n_hands
    uint32_t n_clusters[RS_MAX_ROUNDS] = {0, 0, 0};
    uint64_t *d_mask = nullptr;    // [n_hands]
    uint32_t *d_score = nullptr;   // [n]
    uint32_t *d_cid[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr}, *d_start[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr}, *d_order[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};
    BrGroups groups[RS_MAX_ROUNDS] = {};   // own nodes by groups of run-outs (k_br_own_grouped_jobs), where the round's info sets stay within a few run-outs each
    bool grouped[RS_MAX_ROUNDS] = {false, false, false};
    uint32_t *d_tord[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr}, *d_perm[RS_MAX_ROUNDS] = {nullptr, nullptr, nullptr};   // own nodes by columns (k_br_own_cols_jobs)
    uint32_t n_sets[RS_MAX_ROUNDS] = {0, 0, 0}, kmax[RS_MAX_ROUNDS] = {0, 0, 0};
    size_t group_lds[RS_MAX_ROUNDS] = {0, 0, 0};
    double *d_init_q = nullptr;    // this side's lanes as the OPPONENT's initial reach (its share of the deal probability)
    double *d_pw = nullptr;        // this side's lanes as the TRAVERSER's weight
    uint8_t *d_hands = nullptr;    // [n_hands][2]
    BrIndex index;                 // RS_BR_SORTED: this side as the TRAVERSER against the other side's sorted hands
};

struct BrRun {
    rs_table *t = nullptr;
    const rs_tree *tree = nullptr;
    int mode = RS_BR_MAX;
    bool sorted = false;           // RS_BR_SORTED: showdowns by rank order
    int p = 0;
    uint32_t NB = 1;
    uint64_t *d_bmask = nullptr;
    BrSide side[2];
    std::vector<void *> allocs;
    std::vector<double *> q_level, v_level;   // per tree depth: [max actions][n_pad] children buffers
    double *d_root = nullptr;                 // [n_pad_max]: the root values of the traverser's lanes (depth-first walk)
    bool last_level_plan = false;             // what the last br_execute ran, and its launches (level plan)
    int last_launches = 0;
    double *ws = nullptr;                     // the walk's workspace, allocated by the first br_execute and kept (a 60 GB hipMalloc takes seconds): level plan or depth-first
    size_t ws_bytes = 0, game_bytes = 0;      // game_bytes: what dalloc handed out (the game-only half)
    bool ws_levels = false;
    void release_workspace() {
        if (ws) {
            (void)hipStreamSynchronize(t->stream);
            (void)hipFree(ws);
        }
        ws = nullptr;
        ws_bytes = 0;
        q_level.clear();
        v_level.clear();
        d_root = nullptr;
    }
    size_t n_pad_max = 0;
    hipError_t err = hipSuccess;

    template <typename T> T *dalloc(size_t n) {
        void *ptr = nullptr;
        if (err == hipSuccess) err = hipMalloc(&ptr, (n ? n : 1) * sizeof(T));
        if (err == hipSuccess) {
            allocs.push_back(ptr);
            game_bytes += (n ? n : 1) * sizeof(T);
        }
        return static_cast<T *>(ptr);
    }
    template <typename T> T *upload(const std::vector<T> &h) {
        T *d = dalloc<T>(h.size());
        if (err == hipSuccess && !h.empty()) err = hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);   // blocking: h may be a temporary
        return d;
    }
    ~BrRun() {
        if (ws) (void)hipFree(ws);
        for (void *ptr : allocs) (void)hipFree(ptr);
    }

    // ---- the level plan: every node gets buffers of its own (an action node: [A][n_pad] for its children's values, an opponent's node the same for their reach), so the
    // nodes of one tree depth no longer wait for each other and ONE launch takes all of a depth's nodes of one kind (grid row = node): reach down level by level, every
    // leaf in one launch, values up level by level -- about sixty launches per traverser instead of one or two per tree node (1 864 nodes: 4 300 launches per call).
    // The arithmetic of a node is the depth-first walk's (same kernels' bodies), so are its bits.  Costs memory: see level_plan_bytes.
    int max_depth = 0;
    std::vector<int> depth_of;    // per tree node: action-node ancestors
    void fill_depths() {
        depth_of.assign(tree->nodes.size(), 0);
        max_depth = 0;
        for (size_t id = 0; id < tree->nodes.size(); ++id) {   // parents come before children
            const rs_tree_node &n = tree->nodes[id];
            for (int a = 0; a < n.n_children; ++a) depth_of[size_t(n.children[a])] = depth_of[id] + (n.kind == RS_NODE_ACTION ? 1 : 0);
            if (n.kind == RS_NODE_ACTION) max_depth = std::max(max_depth, depth_of[id] + 1);
        }
    }
    size_t level_plan_bytes(int pl) const {   // workspace of traverser pl's pass
        const BrSide &me = side[pl], &op = side[1 - pl];
        size_t doubles = me.n_pad;   // the root vector
        for (const rs_tree_node &n : tree->nodes)
            if (n.kind == RS_NODE_ACTION && n.n_children > 0) doubles += size_t(n.n_children) * (size_t(me.n_pad) + (int(n.player) != pl ? size_t(op.n_pad) : 0));
        return doubles * sizeof(double);
    }
    int run_levels(double *ws, double **root_out) {
        const BrSide &me = side[p], &op = side[1 - p];
        const size_t N = tree->nodes.size();
        if (depth_of.size() != N) fill_depths();
        std::vector<const double *> q_in(N, nullptr);
        std::vector<double *> v_out(N, nullptr), vch(N, nullptr), qch(N, nullptr);
        double *at = ws;
        *root_out = at;
        at += me.n_pad;
        v_out[0] = *root_out;
        q_in[0] = op.d_init_q;
        std::vector<std::vector<BrJob>> down(size_t(max_depth) + 1), up_own(size_t(max_depth) + 1), up_wave(size_t(max_depth) + 1), up_sum(size_t(max_depth) + 1);
        std::vector<std::pair<int, int>> grp_at(N, {-1, -1});   // an own node taken by groups: (round, index in up_grp[round][depth])
        std::vector<std::vector<BrJob>> up_cols(size_t(max_depth) + 1);   // own nodes by columns
        std::vector<std::vector<BrJob>> up_grp[RS_MAX_ROUNDS], down_grp[RS_MAX_ROUNDS];   // own / opponent nodes taken by groups of run-outs, per round (the groups are the round's)
        for (auto &v : up_grp) v.resize(size_t(max_depth) + 1);
        for (auto &v : down_grp) v.resize(size_t(max_depth) + 1);
        std::vector<BrJob> leaves;
        for (size_t id = 0; id < N; ++id) {   // parents before children: a node's q and v slot are known when it comes up
            const rs_tree_node &n = tree->nodes[id];
            if (n.kind == RS_NODE_TERMINAL) {
                const int unc = n.ttype == RS_TERM_UNCONTESTED;
                const double pot = double(float(n.value));   // tn.value as f32 (cfr.rs:316)
                BrJob j{};
                j.q = q_in[id];
                j.v = v_out[id];
                j.uncontested = unc;
                j.value = unc ? (p == int(n.last_to_act) ? -pot : pot) : pot;
                leaves.push_back(j);
                continue;
            }
            if (n.kind != RS_NODE_ACTION) {   // chance nodes pass through (cfr.rs:306-313)
                if (n.n_children > 0) {
                    q_in[size_t(n.children[0])] = q_in[id];
                    v_out[size_t(n.children[0])] = v_out[id];
                }
                continue;
            }
            if (n.n_children == 0) return fail(RS_ERR_UNSUPPORTED, "rs_best_response: an action node without actions");
            const int r = n.round_idx, d = depth_of[id];
            vch[id] = at;
            at += size_t(n.n_children) * me.n_pad;
            BrJob j{};
            j.row = row_of(t, n.index);
            j.vch = vch[id];
            j.v = v_out[id];
            j.n_children = uint32_t(n.n_children);
            if (int(n.player) == p) {
                j.start = me.d_start[r];
                j.order = me.d_order[r];
                j.n_clusters = me.n_clusters[r];
                if (mode == RS_BR_MAX && me.grouped[r]) {
                    grp_at[id] = {r, int(up_grp[r][size_t(d)].size())};
                    up_grp[r][size_t(d)].push_back(j);
                }
                else if (me.d_tord[r]) {
                    j.tord = me.d_tord[r];
                    j.perm = me.d_perm[r];
                    j.n_sets = me.n_sets[r];
                    j.kmax = me.kmax[r];
                    up_cols[size_t(d)].push_back(j);
                }
                else (size_t(me.n) >= size_t(me.n_clusters[r]) * 32 ? up_wave : up_own)[size_t(d)].push_back(j);   // many lanes per info set: a wave each
                for (int a = 0; a < n.n_children; ++a) q_in[size_t(n.children[a])] = q_in[id];
            } else {
                qch[id] = at;
                at += size_t(n.n_children) * op.n_pad;
                j.cid = op.d_cid[r];
                j.q = q_in[id];
                j.q_out = qch[id];
                (op.grouped[r] ? down_grp[r] : down)[size_t(d)].push_back(j);
                // its value is the sum of its children's; where the parent is an own node taken by groups, that kernel adds the rows up as it stages them
                const int par = n.parent;
                bool taken = false;
                if (par >= 0 && grp_at[size_t(par)].first >= 0) {
                    const rs_tree_node &pn = tree->nodes[size_t(par)];
                    BrJob &pj = up_grp[grp_at[size_t(par)].first][size_t(depth_of[size_t(par)])][size_t(grp_at[size_t(par)].second)];
                    for (int a = 0; a < pn.n_children && !taken; ++a)
                        if (pn.children[a] == int(id)) {
                            pj.sum_src[a] = vch[id];
                            pj.sum_n[a] = uint32_t(n.n_children);
                            taken = true;
                        }
                }
                if (!taken) up_sum[size_t(d)].push_back(j);
                for (int a = 0; a < n.n_children; ++a) q_in[size_t(n.children[a])] = qch[id] + size_t(a) * op.n_pad;
            }
            for (int a = 0; a < n.n_children; ++a) v_out[size_t(n.children[a])] = vch[id] + size_t(a) * me.n_pad;
        }
        // all jobs in one upload
        std::vector<BrJob> all;
        auto put = [&](const std::vector<BrJob> &v) {
            const size_t at_ = all.size();
            all.insert(all.end(), v.begin(), v.end());
            return at_;
        };
        std::vector<size_t> o_down, o_own, o_wave, o_sum, o_cols, o_grp[RS_MAX_ROUNDS], o_dgrp[RS_MAX_ROUNDS];
        for (int d = 0; d <= max_depth; ++d) {
            o_down.push_back(put(down[size_t(d)]));
            o_cols.push_back(put(up_cols[size_t(d)]));
            o_own.push_back(put(up_own[size_t(d)]));
            for (int r = 0; r < RS_MAX_ROUNDS; ++r) o_grp[r].push_back(put(up_grp[r][size_t(d)]));
            for (int r = 0; r < RS_MAX_ROUNDS; ++r) o_dgrp[r].push_back(put(down_grp[r][size_t(d)]));
            o_wave.push_back(put(up_wave[size_t(d)]));
            o_sum.push_back(put(up_sum[size_t(d)]));
        }
        const size_t o_leaves = put(leaves);
        BrJob *d_jobs = nullptr;
        err = hipMalloc((void **)&d_jobs, std::max<size_t>(all.size(), 1) * sizeof(BrJob));
        if (err == hipSuccess) err = hipMemcpyAsync(d_jobs, all.data(), all.size() * sizeof(BrJob), hipMemcpyHostToDevice, t->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(t->stream);   // `all` is a local
        n_launches = 0;
        for (int d = 0; d <= max_depth && err == hipSuccess; ++d) {   // reach, level by level
            for (int r = 0; r < RS_MAX_ROUNDS && err == hipSuccess; ++r)
                if (const uint32_t nj = uint32_t(down_grp[r][size_t(d)].size())) {
                    uint32_t amax = 0;
                    for (const BrJob &j : down_grp[r][size_t(d)]) amax = std::max(amax, j.row.n_actions);
                    const size_t lds = size_t(amax) * (size_t(op.groups[r].max_sets) + 1) * sizeof(float);
#define RS_OPPG(DT_)                                                                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_br_opp_reach_grouped_jobs<DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                    \
    hipLaunchKernelGGL((k_br_opp_reach_grouped_jobs<DT_>), dim3(8u * ((op.groups[r].n_groups + 7u) / 8u), nj), dim3(kBrGroupBlock), lds, t->stream, t->d_ssum, d_jobs + o_dgrp[r][size_t(d)], \
                       op.groups[r], op.n_pad)
                    RS_BR_DT(t->dtype, RS_OPPG);
#undef RS_OPPG
                    err = hipGetLastError();
                    ++n_launches;
                }
            const uint32_t nj = uint32_t(down[size_t(d)].size());
            if (!nj) continue;
            const uint32_t sliced = 8u * grid1(NB * ((op.n_hands + 7u) / 8u));
#define RS_OPPJ(DT_) hipLaunchKernelGGL((k_br_opp_reach_jobs<DT_>), dim3(sliced, nj), dim3(kBrBlock), 0, t->stream, t->d_ssum, d_jobs + o_down[size_t(d)], op.n, op.n_pad, op.n_hands)
            RS_BR_DT(t->dtype, RS_OPPJ);
#undef RS_OPPJ
            err = hipGetLastError();
            ++n_launches;
        }
        for (size_t lo = 0; lo < leaves.size() && err == hipSuccess; lo += 16384) {   // every leaf
            const uint32_t nj = uint32_t(std::min<size_t>(16384, leaves.size() - lo));
            // (small ranges keep a workgroup per (run-out, leaf): the loop's scans are unrolled for 1 326 hands whatever the range holds -- 200 combos: 0.047 against 0.059 s per call)
            const uint32_t most_hands = std::max(me.n_hands, op.n_hands);
            if (sorted && most_hands <= uint32_t(kBrHandsPerThread) * kBrBlock && op.n_hands <= uint32_t(kBrChunkMax) * 64u) {   // a workgroup per run-out (and slice of the leaves, when run-outs alone do not fill the card)
                const size_t lds = (size_t(op.n_hands) * 2 + 52 * kBrCardHolders + 65) * sizeof(double) + (((size_t(op.n_hands) + 3) & ~size_t(3)) + 52 * kBrCardHolders) * sizeof(uint16_t) + 64;
                // slices of the leaves: enough workgroups to fill the card, and a count that leaves the last round of workgroups (two per CU at full ranges: 220 registers;
                // four and more for the small-range forms) nearly full -- 2 352 run-outs on 512 slots are 4.6 rounds (the fifth 59 % full), three slices 13.8
                const int form = most_hands <= 256 ? 0 : (most_hands <= 512 ? 1 : (most_hands <= 1024 ? 2 : 3));
                uint32_t slices = std::max<uint32_t>(1, std::min<uint32_t>(nj, 2048u / std::max<uint32_t>(NB, 1)));
                {
                    int cus = 256;
                    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) != hipSuccess || cus < 1) cus = 256;
                    const double slots = (form == 3 ? 2.0 : 4.0) * cus;
                    double best = 0.0;
                    for (uint32_t sl = slices; sl <= std::min<uint32_t>(nj, slices + 7); ++sl) {
                        const double rounds = double(NB) * sl / slots, eff = rounds / std::ceil(rounds);
                        if (eff > best + 0.02) best = eff, slices = sl;
                        if (eff >= 0.95) break;
                    }
                }
#define RS_LEAF_LOOP(HPT_, CHUNK_)                                                                                                                                     \
    hipLaunchKernelGGL((k_br_terminal_sorted_loop<HPT_, CHUNK_, (HPT_ == kBrHandsPerThread)>), dim3(NB, slices), dim3(kBrBlock), lds, t->stream, me.d_hands, me.d_mask, me.d_pw, me.n_hands, op.n_hands, \
                       d_bmask, me.index, d_jobs + o_leaves + lo, nj)
                if (form == 0) RS_LEAF_LOOP(1, 4);
                else if (form == 1) RS_LEAF_LOOP(2, 8);
                else if (form == 2) RS_LEAF_LOOP(4, 16);
                else RS_LEAF_LOOP(kBrHandsPerThread, kBrChunkMax);
#undef RS_LEAF_LOOP
            } else if (sorted) {
                const size_t lds = (size_t(op.n_hands) * 2 + 52 * kBrCardHolders + 65) * sizeof(double);
                hipLaunchKernelGGL(k_br_terminal_sorted_jobs, dim3(NB, nj), dim3(kBrBlock), lds, t->stream, me.d_hands, me.d_mask, me.d_pw, me.n_hands, op.n_hands, d_bmask, me.index,
                                   d_jobs + o_leaves + lo);
            } else {
                const size_t lds = size_t(op.n_hands) * (sizeof(double) + sizeof(uint64_t) + sizeof(uint32_t));
                hipLaunchKernelGGL(k_br_terminal_boards_jobs, dim3(grid1(me.n_hands), NB, nj), dim3(kBrBlock), lds, t->stream, me.d_mask, me.d_score, me.d_pw, me.n_hands, op.d_mask,
                                   op.d_score, op.n_hands, d_bmask, d_jobs + o_leaves + lo);
            }
            err = hipGetLastError();
            ++n_launches;
        }
        for (int d = max_depth; d >= 0 && err == hipSuccess; --d) {   // values, deepest level first
            if (const uint32_t nj = uint32_t(up_wave[size_t(d)].size())) {
                uint32_t ncl = 0;
                for (const BrJob &j : up_wave[size_t(d)]) ncl = std::max(ncl, j.n_clusters);
                const uint32_t blocks = uint32_t(std::min<size_t>((size_t(ncl) * 64 + kBrBlock - 1) / kBrBlock, 8192));
#define RS_OWNWJ(DT_) hipLaunchKernelGGL((k_br_own_wave_jobs<DT_>), dim3(blocks, nj), dim3(kBrBlock), 0, t->stream, t->d_ssum, d_jobs + o_wave[size_t(d)], me.n_pad, mode)
                RS_BR_DT(t->dtype, RS_OWNWJ);
#undef RS_OWNWJ
                err = hipGetLastError();
                ++n_launches;
            }
            if (const uint32_t nj = uint32_t(up_cols[size_t(d)].size())) {
                uint32_t ns = 0;
                for (const BrJob &j : up_cols[size_t(d)]) ns = std::max(ns, j.n_sets);
#define RS_OWNC(DT_) hipLaunchKernelGGL((k_br_own_cols_jobs<DT_>), dim3(grid1(ns), nj), dim3(kBrBlock), 0, t->stream, t->d_ssum, d_jobs + o_cols[size_t(d)], me.n_pad, mode)
                RS_BR_DT(t->dtype, RS_OWNC);
#undef RS_OWNC
                err = hipGetLastError();
                ++n_launches;
            }
            for (int r = 0; r < RS_MAX_ROUNDS && err == hipSuccess; ++r)
                if (const uint32_t nj = uint32_t(up_grp[r][size_t(d)].size())) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_br_own_grouped_jobs), hipFuncAttributeMaxDynamicSharedMemorySize, int(me.group_lds[r]));
                    hipLaunchKernelGGL(k_br_own_grouped_jobs, dim3(me.groups[r].n_groups, nj), dim3(kBrGroupBlock), me.group_lds[r], t->stream, d_jobs + o_grp[r][size_t(d)],
                                       me.groups[r], me.n_pad);
                    err = hipGetLastError();
                    ++n_launches;
                }
            if (const uint32_t nj = uint32_t(up_own[size_t(d)].size())) {
                uint32_t ncl = 0;
                for (const BrJob &j : up_own[size_t(d)]) ncl = std::max(ncl, j.n_clusters);
#define RS_OWNJ(DT_) hipLaunchKernelGGL((k_br_own_jobs<DT_>), dim3(grid1(ncl), nj), dim3(kBrBlock), 0, t->stream, t->d_ssum, d_jobs + o_own[size_t(d)], me.n_pad, mode)
                RS_BR_DT(t->dtype, RS_OWNJ);
#undef RS_OWNJ
                if (err == hipSuccess) err = hipGetLastError();
                ++n_launches;
            }
            if (const uint32_t nj = uint32_t(up_sum[size_t(d)].size())) {
                hipLaunchKernelGGL(k_br_sum_jobs, dim3(grid1(me.n), nj), dim3(kBrBlock), 0, t->stream, d_jobs + o_sum[size_t(d)], me.n, me.n_pad);
                if (err == hipSuccess) err = hipGetLastError();
                ++n_launches;
            }
        }
        if (err == hipSuccess) err = hipStreamSynchronize(t->stream);   // d_jobs is freed below
        if (d_jobs) (void)hipFree(d_jobs);
        return err == hipSuccess ? RS_OK : hip_fail(err, "rs_best_response (level plan)");
    }
    int n_launches = 0;

    void walk(int id, const double *q, double *v_out, int level) {
        if (err != hipSuccess) return;
        const rs_tree_node &n = tree->nodes[size_t(id)];
        const BrSide &me = side[p], &op = side[1 - p];
        if (n.kind == RS_NODE_TERMINAL) {
            const int unc = n.ttype == RS_TERM_UNCONTESTED;
            const double pot = double(float(n.value));   // tn.value as f32 (cfr.rs:316)
            const double value = unc ? (p == int(n.last_to_act) ? -pot : pot) : pot;
            if (sorted) {
                const size_t lds = (size_t(op.n_hands) * 2 + 52 * kBrCardHolders + 65) * sizeof(double);
                hipLaunchKernelGGL(k_br_terminal_sorted, dim3(NB), dim3(kBrBlock), lds, t->stream, me.d_hands, me.d_mask, me.d_pw, me.n_hands, q, op.n_hands, d_bmask, me.index,
                                   unc, value, v_out);
                err = hipGetLastError();
                return;
            }
            const size_t lds = size_t(op.n_hands) * (sizeof(double) + sizeof(uint64_t) + sizeof(uint32_t));
            hipLaunchKernelGGL(k_br_terminal_boards, dim3(grid1(me.n_hands), NB), dim3(kBrBlock), lds, t->stream, me.d_mask, me.d_score, me.d_pw, me.n_hands, op.d_mask,
                               op.d_score, q, op.n_hands, d_bmask, unc, value, v_out);
            err = hipGetLastError();
            return;
        }
        if (n.kind != RS_NODE_ACTION) return walk(n.children[0], q, v_out, level);   // chance nodes pass through (cfr.rs:306-313)
        const BrNodeRow row = row_of(t, n.index);
        const int r = n.round_idx;
        double *vch = v_level[size_t(level)];
        if (int(n.player) == p) {
            for (int a = 0; a < n.n_children; ++a) walk(n.children[a], q, vch + size_t(a) * me.n_pad, level + 1);
            if (err != hipSuccess) return;
            if (size_t(me.n) >= size_t(me.n_clusters[r]) * 32) {   // many lanes per info set: a wave each
                const uint32_t blocks = uint32_t(std::min<size_t>((size_t(me.n_clusters[r]) * 64 + kBrBlock - 1) / kBrBlock, 8192));
#define RS_OWNW(DT_)                                                                                                                                    \
    hipLaunchKernelGGL((k_br_own_wave<DT_>), dim3(blocks), dim3(kBrBlock), 0, t->stream, t->d_ssum, row, me.d_start[r], me.d_order[r], me.n_clusters[r], \
                       me.n_pad, vch, mode, v_out)
                RS_BR_DT(t->dtype, RS_OWNW);
#undef RS_OWNW
            } else {
#define RS_OWN(DT_)                                                                                                                                            \
    hipLaunchKernelGGL((k_br_own<DT_>), dim3(grid1(me.n_clusters[r])), dim3(kBrBlock), 0, t->stream, t->d_ssum, row, me.d_start[r], me.d_order[r], me.n_clusters[r], \
                       me.n_pad, vch, mode, v_out)
                RS_BR_DT(t->dtype, RS_OWN);
#undef RS_OWN
            }
        } else {
            double *qch = q_level[size_t(level)];
#define RS_OPP(DT_) hipLaunchKernelGGL((k_br_opp_reach<DT_>), dim3(grid1(op.n)), dim3(kBrBlock), 0, t->stream, t->d_ssum, row, op.d_cid[r], op.n, op.n_pad, q, qch)
            RS_BR_DT(t->dtype, RS_OPP);
#undef RS_OPP
            err = hipGetLastError();
            for (int a = 0; a < n.n_children; ++a) walk(n.children[a], qch + size_t(a) * op.n_pad, vch + size_t(a) * me.n_pad, level + 1);
            if (err != hipSuccess) return;
            hipLaunchKernelGGL(k_br_sum, dim3(grid1(me.n)), dim3(kBrBlock), 0, t->stream, vch, uint32_t(n.n_children), me.n, me.n_pad, v_out);
        }
        err = hipGetLastError();
    }
};

static int tree_depth(const std::vector<rs_tree_node> &nodes, int id) {
    const rs_tree_node &n = nodes[size_t(id)];
    int d = 0;
    for (int a = 0; a < n.n_children; ++a) d = std::max(d, tree_depth(nodes, n.children[a]));
    return d + (n.kind == RS_NODE_ACTION ? 1 : 0);   // only action nodes take a level of child buffers
}

// run-outs of an initial board in the enumeration order of the lanes: the first new card most significant, cards ascending among those still in the deck
static size_t enumerate_runouts(const uint8_t *board0, int n_board0, std::vector<uint8_t> *out) {
    uint8_t deck[52];
    int D = 0;
    for (int c = 0; c < 52; ++c) {
        bool used = false;
        for (int i = 0; i < n_board0; ++i) used = used || board0[i] == c;
        if (!used) deck[D++] = uint8_t(c);
    }
    const int K = 5 - n_board0;
    size_t nb = 0;
    auto emit = [&](int c3, int c4) {
        if (out) {
            uint8_t row[5];
            for (int k = 0; k < n_board0; ++k) row[k] = board0[k];
            if (K == 2) row[3] = uint8_t(c3);
            if (K >= 1) row[4] = uint8_t(c4);
            out->insert(out->end(), row, row + 5);
        }
        ++nb;
    };
    if (K == 0) emit(0, 0);
    else if (K == 1)
        for (int i = 0; i < D; ++i) emit(0, deck[i]);
    else
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j)
                if (j != i) emit(deck[i], deck[j]);
    return nb;
}

}  // namespace rs

extern "C" {

size_t rs_br_runouts(const uint8_t *board0, int n_board0, uint8_t *out_cards) {
    if (!board0 || n_board0 < 3 || n_board0 > 5) return 0;
    std::vector<uint8_t> cards;
    const size_t nb = enumerate_runouts(board0, n_board0, out_cards ? &cards : nullptr);
    if (out_cards) std::memcpy(out_cards, cards.data(), cards.size());
    return nb;
}

}  // extern "C"

namespace rs {

// Everything of a best-response pass that depends on the GAME only (tree shape, ranges, board, cluster ids) and not on the table's contents: lane scores and masks, the lanes of
// every info set in ascending order, the rank-order index of the showdowns, the walk's buffers.  A caller that asks again for the same game (the trainer's exploitability
// ticks) keeps the object and pays for the walk alone.
// The run-outs fall into components (two run-outs that hold lanes of one info set belong together); where the largest is small, components are packed into groups of
// `cap` run-outs and the round's own nodes go to k_br_own_grouped_jobs.  Random bucket files tie every run-out to every other: one component, no groups.
static void build_groups(BrRun &run, BrSide &s, int r, size_t NB, size_t H, const std::vector<uint32_t> &start, const std::vector<uint32_t> &order) {
    const uint32_t NC = s.n_clusters[r];
    if (H == 0 || H > size_t(kBrGroupPerThread) * kBrGroupBlock) return;
    std::vector<uint32_t> parent(NB);
    for (size_t b = 0; b < NB; ++b) parent[b] = uint32_t(b);
    auto find = [&](uint32_t b) {
        while (parent[b] != b) b = parent[b] = parent[parent[b]];
        return b;
    };
    for (uint32_t c = 0; c < NC; ++c) {
        const uint32_t lo = start[c], hi = start[size_t(c) + 1];
        if (lo == hi) continue;
        uint32_t b0 = find(uint32_t(order[lo] / H));
        for (uint32_t i = lo + 1; i < hi; ++i) {
            const uint32_t b = find(uint32_t(order[i] / H));
            if (b == b0) continue;
            if (b < b0) parent[b0] = b, b0 = b;
            else parent[b] = b0;
        }
    }
    std::vector<uint32_t> size(NB, 0);
    uint32_t largest = 0;
    for (size_t b = 0; b < NB; ++b) largest = std::max(largest, ++size[find(uint32_t(b))]);
    const uint32_t cap_max = uint32_t(size_t(kBrGroupPerThread) * kBrGroupBlock / H);
    if (largest > cap_max) return;
    const uint32_t cap = std::min(cap_max, std::max<uint32_t>(largest, std::max<uint32_t>(1, uint32_t(kBrGroupLanes / H))));
    // components in the order of their smallest cluster id, packed greedily: neighbours in that order hold neighbouring info sets -- the same 64-byte lines of the
    // strategy-sum rows (a lossless abstraction numbers its clusters hand by hand, then by the larger and the smaller of the two cards to come) -- and the reach kernel hands
    // neighbouring groups to one XCD
    std::vector<uint32_t> comp_order;
    {
        std::vector<char> seen(NB, 0);
        for (uint32_t c = 0; c < NC; ++c) {
            if (start[c] == start[size_t(c) + 1]) continue;
            const uint32_t root = find(uint32_t(order[start[c]] / H));
            if (!seen[root]) seen[root] = 1, comp_order.push_back(root);
        }
        for (size_t b = 0; b < NB; ++b)   // run-outs without a single info set (no hand fits): on their own, last
            if (find(uint32_t(b)) == b && !seen[b]) comp_order.push_back(uint32_t(b));
    }
    std::vector<uint32_t> group_of(NB, 0), slot_of(NB, 0), group_root(NB, 0xffffffffu);
    std::vector<uint32_t> runouts;
    uint32_t n_groups = 0, used = cap;
    for (uint32_t root : comp_order) {
        if (used + size[root] > cap) {
            ++n_groups;
            used = 0;
            runouts.resize(size_t(n_groups) * cap, 0xffffffffu);
        }
        group_root[root] = n_groups - 1;
        used += size[root];
    }
    for (size_t b = 0; b < NB; ++b) {   // members take their group's slots in ascending order
        const uint32_t g = group_root[find(uint32_t(b))];
        group_of[b] = g;
        uint32_t slot = 0;
        while (runouts[size_t(g) * cap + slot] != 0xffffffffu) ++slot;
        runouts[size_t(g) * cap + slot] = uint32_t(b);
        slot_of[b] = slot;
    }
    std::vector<uint32_t> cstart(size_t(n_groups) + 1, 0);
    for (uint32_t c = 0; c < NC; ++c)
        if (start[c] != start[size_t(c) + 1]) cstart[size_t(group_of[order[start[c]] / H]) + 1]++;
    uint32_t max_sets = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        max_sets = std::max(max_sets, cstart[size_t(g) + 1]);
        cstart[size_t(g) + 1] += cstart[g];
    }
    const size_t lds = (size_t(cap) * H + max_sets) * sizeof(double) + size_t(max_sets) * sizeof(uint32_t);
    if (lds > size_t(150) << 10) return;
    const uint32_t n_sets = cstart[n_groups];
    std::vector<uint32_t> lstart(size_t(n_sets) + 1, 0), at(cstart.begin(), cstart.end() - 1), set_of(NC, 0xffffffffu);
    for (uint32_t c = 0; c < NC; ++c)   // ascending cluster id within a group
        if (start[c] != start[size_t(c) + 1]) {
            const uint32_t k = at[group_of[order[start[c]] / H]]++;
            set_of[c] = k;
            lstart[size_t(k) + 1] = start[size_t(c) + 1] - start[c];
        }
    for (uint32_t k = 0; k < n_sets; ++k) lstart[size_t(k) + 1] += lstart[k];
    std::vector<uint16_t> llane(lstart[n_sets]), lane_set(size_t(n_groups) * cap * H, uint16_t(0xffff));
    for (uint32_t c = 0; c < NC; ++c) {
        if (set_of[c] == 0xffffffffu) continue;
        uint32_t o = lstart[set_of[c]];
        const uint32_t grp = group_of[order[start[c]] / H];
        for (uint32_t i = start[c]; i < start[size_t(c) + 1]; ++i) {
            const uint32_t b = uint32_t(order[i] / H), local = uint32_t(slot_of[b] * H + (order[i] - b * H));
            llane[o++] = uint16_t(local);
            lane_set[size_t(grp) * cap * H + local] = uint16_t(set_of[c] - cstart[grp]);
        }
    }
    BrGroups &g = s.groups[r];
    g.runouts = run.upload(runouts);
    g.cstart = run.upload(cstart);
    g.lstart = run.upload(lstart);
    g.llane = run.upload(llane);
    g.lane_set = run.upload(lane_set);
    {
        std::vector<uint32_t> set_cluster(n_sets, 0);
        for (uint32_t c = 0; c < NC; ++c)
            if (set_of[c] != 0xffffffffu) set_cluster[set_of[c]] = c;
        g.set_cluster = run.upload(set_cluster);
    }
    g.n_groups = n_groups;
    g.cap = cap;
    g.n_hands = uint32_t(H);
    g.max_sets = max_sets;
    s.group_lds[r] = lds;
    s.grouped[r] = n_groups > 0;
}

// Own nodes by columns (k_br_own_cols_jobs): the info sets ordered by their first lane, their lane lists transposed.  Taken where neighbours in that order mostly hold
// neighbouring lanes (a lossless abstraction: the same cards under the next hand) and the lists are of a length a single thread walks (32 .. 2 048 lanes per info set).
static void build_columns(BrRun &run, BrSide &s, int r, const std::vector<uint32_t> &start, const std::vector<uint32_t> &order) {
    const uint32_t NC = s.n_clusters[r];
    std::vector<uint32_t> perm;
    uint32_t kmax = 0;
    for (uint32_t c = 0; c < NC; ++c)
        if (start[c] != start[size_t(c) + 1]) {
            perm.push_back(c);
            kmax = std::max(kmax, start[size_t(c) + 1] - start[c]);
        }
    if (perm.size() < 2 || kmax > 2048 || size_t(kmax) * perm.size() > (size_t(1) << 27)) return;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return order[start[a]] < order[start[b]]; });
    size_t side_by_side = 0;
    for (size_t i = 0; i + 1 < perm.size(); ++i) side_by_side += order[start[perm[i + 1]]] == order[start[perm[i]]] + 1;
    if (side_by_side * 2 < perm.size()) return;
    const size_t n_sets = perm.size();
    std::vector<uint32_t> tord(size_t(kmax) * n_sets, 0xffffffffu);
    for (size_t i = 0; i < n_sets; ++i)
        for (uint32_t k = 0, lo = start[perm[i]], n = start[size_t(perm[i]) + 1] - lo; k < n; ++k) tord[size_t(k) * n_sets + i] = order[lo + k];
    s.d_tord[r] = run.upload(tord);
    s.d_perm[r] = run.upload(perm);
    s.n_sets[r] = uint32_t(n_sets);
    s.kmax[r] = kmax;
}

int br_prepare(rs_table *t, const rs_tree *tree, const uint8_t *board0, int n_board0, const uint8_t *hands_p0, size_t n_hands_p0, const uint8_t *hands_p1,
               size_t n_hands_p1, const uint32_t *const *cluster, int n_rounds, bool sorted, BrRun **prepared) {
    if (!t || !tree || !board0 || !hands_p0 || !hands_p1 || !cluster || !prepared) return fail(RS_ERR_INVALID, "rs_best_response: NULL argument");
    *prepared = nullptr;
    if (tree->nodes.empty()) return fail(RS_ERR_INVALID, "rs_best_response: empty tree");
    if (n_board0 < 3 || n_board0 > 5) return fail(RS_ERR_INVALID, "rs_best_response: the initial board has 3, 4 or 5 cards (state.rs:59-64)");
    const int K = 5 - n_board0, D = 52 - n_board0;
    if (n_rounds < 1 || n_rounds > K + 1 || n_rounds > RS_MAX_ROUNDS) return fail(RS_ERR_INVALID, "rs_best_response: 1 .. (6 - board cards) betting rounds");
    if (n_hands_p0 == 0 || n_hands_p1 == 0 || n_hands_p0 > 1326 || n_hands_p1 > 1326) return fail(RS_ERR_INVALID, "rs_best_response: 1..1326 hands per range");
    if (int rc = check_tree_against_table(t, tree, "rs_best_response")) return rc;
    uint32_t n_clusters[RS_MAX_ROUNDS][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (const rs_tree_node &n : tree->nodes) {
        if (n.kind != RS_NODE_ACTION) continue;
        const rs_node_desc &nd = t->nodes[size_t(n.index)];
        if (n.round_idx >= n_rounds) return fail(RS_ERR_INVALID, "rs_best_response: the tree has more betting rounds than n_rounds");
        if (nd.n_boards != 1 || t->tiled(n.index)) return fail(RS_ERR_UNSUPPORTED, "rs_best_response: tables of the reference's shape [action node][cluster] (n_boards = 1)");
        if (n.player > 1) return fail(RS_ERR_INVALID, "rs_best_response: two players");
        uint32_t &nc = n_clusters[n.round_idx][n.player];
        if (nc && nc != nd.n_clusters) return fail(RS_ERR_INVALID, "rs_best_response: a player's nodes of one round differ in cluster count");
        nc = nd.n_clusters;
    }
    uint64_t board_mask = 0;
    for (int i = 0; i < n_board0; ++i) {
        if (board0[i] >= 52 || (board_mask >> board0[i] & 1)) return fail(RS_ERR_INVALID, "rs_best_response: the board is distinct cards below 52");
        board_mask |= 1ull << board0[i];
    }
    const uint8_t *hands[2] = {hands_p0, hands_p1};
    const size_t n_hands[2] = {n_hands_p0, n_hands_p1};
    std::vector<uint64_t> mask[2];
    for (int p = 0; p < 2; ++p) {
        mask[p].resize(n_hands[p]);
        for (size_t h = 0; h < n_hands[p]; ++h) {
            const uint8_t a = hands[p][2 * h], b = hands[p][2 * h + 1];
            if (a >= 52 || b >= 52 || a == b) return fail(RS_ERR_INVALID, "rs_best_response: bad hole cards in a range");
            mask[p][h] = 1ull << a | 1ull << b;
            if (mask[p][h] & board_mask) return fail(RS_ERR_INVALID, "rs_best_response: a range combo uses a board card");
        }
    }
    if (sorted)   // the rank-order index lists the opponent's hands by card (at most kBrCardHolders per card) and knows ONE opponent hand per card pair (`same`)
        for (int p = 0; p < 2; ++p) {
            std::vector<uint64_t> m(mask[p]);
            std::sort(m.begin(), m.end());
            if (std::adjacent_find(m.begin(), m.end()) != m.end())
                return fail(RS_ERR_INVALID, "rs_best_response: RS_BR_SORTED takes ranges of distinct combos (a combo occurs twice in player " + std::to_string(p) + "'s range)");
        }
    std::vector<uint8_t> cards;
    const size_t NB = enumerate_runouts(board0, n_board0, &cards);
    if (NB * std::max(n_hands[0], n_hands[1]) >= (size_t(1) << 31)) return fail(RS_ERR_UNSUPPORTED, "rs_best_response: too many lanes");
    std::vector<uint64_t> bmask(NB, 0);
    std::vector<uint32_t> cnt0(NB, 0);
    for (size_t b = 0; b < NB; ++b) {
        for (int i = n_board0; i < 5; ++i) bmask[b] |= 1ull << cards[b * 5 + size_t(i)];
        for (size_t h = 0; h < n_hands[0]; ++h) cnt0[b] += (mask[0][h] & bmask[b]) == 0;
    }
    size_t per_prefix[RS_MAX_ROUNDS] = {1, 1, 1};
    for (int r = 0; r < n_rounds; ++r) {   // completions left after r new cards
        size_t left = 1;
        for (int i = r; i < K; ++i) left *= size_t(D - i);
        per_prefix[r] = left;
    }
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    std::unique_ptr<BrRun> owner(new BrRun);
    BrRun &run = *owner;
    run.t = t;
    run.tree = tree;
    run.sorted = sorted;
    run.NB = uint32_t(NB);
    run.d_bmask = run.upload(bmask);
    uint8_t *d_boards = run.upload(cards);
    uint32_t *d_cnt0 = run.upload(cnt0);
    for (int p = 0; p < 2; ++p) {
        BrSide &s = run.side[p];
        s.n_hands = uint32_t(n_hands[p]);
        s.n = uint32_t(NB * n_hands[p]);
        s.n_pad = uint32_t(round_up(size_t(s.n), 64));
        std::vector<uint8_t> hv(hands[p], hands[p] + 2 * n_hands[p]);
        uint8_t *d_hands = run.upload(hv);
        s.d_hands = d_hands;
        s.d_mask = run.upload(mask[p]);
        s.d_score = run.dalloc<uint32_t>(s.n);
        if (run.err == hipSuccess) {
            hipLaunchKernelGGL(k_lane_scores, dim3(grid1(s.n)), dim3(kBrBlock), 0, t->stream, d_hands, s.n_hands, d_boards, uint32_t(n_board0), uint32_t(NB), s.d_score);
            run.err = hipGetLastError();
        }
        for (int r = 0; r < n_rounds; ++r) {
            const uint32_t *src = cluster[size_t(r) * 2 + size_t(p)];
            s.n_clusters[r] = n_clusters[r][p];
            if (!s.n_clusters[r]) continue;   // the player has no node in this round
            if (!src) return fail(RS_ERR_INVALID, "rs_best_response: no cluster ids for round " + std::to_string(r) + " player " + std::to_string(p));
            // lane-level ids and, per info set, its lanes in ascending order.  A blocked lane (its hand uses a card of the run-out: no such deal) belongs to no
            // info set: it is listed in an extra segment after the last cluster (k_br_own writes its 0) and looks up cluster 0 where an id is needed but not used
            const uint32_t NC = s.n_clusters[r];
            std::vector<uint32_t> cv(s.n), seg(s.n), start(size_t(NC) + 2, 0), order(s.n);
            for (size_t b = 0; b < NB; ++b)
                for (size_t h = 0; h < n_hands[p]; ++h) {
                    uint32_t k = src[(b / per_prefix[r]) * n_hands[p] + h];
                    const bool blocked = (mask[p][h] & bmask[b]) != 0;
                    if (!blocked && k >= NC)
                        return fail(RS_ERR_OOB, "rs_best_response: cluster id " + std::to_string(k) + " of player " + std::to_string(p) + " is outside the table");
                    cv[b * n_hands[p] + h] = blocked ? 0u : k;
                    seg[b * n_hands[p] + h] = blocked ? NC : k;
                }
            for (uint32_t k : seg) start[size_t(k) + 1]++;
            for (size_t c = 0; c <= NC; ++c) start[c + 1] += start[c];
            {
                std::vector<uint32_t> fill(start.begin(), start.end() - 1);
                for (size_t l = 0; l < seg.size(); ++l) order[fill[seg[l]]++] = uint32_t(l);
            }
            s.d_cid[r] = run.upload(cv);
            s.d_start[r] = run.upload(start);
            s.d_order[r] = run.upload(order);
            if (size_t(s.n) < size_t(NC) * 32) build_groups(run, s, r, NB, n_hands[p], start, order);
            else build_columns(run, s, r, start, order);   // (what neither takes goes a thread per info set, or a wave where the info sets hold many lanes)
        }
    }
    if (sorted)   // the node-independent half of the rank-order showdowns: once per call
        for (int p = 0; p < 2; ++p) {
            BrSide &me = run.side[p], &op = run.side[1 - p];
            BrIndex &ix = me.index;
            ix.ord = run.dalloc<uint16_t>(NB * op.n_hands);
            ix.nv = run.dalloc<uint16_t>(NB);
            ix.cl = run.dalloc<uint16_t>(NB * 52 * kBrCardHolders);
            ix.cc = run.dalloc<uint8_t>(NB * 52);
            ix.nl = run.dalloc<uint16_t>(NB * me.n_hands);
            ix.nle = run.dalloc<uint16_t>(NB * me.n_hands);
            ix.kl = run.dalloc<uint8_t>(NB * me.n_hands * 2);
            ix.kle = run.dalloc<uint8_t>(NB * me.n_hands * 2);
            std::vector<int16_t> same(n_hands[p], -1);
            for (size_t h = 0; h < n_hands[p]; ++h)
                for (size_t g = 0; g < n_hands[1 - p]; ++g)
                    if (mask[p][h] == mask[1 - p][g]) same[h] = int16_t(g);
            ix.same = run.upload(same);
            if (run.err == hipSuccess) {
                hipLaunchKernelGGL(k_br_index, dim3(uint32_t(NB)), dim3(kBrBlock), 0, t->stream, me.d_hands, me.n_hands, me.d_mask, me.d_score, op.d_hands, op.n_hands, op.d_mask,
                                   op.d_score, run.d_bmask, ix);
                run.err = hipGetLastError();
            }
        }
    // the deal distribution of generate_hand (cfr.rs:100-143): the run-out uniform over ordered completions, player 0's combo uniform over the combos of its range
    // that avoid the full board, player 1's over those that avoid board and player 0: P = P(B) [disjoint] / (N0(B) N1(B, h0)); the whole weight rides on player 0's lane
    double pb = 1.0;
    for (int i = 0; i < K; ++i) pb /= double(D - i);
    double *d_w0 = run.dalloc<double>(run.side[0].n);
    if (run.err == hipSuccess) {
        hipLaunchKernelGGL(k_br_weights, dim3(grid1(run.side[0].n)), dim3(kBrBlock), 0, t->stream, run.side[0].d_mask, run.side[0].n_hands, run.side[1].d_mask,
                           run.side[1].n_hands, run.d_bmask, d_cnt0, uint32_t(NB), pb, d_w0);
        run.err = hipGetLastError();
    }
    std::vector<double> ones(run.side[1].n, 1.0);
    double *d_ones = run.upload(ones);
    run.side[0].d_init_q = run.side[0].d_pw = d_w0;
    run.side[1].d_init_q = run.side[1].d_pw = d_ones;
    // the walk's buffers are allocated by every call (br_execute) and freed when it returns: a kept object holds what depends on the game only
    run.n_pad_max = std::max(run.side[0].n_pad, run.side[1].n_pad);
    run.fill_depths();
    if (run.err != hipSuccess) {
        (void)hipStreamSynchronize(t->stream);   // drain the stream before ~BrRun frees what queued kernels may still touch
        return hip_fail(run.err, "rs_best_response");
    }
    *prepared = owner.release();
    return RS_OK;
}

void br_free(BrRun *run) { delete run; }
size_t br_held_bytes(const BrRun *run) { return run ? run->game_bytes + run->ws_bytes : 0; }
void br_release_workspace(BrRun *run) {
    if (run) run->release_workspace();
}
int br_last_launches(const BrRun *run) { return run && run->last_level_plan ? run->last_launches : -1; }

// the walk: both traversers against the table as it stands
int br_execute(BrRun *prepared, int mode, double *out) {
    if (!prepared || !out) return fail(RS_ERR_INVALID, "rs_best_response: NULL argument");
    if (mode != RS_BR_MAX && mode != RS_BR_AVERAGE) return fail(RS_ERR_INVALID, "rs_best_response: mode is RS_BR_MAX or RS_BR_AVERAGE (| RS_BR_SORTED)");
    BrRun &run = *prepared;
    rs_table *t = run.t;
    if (int rc = table_settle(t)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    run.mode = mode;
    run.err = hipSuccess;
    // The level plan wants a buffer per tree edge (full 1 176-combo ranges from a flop on the 706-node tree: 59 GB, and a hipMalloc of that size takes seconds; 200 combos: 10 GB)
    // and buys launches, not kernel time (4 300 -> 66 per call; at full ranges both orders spend 0.24-0.27 s in their kernels): it is taken while it fits kBrLevelPlanBytes and half
    // of the free memory, else the depth-first walk runs with its two buffers per tree depth.  The workspace is allocated by the first call and KEPT with the object
    // (br_workspace_bytes / br_release_workspace: a trainer holds one object per showdown mode).
    if (!run.ws) {
        const size_t need = std::max(run.level_plan_bytes(0), run.level_plan_bytes(1));
        size_t free_b = 0, total_b = 0;
        const bool levels = !knobs_resolve(nullptr).br_depth_first && need <= kBrLevelPlanBytes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 2;
        (void)hipGetLastError();
        if (levels && hipMalloc((void **)&run.ws, need) == hipSuccess) {
            run.ws_bytes = need;
            run.ws_levels = true;
        } else {
            (void)hipGetLastError();
            int max_a = 1;
            for (const rs_tree_node &n : run.tree->nodes) max_a = std::max(max_a, n.n_children);
            const int depth = tree_depth(run.tree->nodes, 0);
            const size_t per = size_t(max_a) * run.n_pad_max;
            run.ws_bytes = (size_t(2) * size_t(depth) * per + run.n_pad_max) * sizeof(double);
            if (hipMalloc((void **)&run.ws, run.ws_bytes) != hipSuccess) {
                run.ws = nullptr;
                run.ws_bytes = 0;
                return fail(RS_ERR_OOM, "rs_best_response: the walk's buffers");
            }
            run.ws_levels = false;
            for (int l = 0; l < depth; ++l) {
                run.q_level.push_back(run.ws + size_t(2 * l) * per);
                run.v_level.push_back(run.ws + size_t(2 * l + 1) * per);
            }
            run.d_root = run.ws + size_t(2) * size_t(depth) * per;
        }
    }
    const bool levels = run.ws_levels;
    double *ws = run.ws;
    int rc = RS_OK;
    run.last_level_plan = levels;
    run.last_launches = 0;
    std::vector<double> root(run.n_pad_max);
    for (int p = 0; p < 2 && run.err == hipSuccess && rc == RS_OK; ++p) {
        run.p = p;
        double *d_root = run.d_root;
        if (levels) rc = run.run_levels(ws, &d_root);
        else run.walk(0, run.side[1 - p].d_init_q, d_root, 0);
        run.last_launches += levels ? run.n_launches : 0;
        if (rc == RS_OK && run.err == hipSuccess) run.err = hipMemcpyAsync(root.data(), d_root, run.side[p].n * sizeof(double), hipMemcpyDeviceToHost, t->stream);
        if (rc == RS_OK && run.err == hipSuccess) run.err = hipStreamSynchronize(t->stream);
        double total = 0.0;
        for (uint32_t l = 0; l < run.side[p].n; ++l) total += root[l];   // ascending, like the oracle
        out[p] = total;
    }
    if (rc != RS_OK || run.err != hipSuccess) (void)hipStreamSynchronize(t->stream);   // on error drain the stream before the caller frees what queued kernels may still touch
    if (rc != RS_OK) return rc;
    if (run.err != hipSuccess) return hip_fail(run.err, "rs_best_response");
    return RS_OK;
}

}  // namespace rs

extern "C" {

int rs_best_response_rounds(rs_table *t, const rs_tree *tree, const uint8_t *board0, int n_board0, const uint8_t *hands_p0, size_t n_hands_p0, const uint8_t *hands_p1,
                            size_t n_hands_p1, const uint32_t *const *cluster, int n_rounds, int mode, double *out) {
    if (!out) return fail(RS_ERR_INVALID, "rs_best_response: NULL argument");
    const bool sorted = (mode & RS_BR_SORTED) != 0;
    mode &= ~RS_BR_SORTED;
    if (mode != RS_BR_MAX && mode != RS_BR_AVERAGE) return fail(RS_ERR_INVALID, "rs_best_response: mode is RS_BR_MAX or RS_BR_AVERAGE (| RS_BR_SORTED)");
    rs::BrRun *run = nullptr;
    if (int rc = rs::br_prepare(t, tree, board0, n_board0, hands_p0, n_hands_p0, hands_p1, n_hands_p1, cluster, n_rounds, sorted, &run)) return rc;
    const int rc = rs::br_execute(run, mode, out);
    rs::br_free(run);
    return rc;
}

// the single-round game on a full board (the configuration the reference ships): NB = 1
int rs_best_response(rs_table *t, const rs_tree *tree, const uint8_t *board, const uint8_t *hands_p0, size_t n_hands_p0, const uint32_t *cluster_p0,
                     const uint8_t *hands_p1, size_t n_hands_p1, const uint32_t *cluster_p1, int mode, double *out) {
    if (!t || !tree || !board || !hands_p0 || !hands_p1 || !cluster_p0 || !cluster_p1 || !out) return fail(RS_ERR_INVALID, "rs_best_response: NULL argument");
    for (const rs_tree_node &n : tree->nodes)
        if (n.kind == RS_NODE_PUBLIC_CHANCE) return fail(RS_ERR_UNSUPPORTED, "rs_best_response: a single-round tree (multi-round games: rs_best_response_rounds)");
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < i; ++j)
            if (board[i] >= 52 || board[i] == board[j]) return fail(RS_ERR_INVALID, "rs_best_response: the board is five distinct cards");
    const uint32_t *cl[2] = {cluster_p0, cluster_p1};
    return rs_best_response_rounds(t, tree, board, 5, hands_p0, n_hands_p0, hands_p1, n_hands_p1, cl, 1, mode, out);
}

}  // extern "C"
