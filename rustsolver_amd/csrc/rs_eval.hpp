// rs_eval.hpp -- the seven-card hold'em evaluator shared by the showdown-sign kernel (rs_kernels.hip) and the best-response
// kernels (rs_br.hip).  Device code only; include inside namespace-free context after <hip/hip_runtime.h>.
#pragma once
#include <cstdint>

namespace rs {

// ---- showdown evaluation on the device (cfr.rs:38-46, :324-333) ------------------------------------------
// card = 4 * rank + suit, rank 0..12 = 2..A (cfr.rs:592, bin/gen_ehs.rs:67-68).  The reference only ever COMPARES the two
// evaluate() scores, so any correct hold'em ranking yields the same sign.  Score = category << 20 | five 4-bit kickers.
__device__ __forceinline__ int straight_high(uint32_t ranks) {   // highest card of the best straight in a 13-bit rank set, -1 if none
    const uint32_t wheel = ranks | ((ranks >> 12) & 1u) << 13;   // not used directly; handled below
    (void)wheel;
    const uint32_t r = (ranks << 1) | ((ranks >> 12) & 1u);      // bit 0 = ace playing low, bit i+1 = rank i
    const uint32_t run = r & (r >> 1) & (r >> 2) & (r >> 3) & (r >> 4);   // bit i set: i..i+4 all present
    if (!run) return -1;
    return (31 - __builtin_clz(run)) + 4 - 1;                     // top bit index -> rank of the high card (wheel gives 3 = five)
}
__device__ __forceinline__ uint32_t top_bits(uint32_t mask, int n) {   // the n highest set ranks, packed 4 bits each, highest first
    uint32_t out = 0;
    for (int i = 0; i < n; i++) {
        const int hi = mask ? 31 - __builtin_clz(mask) : 0;
        out = (out << 4) | (mask ? (uint32_t)hi : 0u);
        mask &= ~(1u << hi);
    }
    return out;
}
// score of the seven cards given as four 13-bit rank masks, one per suit.  Rank multiplicities come from the suit masks directly:
// a rank held in >= 2 suits is in some pairwise AND, in >= 3 suits in some triple AND, in all four in the full AND.
__device__ __forceinline__ uint32_t evaluate_suits(const uint32_t (&suit_mask)[4]) {
    const uint32_t s0 = suit_mask[0], s1 = suit_mask[1], s2 = suit_mask[2], s3 = suit_mask[3];
    const uint32_t ranks = s0 | s1 | s2 | s3;
    const uint32_t ge2 = (s0 & s1) | (s0 & s2) | (s0 & s3) | (s1 & s2) | (s1 & s3) | (s2 & s3);
    const uint32_t ge3 = (s0 & s1 & s2) | (s0 & s1 & s3) | (s0 & s2 & s3) | (s1 & s2 & s3);
    const uint32_t quads = s0 & s1 & s2 & s3, trips = ge3 & ~quads, pairs = ge2 & ~ge3;
    uint32_t flush = 0;
#pragma unroll
    for (int su = 0; su < 4; su++)
        if (__builtin_popcount(suit_mask[su]) >= 5) flush = suit_mask[su];
    if (flush) {
        const int sf = straight_high(flush);
        if (sf >= 0) return (8u << 20) | (uint32_t)sf;                       // straight flush
    }
    if (quads) {
        const int qr = 31 - __builtin_clz(quads);
        return (7u << 20) | ((uint32_t)qr << 4) | top_bits(ranks & ~(1u << qr), 1);
    }
    if (trips && (pairs || (trips & (trips - 1)))) {                           // full house: best trips + best remaining pair/trips
        const int tr = 31 - __builtin_clz(trips);
        const uint32_t rest = (trips & ~(1u << tr)) | pairs;
        return (6u << 20) | ((uint32_t)tr << 4) | (uint32_t)(31 - __builtin_clz(rest));
    }
    if (flush) return (5u << 20) | top_bits(flush, 5);
    const int st = straight_high(ranks);
    if (st >= 0) return (4u << 20) | (uint32_t)st;
    if (trips) {
        const int tr = 31 - __builtin_clz(trips);
        return (3u << 20) | ((uint32_t)tr << 8) | top_bits(ranks & ~(1u << tr), 2);
    }
    if (pairs & (pairs - 1)) {                                                 // two pair (three pairs possible with 7 cards)
        const int p1 = 31 - __builtin_clz(pairs);
        const uint32_t rest = pairs & ~(1u << p1);
        const int p2 = 31 - __builtin_clz(rest);
        return (2u << 20) | ((uint32_t)p1 << 8) | ((uint32_t)p2 << 4) | top_bits(ranks & ~(1u << p1) & ~(1u << p2), 1);
    }
    if (pairs) {
        const int p1 = 31 - __builtin_clz(pairs);
        return (1u << 20) | ((uint32_t)p1 << 12) | top_bits(ranks & ~(1u << p1), 3);
    }
    return top_bits(ranks, 5);
}

__device__ __forceinline__ void add_card(uint32_t (&suit_mask)[4], uint32_t card) {
    const uint32_t bit = (card >> 2) < 13u ? 1u << (card >> 2) : 0u, su = card & 3u;   // a byte that is no card adds nothing
#pragma unroll
    for (int q = 0; q < 4; q++) suit_mask[q] |= su == (uint32_t)q ? bit : 0u;
}

}  // namespace rs
