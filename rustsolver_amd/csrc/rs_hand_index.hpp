// rs_hand_index.hpp -- canonical (suit-isomorphic) hand index, the arithmetic behind rust_poker::hand_indexer_s that
// RustSolver calls in get_cluster() (card_abstraction.rs:204-209, :245-251, :287-293) and generate_maps (:75-184).
// Shared by the host API and the gfx950 kernels of rs_cards.hip: everything here is plain integer code.
//
// rust_poker 0.1.5 (Cargo.toml:18, not vendored in the reference) wraps K. Waugh's indexer ("A Fast and Optimal Hand
// Isomorphism Algorithm", AAAI 2013 workshop).  The index is restated from the paper in a form that suits a GPU
// thread: per suit a mixed-radix index of rank sets relative to the ranks the suit has not used yet; suits ordered by
// ONE 5-comparator network on the key (count word descending, suit index ascending), so that interchangeable suits
// (equal count words) end up adjacent with their indices already sorted; each run of equal words ranked as a multiset;
// plus the offset of the hand's configuration.  Only two tables are read: rank_rank[8192] (u16) and the per-round
// offset table keyed by the ordered per-suit counts (a few thousand u64).  Equal-masks, suit sizes and the sorting
// permutation are computed, not looked up.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RS_HD __host__ __device__ __forceinline__
#else
#define RS_HD inline
#endif

namespace rs {

constexpr int kHiSuits = 4;
constexpr int kHiRanks = 13;
constexpr int kHiMaxRounds = 8;
constexpr int kHiMaxCards = 16;

struct HandIndexView {
    int32_t rounds;
    uint8_t cards_per_round[kHiMaxRounds];
    uint8_t round_start[kHiMaxRounds];
    const uint16_t *rank_rank;                  // [1 << 13] rank of a (shifted) rank set among the sets of its size
    const uint64_t *perm_offset[kHiMaxRounds];  // [round][ordered-count key] -> first index of the hand's configuration
};

RS_HD uint32_t hi_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// C(n, k), k <= 4; exact while n*(n-1)*(n-2)*(n-3) fits 64 bits (n < 65536; multiset group sizes of real poker hands are far
// smaller).  Closed forms with constant divisors: a dynamic-k division loop costs hundreds of GPU instructions per call.
RS_HD uint64_t hi_choose(uint64_t n, int k) {
    if ((uint64_t)k > n) return 0;
    switch (k) {
        case 0: return 1;
        case 1: return n;
        case 2: return (n * (n - 1)) >> 1;
        case 3: return (n * (n - 1) >> 1) * (n - 2) / 3;
        default: return ((n * (n - 1) >> 1) * (n - 2) / 3) * (n - 3) / 4;
    }
}

// C(n, k) for 0 <= n <= 13 (row n of Pascal's triangle, k <= 13)
RS_HD uint32_t hi_choose13(uint32_t n, uint32_t k) {
    constexpr uint16_t kPascal[14][14] = {
        {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 3, 3, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 4, 6, 4, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 5, 10, 10, 5, 1, 0, 0, 0, 0, 0, 0, 0, 0},
        {1, 6, 15, 20, 15, 6, 1, 0, 0, 0, 0, 0, 0, 0},
        {1, 7, 21, 35, 35, 21, 7, 1, 0, 0, 0, 0, 0, 0},
        {1, 8, 28, 56, 70, 56, 28, 8, 1, 0, 0, 0, 0, 0},
        {1, 9, 36, 84, 126, 126, 84, 36, 9, 1, 0, 0, 0, 0},
        {1, 10, 45, 120, 210, 252, 210, 120, 45, 10, 1, 0, 0, 0},
        {1, 11, 55, 165, 330, 462, 462, 330, 165, 55, 11, 1, 0, 0},
        {1, 12, 66, 220, 495, 792, 924, 792, 495, 220, 66, 12, 1, 0},
        {1, 13, 78, 286, 715, 1287, 1716, 1716, 1287, 715, 286, 78, 13, 1}};
    return n <= 13u && k <= 13u ? kPascal[n][k] : 0u;
}

struct HiSuit {
    uint32_t word;   // per-round card counts, round 0 in the top nibble
    uint64_t index;  // mixed-radix index of the suit's rank sets
    uint64_t mult;   // number of values `index` can take = size of the suit's class
};

// before(a, b): a sorts in front of b
RS_HD bool hi_before(const HiSuit &a, const HiSuit &b) { return a.word > b.word || (a.word == b.word && a.index <= b.index); }
RS_HD void hi_cswap(HiSuit &a, HiSuit &b) {
    if (!hi_before(a, b)) {
        const HiSuit t = a;
        a = b;
        b = t;
    }
}

// index of rounds 0..upto of `cards` (cards[round_start[r] + i], card = 4*rank + suit)
RS_HD uint64_t hand_index(const HandIndexView &v, int upto, const uint8_t *cards) {
    HiSuit s[kHiSuits];
    uint32_t used[kHiSuits];
    for (int i = 0; i < kHiSuits; ++i) {
        s[i].word = 0;
        s[i].index = 0;
        s[i].mult = 1;
        used[i] = 0;
    }
    uint32_t key = 0, key_mult = 1;
    for (int r = 0; r <= upto; ++r) {
        uint32_t ranks[kHiSuits] = {0, 0, 0, 0}, shifted[kHiSuits] = {0, 0, 0, 0};
        const uint32_t n_cards = v.cards_per_round[r];
        for (uint32_t i = 0; i < n_cards; ++i) {
            const uint32_t c = cards[v.round_start[r] + i];
            const uint32_t rank = (c >> 2) < (uint32_t)kHiRanks ? (c >> 2) : (uint32_t)kHiRanks - 1;   // a byte that is no card stays in the tables
            const uint32_t bit = 1u << rank;
#pragma unroll
            for (int q = 0; q < kHiSuits; ++q) {   // selects instead of indexed writes: keeps the arrays in registers
                const bool hit = (c & 3u) == (uint32_t)q;
                ranks[q] |= hit ? bit : 0u;
                shifted[q] |= hit ? bit >> hi_popc((bit - 1u) & used[q]) : 0u;
            }
        }
        uint32_t remaining = n_cards;
#pragma unroll
        for (int q = 0; q < kHiSuits; ++q) {
            const uint32_t n = hi_popc(ranks[q]);
            s[q].index += s[q].mult * v.rank_rank[shifted[q]];
            s[q].mult *= hi_choose13(kHiRanks - hi_popc(used[q]), n);
            s[q].word |= n << (4 * (kHiMaxRounds - 1 - r));
            used[q] |= ranks[q];
            if (q < kHiSuits - 1) {
                key += key_mult * n;
                key_mult *= remaining + 1;
                remaining -= n;
            }
        }
    }
    uint64_t index = v.perm_offset[upto][key];
    hi_cswap(s[0], s[1]);
    hi_cswap(s[2], s[3]);
    hi_cswap(s[0], s[2]);
    hi_cswap(s[1], s[3]);
    hi_cswap(s[1], s[2]);
    uint64_t multiplier = 1, part = 0;
    int run = 0;   // suits of the current run of equal words already folded into `part`
#pragma unroll
    for (int q = 0; q < kHiSuits; ++q) {
        part += hi_choose(s[q].index + (uint64_t)run, run + 1);   // multiset rank: sum C(x_t + t, t + 1), x ascending
        ++run;
        if (q == kHiSuits - 1 || s[q + 1].word != s[q].word) {
            index += multiplier * part;
            multiplier *= hi_choose(s[q].mult + (uint64_t)run - 1, run);
            part = 0;
            run = 0;
        }
    }
    return index;
}

// ---- bucket -> dense cluster id: the HashMap<u64, usize> of generate_maps (card_abstraction.rs:36, :208) as an
// open-addressing table: slot = {key + 1 (0 = empty), dense id}
struct DenseSlot {
    uint64_t key1;
    uint64_t value;
};
RS_HD uint64_t hi_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
constexpr uint32_t kDenseMissing = 0xffffffffu;
RS_HD uint32_t dense_lookup(const DenseSlot *slots, uint64_t mask, uint64_t bucket) {
    uint64_t h = hi_mix(bucket) & mask;
    for (uint64_t probes = 0; probes <= mask; ++probes) {
        const uint64_t k = slots[h].key1;
        if (k == bucket + 1) return (uint32_t)slots[h].value;
        if (k == 0) return kDenseMissing;
        h = (h + 1) & mask;
    }
    return kDenseMissing;
}

// ---- random bits of the deal sampler: counter hash (the reference's SmallRng is seeded from thread_rng, cfr.rs:197-199,
// and cannot be reproduced); draw k of deal d under `seed`
RS_HD uint64_t deal_bits(uint64_t seed, uint64_t deal, uint32_t k) {
    return hi_mix(hi_mix(seed ^ (deal + 1) * 0xD1B54A32D192ED03ull) + (uint64_t)k * 0x632BE59BD9B4E019ull);
}

}  // namespace rs
