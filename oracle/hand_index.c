/*
 * hand_index.c -- CPU oracle: the suit-isomorphic hand indexer behind rust_poker::hand_indexer_s
 * (K. Waugh, "A Fast and Optimal Hand Isomorphism Algorithm", AAAI 2013 workshop).  Its model is the algorithm of that
 * paper AS ITS AUTHOR'S PUBLIC C LIBRARY ORGANISES IT ("hand-isomorphism": hand_index.c / hand_index.h), which is what
 * rust_poker's `hand_indexer_s` wraps (the `_s` struct name and init / size / get_index / get_hand signatures at the
 * reference's call sites are that library's): the same table set -- nth_unset, rank_set_to_index, index_to_rank_set,
 * suit_permutations, per-round configuration / permutation tables -- under the same names, written out again here without
 * that library's source at hand (it is not in /root/reference and there is no network).  It is therefore a restatement
 * that deliberately follows the original's decomposition, so that the index ORDER has the best chance of matching
 * rust_poker's; that order is nevertheless UNPINNED (no reference fixture shows an index value), only the partition and the
 * sizes are pinned -- see hand_index.h.  The product's rs_hand_index.hpp is a different decomposition of the same function.
 * TEST INFRASTRUCTURE ONLY.
 *
 * The algorithm in one paragraph.  A hand is, per suit, the sequence of rank sets dealt in every round.
 * (1) Inside a suit, the rank set of a round is re-expressed relative to the ranks that suit has NOT used yet
 * ("shifted" ranks), ranked colexicographically among sets of its size, and the rounds are combined with a
 * mixed radix whose digits are C(13 - used, size): the suit index.  (2) The per-suit card COUNTS of all rounds
 * form the suit's configuration word; suits are ordered by descending word, the sorted 4-tuple is the hand's
 * configuration.  (3) Suits with identical words are interchangeable: their suit indices are sorted and ranked
 * as a multiset (combinatorial number system), different groups are combined by mixed radix, and the
 * configuration's offset (sum of the sizes of all smaller configurations) is added.
 */
#include "hand_index.h"

#include <stdlib.h>
#include <string.h>

#define SUITS ORC_HI_SUITS
#define RANKS ORC_HI_RANKS
#define MAX_ROUNDS ORC_HI_MAX_ROUNDS
#define ROUND_SHIFT 4
#define ROUND_MASK 0xfu
#define PERMS 24

static int tables_ready;
static uint8_t nth_unset[1 << RANKS][RANKS];         /* position of the i-th clear bit of `used` */
static uint32_t ncr_ranks[RANKS + 1][RANKS + 1];
static uint32_t rank_set_to_index[1 << RANKS];       /* colex rank of the set among the sets of its size */
static uint32_t index_to_rank_set[RANKS + 1][1 << RANKS];
static uint8_t suit_permutations[PERMS][SUITS];

/* C(n, k) for the multiset ranks; k <= 4.  The group sizes reached with <= 7 cards are far below 2^16. */
static uint64_t ncr_groups(uint64_t n, unsigned k) {
    if (k > n) return 0;
    unsigned __int128 r = 1;
    for (unsigned i = 0; i < k; ++i) r = r * (n - i) / (i + 1);
    return (uint64_t)r;
}

static void init_tables(void) {
    if (tables_ready) return;
    for (uint32_t used = 0; used < (1u << RANKS); ++used) {
        uint32_t k = 0;
        for (uint32_t bit = 0; bit < RANKS; ++bit)
            if (!(used >> bit & 1)) nth_unset[used][k++] = (uint8_t)bit;
        for (; k < RANKS; ++k) nth_unset[used][k] = 0xff;
    }
    ncr_ranks[0][0] = 1;
    for (int n = 1; n <= RANKS; ++n) {
        ncr_ranks[n][0] = ncr_ranks[n][n] = 1;
        for (int k = 1; k < n; ++k) ncr_ranks[n][k] = ncr_ranks[n - 1][k - 1] + ncr_ranks[n - 1][k];
    }
    /* colexicographic rank of a rank set among the sets of its size: sum over its elements s_1 < s_2 < ... of C(s_j, j); with m
     * cards over n still-free ranks (shifted ranks occupy positions 0..n-1) it runs over exactly C(n, m) values */
    for (uint32_t s = 0; s < (1u << RANKS); ++s) {
        uint32_t idx = 0, j = 1;
        for (uint32_t set = s; set; set &= set - 1, ++j) idx += ncr_ranks[__builtin_ctz(set)][j];
        rank_set_to_index[s] = idx;
        index_to_rank_set[__builtin_popcount(s)][idx] = s;
    }
    for (uint32_t p = 0; p < PERMS; ++p) {
        uint32_t index = p, used = 0;
        for (uint32_t j = 0; j < SUITS; ++j) {
            const uint32_t suit = index % (SUITS - j);
            index /= SUITS - j;
            const uint32_t shifted = nth_unset[used][suit];
            suit_permutations[p][j] = (uint8_t)shifted;
            used |= 1u << shifted;
        }
    }
    tables_ready = 1;
}

static uint32_t nibble(uint32_t word, int round) { return word >> (ROUND_SHIFT * (MAX_ROUNDS - round - 1)) & ROUND_MASK; }

/* ---- configurations: canonical per-suit count words ----------------------------------------------------- */
typedef void (*observe_fn)(int round, const uint32_t *words, void *data);

static void enum_configurations(const orc_hand_indexer *ix, int round, uint32_t remaining, int suit, uint32_t equal, uint32_t *used,
                                uint32_t *words, observe_fn observe, void *data) {
    if (suit == SUITS) {
        observe(round, words, data);
        if (round + 1 < ix->rounds) enum_configurations(ix, round + 1, ix->cards_per_round[round + 1], 0, equal, used, words, observe, data);
        return;
    }
    uint32_t lo = suit == SUITS - 1 ? remaining : 0;
    uint32_t hi = RANKS - used[suit];
    if (remaining < hi) hi = remaining;
    uint32_t previous = RANKS + 1;
    const uint32_t was_equal = equal >> suit & 1;   /* still tied with the suit before it: counts must not increase */
    if (was_equal) {
        previous = nibble(words[suit - 1], round);
        if (previous < hi) hi = previous;
    }
    const uint32_t old_word = words[suit], old_used = used[suit];
    for (uint32_t n = lo; n <= hi; ++n) {
        words[suit] = old_word | n << (ROUND_SHIFT * (MAX_ROUNDS - round - 1));
        used[suit] = old_used + n;
        const uint32_t new_equal = (equal & ~(1u << suit)) | (uint32_t)(was_equal && n == previous) << suit;
        enum_configurations(ix, round, remaining - n, suit + 1, new_equal, used, words, observe, data);
    }
    words[suit] = old_word;
    used[suit] = old_used;
}

static void count_configuration(int round, const uint32_t *words, void *data) {
    (void)words;
    ++((orc_hand_indexer *)data)->configurations[round];
}

static int compare_words(const uint32_t *a, const uint32_t *b) {
    for (int i = 0; i < SUITS; ++i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

/* keeps configuration[round][] sorted ascending (insertion), records suit sizes, the equal-mask and the class count */
static void tabulate_configuration(int round, const uint32_t *words, void *data) {
    orc_hand_indexer *ix = (orc_hand_indexer *)data;
    uint32_t id = ix->configurations[round]++;
    for (; id > 0 && compare_words(words, ix->configuration[round][id - 1]) < 0; --id) {
        memcpy(ix->configuration[round][id], ix->configuration[round][id - 1], sizeof(uint32_t) * SUITS);
        memcpy(ix->configuration_to_suit_size[round][id], ix->configuration_to_suit_size[round][id - 1], sizeof(uint32_t) * SUITS);
        ix->configuration_to_offset[round][id] = ix->configuration_to_offset[round][id - 1];
        ix->configuration_to_equal[round][id] = ix->configuration_to_equal[round][id - 1];
    }
    memcpy(ix->configuration[round][id], words, sizeof(uint32_t) * SUITS);
    uint64_t classes = 1;
    uint32_t equal = 0;
    for (int i = 0; i < SUITS;) {
        uint64_t size = 1;
        uint32_t left = RANKS;
        for (int r = 0; r <= round; ++r) {
            const uint32_t n = nibble(words[i], r);
            size *= ncr_ranks[left][n];
            left -= n;
        }
        int j = i + 1;
        while (j < SUITS && words[j] == words[i]) ++j;
        for (int k = i; k < j; ++k) ix->configuration_to_suit_size[round][id][k] = (uint32_t)size;
        classes *= ncr_groups(size + (uint64_t)(j - i) - 1, (unsigned)(j - i));   /* multisets of j-i suit indices */
        for (int k = i + 1; k < j; ++k) equal |= 1u << k;
        i = j;
    }
    ix->configuration_to_offset[round][id] = classes;   /* turned into a prefix sum by init */
    ix->configuration_to_equal[round][id] = equal >> 1;
}

/* ---- permutations: every ordered per-suit count vector -> (sorting permutation, configuration) ----------- */
static void enum_permutations(const orc_hand_indexer *ix, int round, uint32_t remaining, int suit, uint32_t *count, observe_fn observe,
                              void *data) {
    if (suit == SUITS) {
        observe(round, count, data);
        if (round + 1 < ix->rounds) enum_permutations(ix, round + 1, ix->cards_per_round[round + 1], 0, count, observe, data);
        return;
    }
    const uint32_t lo = suit == SUITS - 1 ? remaining : 0;
    const uint32_t old = count[suit];
    for (uint32_t n = lo; n <= remaining; ++n) {
        count[suit] = old | n << (ROUND_SHIFT * (MAX_ROUNDS - round - 1));
        enum_permutations(ix, round, remaining - n, suit + 1, count, observe, data);
    }
    count[suit] = old;
}

static uint32_t permutation_index(const orc_hand_indexer *ix, int round, const uint32_t *count) {
    uint32_t idx = 0, mult = 1;
    for (int r = 0; r <= round; ++r) {
        uint32_t remaining = ix->cards_per_round[r];
        for (int s = 0; s < SUITS - 1; ++s) {
            const uint32_t n = nibble(count[s], r);
            idx += mult * n;
            mult *= remaining + 1;
            remaining -= n;
        }
    }
    return idx;
}

static void count_permutation(int round, const uint32_t *count, void *data) {
    orc_hand_indexer *ix = (orc_hand_indexer *)data;
    const uint32_t idx = permutation_index(ix, round, count);
    if (ix->permutations[round] < idx + 1) ix->permutations[round] = idx + 1;
}

static void tabulate_permutation(int round, const uint32_t *count, void *data) {
    orc_hand_indexer *ix = (orc_hand_indexer *)data;
    const uint32_t idx = permutation_index(ix, round, count);
    uint32_t pi[SUITS] = {0, 1, 2, 3};
    for (int i = 1; i < SUITS; ++i) {   /* stable sort, descending count words */
        const uint32_t p = pi[i];
        int j = i;
        for (; j > 0 && count[p] > count[pi[j - 1]]; --j) pi[j] = pi[j - 1];
        pi[j] = p;
    }
    uint32_t pi_idx = 0, pi_mult = 1, pi_used = 0;
    for (int i = 0; i < SUITS; ++i) {
        const uint32_t bit = 1u << pi[i];
        const uint32_t smaller = (uint32_t)__builtin_popcount((bit - 1) & pi_used);
        pi_idx += (pi[i] - smaller) * pi_mult;
        pi_mult *= (uint32_t)(SUITS - i);
        pi_used |= bit;
    }
    ix->permutation_to_pi[round][idx] = pi_idx;
    uint32_t sorted[SUITS];
    for (int i = 0; i < SUITS; ++i) sorted[i] = count[pi[i]];
    uint32_t lo = 0, hi = ix->configurations[round];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        const int c = compare_words(sorted, ix->configuration[round][mid]);
        if (c < 0) hi = mid;
        else if (c == 0) lo = hi = mid;
        else lo = mid + 1;
    }
    ix->permutation_to_configuration[round][idx] = lo;
}

int orc_hand_indexer_init(int rounds, const uint8_t *cards_per_round, orc_hand_indexer *ix) {
    if (!ix || !cards_per_round || rounds < 1 || rounds > MAX_ROUNDS) return -1;
    init_tables();
    memset(ix, 0, sizeof(*ix));
    uint32_t total = 0;
    for (int r = 0; r < rounds; ++r) {
        if (cards_per_round[r] == 0 || cards_per_round[r] > 15) return -1;
        ix->round_start[r] = (uint8_t)total;
        total += cards_per_round[r];
        if (total > 52) return -1;
    }
    ix->rounds = rounds;
    memcpy(ix->cards_per_round, cards_per_round, (size_t)rounds);
    uint32_t used[SUITS] = {0}, words[SUITS] = {0};
    enum_configurations(ix, 0, cards_per_round[0], 0, (1u << SUITS) - 2, used, words, count_configuration, ix);
    for (int r = 0; r < rounds; ++r) {
        const size_t n = ix->configurations[r];
        ix->configuration_to_equal[r] = calloc(n, sizeof(uint32_t));
        ix->configuration_to_offset[r] = calloc(n, sizeof(uint64_t));
        ix->configuration[r] = calloc(n, sizeof(uint32_t[SUITS]));
        ix->configuration_to_suit_size[r] = calloc(n, sizeof(uint32_t[SUITS]));
        if (!ix->configuration_to_equal[r] || !ix->configuration_to_offset[r] || !ix->configuration[r] || !ix->configuration_to_suit_size[r]) {
            orc_hand_indexer_free(ix);
            return -1;
        }
    }
    memset(ix->configurations, 0, sizeof(ix->configurations));
    memset(used, 0, sizeof(used));
    memset(words, 0, sizeof(words));
    enum_configurations(ix, 0, cards_per_round[0], 0, (1u << SUITS) - 2, used, words, tabulate_configuration, ix);
    for (int r = 0; r < rounds; ++r) {
        uint64_t accum = 0;
        for (uint32_t c = 0; c < ix->configurations[r]; ++c) {
            const uint64_t next = accum + ix->configuration_to_offset[r][c];
            ix->configuration_to_offset[r][c] = accum;
            accum = next;
        }
        ix->round_size[r] = accum;
    }
    uint32_t count[SUITS] = {0};
    enum_permutations(ix, 0, cards_per_round[0], 0, count, count_permutation, ix);
    for (int r = 0; r < rounds; ++r) {
        ix->permutation_to_configuration[r] = calloc(ix->permutations[r], sizeof(uint32_t));
        ix->permutation_to_pi[r] = calloc(ix->permutations[r], sizeof(uint32_t));
        if (!ix->permutation_to_configuration[r] || !ix->permutation_to_pi[r]) {
            orc_hand_indexer_free(ix);
            return -1;
        }
    }
    memset(count, 0, sizeof(count));
    enum_permutations(ix, 0, cards_per_round[0], 0, count, tabulate_permutation, ix);
    return 0;
}

void orc_hand_indexer_free(orc_hand_indexer *ix) {
    if (!ix) return;
    for (int r = 0; r < MAX_ROUNDS; ++r) {
        free(ix->permutation_to_configuration[r]);
        free(ix->permutation_to_pi[r]);
        free(ix->configuration_to_equal[r]);
        free(ix->configuration_to_offset[r]);
        free(ix->configuration[r]);
        free(ix->configuration_to_suit_size[r]);
    }
    memset(ix, 0, sizeof(*ix));
}

uint64_t orc_hand_indexer_size(const orc_hand_indexer *ix, int round) {
    return ix && round >= 0 && round < ix->rounds ? ix->round_size[round] : 0;
}

#define SORT2(a, b)                      \
    do {                                 \
        if (sidx[a] > sidx[b]) {         \
            const uint64_t t_ = sidx[a]; \
            sidx[a] = sidx[b];           \
            sidx[b] = t_;                \
        }                                \
    } while (0)

uint64_t orc_hand_index_round(const orc_hand_indexer *ix, int upto, const uint8_t *cards) {
    uint64_t suit_index[SUITS] = {0}, suit_mult[SUITS] = {1, 1, 1, 1};
    uint32_t used_ranks[SUITS] = {0};
    uint32_t perm_index = 0, perm_mult = 1;
    uint64_t index = 0;
    for (int round = 0; round <= upto; ++round) {
        uint32_t ranks[SUITS] = {0}, shifted[SUITS] = {0};
        const uint8_t *c = cards + ix->round_start[round];
        for (uint32_t i = 0; i < ix->cards_per_round[round]; ++i) {
            const uint32_t rank = c[i] >> 2, suit = c[i] & 3, bit = 1u << rank;
            ranks[suit] |= bit;
            shifted[suit] |= bit >> __builtin_popcount((bit - 1) & used_ranks[suit]);
        }
        for (int s = 0; s < SUITS; ++s) {
            const uint32_t used_size = (uint32_t)__builtin_popcount(used_ranks[s]), this_size = (uint32_t)__builtin_popcount(ranks[s]);
            suit_index[s] += suit_mult[s] * rank_set_to_index[shifted[s]];
            suit_mult[s] *= ncr_ranks[RANKS - used_size][this_size];
            used_ranks[s] |= ranks[s];
        }
        uint32_t remaining = ix->cards_per_round[round];
        for (int s = 0; s < SUITS - 1; ++s) {
            const uint32_t this_size = (uint32_t)__builtin_popcount(ranks[s]);
            perm_index += perm_mult * this_size;
            perm_mult *= remaining + 1;
            remaining -= this_size;
        }
        if (round < upto) continue;
        const uint32_t conf = ix->permutation_to_configuration[round][perm_index];
        const uint8_t *pi = suit_permutations[ix->permutation_to_pi[round][perm_index]];
        const uint32_t equal = ix->configuration_to_equal[round][conf];
        uint64_t sidx[SUITS], smul[SUITS];
        for (int s = 0; s < SUITS; ++s) {
            sidx[s] = suit_index[pi[s]];
            smul[s] = suit_mult[pi[s]];
        }
        index = ix->configuration_to_offset[round][conf];
        uint64_t multiplier = 1;
        for (int i = 0; i < SUITS;) {
            uint64_t part, size;
            const int e1 = i + 1 < SUITS && (equal >> i & 1);          /* suit i+1 has the same word as suit i */
            const int e2 = e1 && i + 2 < SUITS && (equal >> (i + 1) & 1);
            const int e3 = e2 && i + 3 < SUITS && (equal >> (i + 2) & 1);
            if (e3) {
                SORT2(i, i + 1); SORT2(i + 2, i + 3); SORT2(i, i + 2); SORT2(i + 1, i + 3); SORT2(i + 1, i + 2);
                part = sidx[i] + ncr_groups(sidx[i + 1] + 1, 2) + ncr_groups(sidx[i + 2] + 2, 3) + ncr_groups(sidx[i + 3] + 3, 4);
                size = ncr_groups(smul[i] + 3, 4);
                i += 4;
            } else if (e2) {
                SORT2(i, i + 1); SORT2(i, i + 2); SORT2(i + 1, i + 2);
                part = sidx[i] + ncr_groups(sidx[i + 1] + 1, 2) + ncr_groups(sidx[i + 2] + 2, 3);
                size = ncr_groups(smul[i] + 2, 3);
                i += 3;
            } else if (e1) {
                SORT2(i, i + 1);
                part = sidx[i] + ncr_groups(sidx[i + 1] + 1, 2);
                size = ncr_groups(smul[i] + 1, 2);
                i += 2;
            } else {
                part = sidx[i];
                size = smul[i];
                i += 1;
            }
            index += multiplier * part;
            multiplier *= size;
        }
    }
    return index;
}

uint64_t orc_hand_index_last(const orc_hand_indexer *ix, const uint8_t *cards) { return orc_hand_index_round(ix, ix->rounds - 1, cards); }

int orc_hand_unindex(const orc_hand_indexer *ix, int round, uint64_t index, uint8_t *cards) {
    if (!ix || round < 0 || round >= ix->rounds || index >= ix->round_size[round]) return -1;
    uint32_t lo = 0, hi = ix->configurations[round], conf = 0;
    while (lo < hi) {   /* last configuration whose offset is <= index */
        const uint32_t mid = (lo + hi) / 2;
        if (ix->configuration_to_offset[round][mid] <= index) {
            conf = mid;
            lo = mid + 1;
        } else hi = mid;
    }
    index -= ix->configuration_to_offset[round][conf];
    const uint32_t *words = ix->configuration[round][conf];
    uint64_t suit_index[SUITS];
    for (int i = 0; i < SUITS;) {
        int j = i + 1;
        while (j < SUITS && words[j] == words[i]) ++j;
        const uint64_t suit_size = ix->configuration_to_suit_size[round][conf][i];
        const uint64_t group_size = ncr_groups(suit_size + (uint64_t)(j - i) - 1, (unsigned)(j - i));
        uint64_t group_index = index % group_size;
        index /= group_size;
        for (; i < j - 1; ++i) {   /* peel the largest member: max m with C(m + k - 1, k) <= group_index */
            const unsigned k = (unsigned)(j - i);
            uint64_t l = 0, h = suit_size, m = 0;
            while (l < h) {
                const uint64_t mid = (l + h) / 2;
                if (ncr_groups(mid + k - 1, k) <= group_index) {
                    m = mid;
                    l = mid + 1;
                } else h = mid;
            }
            suit_index[i] = m;
            group_index -= ncr_groups(m + k - 1, k);
        }
        suit_index[i] = group_index;
        ++i;
    }
    uint8_t location[MAX_ROUNDS];
    memcpy(location, ix->round_start, sizeof(location));
    for (int s = 0; s < SUITS; ++s) {
        uint32_t used = 0, m = 0;
        for (int r = 0; r <= round; ++r) {
            const uint32_t n = nibble(words[s], r);
            const uint32_t round_size = ncr_ranks[RANKS - m][n];
            m += n;
            const uint32_t round_idx = (uint32_t)(suit_index[s] % round_size);
            suit_index[s] /= round_size;
            uint32_t shifted = index_to_rank_set[n][round_idx], rank_set = 0;
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t low = shifted & (0u - shifted);
                shifted ^= low;
                const uint32_t rank = nth_unset[used][__builtin_ctz(low)];
                rank_set |= 1u << rank;
                cards[location[r]++] = (uint8_t)(rank << 2 | (uint32_t)s);
            }
            used |= rank_set;
        }
    }
    return 0;
}

/* ---- brute-force canonical form (independent of everything above) --------------------------------------- */
static int cmp_u8(const void *a, const void *b) { return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b; }

void orc_hand_canon(int rounds, const uint8_t *cards_per_round, const uint8_t *cards, uint8_t *canon_out) {
    static const uint8_t perms[24][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {0, 3, 2, 1}, {1, 0, 2, 3}, {1, 0, 3, 2},
                                         {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 0, 2}, {1, 3, 2, 0}, {2, 0, 1, 3}, {2, 0, 3, 1}, {2, 1, 0, 3}, {2, 1, 3, 0},
                                         {2, 3, 0, 1}, {2, 3, 1, 0}, {3, 0, 1, 2}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 1, 2, 0}, {3, 2, 0, 1}, {3, 2, 1, 0}};
    int total = 0;
    for (int r = 0; r < rounds; ++r) total += cards_per_round[r];
    uint8_t best[64], cur[64];
    for (int p = 0; p < 24; ++p) {
        for (int i = 0; i < total; ++i) cur[i] = (uint8_t)((cards[i] & ~3) | perms[p][cards[i] & 3]);
        int at = 0;
        for (int r = 0; r < rounds; ++r) {
            qsort(cur + at, cards_per_round[r], 1, cmp_u8);
            at += cards_per_round[r];
        }
        if (p == 0 || memcmp(cur, best, (size_t)total) < 0) memcpy(best, cur, (size_t)total);
    }
    memcpy(canon_out, best, (size_t)total);
}

/* ---- generate_maps (card_abstraction.rs:75-184), one player, deterministic order ------------------------- */
typedef struct {
    uint64_t *slots;   /* open addressing, key + 1 (0 = empty) */
    size_t cap, n;
} keyset;

static int keyset_add(keyset *ks, uint64_t key) {   /* 1 if new */
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (ks->cap - 1);
    while (ks->slots[h]) {
        if (ks->slots[h] == key + 1) return 0;
        h = (h + 1) & (ks->cap - 1);
    }
    ks->slots[h] = key + 1;
    ++ks->n;
    return 1;
}

size_t orc_generate_map(const orc_hand_indexer *ix, const uint8_t *hands, size_t n_hands, uint64_t initial_board_mask,
                        int n_round_board_cards, const uint32_t *cluster_arr, uint64_t *keys_out, size_t keys_cap) {
    const int n_board = __builtin_popcountll(initial_board_mask);
    const int cards_left = n_round_board_cards - n_board;   /* card_abstraction.rs:102-106 */
    if (!ix || cards_left < 0 || cards_left > 2) return (size_t)-1;   /* :171 panics */
    uint8_t cards[8] = {0};
    uint64_t bm = initial_board_mask;
    for (int i = 0; i < n_board; ++i) {   /* :94-98 ascending card order */
        cards[i + 2] = (uint8_t)__builtin_ctzll(bm);
        bm &= bm - 1;
    }
    const int next = n_board + 2;
    keyset ks;
    ks.cap = 1;
    while (ks.cap < n_hands * (cards_left == 2 ? 2400 : cards_left == 1 ? 96 : 2) + 16) ks.cap <<= 1;
    ks.n = 0;
    ks.slots = calloc(ks.cap, sizeof(uint64_t));
    if (!ks.slots) return (size_t)-1;
    size_t n_keys = 0;
#define EMIT()                                                               \
    do {                                                                     \
        uint64_t b_ = orc_hand_index_last(ix, cards);                        \
        if (cluster_arr) b_ = cluster_arr[b_];                               \
        if (keyset_add(&ks, b_)) {                                           \
            if (keys_out && n_keys < keys_cap) keys_out[n_keys] = b_;        \
            ++n_keys;                                                        \
        }                                                                    \
    } while (0)
    for (size_t h = 0; h < n_hands; ++h) {
        cards[0] = hands[2 * h];
        cards[1] = hands[2 * h + 1];
        const uint64_t used = 1ull << cards[0] | 1ull << cards[1] | initial_board_mask;
        if (cards_left == 0) EMIT();
        else if (cards_left == 1) {
            for (int i = 0; i < 52; ++i) {
                if (used >> i & 1) continue;
                cards[next] = (uint8_t)i;
                EMIT();
            }
        } else {
            for (int i = 0; i < 52; ++i) {
                if (used >> i & 1) continue;
                cards[next] = (uint8_t)i;
                for (int j = 0; j < i; ++j) {
                    if (used >> j & 1) continue;
                    cards[next + 1] = (uint8_t)j;
                    EMIT();
                }
            }
        }
    }
#undef EMIT
    free(ks.slots);
    return n_keys;
}

/* ---- generate_hand (cfr.rs:100-143) over counter-hash bits ------------------------------------------------ */
static uint64_t mix64(uint64_t x) {   /* splitmix64 finaliser */
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t orc_deal_bits(uint64_t seed, uint64_t deal, uint32_t k) {
    return mix64(mix64(seed ^ (deal + 1) * 0xD1B54A32D192ED03ull) + (uint64_t)k * 0x632BE59BD9B4E019ull);
}

/* train()'s per-deal prune decision (cfr.rs:213-221): `let q: f32 = rng.gen()` after generate_hand -- rand 0.7 Standard: the 24 high bits of a
 * u32 times 2^-24 -- drawn as counter 4096 of the deal's hash (generate_hand stops before it); prune = t > PRUNE_THRESHOLD && q > 0.05 with
 * t = the deal's global number */
int orc_deal_prune(uint64_t seed, uint64_t deal, uint64_t prune_threshold) {
    const float q = (float)((uint32_t)orc_deal_bits(seed, deal, 4096) >> 8) * (1.0f / 16777216.0f);
    return deal > prune_threshold && q > 0.05f;
}

/* rand 0.7 UniformInt<u8>::sample for Uniform::from(0..52) (cfr.rs:106,:116): u32 draws, widening multiply, reject the low half above
 * `zone`; returns the card or -1 when the draw is rejected */
static int uniform52(uint32_t draw) {
    const uint32_t range = 52;
    const uint32_t ints_to_reject = (0xffffffffu - range + 1) % range;
    const uint32_t zone = 0xffffffffu - ints_to_reject;
    const uint64_t wide = (uint64_t)draw * range;
    return (uint32_t)wide <= zone ? (int)(wide >> 32) : -1;
}

/* rand 0.7 gen_range(0, len) behind slice::choose (cfr.rs:129): u64 draws, zone = (range << lz) - 1; -1 when rejected */
static int64_t choose_index(uint64_t draw, uint64_t len) {
    const uint64_t zone = (len << __builtin_clzll(len)) - 1;
    const unsigned __int128 wide = (unsigned __int128)draw * len;
    return (uint64_t)wide <= zone ? (int64_t)(uint64_t)(wide >> 64) : -1;
}

int orc_generate_hand(uint64_t seed, uint64_t deal, uint64_t board_mask, const uint8_t *hands0, uint32_t n0, const uint8_t *hands1, uint32_t n1,
                      uint8_t *cards9) {
    uint64_t used = board_mask;
    int i = 0;
    for (uint64_t m = board_mask; m; m &= m - 1) cards9[i++] = (uint8_t)__builtin_ctzll(m);   /* :110-113 */
    uint32_t k = 0;
    while (i < 5) {   /* :115-122 */
        if (k >= 4096) return -1;
        const int c = uniform52((uint32_t)orc_deal_bits(seed, deal, k++));
        if (c < 0 || (used >> c & 1)) continue;
        cards9[i++] = (uint8_t)c;
        used |= 1ull << c;
    }
    for (int p = 0; p < 2; ++p) {   /* :126-137 */
        const uint8_t *hands = p ? hands1 : hands0;
        const uint64_t n = p ? n1 : n0;
        for (;;) {
            if (k >= 4096) return -1;
            const int64_t idx = choose_index(orc_deal_bits(seed, deal, k++), n);
            if (idx < 0) continue;
            const uint64_t combo = 1ull << hands[2 * idx] | 1ull << hands[2 * idx + 1];
            if (combo & used) continue;
            used |= combo;
            cards9[5 + 2 * p] = hands[2 * idx];
            cards9[6 + 2 * p] = hands[2 * idx + 1];
            break;
        }
    }
    return 0;
}
