// rs_cards.hip -- the card side of get-infoset addressing on the device:
//   * rs_hand_indexer: rust_poker::hand_indexer_s (init / size / get_index / get_hand) restated in rs_hand_index.hpp;
//   * rs_card_abs: one round's ICardAbstraction (ISOMORPHIC / EMD / OCHS, card_abstraction.rs:31-298): generate_maps on the
//     host at init, get_cluster for whole deal batches on the GPU (index -> bucket file gather -> dense id);
//   * rs_deals_sample: generate_hand (cfr.rs:100-143) for a batch of deals.
// Integer work only; one thread per hand, card rows are SoA ([row][pitch] u8) so loads and stores coalesce.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <new>

#include "rs_hand_index.hpp"
#include "rs_internal.hpp"
#include "rs_eval.hpp"

using namespace rs;

#define RS_HIP(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

// ---- host object ---------------------------------------------------------------------------------------------------------
struct rs_hand_indexer {
    int rounds = 0;
    uint8_t cards_per_round[kHiMaxRounds] = {0};
    uint8_t round_start[kHiMaxRounds] = {0};
    int total_cards[kHiMaxRounds] = {0};            // cards of rounds 0..r
    uint64_t round_size[kHiMaxRounds] = {0};
    std::vector<uint16_t> rank_rank;                // [8192]
    std::vector<uint16_t> unrank[kHiRanks + 1];     // [set size][rank] -> rank set
    struct Config {
        std::array<uint32_t, kHiSuits> words;       // descending count words
        uint64_t offset, classes;
    };
    std::vector<Config> configs[kHiMaxRounds];      // ascending by words
    std::vector<uint64_t> perm_offset[kHiMaxRounds];
    struct Dev {
        int device;
        void *blob;
        HandIndexView view;
    };
    std::vector<Dev> devs;                          // one mirror per device that used it
    std::mutex mu;

    HandIndexView host_view() const {
        HandIndexView v{};
        v.rounds = rounds;
        std::memcpy(v.cards_per_round, cards_per_round, sizeof(v.cards_per_round));
        std::memcpy(v.round_start, round_start, sizeof(v.round_start));
        v.rank_rank = rank_rank.data();
        for (int r = 0; r < rounds; ++r) v.perm_offset[r] = perm_offset[r].data();
        return v;
    }
};

namespace {

uint32_t nibble(uint32_t word, int round) { return word >> (4 * (kHiMaxRounds - 1 - round)) & 0xfu; }

// size of one suit's class: product over rounds of C(13 - used, n)
uint64_t suit_classes(uint32_t word, int upto) {
    uint64_t size = 1;
    uint32_t left = kHiRanks;
    for (int r = 0; r <= upto; ++r) {
        const uint32_t n = nibble(word, r);
        size *= hi_choose13(left, n);
        left -= n;
    }
    return size;
}

uint32_t count_key(const rs_hand_indexer &ix, int upto, const uint32_t *words) {
    uint32_t key = 0, mult = 1;
    for (int r = 0; r <= upto; ++r) {
        uint32_t remaining = ix.cards_per_round[r];
        for (int s = 0; s < kHiSuits - 1; ++s) {
            const uint32_t n = nibble(words[s], r);
            key += mult * n;
            mult *= remaining + 1;
            remaining -= n;
        }
    }
    return key;
}

// every ordered assignment of per-suit counts for rounds 0..upto (a suit never holds more than 13 cards)
template <class F>
void for_each_count_vector(const rs_hand_indexer &ix, int upto, int round, int suit, uint32_t remaining, uint32_t *words, uint32_t *held, F &&f) {
    if (suit == kHiSuits) {
        if (round == upto) f(words);
        else for_each_count_vector(ix, upto, round + 1, 0, ix.cards_per_round[round + 1], words, held, f);
        return;
    }
    const uint32_t lo = suit == kHiSuits - 1 ? remaining : 0;
    for (uint32_t n = lo; n <= remaining; ++n) {
        if (held[suit] + n > uint32_t(kHiRanks)) break;
        words[suit] |= n << (4 * (kHiMaxRounds - 1 - round));
        held[suit] += n;
        for_each_count_vector(ix, upto, round, suit + 1, remaining - n, words, held, f);
        held[suit] -= n;
        words[suit] &= ~(0xfu << (4 * (kHiMaxRounds - 1 - round)));
    }
}

void build_round(rs_hand_indexer &ix, int upto) {
    // configurations = the distinct descending-sorted count vectors; std::map keeps them ascending
    std::map<std::array<uint32_t, kHiSuits>, uint64_t> classes;
    uint32_t words[kHiSuits] = {0, 0, 0, 0}, held[kHiSuits] = {0, 0, 0, 0};
    uint32_t max_key = 0;
    auto sorted_of = [](const uint32_t *w) {
        std::array<uint32_t, kHiSuits> s = {w[0], w[1], w[2], w[3]};
        std::sort(s.begin(), s.end(), [](uint32_t a, uint32_t b) { return a > b; });
        return s;
    };
    for_each_count_vector(ix, upto, 0, 0, ix.cards_per_round[0], words, held, [&](const uint32_t *w) {
        max_key = std::max(max_key, count_key(ix, upto, w));
        const auto s = sorted_of(w);
        if (classes.count(s)) return;
        uint64_t n = 1;
        for (int i = 0; i < kHiSuits;) {
            int j = i + 1;
            while (j < kHiSuits && s[j] == s[i]) ++j;
            n *= hi_choose(suit_classes(s[i], upto) + uint64_t(j - i) - 1, j - i);   // multisets of j-i interchangeable suits
            i = j;
        }
        classes.emplace(s, n);
    });
    uint64_t accum = 0;
    std::map<std::array<uint32_t, kHiSuits>, uint64_t> offset;
    for (const auto &kv : classes) {
        ix.configs[upto].push_back({kv.first, accum, kv.second});
        offset.emplace(kv.first, accum);
        accum += kv.second;
    }
    ix.round_size[upto] = accum;
    ix.perm_offset[upto].assign(size_t(max_key) + 1, 0);
    for_each_count_vector(ix, upto, 0, 0, ix.cards_per_round[0], words, held,
                          [&](const uint32_t *w) { ix.perm_offset[upto][count_key(ix, upto, w)] = offset.at(sorted_of(w)); });
}

int validate_cards(const rs_hand_indexer *ix, int round, const uint8_t *cards, const char *who) {
    uint64_t seen = 0;
    for (int i = 0; i < ix->total_cards[round]; ++i) {
        if (cards[i] >= 52) return fail(RS_ERR_INVALID, std::string(who) + ": card " + std::to_string(cards[i]) + " is not in 0..51");
        if (seen >> cards[i] & 1) return fail(RS_ERR_INVALID, std::string(who) + ": card " + std::to_string(cards[i]) + " appears twice in one hand");
        seen |= 1ull << cards[i];
    }
    return RS_OK;
}

// device mirror of the tables (created on first use per device)
int device_view(rs_hand_indexer *ix, int device, hipStream_t stream, HandIndexView *out) {
    std::lock_guard<std::mutex> lock(ix->mu);
    for (const auto &d : ix->devs)
        if (d.device == device) {
            *out = d.view;
            return RS_OK;
        }
    size_t bytes = round_up(ix->rank_rank.size() * sizeof(uint16_t), 256);
    size_t off[kHiMaxRounds];
    for (int r = 0; r < ix->rounds; ++r) {
        off[r] = bytes;
        bytes += round_up(ix->perm_offset[r].size() * sizeof(uint64_t), 256);
    }
    std::vector<char> host(bytes, 0);
    std::memcpy(host.data(), ix->rank_rank.data(), ix->rank_rank.size() * sizeof(uint16_t));
    for (int r = 0; r < ix->rounds; ++r) std::memcpy(host.data() + off[r], ix->perm_offset[r].data(), ix->perm_offset[r].size() * sizeof(uint64_t));
    void *blob = nullptr;
    RS_HIP(hipSetDevice(device), "hipSetDevice");
    RS_HIP(hipMalloc(&blob, bytes), "rs_hand_indexer: device tables");
    hipError_t e = hipMemcpyAsync(blob, host.data(), bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // `host` dies at return
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "rs_hand_indexer: table upload");
    }
    HandIndexView v = ix->host_view();
    v.rank_rank = reinterpret_cast<const uint16_t *>(blob);
    for (int r = 0; r < kHiMaxRounds; ++r) v.perm_offset[r] = r < ix->rounds ? reinterpret_cast<const uint64_t *>((char *)blob + off[r]) : nullptr;
    ix->devs.push_back({device, blob, v});
    *out = v;
    return RS_OK;
}

}  // namespace

// ---- kernels ---------------------------------------------------------------------------------------------------------------
namespace rs {

struct CardRows {
    uint8_t row[kHiMaxCards];   // row of the SoA card matrix holding card i of the hand
    int32_t n_cards;
};

// one get_cluster (card_abstraction.rs:204-209 / :245-251 / :287-293) per deal, for one (round, player)
struct ClusterJob {
    HandIndexView view;
    CardRows rows;
    int32_t upto;                  // round of the indexer whose index is taken (its last)
    const uint32_t *cluster_arr;   // bucket file (EMD / OCHS) or nullptr (ISOMORPHIC: bucket = index)
    uint64_t arr_len;
    const DenseSlot *slots;        // cluster_map[player]
    uint64_t mask;
    uint32_t *out;                 // [pitch] dense cluster ids
    const uint32_t *lut;           // [52 * 52] get_cluster by hole cards when the abstraction's whole board is the initial board, else nullptr
    // One or two board cards beyond the initial board (a flop-start game's turn and river): get_cluster is a function of the hole cards and those cards -- tabulated too
    // (k_cluster_xlut, once per device): xlut[pair id of the hole cards * xw + (the one card | pair id of the two cards)], pair ids from pid[52 * 52] (0xffff: a card of the
    // initial board, or twice the same card).  A deal whose first n_fixed board cards are NOT the initial board takes the index path below.  nullptr: no such table.
    const uint32_t *xlut;
    const uint16_t *pid;
    uint64_t fixed_mask;
    uint32_t xw;
    int32_t n_fixed, cards_left;
};
constexpr uint32_t kDenseBeyond = kDenseMissing - 1u;   // table entries: the canonical index lies beyond the bucket file

__global__ __launch_bounds__(kBlock) void k_hand_index(HandIndexView v, int upto, CardRows rows, const uint8_t *__restrict__ cards, uint32_t n,
                                                       uint32_t pitch, uint64_t *__restrict__ out) {
    for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {
        uint8_t c[kHiMaxCards];
        for (int i = 0; i < rows.n_cards; ++i) c[i] = cards[(size_t)rows.row[i] * pitch + l];
        out[l] = hand_index(v, upto, c);
    }
}

// error word: bit 0 = a bucket without a dense id (Rust: unwrap on None, card_abstraction.rs:208), bit 1 = canonical index beyond the
// bucket file (Rust: index out of bounds), bit 2 = the deal sampler gave up
struct ClusterJobs {
    ClusterJob j[2];   // one per player
};
__global__ __launch_bounds__(kBlock) void k_deal_clusters(const ClusterJobs jobs, const uint8_t *__restrict__ cards, uint32_t n, uint32_t pitch,
                                                          uint32_t *__restrict__ err) {
    // The descriptors are kernel ARGUMENTS (by value): the kernarg segment is ordinary memory, so jobs.j[blockIdx.y].rows.row[i] is a scalar
    // load at a uniform address.  Two earlier forms were wrong: copying a descriptor from global memory into a local put the copy in scratch
    // (168 B per lane, 700 MB of scratch writes per 4 M deals, profiles/r01e), and a descriptor slot in device memory shared by every user of
    // the abstraction was rewritten under another stream's launch.
    const ClusterJob *job = &jobs.j[blockIdx.y];
    const int n_cards = job->rows.n_cards;
    const uint32_t *__restrict__ arr = job->cluster_arr;
    const uint64_t arr_len = job->arr_len;
    const DenseSlot *__restrict__ slots = job->slots;
    const uint64_t mask = job->mask;
    uint32_t *__restrict__ dst = job->out;
    const uint32_t *__restrict__ lut = job->lut;
    if (lut) {   // the board of this round is fixed: get_cluster is a function of the two hole cards, tabulated at init
        const uint32_t r0 = job->rows.row[0], r1 = job->rows.row[1];
        // four deals per thread (one 4-byte load per card row, four table reads, one 16-byte store): a quarter of the waves for the same bytes -- the kernel is latency-bound
        if ((pitch & 3u) == 0 && ((uintptr_t)dst & 15u) == 0 && ((uintptr_t)cards & 3u) == 0) {
            const uint32_t nv = n / 4;
            for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < nv; v += gridDim.x * kBlock) {
                const uint32_t a4 = *reinterpret_cast<const uint32_t *>(cards + (size_t)r0 * pitch + 4 * (size_t)v);
                const uint32_t b4 = *reinterpret_cast<const uint32_t *>(cards + (size_t)r1 * pitch + 4 * (size_t)v);
                uint32_t d[4];
                bool bad = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t a = (a4 >> (8 * i)) & 0xffu, b = (b4 >> (8 * i)) & 0xffu;
                    d[i] = a < 52u && b < 52u ? lut[a * 52u + b] : kDenseMissing;
                    if (d[i] == kDenseMissing) {
                        bad = true;
                        d[i] = 0;
                    }
                }
                if (bad) atomicOr(err, 1u);
                *reinterpret_cast<uint4 *>(dst + 4 * (size_t)v) = make_uint4(d[0], d[1], d[2], d[3]);
            }
            for (uint32_t l = nv * 4 + blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {   // the last n % 4 deals
                const uint32_t a = cards[(size_t)r0 * pitch + l], b2 = cards[(size_t)r1 * pitch + l];
                uint32_t dense = a < 52u && b2 < 52u ? lut[a * 52u + b2] : kDenseMissing;
                if (dense == kDenseMissing) {
                    atomicOr(err, 1u);
                    dense = 0;
                }
                dst[l] = dense;
            }
            return;
        }
        for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {
            const uint32_t a = cards[(size_t)r0 * pitch + l], b = cards[(size_t)r1 * pitch + l];
            uint32_t dense = a < 52u && b < 52u ? lut[a * 52u + b] : kDenseMissing;
            if (dense == kDenseMissing) {
                atomicOr(err, 1u);
                dense = 0;
            }
            dst[l] = dense;
        }
        return;
    }
    const uint32_t *__restrict__ xlut = job->xlut;
    const uint16_t *__restrict__ pid = job->pid;
    const int n_fixed = job->n_fixed, cards_left = job->cards_left;
    const uint64_t fixed_mask = job->fixed_mask;
    const uint32_t xw = job->xw;
    for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {
        uint8_t c[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) c[i] = i < n_cards ? cards[(size_t)job->rows.row[i] * pitch + l] : (uint8_t)0;
        if (xlut) {   // the deal's board starts with the abstraction's initial board (any order): one table read instead of the index, the bucket file and the hash probe
            uint64_t seen = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < n_fixed) seen |= 1ull << (c[2 + i] & 63u);
            if (seen == fixed_mask && c[0] < 52u && c[1] < 52u) {
                const uint32_t hp = pid[(uint32_t)c[0] * 52u + c[1]];
                const uint32_t f0 = c[2 + n_fixed], f1 = cards_left == 2 ? c[3 + n_fixed] : 0u;
                uint32_t col = 0xffffu;
                if (cards_left == 1) col = f0 < 52u ? f0 : 0xffffu;
                else if (f0 < 52u && f1 < 52u) col = pid[f0 * 52u + f1];
                uint32_t dense = (hp != 0xffffu && col != 0xffffu) ? xlut[(size_t)hp * xw + col] : kDenseMissing;
                if (dense >= kDenseBeyond) {
                    atomicOr(err, dense == kDenseBeyond ? 2u : 1u);
                    dense = 0;
                }
                dst[l] = dense;
                continue;
            }
        }
        uint64_t bucket = hand_index(job->view, job->upto, c);
        if (arr) {
            if (bucket < arr_len) bucket = arr[bucket];
            else {
                atomicOr(err, 2u);
                bucket = ~0ull - 1;
            }
        }
        uint32_t dense = dense_lookup(slots, mask, bucket);
        if (dense == kDenseMissing) {
            atomicOr(err, 1u);
            dense = 0;   // stays inside the table; the error word makes the host refuse the batch
        }
        dst[l] = dense;
    }
}

// The table behind ClusterJob::xlut, filled once per device: entry (hole pair, free card(s)) = what the index path gives for that hand on the initial board.
// pair_cards[id] = (x << 8) | y, x > y: the cards of pair `id` (cards of the initial board have no pairs).  Hands that share a card: kDenseMissing.
__global__ __launch_bounds__(kBlock) void k_cluster_xlut(ClusterJob job, const uint16_t *__restrict__ pair_cards, uint32_t n_pairs, uint32_t *__restrict__ out) {
    const uint64_t total = (uint64_t)n_pairs * job.xw;
    for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (uint64_t)gridDim.x * kBlock) {
        const uint32_t hp = (uint32_t)(e / job.xw), col = (uint32_t)(e % job.xw);
        uint8_t c[7] = {0, 0, 0, 0, 0, 0, 0};
        c[0] = (uint8_t)(pair_cards[hp] >> 8);
        c[1] = (uint8_t)(pair_cards[hp] & 0xffu);
        int at = 2;
        for (uint64_t m = job.fixed_mask; m; m &= m - 1) c[at++] = (uint8_t)__builtin_ctzll(m);
        bool ok = true;
        if (job.cards_left == 1) {
            ok = col < 52u && !((job.fixed_mask >> col) & 1ull);
            c[at++] = (uint8_t)col;
        } else {
            ok = col < n_pairs;
            const uint16_t bc = ok ? pair_cards[col] : (uint16_t)0;
            c[at++] = (uint8_t)(bc >> 8);
            c[at++] = (uint8_t)(bc & 0xffu);
        }
        uint64_t used = 0;
        for (int i = 0; i < at && ok; ++i) {
            ok = !((used >> c[i]) & 1ull);
            used |= 1ull << c[i];
        }
        uint32_t dense = kDenseMissing;
        if (ok) {
            uint64_t bucket = hand_index(job.view, job.upto, c);
            bool beyond = false;
            if (job.cluster_arr) {
                if (bucket < job.arr_len) bucket = job.cluster_arr[bucket];
                else beyond = true;
            }
            dense = beyond ? kDenseBeyond : dense_lookup(job.slots, job.mask, bucket);
        }
        out[e] = dense;
    }
}

// generate_hand (cfr.rs:100-143): the missing board cards uniformly without replacement by rejection (:115-122), then one combo per
// range by rejection against everything dealt so far (:126-137).  Uniform::from(0..52) and slice::choose are rand 0.7's widening-
// multiply samplers (Cargo.toml:23): a u32 draw for the u8 range, a u64 draw for the usize range, accepted when the low half of
// draw*range is <= zone.
constexpr uint32_t kSampleMaxDraws = 4096;   // every loop below ends: the reference would spin forever on an impossible range

__device__ __forceinline__ void sample_deal(uint64_t seed, uint64_t deal, uint64_t board_mask, const uint8_t *__restrict__ hands0, uint32_t n0,
                                            const uint8_t *__restrict__ hands1, uint32_t n1, uint8_t (&out)[9], bool &gave_up) {
    uint64_t used = board_mask;
    int i = 0;
    for (uint64_t m = board_mask; m; m &= m - 1) out[i++] = (uint8_t)__builtin_ctzll(m);   // :110-113 ascending
    uint32_t k = 0;
    const uint32_t zone52 = 0xffffffffu - (0xffffffffu - 52u + 1u) % 52u;
    while (i < 5 && k < kSampleMaxDraws) {
        const uint32_t v = (uint32_t)deal_bits(seed, deal, k++);
        const uint64_t wide = (uint64_t)v * 52u;
        if ((uint32_t)wide > zone52) continue;
        const uint32_t c = (uint32_t)(wide >> 32);
        if (used >> c & 1) continue;
        out[i++] = (uint8_t)c;
        used |= 1ull << c;
    }
    for (int p = 0; p < 2; ++p) {
        const uint8_t *hands = p == 0 ? hands0 : hands1;
        const uint64_t n = p == 0 ? n0 : n1;
        const uint64_t zone = (n << __builtin_clzll(n)) - 1;
        bool done = false;
        while (!done && k < kSampleMaxDraws) {
            const uint64_t v = deal_bits(seed, deal, k++);
            if (v * n > zone) continue;
            const uint64_t idx = __umul64hi(v, n);
            const uint8_t a = hands[2 * idx], b = hands[2 * idx + 1];
            const uint64_t combo = 1ull << a | 1ull << b;
            if (combo & used) continue;
            used |= combo;
            out[5 + 2 * p] = a;
            out[6 + 2 * p] = b;
            done = true;
        }
        if (!done) i = -100;
    }
    gave_up = i != 5;
    if (gave_up) {   // keep everything downstream in bounds: the lowest nine free cards
        uint64_t u = board_mask;
        int j = __builtin_popcountll(board_mask);
        for (uint32_t c = 0; c < 52 && j < 9; ++c)
            if (!(u >> c & 1)) {
                out[j++] = (uint8_t)c;
                u |= 1ull << c;
            }
    }
}

__global__ __launch_bounds__(kBlock) void k_deal_sample(uint64_t seed, uint64_t first_deal, uint64_t board_mask, const uint8_t *__restrict__ hands0,
                                                        uint32_t n0, const uint8_t *__restrict__ hands1, uint32_t n1, uint32_t n, uint32_t pitch,
                                                        uint8_t *__restrict__ cards, uint32_t *__restrict__ err, float *__restrict__ sign,
                                                        uint8_t *__restrict__ flags, uint64_t prune_threshold) {
    for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {
        uint8_t c[9];
        bool gave_up;
        sample_deal(seed, first_deal + l, board_mask, hands0, n0, hands1, n1, c, gave_up);
        if (gave_up) atomicOr(err, 4u);
#pragma unroll
        for (int i = 0; i < 9; ++i) cards[(size_t)i * pitch + l] = c[i];
        // the trainer's fused form: what k_showdown_sign and k_deal_prune_flags would compute from the same cards / deal number
        if (sign) {
            uint32_t m0[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 5; i++) add_card(m0, c[i]);
            uint32_t m1[4] = {m0[0], m0[1], m0[2], m0[3]};
            add_card(m0, c[5]);
            add_card(m0, c[6]);
            add_card(m1, c[7]);
            add_card(m1, c[8]);
            const uint32_t s0 = evaluate_suits(m0), s1 = evaluate_suits(m1);
            sign[l] = s0 == s1 ? 0.0f : (s0 > s1 ? 1.0f : -1.0f);   // cfr.rs:326-333
        }
        if (flags) {
            const uint64_t deal = first_deal + l;
            const float q = (float)((uint32_t)deal_bits(seed, deal, kSampleMaxDraws) >> 8) * 5.9604644775390625e-08f;
            flags[l] = (deal > prune_threshold && q > 0.05f) ? 1 : 0;   // cfr.rs:213-221
        }
    }
}

}  // namespace rs

namespace {
dim3 lane_grid(uint32_t n, uint32_t y = 1) { return dim3(std::max(1u, std::min((n + kBlock - 1) / kBlock, 8192u)), y); }
}  // namespace

// ---- card abstraction of one round ------------------------------------------------------------------------------------------
struct rs_card_abs {
    int round = 0;                              // BettingRound: 0 flop, 1 turn, 2 river
    rs_hand_indexer *ix = nullptr;              // hand_indexer_s::init(2, [2, 3 + round]) (card_abstraction.rs:88-90)
    std::vector<uint32_t> cluster_arr;          // EMD / OCHS bucket file; unused for ISOMORPHIC
    bool has_arr = false;
    std::vector<DenseSlot> slots[2];            // cluster_map[player]
    std::vector<uint64_t> keys[2];              // dense id -> bucket
    std::vector<uint32_t> lut[2];               // [52 * 52] get_cluster by hole cards, when the round's board IS the initial board (else empty)
    uint64_t initial_mask = 0;                  // the board the abstraction was generated from (card_abstraction.rs:94-98)
    int n_fixed = 0, cards_left = 0;            // its cards, and the round's board cards beyond it (0, 1 or 2)
    std::vector<uint16_t> pid, pair_cards;      // cards_left > 0: pair id by two cards [52 * 52] (0xffff: none) and back (ClusterJob::xlut)
    struct Dev {                                // mirror on one GPU, created on first use
        int device;
        uint32_t *cluster_arr;
        DenseSlot *slots[2];
        uint32_t *lut[2];
        uint32_t *err;
        uint32_t *xlut[2];                      // cards_left > 0: get_cluster by (hole pair, free board cards), filled on the device at first use
        uint16_t *pid;
        uint32_t xw;
    };
    std::vector<Dev> devs;
    std::mutex mu;
};

namespace rs {
hipError_t build_xlut(rs_card_abs *a, rs_table *t, rs_card_abs::Dev *d);   // below the kernels
}
namespace {
int abs_device(rs_card_abs *a, rs_table *t, rs_card_abs::Dev *out) {
    std::lock_guard<std::mutex> lock(a->mu);
    for (const auto &d : a->devs)
        if (d.device == t->device) {
            *out = d;
            return RS_OK;
        }
    rs_card_abs::Dev d{};
    d.device = t->device;
    hipError_t e = hipSetDevice(t->device);
    if (e == hipSuccess && a->has_arr && !a->cluster_arr.empty()) {
        e = hipMalloc(reinterpret_cast<void **>(&d.cluster_arr), a->cluster_arr.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d.cluster_arr, a->cluster_arr.data(), a->cluster_arr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream);
    }
    for (int p = 0; e == hipSuccess && p < 2; ++p) {
        e = hipMalloc(reinterpret_cast<void **>(&d.slots[p]), a->slots[p].size() * sizeof(DenseSlot));
        if (e == hipSuccess) e = hipMemcpyAsync(d.slots[p], a->slots[p].data(), a->slots[p].size() * sizeof(DenseSlot), hipMemcpyHostToDevice, t->stream);
    }
    for (int p = 0; e == hipSuccess && p < 2; ++p) {
        if (a->lut[p].empty()) continue;
        e = hipMalloc(reinterpret_cast<void **>(&d.lut[p]), a->lut[p].size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d.lut[p], a->lut[p].data(), a->lut[p].size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d.err), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d.err, 0, sizeof(uint32_t), t->stream);
    if (e == hipSuccess && a->cards_left > 0 && !a->pair_cards.empty()) e = build_xlut(a, t, &d);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) {
        (void)hipFree(d.cluster_arr);
        (void)hipFree(d.slots[0]);
        (void)hipFree(d.slots[1]);
        (void)hipFree(d.lut[0]);
        (void)hipFree(d.lut[1]);
        (void)hipFree(d.xlut[0]);
        (void)hipFree(d.xlut[1]);
        (void)hipFree(d.pid);
        (void)hipFree(d.err);
        return hip_fail(e, "rs_card_abs: device mirror");
    }
    a->devs.push_back(d);
    *out = d;
    return RS_OK;
}
}  // namespace

extern "C" {

// ---- rust_poker::hand_indexer_s ----------------------------------------------------------------------------------------------
int rs_hand_indexer_create(int rounds, const uint8_t *cards_per_round, rs_hand_indexer **out) {
    if (!cards_per_round || !out) return fail(RS_ERR_INVALID, "rs_hand_indexer_create: NULL argument");
    if (rounds < 1 || rounds > kHiMaxRounds) return fail(RS_ERR_INVALID, "rs_hand_indexer_create: rounds must be 1..8");
    rs_hand_indexer *ix = new (std::nothrow) rs_hand_indexer();
    if (!ix) return fail(RS_ERR_OOM, "rs_hand_indexer_create: out of memory");
    int total = 0;
    for (int r = 0; r < rounds; ++r) {
        if (cards_per_round[r] < 1 || cards_per_round[r] > 7) {
            delete ix;
            return fail(RS_ERR_INVALID, "rs_hand_indexer_create: 1..7 cards per round");
        }
        ix->cards_per_round[r] = cards_per_round[r];
        ix->round_start[r] = uint8_t(total);
        total += cards_per_round[r];
        ix->total_cards[r] = total;
    }
    if (total > kHiMaxCards) {
        delete ix;
        return fail(RS_ERR_INVALID, "rs_hand_indexer_create: at most 16 cards per hand");
    }
    ix->rounds = rounds;
    // colexicographic rank of a rank set among the sets of its size: element number j (ascending, from 1) at position b adds C(b, j)
    ix->rank_rank.assign(1u << kHiRanks, 0);
    for (int n = 0; n <= kHiRanks; ++n) ix->unrank[n].assign(hi_choose13(kHiRanks, uint32_t(n)), 0);
    for (uint32_t set = 0; set < (1u << kHiRanks); ++set) {
        uint32_t rank = 0, j = 0;
        for (uint32_t b = 0; b < uint32_t(kHiRanks); ++b)
            if (set >> b & 1) rank += hi_choose13(b, ++j);
        ix->rank_rank[set] = uint16_t(rank);
        ix->unrank[hi_popc(set)][rank] = uint16_t(set);
    }
    for (int r = 0; r < rounds; ++r) build_round(*ix, r);
    *out = ix;
    return RS_OK;
}

void rs_hand_indexer_destroy(rs_hand_indexer *ix) {
    if (!ix) return;
    for (auto &d : ix->devs)
        if (hipSetDevice(d.device) == hipSuccess) {
            (void)hipDeviceSynchronize();   // a launch may still read the tables
            (void)hipFree(d.blob);
        }
    delete ix;
}

uint64_t rs_hand_indexer_size(const rs_hand_indexer *ix, int round) { return ix && round >= 0 && round < ix->rounds ? ix->round_size[round] : 0; }
int rs_hand_indexer_rounds(const rs_hand_indexer *ix) { return ix ? ix->rounds : 0; }
int rs_hand_indexer_n_cards(const rs_hand_indexer *ix, int round) { return ix && round >= 0 && round < ix->rounds ? ix->total_cards[round] : 0; }

// get_index for a batch on the host: cards[n][n_cards(round)], out[n]
int rs_hand_index(const rs_hand_indexer *ix, int round, const uint8_t *cards, size_t n, uint64_t *out) {
    if (!ix || (!cards && n) || (!out && n)) return fail(RS_ERR_INVALID, "rs_hand_index: NULL argument");
    if (round < 0 || round >= ix->rounds) return fail(RS_ERR_OOB, "rs_hand_index: round out of range");
    const HandIndexView v = ix->host_view();
    const int nc = ix->total_cards[round];
    for (size_t i = 0; i < n; ++i) {
        if (int rc = validate_cards(ix, round, cards + i * nc, "rs_hand_index")) return rc;
        out[i] = hand_index(v, round, cards + i * nc);
    }
    return RS_OK;
}

// The self-check an integrator runs before mixing bucket files written by the reference's gen_abstraction (indexed by rust_poker's hand_indexer_s::get_index,
// card_abstraction.rs:204-209, :227-229) with indices computed here: `expect[i]` = what THEIR indexer returned for cards[i].  The ORDER of indices inside the
// isomorphism partition is not pinned by any reference fixture (the crate is not vendored), so this is the place where a disagreement shows before it silently mis-buckets.
// *first_bad = index of the first disagreeing hand, or n when all agree; *got_at_first_bad = this library's index for it.  RS_OK when all agree, RS_ERR_MISMATCH otherwise.
int rs_hand_index_verify(const rs_hand_indexer *ix, int round, const uint8_t *cards, size_t n, const uint64_t *expect, size_t *first_bad, uint64_t *got_at_first_bad) {
    if (!ix || (!cards && n) || (!expect && n)) return fail(RS_ERR_INVALID, "rs_hand_index_verify: NULL argument");
    if (round < 0 || round >= ix->rounds) return fail(RS_ERR_OOB, "rs_hand_index_verify: round out of range");
    const HandIndexView v = ix->host_view();
    const int nc = ix->total_cards[round];
    if (first_bad) *first_bad = n;
    if (got_at_first_bad) *got_at_first_bad = 0;
    for (size_t i = 0; i < n; ++i) {
        if (int rc = validate_cards(ix, round, cards + i * nc, "rs_hand_index_verify")) return rc;
        const uint64_t got = hand_index(v, round, cards + i * nc);
        if (got != expect[i]) {
            if (first_bad) *first_bad = i;
            if (got_at_first_bad) *got_at_first_bad = got;
            std::string hand;
            for (int c = 0; c < nc; ++c) hand += (c ? " " : "") + std::to_string(int(cards[i * nc + c]));
            return fail(RS_ERR_MISMATCH, "rs_hand_index_verify: hand " + std::to_string(i) + " (cards " + hand + "): this library indexes it " + std::to_string(got) +
                                             ", the caller's indexer " + std::to_string(expect[i]) + " -- bucket files indexed by the other side must not be loaded");
        }
    }
    return RS_OK;
}

// get_hand (hand_unindex): a representative hand of every index; cards_out[n][n_cards(round)]
int rs_hand_unindex(const rs_hand_indexer *ix, int round, const uint64_t *indices, size_t n, uint8_t *cards_out) {
    if (!ix || (!indices && n) || (!cards_out && n)) return fail(RS_ERR_INVALID, "rs_hand_unindex: NULL argument");
    if (round < 0 || round >= ix->rounds) return fail(RS_ERR_OOB, "rs_hand_unindex: round out of range");
    const int nc = ix->total_cards[round];
    const auto &configs = ix->configs[round];
    for (size_t h = 0; h < n; ++h) {
        uint64_t index = indices[h];
        if (index >= ix->round_size[round])
            return fail(RS_ERR_OOB, "rs_hand_unindex: index " + std::to_string(index) + " is not below size(round) = " + std::to_string(ix->round_size[round]));
        // the configuration whose [offset, offset + classes) holds the index
        auto it = std::upper_bound(configs.begin(), configs.end(), index, [](uint64_t x, const rs_hand_indexer::Config &c) { return x < c.offset; });
        const rs_hand_indexer::Config &cf = *(it - 1);
        index -= cf.offset;
        uint64_t suit_index[kHiSuits];
        for (int i = 0; i < kHiSuits;) {
            int j = i + 1;
            while (j < kHiSuits && cf.words[j] == cf.words[i]) ++j;
            const uint64_t size = suit_classes(cf.words[i], round);
            const uint64_t group = hi_choose(size + uint64_t(j - i) - 1, j - i);
            uint64_t rest = index % group;
            index /= group;
            for (int k = j - i; k >= 1; --k) {   // undo the multiset rank, largest member first (it goes to the group's first suit): max x with C(x + k - 1, k) <= rest
                uint64_t lo = 0, hi = size, x = 0;
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) / 2;
                    if (hi_choose(mid + uint64_t(k) - 1, k) <= rest) {
                        x = mid;
                        lo = mid + 1;
                    } else hi = mid;
                }
                suit_index[j - k] = x;
                rest -= hi_choose(x + uint64_t(k) - 1, k);
            }
            i = j;
        }
        uint8_t *cards = cards_out + h * nc;
        int at[kHiMaxRounds];
        for (int r = 0; r <= round; ++r) at[r] = ix->round_start[r];
        for (int s = 0; s < kHiSuits; ++s) {
            uint32_t used = 0, left = kHiRanks;
            for (int r = 0; r <= round; ++r) {
                const uint32_t nn = nibble(cf.words[s], r);
                const uint32_t radix = hi_choose13(left, nn);
                left -= nn;
                uint32_t shifted = ix->unrank[nn][suit_index[s] % radix];
                suit_index[s] /= radix;
                uint32_t now = 0;
                for (; shifted; shifted &= shifted - 1) {
                    // the t-th rank this suit has not used yet
                    uint32_t t = uint32_t(__builtin_ctz(shifted)), rank = 0;
                    for (;; ++rank)
                        if (!(used >> rank & 1) && t-- == 0) break;
                    now |= 1u << rank;
                    cards[at[r]++] = uint8_t(rank << 2 | uint32_t(s));
                }
                used |= now;
            }
        }
    }
    return RS_OK;
}

// get_index for a batch on the device: d_cards[n_cards(round)][pitch] u8 (row i = card i of the hand), pitch = round_up(n, 64)
int rs_hand_index_device(rs_table *t, rs_hand_indexer *ix, int round, const uint8_t *d_cards, uint32_t n, uint64_t *d_out) {
    if (!t || !ix || !d_cards || !d_out) return fail(RS_ERR_INVALID, "rs_hand_index_device: NULL argument");
    if (round < 0 || round >= ix->rounds) return fail(RS_ERR_OOB, "rs_hand_index_device: round out of range");
    HandIndexView v;
    if (int rc = device_view(ix, t->device, t->stream, &v)) return rc;
    CardRows rows{};
    rows.n_cards = ix->total_cards[round];
    for (int i = 0; i < rows.n_cards; ++i) rows.row[i] = uint8_t(i);
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    if (n == 0) return RS_OK;
    hipLaunchKernelGGL(k_hand_index, lane_grid(n), dim3(kBlock), 0, t->stream, v, round, rows, d_cards, n, uint32_t(round_up(n, kLanePad)), d_out);
    RS_HIP(hipGetLastError(), "k_hand_index");
    return RS_OK;
}

// ---- ISOMORPHIC / EMD / OCHS ::init (card_abstraction.rs:186-202, :216-243, :258-285) ----------------------------------------
static void dense_insert(std::vector<DenseSlot> &slots, std::vector<uint64_t> &keys, uint64_t bucket) {
    if ((keys.size() + 1) * 2 > slots.size()) {   // grow and rehash
        std::vector<DenseSlot> bigger(std::max<size_t>(64, slots.size() * 2), DenseSlot{0, 0});
        const uint64_t mask = bigger.size() - 1;
        for (size_t id = 0; id < keys.size(); ++id) {
            uint64_t h = hi_mix(keys[id]) & mask;
            while (bigger[h].key1) h = (h + 1) & mask;
            bigger[h] = DenseSlot{keys[id] + 1, id};
        }
        slots.swap(bigger);
    }
    const uint64_t mask = slots.size() - 1;
    uint64_t h = hi_mix(bucket) & mask;
    while (slots[h].key1) {
        if (slots[h].key1 == bucket + 1) return;   // seen before (card_abstraction.rs:118)
        h = (h + 1) & mask;
    }
    slots[h] = DenseSlot{bucket + 1, uint64_t(keys.size())};   // :119-120 next dense id
    keys.push_back(bucket);
}

void rs_card_abs_destroy(rs_card_abs *a) {
    if (!a) return;
    for (auto &d : a->devs)
        if (hipSetDevice(d.device) == hipSuccess) {
            (void)hipDeviceSynchronize();   // a launch may still read the mirror
            (void)hipFree(d.cluster_arr);
            (void)hipFree(d.slots[0]);
            (void)hipFree(d.slots[1]);
            (void)hipFree(d.lut[0]);
            (void)hipFree(d.lut[1]);
            (void)hipFree(d.xlut[0]);
            (void)hipFree(d.xlut[1]);
            (void)hipFree(d.pid);
            (void)hipFree(d.err);
        }
    rs_hand_indexer_destroy(a->ix);
    delete a;
}

int rs_card_abs_create(int betting_round, const uint8_t *hands_p0, size_t n_hands_p0, const uint8_t *hands_p1, size_t n_hands_p1,
                       uint64_t initial_board_mask, const uint32_t *cluster_arr, size_t arr_len, rs_card_abs **out) {
    if (!out || (!hands_p0 && n_hands_p0) || (!hands_p1 && n_hands_p1)) return fail(RS_ERR_INVALID, "rs_card_abs_create: NULL argument");
    if (betting_round < 0 || betting_round > 2) return fail(RS_ERR_INVALID, "rs_card_abs_create: betting_round is 0 (flop), 1 (turn) or 2 (river)");
    if (initial_board_mask >> 52) return fail(RS_ERR_INVALID, "rs_card_abs_create: board mask has bits beyond card 51");
    const int n_board = __builtin_popcountll(initial_board_mask);
    const int cards_left = 3 + betting_round - n_board;                    // card_abstraction.rs:102-106
    if (cards_left < 0 || cards_left > 2) return fail(RS_ERR_INVALID, "invalid number of board cards");   // the panic of :171
    rs_card_abs *a = new (std::nothrow) rs_card_abs();
    if (!a) return fail(RS_ERR_OOM, "rs_card_abs_create: out of memory");
    a->round = betting_round;
    a->initial_mask = initial_board_mask;
    a->n_fixed = n_board;
    a->cards_left = cards_left;
    if (cards_left > 0) {   // pair ids over the cards that are not on the initial board: id(x, y) = id(y, x), x > y in pair_cards
        a->pid.assign(52 * 52, uint16_t(0xffff));
        for (int x = 0; x < 52; ++x)
            for (int y = 0; y < x; ++y) {
                if ((1ull << x | 1ull << y) & initial_board_mask) continue;
                a->pid[size_t(x) * 52 + y] = a->pid[size_t(y) * 52 + x] = uint16_t(a->pair_cards.size());
                a->pair_cards.push_back(uint16_t(x << 8 | y));
            }
    }
    const uint8_t cpr[2] = {2, uint8_t(3 + betting_round)};
    int rc = rs_hand_indexer_create(2, cpr, &a->ix);
    if (rc == RS_OK && cluster_arr) {
        a->has_arr = true;
        a->cluster_arr.assign(cluster_arr, cluster_arr + arr_len);
    }
    // generate_maps (card_abstraction.rs:75-184): hands x missing board cards, in the reference's loop order; dense ids in first-appearance
    // order (the reference's channel-arrival order is not deterministic)
    const HandIndexView v = a->ix ? a->ix->host_view() : HandIndexView{};
    for (int p = 0; rc == RS_OK && p < 2; ++p) {
        const uint8_t *hands = p == 0 ? hands_p0 : hands_p1;
        const size_t n_hands = p == 0 ? n_hands_p0 : n_hands_p1;
        uint8_t cards[7] = {0};
        int i = 2;
        for (uint64_t m = initial_board_mask; m; m &= m - 1) cards[i++] = uint8_t(__builtin_ctzll(m));   // :94-98
        const int next = n_board + 2;
        auto emit = [&]() -> int {
            uint64_t bucket = hand_index(v, 1, cards);
            if (a->has_arr) {
                if (bucket >= a->cluster_arr.size())
                    return fail(RS_ERR_OOB, "index out of bounds: the len is " + std::to_string(a->cluster_arr.size()) + " but the index is " + std::to_string(bucket));
                bucket = a->cluster_arr[bucket];   // index_to_cluster, :20-29
            }
            dense_insert(a->slots[p], a->keys[p], bucket);
            return RS_OK;
        };
        for (size_t h = 0; rc == RS_OK && h < n_hands; ++h) {
            cards[0] = hands[2 * h];
            cards[1] = hands[2 * h + 1];
            if (cards[0] >= 52 || cards[1] >= 52 || cards[0] == cards[1]) rc = fail(RS_ERR_INVALID, "rs_card_abs_create: bad hole cards in a range");
            if (rc != RS_OK) break;
            const uint64_t used = 1ull << cards[0] | 1ull << cards[1] | initial_board_mask;
            if ((1ull << cards[0] | 1ull << cards[1]) & initial_board_mask) {
                rc = fail(RS_ERR_INVALID, "rs_card_abs_create: a range combo uses a board card (remove_invalid_combos first, cfr.rs:163)");
                break;
            }
            if (cards_left == 0) rc = emit();                                   // :131-135
            else if (cards_left == 1) {                                         // :136-149
                for (int c = 0; rc == RS_OK && c < 52; ++c) {
                    if (used >> c & 1) continue;
                    cards[next] = uint8_t(c);
                    rc = emit();
                }
            } else {                                                            // :150-169
                for (int c = 0; rc == RS_OK && c < 52; ++c) {
                    if (used >> c & 1) continue;
                    cards[next] = uint8_t(c);
                    for (int d = 0; rc == RS_OK && d < c; ++d) {
                        if (used >> d & 1) continue;
                        cards[next + 1] = uint8_t(d);
                        rc = emit();
                    }
                }
            }
        }
        if (rc == RS_OK && a->slots[p].empty()) a->slots[p].assign(64, DenseSlot{0, 0});
        if (rc == RS_OK && cards_left == 0) {   // the round's board is the initial board: tabulate get_cluster over the hole cards (memoisation only)
            a->lut[p].assign(52 * 52, kDenseMissing);
            for (int x = 0; x < 52; ++x)
                for (int y = 0; y < 52; ++y) {
                    if (x == y || ((1ull << x | 1ull << y) & initial_board_mask)) continue;
                    cards[0] = uint8_t(x);
                    cards[1] = uint8_t(y);
                    uint64_t bucket = hand_index(v, 1, cards);
                    if (a->has_arr) bucket = bucket < a->cluster_arr.size() ? a->cluster_arr[bucket] : ~0ull - 1;
                    a->lut[p][size_t(x) * 52 + y] = dense_lookup(a->slots[p].data(), a->slots[p].size() - 1, bucket);
                }
        }
    }
    if (rc != RS_OK) {
        rs_card_abs_destroy(a);
        return rc;
    }
    *out = a;
    return RS_OK;
}

int rs_card_abs_round(const rs_card_abs *a) { return a ? a->round : -1; }
size_t rs_card_abs_size(const rs_card_abs *a, int player) { return a && (player == 0 || player == 1) ? a->keys[player].size() : 0; }   // get_size
uint64_t rs_card_abs_index_size(const rs_card_abs *a) { return a ? a->ix->round_size[1] : 0; }
int rs_card_abs_keys(const rs_card_abs *a, int player, uint64_t *keys_out) {
    if (!a || !keys_out || (player != 0 && player != 1)) return fail(RS_ERR_INVALID, "rs_card_abs_keys: bad argument");
    std::memcpy(keys_out, a->keys[player].data(), a->keys[player].size() * sizeof(uint64_t));
    return RS_OK;
}

// get_cluster(&cards, player) on the host for n hands: cards[n][5 + round] = hole cards then the board (cfr.rs:357-365)
int rs_card_abs_get_cluster(const rs_card_abs *a, const uint8_t *cards, size_t n, int player, uint32_t *out) {
    if (!a || (!cards && n) || (!out && n)) return fail(RS_ERR_INVALID, "rs_card_abs_get_cluster: NULL argument");
    if (player != 0 && player != 1) return fail(RS_ERR_OOB, "rs_card_abs_get_cluster: player must be 0 or 1");
    const HandIndexView v = a->ix->host_view();
    const int nc = 5 + a->round;
    for (size_t i = 0; i < n; ++i) {
        if (int rc = validate_cards(a->ix, 1, cards + i * nc, "rs_card_abs_get_cluster")) return rc;
        uint64_t bucket = hand_index(v, 1, cards + i * nc);
        if (a->has_arr) {
            if (bucket >= a->cluster_arr.size())
                return fail(RS_ERR_OOB, "index out of bounds: the len is " + std::to_string(a->cluster_arr.size()) + " but the index is " + std::to_string(bucket));
            bucket = a->cluster_arr[bucket];
        }
        const uint32_t dense = dense_lookup(a->slots[player].data(), a->slots[player].size() - 1, bucket);
        if (dense == kDenseMissing)
            return fail(RS_ERR_OOB, "rs_card_abs_get_cluster: bucket " + std::to_string(bucket) + " has no cluster id (Rust: unwrap on None, card_abstraction.rs:208)");
        out[i] = dense;
    }
    return RS_OK;
}

// get_cluster for every deal of a batch on the device.  d_cards[9][pitch]: rows 0-4 the board, 5-6 player 0's hole cards, 7-8 player 1's
// (the layout of rs_showdown_sign).  d_cluster_p0 / d_cluster_p1 [pitch] (either may be NULL).  Asynchronous; a deal whose bucket has no
// dense id raises the abstraction's error word, reported by rs_card_abs_status.
int rs_card_abs_clusters_device(rs_card_abs *a, rs_table *t, const uint8_t *d_cards, uint32_t n_deals, uint32_t *d_cluster_p0, uint32_t *d_cluster_p1) {
    return rs::card_abs_clusters_on(a, t, t ? t->stream : nullptr, d_cards, n_deals, d_cluster_p0, d_cluster_p1);
}
}  // extern "C" (interrupted: the stream-taking forms below are internal)

namespace rs {
// get_cluster tabulated over (hole pair, the one or two board cards beyond the initial board): 1 176 x 52 entries for a flop-start game's turn, 1 176 x 1 176 (5.5 MB) for its
// river, per player.  The index path (hand index: ~1 000 vector instructions of 64-bit arithmetic, a 4-byte read from a bucket file of up to 492 MB, a hash probe) then runs
// once per table entry instead of once per deal and sweep: 4 M deals, flop start, river: 219 -> ~20 us per call.
hipError_t build_xlut(rs_card_abs *a, rs_table *t, rs_card_abs::Dev *d) {
    HandIndexView v;
    if (device_view(a->ix, t->device, t->stream, &v) != RS_OK) return hipErrorUnknown;
    const uint32_t n_pairs = uint32_t(a->pair_cards.size());
    d->xw = a->cards_left == 1 ? 52u : n_pairs;
    uint16_t *d_pairs = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d->pid), a->pid.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d->pid, a->pid.data(), a->pid.size() * sizeof(uint16_t), hipMemcpyHostToDevice, t->stream);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_pairs), n_pairs * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_pairs, a->pair_cards.data(), n_pairs * sizeof(uint16_t), hipMemcpyHostToDevice, t->stream);
    const size_t entries = size_t(n_pairs) * d->xw;
    for (int p = 0; e == hipSuccess && p < 2; ++p) {
        e = hipMalloc(reinterpret_cast<void **>(&d->xlut[p]), entries * sizeof(uint32_t));
        if (e != hipSuccess) break;
        ClusterJob j;
        std::memset(&j, 0, sizeof(j));
        j.view = v;
        j.upto = 1;
        j.cluster_arr = d->cluster_arr;
        j.arr_len = a->cluster_arr.size();
        j.slots = d->slots[p];
        j.mask = a->slots[p].size() - 1;
        j.fixed_mask = a->initial_mask;
        j.xw = d->xw;
        j.n_fixed = a->n_fixed;
        j.cards_left = a->cards_left;
        hipLaunchKernelGGL(k_cluster_xlut, dim3(uint32_t(std::min<size_t>((entries + kBlock - 1) / kBlock, 8192))), dim3(kBlock), 0, t->stream, j, d_pairs, n_pairs, d->xlut[p]);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);   // d_pairs is freed below
    (void)hipFree(d_pairs);
    return e;
}
// the same on a stream of the caller's choice (the trainer deals the NEXT batch beside the sweeps of the current one)
int card_abs_clusters_on(rs_card_abs *a, rs_table *t, hipStream_t stream, const uint8_t *d_cards, uint32_t n_deals, uint32_t *d_cluster_p0,
                         uint32_t *d_cluster_p1) {
    if (!a || !t || !d_cards) return fail(RS_ERR_INVALID, "rs_card_abs_clusters_device: NULL argument");
    if (!d_cluster_p0 && !d_cluster_p1) return RS_OK;
    HandIndexView v;
    if (int rc = device_view(a->ix, t->device, t->stream, &v)) return rc;
    rs_card_abs::Dev dev;
    if (int rc = abs_device(a, t, &dev)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    ClusterJobs jobs;
    std::memset(&jobs, 0, sizeof(jobs));
    int n_jobs = 0;
    for (int p = 0; p < 2; ++p) {
        uint32_t *dst = p == 0 ? d_cluster_p0 : d_cluster_p1;
        if (!dst) continue;
        ClusterJob &j = jobs.j[n_jobs++];
        j.view = v;
        j.upto = 1;
        j.rows.n_cards = 5 + a->round;
        j.rows.row[0] = uint8_t(5 + 2 * p);   // hand.board[0..2] = the acting player's hole cards (cfr.rs:357-358)
        j.rows.row[1] = uint8_t(6 + 2 * p);
        for (int i = 0; i < 3 + a->round; ++i) j.rows.row[2 + i] = uint8_t(i);
        j.cluster_arr = dev.cluster_arr;
        j.arr_len = a->cluster_arr.size();
        j.slots = dev.slots[p];
        j.lut = dev.lut[p];
        j.mask = a->slots[p].size() - 1;
        j.out = dst;
        j.xlut = dev.xlut[p];
        j.pid = dev.pid;
        j.fixed_mask = a->initial_mask;
        j.xw = dev.xw;
        j.n_fixed = a->n_fixed;
        j.cards_left = a->cards_left;
    }
    if (n_deals == 0) return RS_OK;
    hipLaunchKernelGGL(k_deal_clusters, lane_grid(n_deals, uint32_t(n_jobs)), dim3(kBlock), 0, stream, jobs, d_cards, n_deals,
                       uint32_t(round_up(n_deals, kLanePad)), dev.err);
    RS_HIP(hipGetLastError(), "k_deal_clusters");
    return RS_OK;
}
}  // namespace rs

extern "C" {

// synchronises and reports (then clears) what the device kernels flagged since the last call
int rs_card_abs_status(rs_card_abs *a, rs_table *t) {
    if (!a || !t) return fail(RS_ERR_INVALID, "rs_card_abs_status: NULL argument");
    rs_card_abs::Dev dev;
    if (int rc = abs_device(a, t, &dev)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    uint32_t err = 0;
    RS_HIP(hipMemcpyAsync(&err, dev.err, sizeof(err), hipMemcpyDeviceToHost, t->stream), "rs_card_abs_status");
    RS_HIP(hipStreamSynchronize(t->stream), "rs_card_abs_status");
    if (!err) return RS_OK;
    RS_HIP(hipMemsetAsync(dev.err, 0, sizeof(uint32_t), t->stream), "rs_card_abs_status");
    if (err & 2u) return fail(RS_ERR_OOB, "get_cluster: a canonical hand index lies beyond the bucket file (Rust: index out of bounds, card_abstraction.rs:247)");
    return fail(RS_ERR_OOB, "get_cluster: a deal's bucket has no cluster id for its player (Rust: unwrap on None, card_abstraction.rs:208)");
}

// ---- train()'s per-deal prune flag (cfr.rs:213-221) --------------------------------------------------------------------------
// `let q: f32 = rng.gen()` right after generate_hand, then prune = t.load() > PRUNE_THRESHOLD && q > 0.05 for BOTH traversals of the deal.
// rand 0.7 Standard for f32: 24 high bits of a u32 times 2^-24.  The draw is counter kSampleMaxDraws of the deal's hash (generate_hand never
// gets that far); t = the deal's global number (iterations completed before it in the sequential reading).
int rs_deals_prune_flags(rs_table *t, uint64_t seed, uint64_t first_deal, uint64_t prune_threshold, uint32_t n_deals, uint8_t *d_flags) {
    return rs::deal_prune_flags_on(t, t ? t->stream : nullptr, seed, first_deal, prune_threshold, n_deals, d_flags);
}

// ---- generate_hand for a batch (cfr.rs:100-143) ------------------------------------------------------------------------------
// d_hands_p*: device arrays of (u8, u8) combos = HandRange.hands after remove_invalid_combos (cfr.rs:161-163).  Deal i of the call draws from
// the counter hash (seed, first_deal + i).  d_cards[9][pitch] as above.  d_err (may be NULL): bit 2 raised when a deal found no valid combo.
int rs_deals_sample(rs_table *t, uint64_t seed, uint64_t first_deal, uint64_t board_mask, const uint8_t *d_hands_p0, uint32_t n_hands_p0,
                    const uint8_t *d_hands_p1, uint32_t n_hands_p1, uint32_t n_deals, uint8_t *d_cards, uint32_t *d_err) {
    return rs::deals_sample_on(t, t ? t->stream : nullptr, seed, first_deal, board_mask, d_hands_p0, n_hands_p0, d_hands_p1, n_hands_p1, n_deals, d_cards, d_err,
                               nullptr, nullptr, 0);
}
}  // extern "C"

namespace rs {
__global__ __launch_bounds__(kBlock) void k_deal_prune_flags(uint64_t seed, uint64_t first_deal, uint64_t threshold, uint32_t n, uint32_t pitch,
                                                             uint8_t *__restrict__ flags) {
    for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < pitch; l += gridDim.x * kBlock) {
        const uint64_t deal = first_deal + l;
        const float q = (float)((uint32_t)deal_bits(seed, deal, kSampleMaxDraws) >> 8) * 5.9604644775390625e-08f;   // 2^-24
        flags[l] = (l < n && deal > threshold && q > 0.05f) ? 1 : 0;
    }
}
int deal_prune_flags_on(rs_table *t, hipStream_t stream, uint64_t seed, uint64_t first_deal, uint64_t prune_threshold, uint32_t n_deals, uint8_t *d_flags) {
    if (!t || !d_flags) return fail(RS_ERR_INVALID, "rs_deals_prune_flags: NULL argument");
    if (n_deals == 0) return RS_OK;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    hipLaunchKernelGGL(k_deal_prune_flags, lane_grid(n_deals), dim3(kBlock), 0, stream, seed, first_deal, prune_threshold, n_deals, uint32_t(round_up(n_deals, kLanePad)), d_flags);
    RS_HIP(hipGetLastError(), "k_deal_prune_flags");
    return RS_OK;
}
int deals_sample_on(rs_table *t, hipStream_t stream, uint64_t seed, uint64_t first_deal, uint64_t board_mask, const uint8_t *d_hands_p0, uint32_t n_hands_p0,
                    const uint8_t *d_hands_p1, uint32_t n_hands_p1, uint32_t n_deals, uint8_t *d_cards, uint32_t *d_err, float *d_sign, uint8_t *d_flags,
                    uint64_t prune_threshold) {
    if (!t || !d_hands_p0 || !d_hands_p1 || !d_cards) return fail(RS_ERR_INVALID, "rs_deals_sample: NULL argument");
    if (n_hands_p0 == 0 || n_hands_p1 == 0) return fail(RS_ERR_INVALID, "rs_deals_sample: empty hand range (Rust: choose().unwrap() on None, cfr.rs:129)");
    if (board_mask >> 52) return fail(RS_ERR_INVALID, "rs_deals_sample: board mask has bits beyond card 51");
    const int n_board = __builtin_popcountll(board_mask);
    if (n_board < 3 || n_board > 5) return fail(RS_ERR_INVALID, "invalid board mask");   // options.rs:41
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    if (!d_err) {
        if (!t->d_err_sink) RS_HIP(hipMalloc(reinterpret_cast<void **>(&t->d_err_sink), sizeof(uint32_t)), "rs_deals_sample: error word");
        d_err = t->d_err_sink;   // nobody reads it
    }
    if (n_deals == 0) return RS_OK;
    hipLaunchKernelGGL(k_deal_sample, lane_grid(n_deals), dim3(kBlock), 0, stream, seed, first_deal, board_mask, d_hands_p0, n_hands_p0, d_hands_p1,
                       n_hands_p1, n_deals, uint32_t(round_up(n_deals, kLanePad)), d_cards, d_err, d_sign, d_flags, prune_threshold);
    RS_HIP(hipGetLastError(), "k_deal_sample");
    return RS_OK;
}
}  // namespace rs
