// rs_kmeans.hip -- the abstraction generator's distance sweep on the GPU: Kmeans::predict
// (gen_abstraction/kmeans.rs:173-211) and update_min_dists (kmeans.rs:603-619) with dist_func = emd_1d (emd.rs:53-113) or l2_dist
// (kmeans.rs:622-630).  predict over all canonical hands is the sweep that produces the bucket file EMD::init reads
// (gen_abstraction/main.rs:370-380 -> card_abstraction.rs:269-271).
//
// One thread owns one histogram of the dataset and walks all centers; centers are wave-uniform (scalar loads).
// emd_1d is evaluated EXACTLY as written (f32, no FMA, same operation order) but not as written:
//   * normalisation is hoisted: the datum is divided by its sum once per thread, every center once on the host (same f32 divisions);
//   * after the same-bin pass (emd.rs:72-77) a bin has mass left on at most one side: p' = p - min(p, q) and q' = q - min(p, q) are
//     max(p - q, 0) and max(q - p, 0), so ONE signed residual r = p - q holds both (negation is exact);
//   * the cross-bin loop (emd.rs:96-110) visits offsets -1, +1, -2, +2, .. and, per offset, bins in ascending order; inside one offset
//     every p[j] and every q[k] is touched at most once, so the pairs that move mass are exactly the set bits of
//     P & (Q << d) resp. P & (Q >> d) (P / Q = bins with residual on the p / q side) taken in ascending order.  Each transfer empties
//     one bin, hence at most n_bins transfers per pair instead of 2(u-1) * n_bins probes.
// Residuals are indexed dynamically only during transfers: they live in LDS, one column per thread (bank-conflict free).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rs_internal.hpp"

using namespace rs;

#define RS_HIP(call, what)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

namespace rs {

constexpr int kKmBlock = 256;

// emd_1d of a normalised datum p (registers) against a normalised center q (uniform); r_col = this thread's LDS column, element b at
// r_col[b * kKmBlock]
template <int NB>
__device__ __forceinline__ float emd_pair(const float (&p)[NB], const float *__restrict__ q, int n_bins, float *r_col) {
    float w = 0.0f, cost = 0.0f;
    unsigned long long P = 0, Q = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float qb = q[b], pb = p[b];
        const float mass = pb < qb ? pb : qb;    // min!(p[i], q[i]), emd.rs:10-21
        w += mass;
        const float r = pb - qb;
        r_col[b * kKmBlock] = r;
        P |= (unsigned long long)(r > 0.0f) << b;
        Q |= (unsigned long long)(r < 0.0f) << b;
    }
    float factor = 4.45f * w - 1.5f;             // emd.rs:83-88
    factor = factor < 1.0f ? 1.0f : (factor > 4.0f ? 4.0f : factor);
    const int u = (int)roundf((float)n_bins / factor);
    for (int d = 1; d < u && P && Q; ++d) {      // |b| < u (emd.rs:31,38,47); nothing moves once a side is empty
#pragma unroll
        for (int side = 0; side < 2; ++side) {   // offset -d (k = j - d), then +d (k = j + d): the stable sort by |b| of emd.rs:93
            unsigned long long m = side == 0 ? P & (Q << d) : P & (Q >> d);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int k = side == 0 ? j - d : j + d;
                const float pj = r_col[j * kKmBlock], qk = -r_col[k * kKmBlock];
                const float mass = pj < qk ? pj : qk;
                w += mass;
                cost += mass * (float)d;         // |j as f32 - k as f32|
                const float pn = pj - mass, qn = qk - mass;
                r_col[j * kKmBlock] = pn;
                r_col[k * kKmBlock] = -qn;
                if (pn == 0.0f) P &= ~(1ull << j);
                if (qn == 0.0f) Q &= ~(1ull << k);
            }
        }
    }
    return fabsf(cost + (1.0f - w) * (float)u);  // emd.rs:112
}

template <int NB>
__device__ __forceinline__ float l2_pair(const float (&a)[NB], const float *__restrict__ b) {   // kmeans.rs:622-630
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float d = a[i] - b[i];
        sum += d * d;
    }
    return sqrtf(sum);
}

// loads one histogram (row-major dataset, kmeans.rs Vec<Histogram>), zero-padded to NB; EMD: normalised by its sequential f32 sum
template <int NB, int DIST>
__device__ __forceinline__ bool load_datum(const float *__restrict__ dataset, size_t i, int n_bins, float (&p)[NB]) {
    float sum = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        p[b] = b < n_bins ? dataset[i * (size_t)n_bins + b] : 0.0f;
        if (b < n_bins) sum += p[b];
    }
    if (DIST == RS_DIST_EMD) {
        if (sum == 0.0f) return false;           // emd.rs:59-61: the distance to everything is 0
#pragma unroll
        for (int b = 0; b < NB; ++b) p[b] = b < n_bins ? p[b] / sum : 0.0f;
    }
    return true;
}

// centers: [n_centers][NB] prepared on the host (EMD: normalised, zero-padded); center_zero[c] != 0: its sum was 0 (distance 0)
template <int NB, int DIST>
__global__ __launch_bounds__(kKmBlock) void k_kmeans_predict(const float *__restrict__ dataset, size_t n, int n_bins, const float *__restrict__ centers,
                                                             const unsigned char *__restrict__ center_zero, int n_centers,
                                                             unsigned *__restrict__ clusters, float *__restrict__ min_dist) {
    extern __shared__ float lds[];
    float *r_col = lds + threadIdx.x;
    for (size_t i = (size_t)blockIdx.x * kKmBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kKmBlock) {
        float p[NB];
        const bool live = load_datum<NB, DIST>(dataset, i, n_bins, p);
        int best = 0;
        float best_v = 0.0f;
        if (live) {
            for (int c = 0; c < n_centers; ++c) {   // kmeans.rs:194-202
                const float *q = centers + (size_t)c * NB;
                float v;
                if (DIST == RS_DIST_EMD) v = center_zero[c] ? 0.0f : emd_pair<NB>(p, q, n_bins, r_col);
                else v = l2_pair<NB>(p, q);
                if (c == 0 || v < best_v) {          // first center, then strictly smaller only
                    best_v = v;
                    best = c;
                }
            }
        }
        if (clusters) clusters[i] = (unsigned)best;
        if (min_dist) min_dist[i] = best_v;
    }
}

// update_min_dists (kmeans.rs:603-619): d = dist(x, new_center); d = d * d; min_dists[i] = min
template <int NB, int DIST>
__global__ __launch_bounds__(kKmBlock) void k_update_min_dists(const float *__restrict__ dataset, size_t n, int n_bins, const float *__restrict__ center,
                                                               int center_is_zero, float *__restrict__ min_dists) {
    extern __shared__ float lds[];
    float *r_col = lds + threadIdx.x;
    for (size_t i = (size_t)blockIdx.x * kKmBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kKmBlock) {
        float p[NB];
        const bool live = load_datum<NB, DIST>(dataset, i, n_bins, p);
        float d = 0.0f;
        if (live) d = DIST == RS_DIST_EMD ? (center_is_zero ? 0.0f : emd_pair<NB>(p, center, n_bins, r_col)) : l2_pair<NB>(p, center);
        d = d * d;
        if (d < min_dists[i]) min_dists[i] = d;
    }
}

// ---- the training loops (kmeans.rs:213-601): Hamerly-bounded assignment, center sums in data order, bound shifts -----------------------------------------
// distance of a loaded datum to prepared center c, with emd_1d's zero-sum shortcut (emd.rs:59-61) on either side
template <int NB, int DIST>
__device__ __forceinline__ float datum_dist(bool live, const float (&p)[NB], const float *__restrict__ centers, const unsigned char *__restrict__ center_zero, int c,
                                            int n_bins, float *r_col) {
    if (DIST == RS_DIST_EMD) return (!live || center_zero[c]) ? 0.0f : emd_pair<NB>(p, centers + (size_t)c * NB, n_bins, r_col);
    return l2_pair<NB>(p, centers + (size_t)c * NB);
}

// Kmeans::init_s (kmeans.rs:267-285): s[i] = min(s[i], min over j != i of dist(c_i, c_j)) / 2 -- s is IN/OUT: the reference creates it once with f32::MAX
// (kmeans.rs:518) and init_s only ever lowers it before halving it again.  raw = the centers as Vec<Histogram> (the datum side), centers = prepared (q side)
template <int NB, int DIST>
__global__ __launch_bounds__(kKmBlock) void k_kmeans_init_s(const float *__restrict__ raw, int n_bins, const float *__restrict__ centers,
                                                            const unsigned char *__restrict__ center_zero, int k, float *__restrict__ s) {
    extern __shared__ float lds[];
    float *r_col = lds + threadIdx.x;
    const int i = blockIdx.x * kKmBlock + threadIdx.x;
    if (i >= k) return;
    float p[NB];
    const bool live = load_datum<NB, DIST>(raw, (size_t)i, n_bins, p);
    float si = s[i];
    for (int j = 0; j < k; ++j) {
        if (j == i) continue;
        const float d = datum_dist<NB, DIST>(live, p, centers, center_zero, j, n_bins, r_col);
        if (d < si) si = d;
    }
    s[i] = si / 2.0f;
}

// Kmeans::init_random's scoring loop (kmeans.rs:133-147): distances[i] = sum over j != i, ascending, of dist(c_i, c_j) -- one thread per center of one restart
template <int NB, int DIST>
__global__ __launch_bounds__(kKmBlock) void k_kmeans_row_sums(const float *__restrict__ raw, int n_bins, const float *__restrict__ centers,
                                                              const unsigned char *__restrict__ center_zero, int k, float *__restrict__ out) {
    extern __shared__ float lds[];
    float *r_col = lds + threadIdx.x;
    const int i = blockIdx.x * kKmBlock + threadIdx.x;
    if (i >= k) return;
    float p[NB];
    const bool live = load_datum<NB, DIST>(raw, (size_t)i, n_bins, p);
    float acc = 0.0f;
    for (int j = 0; j < k; ++j) {
        if (j == i) continue;
        acc += datum_dist<NB, DIST>(live, p, centers, center_zero, j, n_bins, r_col);
    }
    out[i] = acc;
}

// Kmeans::reassign_clusters / assignment_with_bounds (kmeans.rs:287-334, :213-265): one thread per datum; order != nullptr: datum i is dataset[order[i]]
// (fit_growbatch's shuffled_data).  bounds[i] = (lower, upper).
template <int NB, int DIST>
__global__ __launch_bounds__(kKmBlock) void k_kmeans_reassign(const float *__restrict__ dataset, const unsigned *__restrict__ order, size_t n, int n_bins,
                                                              const float *__restrict__ centers, const unsigned char *__restrict__ center_zero, int k,
                                                              const float *__restrict__ s, unsigned *__restrict__ clusters, float *__restrict__ bounds) {
    extern __shared__ float lds[];
    float *r_col = lds + threadIdx.x;
    for (size_t i = (size_t)blockIdx.x * kKmBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kKmBlock) {
        const int ci = (int)clusters[i];
        const float lb = bounds[2 * i], ub = bounds[2 * i + 1];
        const float ucb = fmaxf(s[ci], lb);                      // s[min_cluster].max(bi.0)
        if (ub <= ucb) continue;
        float p[NB];
        const bool live = load_datum<NB, DIST>(dataset, order ? (size_t)order[i] : i, n_bins, p);
        float u2 = datum_dist<NB, DIST>(live, p, centers, center_zero, ci, n_bins, r_col);
        bounds[2 * i + 1] = u2;                                  // bi.1 = u2
        if (u2 <= ucb) continue;
        float l2 = 3.40282347e+38f;                              // f32::MAX
        int min_cluster = ci;
        for (int j = 0; j < k; ++j) {
            if (j == min_cluster) continue;                      // the CURRENT best, as coded (kmeans.rs:311)
            const float d = datum_dist<NB, DIST>(live, p, centers, center_zero, j, n_bins, r_col);
            if (d < u2) {
                l2 = u2;
                u2 = d;
                min_cluster = j;
            } else if (d < l2) {
                l2 = d;
            }
        }
        bounds[2 * i] = l2;
        if (ci != min_cluster) {
            bounds[2 * i + 1] = u2;
            clusters[i] = (unsigned)min_cluster;
        }
    }
}

// ---- member lists: a STABLE counting sort of the data indices by cluster (histogram per tile -> exclusive scan -> stable scatter), once per training round.  Inside a
// cluster the members keep their data order, which is the order of the reference's sequential f32 `+=` (kmeans.rs:525-530, :393-400).  Round 2 let ONE wave per
// cluster scan the whole assignment vector for its members: k x n reads per round (500 x 1.29 M = 643 M), 267 ms per round at the reference's size.
constexpr int kKmTile = 512;   // data per tile of the counting sort (kKmBlock threads x 2 consecutive data)
__global__ __launch_bounds__(kKmBlock) void k_kmeans_tile_hist(const unsigned *__restrict__ clusters, size_t n, int k, unsigned *__restrict__ tile_hist) {
    extern __shared__ unsigned km_lds[];
    for (int c = threadIdx.x; c < k; c += kKmBlock) km_lds[c] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kKmTile;
    for (int q = threadIdx.x; q < kKmTile; q += kKmBlock)
        if (base + q < n) atomicAdd(&km_lds[min(clusters[base + q], (unsigned)(k - 1))], 1u);
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += kKmBlock) tile_hist[(size_t)blockIdx.x * k + c] = km_lds[c];
}
// per cluster: members in the tiles before tile t (in place), and the cluster's total
__global__ __launch_bounds__(kKmBlock) void k_kmeans_tile_scan(unsigned *__restrict__ tile_hist, size_t n_tiles, int k, unsigned *__restrict__ total) {
    const int c = blockIdx.x * kKmBlock + threadIdx.x;
    if (c >= k) return;
    unsigned run = 0;
    size_t t = 0;
    for (; t + 8 <= n_tiles; t += 8) {   // eight loads in flight, then the dependent adds
        unsigned x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = tile_hist[(t + q) * k + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            tile_hist[(t + q) * k + c] = run;
            run += x[q];
        }
    }
    for (; t < n_tiles; ++t) {
        const unsigned x = tile_hist[t * k + c];
        tile_hist[t * k + c] = run;
        run += x;
    }
    total[c] = run;
}
// exclusive scan of the totals (one workgroup; k <= a few thousand), start[k] = n
__global__ __launch_bounds__(kKmBlock) void k_kmeans_start_scan(const unsigned *__restrict__ total, int k, unsigned *__restrict__ start) {
    __shared__ unsigned part[kKmBlock];
    const int per = (k + kKmBlock - 1) / kKmBlock, c0 = min(k, (int)threadIdx.x * per), c1 = min(k, c0 + per);
    unsigned sum = 0;
    for (int c = c0; c < c1; ++c) sum += total[c];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kKmBlock; d <<= 1) {
        const unsigned x = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    for (int c = c0; c < c1; ++c) {
        start[c] = run;
        run += total[c];
    }
    if (threadIdx.x == kKmBlock - 1) start[k] = part[kKmBlock - 1];
}
// stable scatter: datum i goes to start[c] + (members of c in earlier tiles) + (members of c earlier in its own tile)
__global__ __launch_bounds__(kKmBlock) void k_kmeans_scatter(const unsigned *__restrict__ clusters, size_t n, int k, const unsigned *__restrict__ tile_hist,
                                                             const unsigned *__restrict__ start, unsigned *__restrict__ members) {
    __shared__ unsigned key[kKmTile];
    const size_t base = (size_t)blockIdx.x * kKmTile;
    for (int q = threadIdx.x; q < kKmTile; q += kKmBlock) key[q] = base + q < n ? min(clusters[base + q], (unsigned)(k - 1)) : 0xffffffffu;
    __syncthreads();
    for (int q = threadIdx.x; q < kKmTile; q += kKmBlock) {
        if (base + q >= n) continue;
        const unsigned c = key[q];
        unsigned rank = 0;
        for (int e = 0; e < q; ++e) rank += key[e] == c ? 1u : 0u;   // earlier data of the same cluster inside the tile
        members[start[c] + tile_hist[(size_t)blockIdx.x * k + c] + rank] = (unsigned)(base + q);
    }
}
typedef float km_f32x4 __attribute__((ext_vector_type(4)));
size_t member_list_tiles(size_t n) { return (n + kKmTile - 1) / kKmTile; }
hipError_t launch_member_lists(const uint32_t *keys, size_t n, int k, uint32_t *tile_hist, uint32_t *total, uint32_t *start, uint32_t *members, hipStream_t stream) {
    if (size_t(k) * 4 > 64 * 1024) return hipErrorInvalidValue;
    const size_t n_tiles = member_list_tiles(n);
    hipLaunchKernelGGL(k_kmeans_tile_hist, dim3((unsigned)n_tiles), dim3(kKmBlock), size_t(k) * 4, stream, (const unsigned *)keys, n, k, (unsigned *)tile_hist);
    hipLaunchKernelGGL(k_kmeans_tile_scan, dim3((unsigned)((k + kKmBlock - 1) / kKmBlock)), dim3(kKmBlock), 0, stream, (unsigned *)tile_hist, n_tiles, k, (unsigned *)total);
    hipLaunchKernelGGL(k_kmeans_start_scan, dim3(1), dim3(kKmBlock), 0, stream, (const unsigned *)total, k, (unsigned *)start);
    hipLaunchKernelGGL(k_kmeans_scatter, dim3((unsigned)n_tiles), dim3(kKmBlock), 0, stream, (const unsigned *)keys, n, k, (const unsigned *)tile_hist, (const unsigned *)start,
                       (unsigned *)members);
    return hipGetLastError();
}

// The ordered sums are a chain of dependent f32 additions per (cluster, bin); what need NOT be ordered are the loads.  So the members' histograms are first copied,
// by every CU at once, into a staging buffer laid out per cluster and BIN-major -- staged[start[c] * rows + b * count_c + m] = bin b of the cluster's m-th member (rows =
// n_bins, + 1 row of upper bounds for growbatch) -- and then ONE wave per cluster streams it: lane b walks its own contiguous row with 16-byte loads and adds the values
// strictly in member order (kmeans.rs:525-530, :393-400).  With the EMD heuristic a handful of clusters end up holding most of the data (kmeans.rs:287-334 as coded):
// walking a 10^6-member list row by row was a chain of 10^6 dependent HBM round trips (79 ms per round at the reference's size).
__global__ __launch_bounds__(kKmBlock) void k_kmeans_stage(const float *__restrict__ dataset, const unsigned *__restrict__ order, const unsigned *__restrict__ clusters,
                                                           const unsigned *__restrict__ members, const unsigned *__restrict__ start, const float *__restrict__ bounds, size_t n,
                                                           int n_bins, int k, int rows, float *__restrict__ staged) {
    for (size_t j = (size_t)blockIdx.x * kKmBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kKmBlock) {
        const unsigned jj = members[j];                               // position in the (shuffled) data
        const unsigned c = min(clusters[jj], (unsigned)(k - 1));
        const unsigned lo = start[c], cnt = start[c + 1] - lo, m = (unsigned)j - lo;
        const size_t row = order ? (size_t)order[jj] : (size_t)jj;
        float *__restrict__ out = staged + (size_t)lo * rows + m;
        for (int b = 0; b < n_bins; ++b) out[(size_t)b * cnt] = dataset[row * (size_t)n_bins + b];
        if (rows > n_bins) out[(size_t)n_bins * cnt] = bounds[2 * (size_t)jj + 1];
    }
}
__global__ __launch_bounds__(64) void k_kmeans_center_sums(const float *__restrict__ staged, const unsigned *__restrict__ start, int n_bins, int rows, float *__restrict__ sums,
                                                           float *__restrict__ counts, float *__restrict__ sq) {
    const unsigned c = blockIdx.x, lane = threadIdx.x;
    const unsigned lo = start[c], cnt = start[c + 1] - lo;
    if ((int)lane < rows) {
        const float *__restrict__ row = staged + (size_t)lo * rows + (size_t)lane * cnt;
        const bool square = (int)lane == n_bins;                      // the growbatch row of upper bounds: sum of bounds[i].1.powf(2.0)
        float acc = 0.0f, n_f = 0.0f;
        unsigned m = 0;
        for (; m < cnt && ((size_t)(row + m) & 15u) != 0; ++m) {      // up to the first 16-byte boundary of this lane's row
            const float v = row[m];
            acc += square ? v * v : v;
            n_f += 1.0f;
        }
        for (; m + 32 <= cnt; m += 32) {                              // eight 16-byte loads in flight, then 32 ordered additions
            km_f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const km_f32x4 *>(row + m + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc += square ? v[q].x * v[q].x : v[q].x;
                acc += square ? v[q].y * v[q].y : v[q].y;
                acc += square ? v[q].z * v[q].z : v[q].z;
                acc += square ? v[q].w * v[q].w : v[q].w;
                n_f += 1.0f; n_f += 1.0f; n_f += 1.0f; n_f += 1.0f;   // f32 `+= 1.0` per member (sticks at 2^24 exactly like the reference)
            }
        }
        for (; m < cnt; ++m) {
            const float v = row[m];
            acc += square ? v * v : v;
            n_f += 1.0f;
        }
        if ((int)lane < n_bins) sums[(size_t)c * n_bins + lane] = acc;
        else if (sq) sq[c] = acc;
        if (lane == 0) counts[c] = n_f;
    }
}

// bounds after the centers moved (kmeans.rs:571-578 = :445-452): upper += movement[own], lower -= the longest movement of any OTHER center
__global__ __launch_bounds__(kKmBlock) void k_kmeans_shift_bounds(const unsigned *__restrict__ clusters, size_t n, const float *__restrict__ movement, int longest_idx,
                                                                  float longest, float second, float *__restrict__ bounds) {
    for (size_t i = (size_t)blockIdx.x * kKmBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kKmBlock) {
        const unsigned c = clusters[i];
        bounds[2 * i + 1] += movement[c];
        bounds[2 * i] -= ((int)c == longest_idx) ? second : longest;
    }
}
__global__ __launch_bounds__(kKmBlock) void k_kmeans_init_state(size_t n, unsigned *__restrict__ clusters, float *__restrict__ bounds) {
    for (size_t i = (size_t)blockIdx.x * kKmBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kKmBlock) {
        clusters[i] = 0;
        bounds[2 * i] = 0.0f;
        bounds[2 * i + 1] = 3.40282347e+38f;                     // (0f32, f32::MAX), kmeans.rs:520
    }
}

}  // namespace rs

namespace {

// host-side preparation of the centers: zero padding to NB and, for EMD, the normalisation emd_1d applies to its copy of q (emd.rs:57,65)
void prepare_centers(int dist, const float *centers, int n_centers, int n_bins, int nb, std::vector<float> &out, std::vector<unsigned char> &zero) {
    out.assign(size_t(n_centers) * nb, 0.0f);
    zero.assign(size_t(n_centers), 0);
    for (int c = 0; c < n_centers; ++c) {
        const float *q = centers + size_t(c) * n_bins;
        float sum = 0.0f;
        for (int b = 0; b < n_bins; ++b) sum += q[b];
        if (dist == RS_DIST_EMD && sum == 0.0f) {
            zero[size_t(c)] = 1;
            continue;
        }
        for (int b = 0; b < n_bins; ++b) out[size_t(c) * nb + b] = dist == RS_DIST_EMD ? q[b] / sum : q[b];
    }
}

int check_args(const char *who, const rs_table *t, int dist, const void *d_dataset, size_t n, const float *centers, int n_centers, int n_bins) {
    if (!t || (!d_dataset && n) || !centers) return fail(RS_ERR_INVALID, std::string(who) + ": NULL argument");
    if (dist != RS_DIST_EMD && dist != RS_DIST_L2) return fail(RS_ERR_INVALID, std::string(who) + ": dist is RS_DIST_EMD or RS_DIST_L2");
    if (n_bins < 1 || n_bins > 64) return fail(RS_ERR_UNSUPPORTED, std::string(who) + ": 1..64 bins per histogram");
    if (n_centers < 1) return fail(RS_ERR_INVALID, std::string(who) + ": no centers (Rust: index out of bounds on centers[0], kmeans.rs:196)");
    return RS_OK;
}

// Centers are staged in a scratch buffer owned by the table.  It only ever grows, and it is rewritten only after the stream has drained, so a
// sweep still in flight never sees the next call's centers.  (Stream-ordered pool allocations -- hipMallocAsync / hipFreeAsync per call -- were
// tried first and produced sporadically wrong clusters: the staged centers were not reliably in place when the kernel ran.)
int stage_centers(rs_table *t, const std::vector<float> &prepared, const std::vector<unsigned char> &zero, float **d_centers, unsigned char **d_zero) {
    const size_t c_bytes = round_up(prepared.size() * sizeof(float), 256), need = c_bytes + round_up(zero.size(), 256);
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    if (need > t->km_scratch_bytes) {
        if (t->d_km_scratch) RS_HIP(hipFree(t->d_km_scratch), "hipFree(k-means scratch)");
        t->d_km_scratch = nullptr;
        t->km_scratch_bytes = 0;
        RS_HIP(hipMalloc(&t->d_km_scratch, need), "hipMalloc(k-means scratch)");
        t->km_scratch_bytes = need;
    }
    *d_centers = reinterpret_cast<float *>(t->d_km_scratch);
    *d_zero = reinterpret_cast<unsigned char *>((char *)t->d_km_scratch + c_bytes);
    RS_HIP(hipMemcpy(*d_centers, prepared.data(), prepared.size() * sizeof(float), hipMemcpyHostToDevice), "k-means centers upload");
    RS_HIP(hipMemcpy(*d_zero, zero.data(), zero.size(), hipMemcpyHostToDevice), "k-means centers upload");
    return RS_OK;
}

// histograms are padded with empty bins to the next instantiated width (registers and LDS rows per thread)
int padded_bins(int n_bins) {
    for (int nb : {8, 16, 24, 32, 48, 64})
        if (n_bins <= nb) return nb;
    return 64;
}

dim3 km_grid(size_t n) { return dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n + kKmBlock - 1) / kKmBlock, 16384))); }

}  // namespace

extern "C" {

// one emd_1d / l2_dist on the host (the same arithmetic as the kernels, for callers that need single distances: init_s, center movement)
int rs_histogram_distance(int dist, const float *p, const float *q, int n_bins, float *out) {
    if (!p || !q || !out) return fail(RS_ERR_INVALID, "rs_histogram_distance: NULL argument");
    if (n_bins < 1 || n_bins > 64) return fail(RS_ERR_UNSUPPORTED, "rs_histogram_distance: 1..64 bins per histogram");
    if (dist == RS_DIST_L2) {
        float sum = 0.0f;
        for (int i = 0; i < n_bins; ++i) {
            const float d = p[i] - q[i];
            sum += d * d;
        }
        *out = std::sqrt(sum);
        return RS_OK;
    }
    if (dist != RS_DIST_EMD) return fail(RS_ERR_INVALID, "rs_histogram_distance: dist is RS_DIST_EMD or RS_DIST_L2");
    float ps = 0.0f, qs = 0.0f, r[64];
    for (int i = 0; i < n_bins; ++i) ps += p[i];
    for (int i = 0; i < n_bins; ++i) qs += q[i];
    if (ps == 0.0f || qs == 0.0f) {
        *out = 0.0f;
        return RS_OK;
    }
    float w = 0.0f, cost = 0.0f;
    unsigned long long P = 0, Q = 0;
    for (int b = 0; b < n_bins; ++b) {
        const float pb = p[b] / ps, qb = q[b] / qs;
        w += pb < qb ? pb : qb;
        r[b] = pb - qb;
        P |= (unsigned long long)(r[b] > 0.0f) << b;
        Q |= (unsigned long long)(r[b] < 0.0f) << b;
    }
    float factor = 4.45f * w - 1.5f;
    factor = factor < 1.0f ? 1.0f : (factor > 4.0f ? 4.0f : factor);
    const int u = (int)std::round((float)n_bins / factor);
    for (int d = 1; d < u && P && Q; ++d)
        for (int side = 0; side < 2; ++side) {
            unsigned long long m = side == 0 ? P & (Q << d) : P & (Q >> d);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int k = side == 0 ? j - d : j + d;
                const float pj = r[j], qk = -r[k];
                const float mass = pj < qk ? pj : qk;
                w += mass;
                cost += mass * (float)d;
                r[j] = pj - mass;
                r[k] = -(qk - mass);
                if (r[j] == 0.0f) P &= ~(1ull << j);
                if (r[k] == 0.0f) Q &= ~(1ull << k);
            }
        }
    *out = std::fabs(cost + (1.0f - w) * (float)u);
    return RS_OK;
}

// Kmeans::predict (kmeans.rs:173-211).  d_dataset: DEVICE [n][n_bins] f32 row-major (Vec<Histogram>); centers: HOST [n_centers][n_bins];
// d_clusters[n] (u32, the value written to the bucket file, main.rs:378-380) and d_min_dist[n] (the distance to that center) are DEVICE
// buffers, either may be NULL.  Asynchronous on the table's stream (the table only lends its device and stream).
int rs_kmeans_predict(rs_table *t, int dist, const float *d_dataset, size_t n, const float *centers, int n_centers, int n_bins, uint32_t *d_clusters,
                      float *d_min_dist) {
    if (int rc = check_args("rs_kmeans_predict", t, dist, d_dataset, n, centers, n_centers, n_bins)) return rc;
    if (!d_clusters && !d_min_dist) return RS_OK;
    const int nb = padded_bins(n_bins);
    std::vector<float> prepared;
    std::vector<unsigned char> zero;
    prepare_centers(dist, centers, n_centers, n_bins, nb, prepared, zero);
    float *d_centers = nullptr;
    unsigned char *d_zero = nullptr;
    if (int rc = stage_centers(t, prepared, zero, &d_centers, &d_zero)) return rc;
    hipError_t e = hipSuccess;
    if (e == hipSuccess && n > 0) {
        const size_t lds = size_t(nb) * kKmBlock * sizeof(float);
        const dim3 grid = km_grid(n), block(kKmBlock);
#define RS_KM_PREDICT(NB_)                                                                                                                          \
    if (nb == NB_) {                                                                                                                           \
        if (dist == RS_DIST_EMD)                                                                                                               \
            hipLaunchKernelGGL((k_kmeans_predict<NB_, RS_DIST_EMD>), grid, block, lds, t->stream, d_dataset, n, n_bins, d_centers, d_zero, n_centers, \
                               d_clusters, d_min_dist);                                                                                        \
        else                                                                                                                                   \
            hipLaunchKernelGGL((k_kmeans_predict<NB_, RS_DIST_L2>), grid, block, 0, t->stream, d_dataset, n, n_bins, d_centers, d_zero, n_centers,    \
                               d_clusters, d_min_dist);                                                                                        \
    }
        RS_KM_PREDICT(8) RS_KM_PREDICT(16) RS_KM_PREDICT(24) RS_KM_PREDICT(32) RS_KM_PREDICT(48) RS_KM_PREDICT(64)
#undef RS_KM_PREDICT
        e = hipGetLastError();
    }
    RS_HIP(e, "k_kmeans_predict");
    return RS_OK;
}

// update_min_dists (kmeans.rs:603-619), the kmeans++ step: d_min_dists[i] = min(d_min_dists[i], dist(dataset[i], new_center)^2); center on the HOST
int rs_update_min_dists(rs_table *t, int dist, float *d_min_dists, const float *d_dataset, size_t n, const float *new_center, int n_bins) {
    if (int rc = check_args("rs_update_min_dists", t, dist, d_dataset, n, new_center, 1, n_bins)) return rc;
    if (!d_min_dists && n) return fail(RS_ERR_INVALID, "rs_update_min_dists: NULL argument");
    const int nb = padded_bins(n_bins);
    std::vector<float> prepared;
    std::vector<unsigned char> zero;
    prepare_centers(dist, new_center, 1, n_bins, nb, prepared, zero);
    float *d_center = nullptr;
    unsigned char *d_zero = nullptr;
    if (int rc = stage_centers(t, prepared, zero, &d_center, &d_zero)) return rc;
    hipError_t e = hipSuccess;
    if (e == hipSuccess && n > 0) {
        const size_t lds = size_t(nb) * kKmBlock * sizeof(float);
        const dim3 grid = km_grid(n), block(kKmBlock);
        const int z = zero[0];
#define RS_KM_MIND(NB_)                                                                                                                                 \
    if (nb == NB_) {                                                                                                                              \
        if (dist == RS_DIST_EMD)                                                                                                                  \
            hipLaunchKernelGGL((k_update_min_dists<NB_, RS_DIST_EMD>), grid, block, lds, t->stream, d_dataset, n, n_bins, d_center, z, d_min_dists);       \
        else                                                                                                                                      \
            hipLaunchKernelGGL((k_update_min_dists<NB_, RS_DIST_L2>), grid, block, 0, t->stream, d_dataset, n, n_bins, d_center, z, d_min_dists);          \
    }
        RS_KM_MIND(8) RS_KM_MIND(16) RS_KM_MIND(24) RS_KM_MIND(32) RS_KM_MIND(48) RS_KM_MIND(64)
#undef RS_KM_MIND
        e = hipGetLastError();
    }
    RS_HIP(e, "k_update_min_dists");
    return RS_OK;
}

}  // extern "C"

// ---- Kmeans::init_s / reassign_clusters / fit_regular / fit_growbatch (kmeans.rs:213-601) --------------------------------------------------------------
namespace {

struct KmDevice {   // per-call device workspace of the training loops
    float *raw = nullptr, *s = nullptr, *mv = nullptr, *sums = nullptr, *counts = nullptr, *sq = nullptr;
    unsigned *members = nullptr, *tile_hist = nullptr, *total = nullptr, *start = nullptr;   // the member lists of a round (stable counting sort by cluster)
    float *staged = nullptr;                                                                 // the members' histograms per cluster, bin-major
    size_t n_tiles = 0, staged_floats = 0;
    ~KmDevice() {
        for (float *q : {raw, s, mv, sums, counts, sq})
            if (q) (void)hipFree(q);
        for (unsigned *q : {members, tile_hist, total, start})
            if (q) (void)hipFree(q);
        if (staged) (void)hipFree(staged);
    }
    int alloc_lists(size_t n, int k) {
        n_tiles = (n + kKmTile - 1) / kKmTile;
        hipError_t e = hipMalloc((void **)&members, std::max<size_t>(n, 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&tile_hist, std::max<size_t>(n_tiles * size_t(k), 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&total, size_t(k) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&start, (size_t(k) + 1) * 4);
        return e == hipSuccess ? RS_OK : hip_fail(e, "k-means member lists");
    }
    int alloc(int k, int n_bins) {
        hipError_t e = hipMalloc((void **)&raw, size_t(k) * n_bins * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&s, size_t(k) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&mv, size_t(k) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&sums, size_t(k) * n_bins * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&counts, size_t(k) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&sq, size_t(k) * 4);
        return e == hipSuccess ? RS_OK : hip_fail(e, "k-means workspace");
    }
};

// stages the (prepared) centers for the distance kernels; returns the device pointers
int stage_for(rs_table *t, int dist, const float *centers, int k, int n_bins, int nb, float **d_centers, unsigned char **d_zero) {
    std::vector<float> prepared;
    std::vector<unsigned char> zero;
    prepare_centers(dist, centers, k, n_bins, nb, prepared, zero);
    return stage_centers(t, prepared, zero, d_centers, d_zero);
}

#define RS_KM_DISPATCH(KERNEL, GRID, BLOCK, ...)                                                                                   \
    do {                                                                                                                            \
        const size_t lds_ = dist == RS_DIST_EMD ? size_t(nb) * kKmBlock * sizeof(float) : 0;                                        \
        if (dist == RS_DIST_EMD) {                                                                                                  \
            if (nb == 8) hipLaunchKernelGGL((KERNEL<8, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);                   \
            else if (nb == 16) hipLaunchKernelGGL((KERNEL<16, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);            \
            else if (nb == 24) hipLaunchKernelGGL((KERNEL<24, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);            \
            else if (nb == 32) hipLaunchKernelGGL((KERNEL<32, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);            \
            else if (nb == 48) hipLaunchKernelGGL((KERNEL<48, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERNEL<64, RS_DIST_EMD>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);                          \
        } else {                                                                                                                    \
            if (nb == 8) hipLaunchKernelGGL((KERNEL<8, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);                    \
            else if (nb == 16) hipLaunchKernelGGL((KERNEL<16, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);             \
            else if (nb == 24) hipLaunchKernelGGL((KERNEL<24, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);             \
            else if (nb == 32) hipLaunchKernelGGL((KERNEL<32, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);             \
            else if (nb == 48) hipLaunchKernelGGL((KERNEL<48, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);             \
            else hipLaunchKernelGGL((KERNEL<64, RS_DIST_L2>), GRID, BLOCK, lds_, t->stream, __VA_ARGS__);                           \
        }                                                                                                                           \
    } while (0)

// init_s on the device: s (device, in/out) against `centers` (host); the raw centers go to w.raw (the datum side of dist_func(&centers[i], &centers[j]))
int init_s_device(rs_table *t, int dist, const float *centers, int k, int n_bins, KmDevice &w) {
    const int nb = padded_bins(n_bins);
    float *d_centers = nullptr;
    unsigned char *d_zero = nullptr;
    if (int rc = stage_for(t, dist, centers, k, n_bins, nb, &d_centers, &d_zero)) return rc;
    RS_HIP(hipMemcpy(w.raw, centers, size_t(k) * n_bins * 4, hipMemcpyHostToDevice), "k-means centers upload");
    const dim3 grid((unsigned)((k + kKmBlock - 1) / kKmBlock)), block(kKmBlock);
    RS_KM_DISPATCH(k_kmeans_init_s, grid, block, (const float *)w.raw, n_bins, (const float *)d_centers, (const unsigned char *)d_zero, k, w.s);
    RS_HIP(hipGetLastError(), "k_kmeans_init_s");
    return RS_OK;
}

int reassign_device(rs_table *t, int dist, const float *d_dataset, const uint32_t *d_order, size_t n, int k, int n_bins, const float *d_s, uint32_t *d_clusters,
                    float *d_bounds) {
    // the centers were staged by init_s_device just before (same stream, same scratch): reuse them
    const int nb = padded_bins(n_bins);
    float *d_centers = reinterpret_cast<float *>(t->d_km_scratch);
    unsigned char *d_zero = reinterpret_cast<unsigned char *>((char *)t->d_km_scratch + round_up(size_t(k) * nb * sizeof(float), 256));
    if (n == 0) return RS_OK;
    const dim3 grid = km_grid(n), block(kKmBlock);
    RS_KM_DISPATCH(k_kmeans_reassign, grid, block, d_dataset, (const unsigned *)d_order, n, n_bins, (const float *)d_centers, (const unsigned char *)d_zero, k, d_s,
                   (unsigned *)d_clusters, d_bounds);
    RS_HIP(hipGetLastError(), "k_kmeans_reassign");
    return RS_OK;
}

// kmeans.rs:552-569 (= :427-444)
void two_longest(const std::vector<float> &mv, int *longest_idx, float *longest, float *second) {
    *longest_idx = 0;
    *longest = mv[0];
    *second = mv[1];
    if (*longest < *second) {
        *longest = mv[1];
        *second = mv[0];
        *longest_idx = 1;
    }
    for (size_t i = 2; i < mv.size(); ++i) {
        if (*longest < mv[i]) {
            *second = *longest;
            *longest = mv[i];
            *longest_idx = int(i);
        } else if (*second < mv[i]) {
            *second = mv[i];
        }
    }
}

int fit_check(const char *who, const rs_table *t, int dist, const void *d_dataset, size_t n, const float *centers, int k, int n_bins) {
    if (int rc = check_args(who, t, dist, d_dataset, n, centers, k, n_bins)) return rc;
    if (k < 2) return fail(RS_ERR_INVALID, std::string(who) + ": at least two centers (Rust: index out of bounds on center_movement[1], kmeans.rs:554)");
    if (n == 0) return fail(RS_ERR_INVALID, std::string(who) + ": empty dataset");
    return RS_OK;
}

// one training step after the assignment: sums in data order on the device, means / movements on the host (k * n_bins numbers), bounds shifted on the device.
// growbatch: the `&& count > 0` form of the mean and the squared upper bounds per cluster.  new_centers / counts / sq are host outputs.
int update_step(rs_table *t, int dist, const float *d_dataset, const uint32_t *d_order, size_t n, std::vector<float> &centers, int k, int n_bins, bool growbatch,
                const uint32_t *d_clusters, float *d_bounds, KmDevice &w, std::vector<float> &mv, std::vector<float> &counts, std::vector<float> &sq) {
    if (!w.members || w.n_tiles != (n + kKmTile - 1) / kKmTile)
        if (int rc = w.alloc_lists(n, k)) return rc;
    if (size_t(k) * 4 > 64 * 1024) return fail(RS_ERR_UNSUPPORTED, "k-means training loops: at most 16 384 centers (the counting sort's LDS histogram)");
    hipLaunchKernelGGL(k_kmeans_tile_hist, dim3((unsigned)w.n_tiles), dim3(kKmBlock), size_t(k) * 4, t->stream, (const unsigned *)d_clusters, n, k, w.tile_hist);
    hipLaunchKernelGGL(k_kmeans_tile_scan, dim3((unsigned)((k + kKmBlock - 1) / kKmBlock)), dim3(kKmBlock), 0, t->stream, w.tile_hist, w.n_tiles, k, w.total);
    hipLaunchKernelGGL(k_kmeans_start_scan, dim3(1), dim3(kKmBlock), 0, t->stream, (const unsigned *)w.total, k, w.start);
    hipLaunchKernelGGL(k_kmeans_scatter, dim3((unsigned)w.n_tiles), dim3(kKmBlock), 0, t->stream, (const unsigned *)d_clusters, n, k, (const unsigned *)w.tile_hist,
                       (const unsigned *)w.start, w.members);
    RS_HIP(hipGetLastError(), "k-means member lists");
    const int rows = n_bins + (growbatch ? 1 : 0);
    if (rows > 64) return fail(RS_ERR_UNSUPPORTED, "k-means training loops: at most 63 bins (one lane of a wave per bin)");
    if (w.staged_floats < n * size_t(rows) + 4) {
        if (w.staged) (void)hipFree(w.staged);
        w.staged = nullptr;
        w.staged_floats = n * size_t(rows) + 4;
        RS_HIP(hipMalloc((void **)&w.staged, w.staged_floats * 4), "k-means staging buffer");
    }
    hipLaunchKernelGGL(k_kmeans_stage, km_grid(n), dim3(kKmBlock), 0, t->stream, d_dataset, (const unsigned *)d_order, (const unsigned *)d_clusters, (const unsigned *)w.members,
                       (const unsigned *)w.start, (const float *)d_bounds, n, n_bins, k, rows, w.staged);
    hipLaunchKernelGGL(k_kmeans_center_sums, dim3((unsigned)k), dim3(64), 0, t->stream, (const float *)w.staged, (const unsigned *)w.start, n_bins, rows, w.sums, w.counts,
                       growbatch ? w.sq : (float *)nullptr);
    RS_HIP(hipGetLastError(), "k_kmeans_center_sums");
    std::vector<float> mass(size_t(k) * n_bins);
    counts.assign(size_t(k), 0.0f);
    sq.assign(size_t(k), 0.0f);
    RS_HIP(hipMemcpyAsync(mass.data(), w.sums, mass.size() * 4, hipMemcpyDeviceToHost, t->stream), "k-means sums download");
    RS_HIP(hipMemcpyAsync(counts.data(), w.counts, size_t(k) * 4, hipMemcpyDeviceToHost, t->stream), "k-means counts download");
    if (growbatch) RS_HIP(hipMemcpyAsync(sq.data(), w.sq, size_t(k) * 4, hipMemcpyDeviceToHost, t->stream), "k-means squares download");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    for (int j = 0; j < k; ++j)        // kmeans.rs:531-543 / :402-414
        for (int b = 0; b < n_bins; ++b) {
            float &m = mass[size_t(j) * n_bins + b];
            if (m > 0.0f && (!growbatch || counts[size_t(j)] > 0.0f)) m /= counts[size_t(j)];
        }
    mv.assign(size_t(k), 0.0f);
    for (int j = 0; j < k; ++j)        // dist_func(&new_centers[j], &self.centers[j]), kmeans.rs:546-549
        if (int rc = rs_histogram_distance(dist, mass.data() + size_t(j) * n_bins, centers.data() + size_t(j) * n_bins, n_bins, &mv[size_t(j)])) return rc;
    int longest_idx;
    float longest, second;
    two_longest(mv, &longest_idx, &longest, &second);
    RS_HIP(hipMemcpyAsync(w.mv, mv.data(), size_t(k) * 4, hipMemcpyHostToDevice, t->stream), "k-means movement upload");
    hipLaunchKernelGGL(k_kmeans_shift_bounds, km_grid(n), dim3(kKmBlock), 0, t->stream, (const unsigned *)d_clusters, n, (const float *)w.mv, longest_idx, longest, second,
                       d_bounds);
    RS_HIP(hipGetLastError(), "k_kmeans_shift_bounds");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");   // mv (host vector) was the source of an async copy
    centers = mass;                    // self.centers = new_centers
    return RS_OK;
}

// sum of the upper bounds in data order (f32, sequential), as the reference prints it
int upper_bound_sum(rs_table *t, const float *d_bounds, size_t n, float *out) {
    std::vector<float> b(2 * n);
    RS_HIP(hipMemcpyAsync(b.data(), d_bounds, b.size() * 4, hipMemcpyDeviceToHost, t->stream), "k-means bounds download");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum += b[2 * i + 1];
    *out = sum;
    return RS_OK;
}

}  // namespace

extern "C" {

// Kmeans::init_random's choice among restarts (kmeans.rs:104-165): the caller draws `n_restarts` candidate center sets with its rng (`choose_multiple`, :120); this
// scores them exactly as coded -- per restart, distances[i] = sum over j != i of dist(c_i, c_j), sum over i, divided by the k (k - 1) pairs (:133-147) -- and returns
// the index of the maximum (the LAST of equal maxima: Iterator::max_by, :151-156).  centers: HOST [n_restarts][k][n_bins]; cluster_dists: HOST out [n_restarts], may be NULL.
int rs_kmeans_pick_restart(rs_table *t, int dist, const float *centers, int n_restarts, int n_centers, int n_bins, float *cluster_dists, int *best) {
    if (!best || n_restarts < 1) return fail(RS_ERR_INVALID, "rs_kmeans_pick_restart: bad argument");
    if (int rc = check_args("rs_kmeans_pick_restart", t, dist, centers, 0, centers, n_centers, n_bins)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    KmDevice w;
    if (int rc = w.alloc(n_centers, n_bins)) return rc;
    const int nb = padded_bins(n_bins);
    std::vector<float> row(static_cast<size_t>(n_centers), 0.0f), cd(static_cast<size_t>(n_restarts), 0.0f);
    for (int r = 0; r < n_restarts; ++r) {
        const float *cr = centers + size_t(r) * n_centers * n_bins;
        float *d_centers = nullptr;
        unsigned char *d_zero = nullptr;
        if (int rc = stage_for(t, dist, cr, n_centers, n_bins, nb, &d_centers, &d_zero)) return rc;
        RS_HIP(hipMemcpy(w.raw, cr, size_t(n_centers) * n_bins * 4, hipMemcpyHostToDevice), "k-means centers upload");
        const dim3 grid((unsigned)((n_centers + kKmBlock - 1) / kKmBlock)), block(kKmBlock);
        const int k = n_centers;
        RS_KM_DISPATCH(k_kmeans_row_sums, grid, block, (const float *)w.raw, n_bins, (const float *)d_centers, (const unsigned char *)d_zero, k, w.s);
        RS_HIP(hipGetLastError(), "k_kmeans_row_sums");
        RS_HIP(hipMemcpyAsync(row.data(), w.s, size_t(n_centers) * 4, hipMemcpyDeviceToHost, t->stream), "row sums download");
        RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
        float sum = 0.0f;
        for (int i = 0; i < n_centers; ++i) sum += row[size_t(i)];                  // sum += distances[i], kmeans.rs:146
        cd[size_t(r)] = sum / float(size_t(n_centers) * size_t(n_centers - 1));   // count = k (k - 1) pairs, :148
    }
    int arg = 0;
    for (int r = 1; r < n_restarts; ++r)   // max_by(partial_cmp): the last of equal maxima; NaN compares Equal (unwrap_or) and so replaces too
        if (!(cd[size_t(r)] < cd[size_t(arg)])) arg = r;
    if (cluster_dists) std::memcpy(cluster_dists, cd.data(), cd.size() * 4);
    *best = arg;
    return RS_OK;
}

// Kmeans::init_s (kmeans.rs:267-285) for callers that drive the loop themselves: s (HOST, in/out, n_centers floats) is only ever lowered, then halved
int rs_kmeans_init_s(rs_table *t, int dist, const float *centers, int n_centers, int n_bins, float *s) {
    if (int rc = check_args("rs_kmeans_init_s", t, dist, centers, 0, centers, n_centers, n_bins)) return rc;
    if (!s) return fail(RS_ERR_INVALID, "rs_kmeans_init_s: s is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    KmDevice w;
    if (int rc = w.alloc(n_centers, n_bins)) return rc;
    RS_HIP(hipMemcpy(w.s, s, size_t(n_centers) * 4, hipMemcpyHostToDevice), "s upload");
    if (int rc = init_s_device(t, dist, centers, n_centers, n_bins, w)) return rc;
    RS_HIP(hipMemcpyAsync(s, w.s, size_t(n_centers) * 4, hipMemcpyDeviceToHost, t->stream), "s download");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    return RS_OK;
}

// Kmeans::reassign_clusters (kmeans.rs:287-334) = assignment_with_bounds (:213-265): d_clusters[n] and d_bounds[n][2] = (lower, upper) are DEVICE arrays updated
// in place; d_order (DEVICE, may be NULL): datum i is dataset[d_order[i]]; centers and s on the HOST.  Synchronises.
int rs_kmeans_reassign(rs_table *t, int dist, const float *d_dataset, const uint32_t *d_order, size_t n, const float *centers, int n_centers, int n_bins, const float *s,
                       uint32_t *d_clusters, float *d_bounds) {
    if (int rc = check_args("rs_kmeans_reassign", t, dist, d_dataset, n, centers, n_centers, n_bins)) return rc;
    if (!s || ((!d_clusters || !d_bounds) && n)) return fail(RS_ERR_INVALID, "rs_kmeans_reassign: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const int nb = padded_bins(n_bins);
    float *d_centers = nullptr;
    unsigned char *d_zero = nullptr;
    if (int rc = stage_for(t, dist, centers, n_centers, n_bins, nb, &d_centers, &d_zero)) return rc;
    float *d_s = nullptr;
    RS_HIP(hipMalloc((void **)&d_s, size_t(n_centers) * 4), "hipMalloc(s)");
    hipError_t e = hipMemcpy(d_s, s, size_t(n_centers) * 4, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? reassign_device(t, dist, d_dataset, d_order, n, n_centers, n_bins, d_s, d_clusters, d_bounds) : hip_fail(e, "s upload");
    if (rc == RS_OK && (e = hipStreamSynchronize(t->stream)) != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    (void)hipFree(d_s);
    return rc;
}

// Kmeans::fit_regular (kmeans.rs:497-600): `iterations` rounds (the reference: 10, :589) of init_s -> reassign_clusters -> means -> bound shifts.
// centers: HOST in/out [n_centers][n_bins]; d_clusters: DEVICE out [n] (the returned Vec<usize>); d_bounds: DEVICE out [n][2], may be NULL;
// inertia: HOST out, may be NULL (sum of the upper bounds / n as printed at :594).  Synchronises.
int rs_kmeans_fit_regular(rs_table *t, int dist, const float *d_dataset, size_t n, float *centers, int n_centers, int n_bins, int iterations, uint32_t *d_clusters,
                          float *d_bounds, float *inertia) {
    if (int rc = fit_check("rs_kmeans_fit_regular", t, dist, d_dataset, n, centers, n_centers, n_bins)) return rc;
    if (!d_clusters || iterations < 1) return fail(RS_ERR_INVALID, "rs_kmeans_fit_regular: d_clusters is NULL or iterations < 1");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    KmDevice w;
    if (int rc = w.alloc(n_centers, n_bins)) return rc;
    float *own_bounds = nullptr;
    if (!d_bounds) {
        RS_HIP(hipMalloc((void **)&own_bounds, n * 8), "hipMalloc(bounds)");
        d_bounds = own_bounds;
    }
    std::vector<float> c(centers, centers + size_t(n_centers) * n_bins), mv, counts, sq;
    std::vector<float> s0(size_t(n_centers), 3.40282347e+38f);           // vec![f32::MAX; k], created ONCE (kmeans.rs:518)
    int rc = RS_OK;
    hipError_t e = hipMemcpy(w.s, s0.data(), s0.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "s upload");
    if (rc == RS_OK) {
        hipLaunchKernelGGL(k_kmeans_init_state, km_grid(n), dim3(kKmBlock), 0, t->stream, n, (unsigned *)d_clusters, d_bounds);
        if ((e = hipGetLastError()) != hipSuccess) rc = hip_fail(e, "k_kmeans_init_state");
    }
    for (int it = 0; rc == RS_OK && it < iterations; ++it) {
        rc = init_s_device(t, dist, c.data(), n_centers, n_bins, w);
        if (rc == RS_OK) rc = reassign_device(t, dist, d_dataset, nullptr, n, n_centers, n_bins, w.s, d_clusters, d_bounds);
        if (rc == RS_OK) rc = update_step(t, dist, d_dataset, nullptr, n, c, n_centers, n_bins, false, d_clusters, d_bounds, w, mv, counts, sq);
    }
    if (rc == RS_OK && inertia) {
        float sum = 0.0f;
        rc = upper_bound_sum(t, d_bounds, n, &sum);
        *inertia = sum / float(n);
    }
    if (rc == RS_OK) std::memcpy(centers, c.data(), c.size() * 4);
    if (own_bounds) (void)hipFree(own_bounds);
    return rc;
}

// Kmeans::fit_growbatch AS CODED (kmeans.rs:336-495: the loop body ends in an unconditional `break`, :492): one pass over the first `batch` items of the
// shuffled data.  d_order: DEVICE [>= batch], shuffled_data[i] = dataset[d_order[i]] (the reference shuffles with its rng, :352: the permutation is an input);
// centers: HOST in/out; d_clusters [batch] / d_bounds [batch][2]: DEVICE out; stats: HOST out {min_change p (:466-471), inertia as printed (:478)}, may be NULL.
int rs_kmeans_fit_growbatch(rs_table *t, int dist, const float *d_dataset, size_t n, const uint32_t *d_order, size_t batch, float *centers, int n_centers, int n_bins,
                            uint32_t *d_clusters, float *d_bounds, float *stats) {
    if (int rc = fit_check("rs_kmeans_fit_growbatch", t, dist, d_dataset, n, centers, n_centers, n_bins)) return rc;
    if (!d_order || !d_clusters || !d_bounds || batch == 0 || batch > n) return fail(RS_ERR_INVALID, "rs_kmeans_fit_growbatch: NULL argument or batch outside 1..n");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    KmDevice w;
    if (int rc = w.alloc(n_centers, n_bins)) return rc;
    std::vector<float> c(centers, centers + size_t(n_centers) * n_bins), mv, counts, sq;
    std::vector<float> s0(size_t(n_centers), 3.40282347e+38f);
    RS_HIP(hipMemcpy(w.s, s0.data(), s0.size() * 4, hipMemcpyHostToDevice), "s upload");
    hipLaunchKernelGGL(k_kmeans_init_state, km_grid(batch), dim3(kKmBlock), 0, t->stream, batch, (unsigned *)d_clusters, d_bounds);
    RS_HIP(hipGetLastError(), "k_kmeans_init_state");
    if (int rc = init_s_device(t, dist, c.data(), n_centers, n_bins, w)) return rc;
    if (int rc = reassign_device(t, dist, d_dataset, d_order, batch, n_centers, n_bins, w.s, d_clusters, d_bounds)) return rc;
    if (int rc = update_step(t, dist, d_dataset, d_order, batch, c, n_centers, n_bins, true, d_clusters, d_bounds, w, mv, counts, sq)) return rc;
    if (stats) {
        float min_change = 0.0f;
        for (int j = 0; j < n_centers; ++j) {   // kmeans.rs:454-471
            const float cn = counts[size_t(j)];
            const float sd = cn <= 1.0f ? INFINITY : std::sqrt(std::fabs(sq[size_t(j)] / (cn * (cn - 1.0f))));
            const float ch = sd / (mv[size_t(j)] + 1e-9f);
            if (j == 0 || ch < min_change) min_change = ch;
        }
        float sum = 0.0f;
        if (int rc = upper_bound_sum(t, d_bounds, batch, &sum)) return rc;
        stats[0] = min_change;
        stats[1] = sum / float(std::min(n, 2 * batch));   // current_batch_idx has already doubled when the inertia is formed (:476-478)
    }
    std::memcpy(centers, c.data(), c.size() * 4);
    return RS_OK;
}

}  // extern "C"
