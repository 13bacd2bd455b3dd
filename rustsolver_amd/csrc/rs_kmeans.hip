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
