/*
 * kmeans_emd.c -- CPU oracle for the abstraction generator's distance sweep:
 * emd_1d (gen_abstraction/emd.rs:53-113), l2_dist (kmeans.rs:622-630), Kmeans::predict (kmeans.rs:173-211) and
 * update_min_dists (kmeans.rs:603-619).  TEST INFRASTRUCTURE ONLY (tests/, smoke, bench.py's cpu_baseline).
 *
 * Pinned by the reference's own known answers (emd.rs:122-180): emd(h, h) == 0.0, emd(66, JT) = 2.7094990435 +- 0.01,
 * emd(27, AA) = 14.2204956945 +- 0.01 (tests/golden/kmeans_emd.json holds the six histograms).
 *
 * Restated literally, f32 throughout, no FMA (-ffp-contract=off): the two histograms are COPIED and normalised by their sums
 * (sequential f32 sums from 0.0); same-bin mass is matched first; u = round(len / clamp(4.45*w - 1.5, 1, 4)) (f32::round: half away
 * from zero); the offsets are the list get_bins_1d(0) builds -- -1, -2, .., -(u-1), then 1, 2, .., u-1 -- stably sorted by |b|,
 * i.e. -1, 1, -2, 2, ..; for every offset all bins j in ascending order move min(p[j], q[j+b]) when both are non-zero;
 * result |cost + (1 - w) * u|.  min!(x, y) is `if x < y {x} else {y}` (emd.rs:10-21).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EMD_MAX_BINS 64

/* get_bins_1d (emd.rs:26-48): recursion kept as written */
static void get_bins_1d(long b, long *bins, int *n, long u) {
    long bp = b;
    if (bp == 0) {
        bp -= 1;
        if (labs(bp) < u) {
            bins[(*n)++] = bp;
            get_bins_1d(bp, bins, n, u);
        }
        bp = b;
        bp += 1;
        if (labs(bp) < u) {
            bins[(*n)++] = bp;
            get_bins_1d(bp, bins, n, u);
        }
    } else {
        if (bp < 0) bp -= 1;
        else bp += 1;
        if (labs(bp) < u) {
            bins[(*n)++] = bp;
            get_bins_1d(bp, bins, n, u);
        }
    }
}

static float min2(float x, float y) { return x < y ? x : y; }

float orc_emd_1d(const float *p_in, const float *q_in, int len) {
    float p[ORC_EMD_MAX_BINS], q[ORC_EMD_MAX_BINS];
    float p_sum = 0.0f, q_sum = 0.0f, cost = 0.0f, w = 0.0f, factor;
    long bins[2 * ORC_EMD_MAX_BINS + 2], u;
    int n_b = 0, i, j;
    if (len < 1 || len > ORC_EMD_MAX_BINS) return NAN;
    memcpy(p, p_in, (size_t)len * sizeof(float));
    memcpy(q, q_in, (size_t)len * sizeof(float));
    for (i = 0; i < len; i++) p_sum += p[i];   /* iter().sum::<f32>() */
    for (i = 0; i < len; i++) q_sum += q[i];
    if (p_sum == 0.0f || q_sum == 0.0f) return 0.0f;   /* emd.rs:59-61 */
    for (i = 0; i < len; i++) {
        p[i] /= p_sum;
        q[i] /= q_sum;
    }
    for (i = 0; i < len; i++) {   /* corresponding bins, emd.rs:72-77 */
        const float mass = min2(p[i], q[i]);
        w += mass;
        p[i] -= mass;
        q[i] -= mass;
    }
    factor = 4.45f * w - 1.5f;   /* emd.rs:83-88 */
    if (factor < 1.0f) factor = 1.0f;
    else if (factor > 4.0f) factor = 4.0f;
    u = (long)roundf((float)len / factor);   /* f32::round: ties away from zero, like roundf */
    get_bins_1d(0, bins, &n_b, u);
    /* sort_by(|x, y| x.abs().partial_cmp(&y.abs())): stable insertion sort on |b| */
    for (i = 1; i < n_b; i++) {
        const long v = bins[i];
        for (j = i; j > 0 && labs(bins[j - 1]) > labs(v); j--) bins[j] = bins[j - 1];
        bins[j] = v;
    }
    for (i = 0; i < n_b; i++)   /* cross bin, emd.rs:96-110 */
        for (j = 0; j < len; j++)
            if (p[j] != 0.0f && (long)j + bins[i] >= 0) {
                const long k = (long)j + bins[i];
                if (k < len && q[k] != 0.0f) {
                    const float mass = min2(p[j], q[k]);
                    w += mass;
                    cost += mass * fabsf((float)j - (float)k);
                    p[j] -= mass;
                    q[k] -= mass;
                }
            }
    return fabsf(cost + (1.0f - w) * (float)u);
}

float orc_l2_dist(const float *a, const float *b, int len) {   /* kmeans.rs:622-630 */
    float sum = 0.0f;
    int i;
    for (i = 0; i < len; i++) {
        const float d = a[i] - b[i];
        sum += d * d;
    }
    return sqrtf(sum);
}

static float dist(int kind, const float *a, const float *b, int len) { return kind == 0 ? orc_emd_1d(a, b, len) : orc_l2_dist(a, b, len); }

/* Kmeans::predict (kmeans.rs:173-211) for rows [lo, hi): clusters[i] = first center with the strictly smallest distance; min_dist[i] (may
 * be NULL) = that distance.  (The reference's `inertia` is a racy AtomicCell load+store, kmeans.rs:205: not a defined value; the sum of
 * min_dist is what it means.) */
void orc_kmeans_predict(int kind, const float *dataset, size_t lo, size_t hi, const float *centers, int n_centers, int len, uint32_t *clusters,
                        float *min_dist) {
    size_t i;
    for (i = lo; i < hi; i++) {
        const float *x = dataset + i * (size_t)len;
        int min_cluster = 0, k;
        float min_variance = dist(kind, x, centers, len);
        for (k = 1; k < n_centers; k++) {
            const float v = dist(kind, x, centers + (size_t)k * (size_t)len, len);
            if (v < min_variance) {
                min_variance = v;
                min_cluster = k;
            }
        }
        clusters[i] = (uint32_t)min_cluster;
        if (min_dist) min_dist[i] = min_variance;
    }
}

/* update_min_dists (kmeans.rs:603-619): d = dist(x, new_center); d = d*d; keep the smaller */
void orc_update_min_dists(int kind, float *min_dists, const float *dataset, size_t n, const float *new_center, int len) {
    size_t i;
    for (i = 0; i < n; i++) {
        float d = dist(kind, dataset + i * (size_t)len, new_center, len);
        d = d * d;
        if (d < min_dists[i]) min_dists[i] = d;
    }
}

/* threaded predict for the timed CPU baseline (rayon par_iter_mut in the reference, kmeans.rs:190-208) */
#include <pthread.h>
typedef struct {
    int kind, n_centers, len;
    const float *dataset, *centers;
    size_t lo, hi;
    uint32_t *clusters;
    float *min_dist;
} predict_job;
static void *predict_worker(void *arg) {
    predict_job *j = (predict_job *)arg;
    orc_kmeans_predict(j->kind, j->dataset, j->lo, j->hi, j->centers, j->n_centers, j->len, j->clusters, j->min_dist);
    return NULL;
}
void orc_kmeans_predict_mt(int kind, const float *dataset, size_t n, const float *centers, int n_centers, int len, uint32_t *clusters,
                           float *min_dist, int n_threads) {
    pthread_t th[256];
    predict_job jobs[256];
    int t;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    for (t = 0; t < n_threads; t++) {
        predict_job jb = {kind, n_centers, len, dataset, centers, n * (size_t)t / (size_t)n_threads, n * (size_t)(t + 1) / (size_t)n_threads, clusters,
                          min_dist};
        jobs[t] = jb;
        pthread_create(&th[t], NULL, predict_worker, &jobs[t]);
    }
    for (t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}
