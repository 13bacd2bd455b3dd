/*
 * kmeans_fit.c -- CPU restatement of the abstraction generator's k-means TRAINING loops:
 *   Kmeans::init_random (scoring)   gen_abstraction/kmeans.rs:124-156
 *   Kmeans::init_s                  gen_abstraction/kmeans.rs:267-285
 *   Kmeans::reassign_clusters       gen_abstraction/kmeans.rs:287-334   (= assignment_with_bounds :213-265, same body)
 *   Kmeans::fit_regular             gen_abstraction/kmeans.rs:497-600
 *   Kmeans::fit_growbatch           gen_abstraction/kmeans.rs:336-495   (as coded: the loop body ends in an unconditional `break`, :492)
 * with dist_func = emd_1d / l2_dist of kmeans_emd.c.
 *
 * TEST INFRASTRUCTURE ONLY (see rs_oracle.h).  PARITY UNPINNED for the training loops: the reference has no test for them; the distance
 * functions underneath ARE pinned by the reference's own known answers (emd.rs:122-180, tests/test_kmeans_cpu.py).
 *
 * Things restated exactly as coded, bugs included:
 *   - `s` is created ONCE with f32::MAX (kmeans.rs:518) and init_s only ever lowers it (`if d < *s`) before halving it again, so from the second
 *     iteration on s[i] = min(s_prev[i], min_j d(c_i, c_j)) / 2: it shrinks every round instead of being recomputed;
 *   - the scan over the other centers skips j == min_cluster where min_cluster is the CURRENT best (it moves during the scan), kmeans.rs:311;
 *   - means: `if cbm[k] > 0.0 { cbm[k] /= count }` (fit_regular, :537) and `if cs[j] > 0.0 && count > 0.0` (fit_growbatch, :412); sums and counts are
 *     f32 accumulated in DATA ORDER (:525-530, :399-406);
 *   - rayon's par_iter order does not matter anywhere: every parallel loop writes only its own element.
 * fit_growbatch shuffles the data with the caller's rng (`shuffled_data.shuffle(rng)`, :352): not reproducible, so the permutation is an INPUT here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float orc_emd_1d(const float *p_in, const float *q_in, int len);
float orc_l2_dist(const float *a, const float *b, int len);

static float dist(int kind, const float *a, const float *b, int len) { return kind == 0 ? orc_emd_1d(a, b, len) : orc_l2_dist(a, b, len); }

#define ORC_F32_MAX 3.40282347e+38f

/* kmeans.rs:267-285; s is in/out (see the header) */
void orc_kmeans_init_s(int kind, const float *centers, int k, int len, float *s) {
    int i, j;
    for (i = 0; i < k; i++) {
        for (j = 0; j < k; j++) {
            float d;
            if (i == j) continue;
            d = dist(kind, centers + (size_t)i * len, centers + (size_t)j * len, len);
            if (d < s[i]) s[i] = d;
        }
        s[i] /= 2.0f;
    }
}

/* Kmeans::init_random's scoring of the candidate center sets (kmeans.rs:124-156): cluster_dists[r] = (sum_i sum_{j != i} dist(c_i, c_j)) / count, the index of the
 * maximum is returned (Iterator::max_by keeps the LAST of equal maxima; partial_cmp(..).unwrap_or(Equal) makes a NaN "equal") */
int orc_kmeans_pick_restart(int kind, const float *centers, int n_restarts, int k, int len, float *cluster_dists) {
    int r, i, j, arg = 0;
    for (r = 0; r < n_restarts; r++) {
        const float *c = centers + (size_t)r * k * len;
        float sum = 0.0f;
        size_t count = 0;
        for (i = 0; i < k; i++) {
            float di = 0.0f;                       /* distances[i], kmeans.rs:137 */
            for (j = 0; j < k; j++) {
                if (j == i) continue;
                di += dist(kind, c + (size_t)i * len, c + (size_t)j * len, len);
                count += 1;
            }
            sum += di;
        }
        cluster_dists[r] = sum / (float)count;
    }
    for (r = 1; r < n_restarts; r++)
        if (!(cluster_dists[r] < cluster_dists[arg])) arg = r;
    return arg;
}

/* kmeans.rs:287-334 (and :213-265).  order != NULL: datum i is dataset[order[i]] (fit_growbatch's shuffled_data).  bounds = (lower, upper) pairs. */
void orc_kmeans_reassign(int kind, const float *dataset, const uint32_t *order, size_t n, const float *centers, int k, int len, const float *s,
                         uint32_t *clusters, float *bounds) {
    size_t i;
    for (i = 0; i < n; i++) {
        const float *x = dataset + (size_t)(order ? order[i] : i) * len;
        float *bi = bounds + 2 * i;
        int min_cluster = (int)clusters[i], j;
        const float upper_comp_bound = s[min_cluster] > bi[0] ? s[min_cluster] : bi[0];   /* f32::max */
        float u2, l2;
        if (bi[1] <= upper_comp_bound) continue;
        u2 = dist(kind, x, centers + (size_t)min_cluster * len, len);
        bi[1] = u2;
        if (bi[1] <= upper_comp_bound) continue;
        l2 = ORC_F32_MAX;
        for (j = 0; j < k; j++) {
            float dist2;
            if (j == min_cluster) continue;
            dist2 = dist(kind, x, centers + (size_t)j * len, len);
            if (dist2 < u2) {
                l2 = u2;
                u2 = dist2;
                min_cluster = j;
            } else if (dist2 < l2) {
                l2 = dist2;
            }
        }
        bi[0] = l2;
        if ((int)clusters[i] != min_cluster) {
            bi[1] = u2;
            clusters[i] = (uint32_t)min_cluster;
        }
    }
}

/* longest / second longest center movement, kmeans.rs:552-569 (= :427-444) */
static void two_longest(const float *mv, int k, int *longest_idx, float *longest, float *second) {
    int i;
    *longest_idx = 0;
    *longest = mv[0];
    *second = mv[1];
    if (*longest < *second) {
        *longest = mv[1];
        *second = mv[0];
        *longest_idx = 1;
    }
    for (i = 2; i < k; i++) {
        if (*longest < mv[i]) {
            *second = *longest;
            *longest = mv[i];
            *longest_idx = i;
        } else if (*second < mv[i]) {
            *second = mv[i];
        }
    }
}

/* Kmeans::fit_regular (kmeans.rs:497-600): `iterations` = 10 in the reference (:589).  centers in/out [k][len]; clusters out [n]; bounds out [n][2];
 * returns the inertia printed at the end (:594).  k >= 2 (center_movement[1] is read, :554). */
float orc_kmeans_fit_regular(int kind, const float *dataset, size_t n, float *centers, int k, int len, int iterations, uint32_t *clusters, float *bounds) {
    float *s = (float *)malloc((size_t)k * sizeof(float)), *count = (float *)malloc((size_t)k * sizeof(float));
    float *mass = (float *)malloc((size_t)k * len * sizeof(float)), *mv = (float *)malloc((size_t)k * sizeof(float));
    size_t i;
    int j, b, t = 0;
    float inertia = 0.0f;
    for (j = 0; j < k; j++) s[j] = ORC_F32_MAX;
    for (i = 0; i < n; i++) {
        clusters[i] = 0;
        bounds[2 * i] = 0.0f;
        bounds[2 * i + 1] = ORC_F32_MAX;
    }
    for (;;) {
        int longest_idx;
        float longest, second;
        orc_kmeans_init_s(kind, centers, k, len, s);
        orc_kmeans_reassign(kind, dataset, NULL, n, centers, k, len, s, clusters, bounds);
        memset(count, 0, (size_t)k * sizeof(float));
        memset(mass, 0, (size_t)k * len * sizeof(float));
        for (i = 0; i < n; i++) {   /* :525-530 */
            count[clusters[i]] += 1.0f;
            for (b = 0; b < len; b++) mass[(size_t)clusters[i] * len + b] += dataset[i * (size_t)len + b];
        }
        for (j = 0; j < k; j++)     /* :531-543 */
            for (b = 0; b < len; b++)
                if (mass[(size_t)j * len + b] > 0.0f) mass[(size_t)j * len + b] /= count[j];
        for (j = 0; j < k; j++) mv[j] = dist(kind, mass + (size_t)j * len, centers + (size_t)j * len, len);   /* :546-549 */
        two_longest(mv, k, &longest_idx, &longest, &second);
        for (i = 0; i < n; i++) {   /* :571-578 */
            bounds[2 * i + 1] += mv[clusters[i]];
            bounds[2 * i] -= ((int)clusters[i] == longest_idx) ? second : longest;
        }
        memcpy(centers, mass, (size_t)k * len * sizeof(float));   /* :586 */
        t += 1;
        if (t == iterations) break;
    }
    for (i = 0; i < n; i++) inertia += bounds[2 * i + 1];   /* :594 */
    inertia = inertia / (float)n;
    free(s);
    free(count);
    free(mass);
    free(mv);
    return inertia;
}

/* Kmeans::fit_growbatch as coded (kmeans.rs:336-495): ONE pass over the first `batch` shuffled items.  order[i] = index of shuffled_data[i];
 * centers in/out; clusters / bounds out [batch]; stats out = {min_change (p), inertia as printed (sum of upper bounds / min(n, 2 batch))} */
void orc_kmeans_fit_growbatch(int kind, const float *dataset, size_t n, const uint32_t *order, size_t batch, float *centers, int k, int len,
                              uint32_t *clusters, float *bounds, float *stats) {
    float *s = (float *)malloc((size_t)k * sizeof(float)), *count = (float *)calloc((size_t)k, sizeof(float));
    float *sums = (float *)calloc((size_t)k * len, sizeof(float)), *mv = (float *)malloc((size_t)k * sizeof(float));
    float *sq = (float *)calloc((size_t)k, sizeof(float));
    size_t i, next_batch;
    int j, b, longest_idx;
    float longest, second, min_change = 0.0f, inertia = 0.0f;
    for (j = 0; j < k; j++) s[j] = ORC_F32_MAX;
    orc_kmeans_init_s(kind, centers, k, len, s);                       /* :367 */
    for (i = 0; i < batch; i++) {                                      /* :369-372 */
        bounds[2 * i] = 0.0f;
        bounds[2 * i + 1] = ORC_F32_MAX;
        clusters[i] = 0;
    }
    orc_kmeans_reassign(kind, dataset, order, batch, centers, k, len, s, clusters, bounds);   /* :380-386 */
    for (i = 0; i < batch; i++) {                                      /* :393-400 */
        const uint32_t a = clusters[i];
        sq[a] += bounds[2 * i + 1] * bounds[2 * i + 1];   /* .powf(2.0): LLVM folds pow(x, 2.0) to x * x */
        count[a] += 1.0f;
        for (b = 0; b < len; b++) sums[(size_t)a * len + b] += dataset[(size_t)order[i] * len + b];
    }
    for (j = 0; j < k; j++)                                            /* :402-414 */
        for (b = 0; b < len; b++)
            if (sums[(size_t)j * len + b] > 0.0f && count[j] > 0.0f) sums[(size_t)j * len + b] /= count[j];
    for (j = 0; j < k; j++) mv[j] = dist(kind, sums + (size_t)j * len, centers + (size_t)j * len, len);   /* :417-420 */
    two_longest(mv, k, &longest_idx, &longest, &second);
    for (i = 0; i < batch; i++) {                                      /* :445-452 */
        bounds[2 * i + 1] += mv[clusters[i]];
        bounds[2 * i] -= ((int)clusters[i] == longest_idx) ? second : longest;
    }
    for (j = 0; j < k; j++) {                                          /* :454-471: min over i of std_dev[i] / (movement[i] + 1e-9) */
        float sd, c;
        if (count[j] <= 1.0f) sd = INFINITY;
        else sd = sqrtf(fabsf(sq[j] / (count[j] * (count[j] - 1.0f))));
        c = sd / (mv[j] + 1e-9f);
        if (j == 0 || c < min_change) min_change = c;                  /* min_by(partial_cmp): the first of equal minima */
    }
    memcpy(centers, sums, (size_t)k * len * sizeof(float));            /* :473 */
    next_batch = 2 * batch < n ? 2 * batch : n;                        /* :476 */
    for (i = 0; i < batch; i++) inertia += bounds[2 * i + 1];          /* :478 */
    inertia = inertia / (float)next_batch;
    stats[0] = min_change;
    stats[1] = inertia;
    free(s);
    free(count);
    free(sums);
    free(mv);
    free(sq);
}
