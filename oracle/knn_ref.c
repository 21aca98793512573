/*
 * knn_ref.c - ORACLE (test infrastructure only): the bit-exact CPU statement of the kNN /
 * radius-graph semantics the HIP kernels implement.
 *
 * Follows the reference call sites models/graph_construction.py:222-237
 * (knn_with_max_radius = torch_cluster.knn_graph(x, k) then keep ||x_j - x_i|| < r) and
 * metrics/losses/oc.py:115-117 (radius_graph).  torch_cluster itself is not in
 * /root/reference (un-vendored, unpinned dependency); its documented
 * semantics are restated here with every rounding fixed:
 *   d2(q,c) = fma chain over dimensions in order: t = x[q][d] - x[c][d]; d2 = fmaf(t,t,d2)
 *   neighbours of q: the k smallest (d2, c) pairs, c != q, lexicographic (ties -> lower
 *   index), ascending; radius filter: sqrtf(d2) < r (strict), applied to that list.
 * Output: edges [neighbour, query] grouped by query ascending.
 *
 * Built by oracle/build_oracle.py into oracle/_build/libknn_ref.so; compiled with
 * -ffp-contract=off so the only fused operations are the explicit fmaf calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d2; int32_t idx; } ent_t;

static int ent_less(const ent_t *a, const ent_t *b) {
    return a->d2 < b->d2 || (a->d2 == b->d2 && a->idx < b->idx);
}

/* nbr[q*k .. q*k+cnt[q]) neighbour ids ascending; returns total edge count */
int64_t knn_ref_search(const float *x, int64_t n, int32_t dim, int32_t k, float max_radius,
                       int32_t *nbr, float *nbr_d2, int32_t *cnt) {
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int64_t q = 0; q < n; ++q) {
        ent_t *best = (ent_t *)malloc(sizeof(ent_t) * (size_t)(k + 1));
        int m = 0;
        for (int64_t c = 0; c < n; ++c) {
            if (c == q) continue;
            float d2 = 0.f;
            for (int d = 0; d < dim; ++d) {
                const float t = x[q * dim + d] - x[c * dim + d];
                d2 = fmaf(t, t, d2);
            }
            ent_t e = {d2, (int32_t)c};
            if (m == k && !ent_less(&e, &best[k - 1])) continue;
            int pos = m < k ? m : k - 1; /* insertion sort into the (<= k)-list */
            while (pos > 0 && ent_less(&e, &best[pos - 1])) {
                if (pos < k) best[pos] = best[pos - 1];
                --pos;
            }
            best[pos] = e;
            if (m < k) ++m;
        }
        int out = 0;
        for (int i = 0; i < m; ++i) {
            if (max_radius > 0.f && !(sqrtf(best[i].d2) < max_radius)) continue;
            nbr[q * k + out] = best[i].idx;
            if (nbr_d2) nbr_d2[q * k + out] = best[i].d2;
            ++out;
        }
        cnt[q] = out;
        total += out;
        free(best);
    }
    return total;
}
