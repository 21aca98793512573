// DBSCAN post-processing on the device: the radius-neighbourhood graph
// of postprocessing/fastrescanner.py:25-39 (sklearn NearestNeighbors.radius_neighbors) and
// the clustering of :41-66 (sklearn's dbscan_inner) for any (eps, min_pts) <= the graph radius.
//
// Arithmetic contract = sklearn's kd-tree path (what NearestNeighbors picks for <= 15
// features): coordinates widened to fp64, squared distance as a sequential sum over the
// features (no FMA), membership d2 <= r*r, stored distance sqrt(d2); the rescan keeps edges
// with dist <= eps.  Neighbourhoods include the point itself, so min_pts counts it.
//
// dbscan_inner's result does not depend on its traversal order: clusters are the connected
// components of the core points under the eps graph, numbered by their lowest core index;
// a border point takes the lowest-numbered cluster among its core neighbours; everything
// else is noise (-1).  That is what the kernels compute: core flags -> min-label propagation
// with pointer jumping (monotone, so races are benign and the fixpoint is unique) -> root
// compaction -> labels.
#include "host_util.h"

namespace gnntrk {

constexpr int kRTpb = 256;  // one query per thread, candidate tiles of 256 points in LDS

// d2 of query q (registers) against candidate j of the LDS tile [D][256]
template <int D>
__device__ __forceinline__ double dist2(const double (&q)[D], const double *tile, int j) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double t = q[d] - tile[d * kRTpb + j];
        s = s + t * t;  // (-ffp-contract=off: separate multiply and add, as the reference computes)
    }
    return s;
}

// FILL = false: cnt[q] = |{j : d2(q, j) <= r2}|;  FILL = true: the lists themselves, ascending j
template <int D, bool FILL>
__global__ __launch_bounds__(kRTpb) void radius_kernel(const float *__restrict__ x, int64_t n, int dim, int stride,
                                                       double r2, int32_t *__restrict__ cnt,
                                                       const int64_t *__restrict__ off, int32_t *__restrict__ nbr,
                                                       double *__restrict__ dist) {
    __shared__ double s_tile[D * kRTpb];
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * kRTpb + tid;
    const int64_t qc = q < n ? q : n - 1;
    double xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xq[d] = d < dim ? (double)x[qc * stride + d] : 0.0;  // padding adds exactly 0
    int64_t pos = FILL ? off[qc] : 0;
    int32_t c = 0;
    for (int64_t j0 = 0; j0 < n; j0 += kRTpb) {
        __syncthreads();
        {
            const int64_t j = j0 + tid;
            const int64_t jc = j < n ? j : n - 1;
#pragma unroll
            for (int d = 0; d < D; ++d) s_tile[d * kRTpb + tid] = d < dim ? (double)x[jc * stride + d] : 0.0;
        }
        __syncthreads();
        const int lim = (int)(n - j0 < kRTpb ? n - j0 : kRTpb);
        for (int j = 0; j < lim; ++j) {
            const double d2 = dist2<D>(xq, s_tile, j);
            if (d2 <= r2) {
                if (FILL) {
                    if (q < n) {
                        nbr[pos] = (int32_t)(j0 + j);
                        dist[pos] = sqrt(d2);
                    }
                    ++pos;
                } else {
                    ++c;
                }
            }
        }
    }
    if (!FILL && q < n) cnt[q] = c;
}

// offsets[0..n] = exclusive scan of cnt (one workgroup)
__global__ __launch_bounds__(1024) void scan_i32_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                        int64_t *__restrict__ off) {
    __shared__ long long s_part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = t * per, e = (b + per < n) ? b + per : n;
    long long s = 0;
    for (int64_t i = b; i < e; ++i) s += cnt[i];
    s_part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) {
            const long long v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        off[n] = run;
    }
    __syncthreads();
    long long run = s_part[t];
    for (int64_t i = b; i < e; ++i) {
        off[i] = run;
        run += cnt[i];
    }
}

// core[i] = |{e in list(i): dist[e] <= eps}| >= min_pts;  root[i] = i for core points, else -1
__global__ __launch_bounds__(256) void dbscan_init_kernel(const int64_t *__restrict__ off,
                                                          const double *__restrict__ dist, int64_t n, double eps,
                                                          int min_pts, uint8_t *__restrict__ core,
                                                          int32_t *__restrict__ root) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int32_t c = 0;
        for (int64_t e = off[i]; e < off[i + 1]; ++e) c += dist[e] <= eps ? 1 : 0;
        const bool is_core = c >= min_pts;
        core[i] = is_core ? 1 : 0;
        root[i] = is_core ? (int32_t)i : -1;
    }
}

// one round: root[i] <- min over core eps-neighbours of root[.], then pointer jumping.
// root[.] only ever decreases and always names a core point of the same component, so
// stale reads are harmless; *changed is set when anything moved.
__global__ __launch_bounds__(256) void dbscan_propagate_kernel(const int64_t *__restrict__ off,
                                                               const int32_t *__restrict__ nbr,
                                                               const double *__restrict__ dist, int64_t n,
                                                               double eps, const uint8_t *__restrict__ core,
                                                               int32_t *root, int32_t *__restrict__ changed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!core[i]) continue;
        const int32_t old = root[i];
        int32_t m = old;
        for (int64_t e = off[i]; e < off[i + 1]; ++e) {
            const int32_t j = nbr[e];
            if (dist[e] <= eps && core[j]) {
                const int32_t r = root[j];
                m = r < m ? r : m;
            }
        }
        for (int hop = 0; hop < 32; ++hop) {  // follow the chain towards its current end
            const int32_t r = root[m];
            if (r >= m) break;
            m = r;
        }
        if (m < old) {
            root[i] = m;
            *changed = 1;
        }
    }
}

__global__ __launch_bounds__(256) void dbscan_roots_kernel(const uint8_t *__restrict__ core,
                                                           const int32_t *__restrict__ root, int64_t n,
                                                           uint8_t *__restrict__ is_root) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        is_root[i] = (core[i] && root[i] == (int32_t)i) ? 1 : 0;
}

// cluster number = rank of the component's root among the roots (ascending index);
// border point = lowest cluster number among its core neighbours; otherwise -1
__global__ __launch_bounds__(256) void dbscan_labels_kernel(const int64_t *__restrict__ off,
                                                            const int32_t *__restrict__ nbr,
                                                            const double *__restrict__ dist, int64_t n,
                                                            double eps, const uint8_t *__restrict__ core,
                                                            const int32_t *__restrict__ root,
                                                            const int32_t *__restrict__ rank,
                                                            int64_t *__restrict__ labels) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int64_t lab = -1;
        if (core[i]) {
            lab = rank[root[i]];
        } else {
            for (int64_t e = off[i]; e < off[i + 1]; ++e) {
                const int32_t j = nbr[e];
                if (dist[e] <= eps && core[j]) {
                    const int64_t l = rank[root[j]];
                    lab = (lab < 0 || l < lab) ? l : lab;
                }
            }
        }
        labels[i] = lab;
    }
}

static int node_blocks(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, 256);
    const int64_t cap = (int64_t)cu_count() * 16;
    return (int)(g > cap ? cap : g);
}

static int check_points(const float *x, int64_t n, int dim, int stride, double radius, const char *who) {
    if (n < 0 || n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "radius graph: n must fit int32");
    if (dim < 1 || dim > 32) return fail(GNNTRK_EUNSUPPORTED, "radius graph: 1 <= dim <= 32");
    if (n > 0 && (!x || stride < dim)) return fail(GNNTRK_EINVAL, "radius graph: bad points");
    if (!(radius >= 0.0)) return fail(GNNTRK_EINVAL, "radius graph: radius must be >= 0");
    (void)who;
    return GNNTRK_OK;
}

#define GNNTRK_RADIUS_CALL(D_, FILL_)                                                                     \
    hipLaunchKernelGGL((radius_kernel<D_, FILL_>), dim3((unsigned)ceil_div(n, kRTpb)), dim3(kRTpb), 0, \
                       stream, x, n, dim, stride, r2, cnt, off, nbr, dist)
#define GNNTRK_RADIUS_DISPATCH(FILL_)        \
    if (dim <= 2) GNNTRK_RADIUS_CALL(2, FILL_);       \
    else if (dim <= 4) GNNTRK_RADIUS_CALL(4, FILL_);  \
    else if (dim <= 8) GNNTRK_RADIUS_CALL(8, FILL_);  \
    else if (dim <= 16) GNNTRK_RADIUS_CALL(16, FILL_); \
    else GNNTRK_RADIUS_CALL(32, FILL_)

int radius_count_launch(const float *x, int64_t n, int dim, int stride, double radius, int32_t *cnt,
                        int64_t *offsets, hipStream_t stream) {
    int rc = check_points(x, n, dim, stride, radius, "radius_count");
    if (rc) return rc;
    if (!offsets) return fail(GNNTRK_EINVAL, "radius_count: NULL offsets");
    if (n == 0) return check_hip(hipMemsetAsync(offsets, 0, sizeof(int64_t), stream), "radius_count");
    if (!cnt) return fail(GNNTRK_EINVAL, "radius_count: NULL counts");
    const double r2 = radius * radius;
    const int64_t *off = nullptr;
    int32_t *nbr = nullptr;
    double *dist = nullptr;
    GNNTRK_RADIUS_DISPATCH(false);
    hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, stream, cnt, n, offsets);
    return check_launch("radius_count");
}

int radius_fill_launch(const float *x, int64_t n, int dim, int stride, double radius, const int64_t *off,
                       int32_t *nbr, double *dist, hipStream_t stream) {
    int rc = check_points(x, n, dim, stride, radius, "radius_fill");
    if (rc || n == 0) return rc;
    if (!off || !nbr || !dist) return fail(GNNTRK_EINVAL, "radius_fill: NULL argument");
    const double r2 = radius * radius;
    int32_t *cnt = nullptr;
    GNNTRK_RADIUS_DISPATCH(true);
    return check_launch("radius_fill");
}

int dbscan_init_launch(const int64_t *off, const double *dist, int64_t n, double eps, int min_pts, uint8_t *core,
                       int32_t *root, hipStream_t stream) {
    if (n < 0 || n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "dbscan: n must fit int32");
    if (n == 0) return GNNTRK_OK;
    if (!off || !core || !root) return fail(GNNTRK_EINVAL, "dbscan_init: NULL argument");
    hipLaunchKernelGGL(dbscan_init_kernel, dim3(node_blocks(n)), dim3(256), 0, stream, off, dist, n, eps, min_pts,
                       core, root);
    return check_launch("dbscan_init");
}

int dbscan_propagate_launch(const int64_t *off, const int32_t *nbr, const double *dist, int64_t n, double eps,
                            const uint8_t *core, int32_t *root, int rounds, int32_t *changed,
                            hipStream_t stream) {
    if (!changed) return fail(GNNTRK_EINVAL, "dbscan_propagate: NULL flag");
    if (n == 0) return check_hip(hipMemsetAsync(changed, 0, sizeof(int32_t), stream), "dbscan_propagate");
    if (!off || !core || !root || rounds < 1) return fail(GNNTRK_EINVAL, "dbscan_propagate: bad argument");
    for (int r = 0; r < rounds; ++r) {
        // the flag reports the LAST round only: zero means the fixpoint has been reached
        int rc = check_hip(hipMemsetAsync(changed, 0, sizeof(int32_t), stream), "dbscan_propagate(memset)");
        if (rc) return rc;
        hipLaunchKernelGGL(dbscan_propagate_kernel, dim3(node_blocks(n)), dim3(256), 0, stream, off, nbr, dist, n,
                           eps, core, root, changed);
    }
    return check_launch("dbscan_propagate");
}

size_t dbscan_ws_bytes(int64_t n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return align_up(nn, 256) /* is_root */ + align_up(nn * 4, 256) /* root list */ + align_up(nn * 4, 256) /* rank */ +
           compact_ws_bytes(n);
}

int dbscan_labels_launch(const int64_t *off, const int32_t *nbr, const double *dist, int64_t n, double eps,
                         const uint8_t *core, const int32_t *root, int64_t *labels, int64_t *n_clusters, void *ws,
                         size_t ws_bytes, hipStream_t stream) {
    if (!n_clusters) return fail(GNNTRK_EINVAL, "dbscan_labels: NULL count output");
    if (n == 0) return check_hip(hipMemsetAsync(n_clusters, 0, sizeof(int64_t), stream), "dbscan_labels");
    if (!off || !core || !root || !labels) return fail(GNNTRK_EINVAL, "dbscan_labels: NULL argument");
    if (!ws || ws_bytes < dbscan_ws_bytes(n)) return fail(GNNTRK_EINVAL, "dbscan_labels: workspace too small");
    char *p = reinterpret_cast<char *>(ws);
    uint8_t *is_root = reinterpret_cast<uint8_t *>(p);
    p += align_up((size_t)n, 256);
    int32_t *root_list = reinterpret_cast<int32_t *>(p);
    p += align_up((size_t)n * 4, 256);
    int32_t *rank = reinterpret_cast<int32_t *>(p);
    p += align_up((size_t)n * 4, 256);
    hipLaunchKernelGGL(dbscan_roots_kernel, dim3(node_blocks(n)), dim3(256), 0, stream, core, root, n, is_root);
    int rc = compact_bytes_launch(is_root, n, root_list, rank, n_clusters, p, compact_ws_bytes(n), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(dbscan_labels_kernel, dim3(node_blocks(n)), dim3(256), 0, stream, off, nbr, dist, n, eps,
                       core, root, rank, labels);
    return check_launch("dbscan_labels");
}

}  // namespace gnntrk
