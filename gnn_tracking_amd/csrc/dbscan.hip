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

// ---- the same graph without the N^2 walk --------------------------------------------------------
// Points sorted by Morton code into chunks of 64 with bounding boxes (knn.hip: spatial_chunks_build,
// dim <= 16).  A wave owns four (two beyond 8 dimensions) consecutive sorted QUERIES (coordinates in registers); the candidate
// chunks are tested 64 at a time (lane = chunk) query point against box,
//     LB = sum over d of max(lo_d - q_d, q_d - hi_d, 0)^2     in fp32, compared with r^2 (1 + 1e-5)
// (the fp32 evaluation is within a few 1e-7 of the exact bound, which in turn is <= the fp64 d2 of
// every candidate in the box: with the margin no neighbour can be lost), and only the surviving
// (query, chunk) pairs run the graph's own arithmetic (lane = candidate): fp64 sequential sum,
// member iff d2 <= r^2.  A box-against-box test per chunk of queries does NOT prune in 8 dimensions
// (measured: 92 -> 80 ms; both boxes are wide), the per-query test does (a few per cent survive).
// Lists come out in the order of the sorted chunks; radius_order_kernel puts every list into
// ascending neighbour index (the contract of radius_fill) on its way from the staging arrays to the
// output.
constexpr int kRWaves = kRTpb / 64;
constexpr int radius_queries_per_wave(int d) { return d > 8 ? 2 : 4; }  // (register budget: fp32 + fp64 copies)

template <int D, bool FILL>
__global__ __launch_bounds__(kRTpb) void radius_pruned_kernel(const float *__restrict__ xs,
                                                              const int32_t *__restrict__ sidx,
                                                              const float *__restrict__ box, int64_t n, int n_chunks,
                                                              double r2, int32_t *__restrict__ cnt,
                                                              const int64_t *__restrict__ off,
                                                              int32_t *__restrict__ nbr, double *__restrict__ dist) {
    constexpr int kRQ = radius_queries_per_wave(D);
    const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q0 = ((int64_t)blockIdx.x * kRWaves + wv) * kRQ;  // position in the sorted order
    if (q0 >= n) return;
    const int nq = (int)(n - q0 < kRQ ? n - q0 : kRQ);
    int lane_zero;  // (per-lane copies of the query coordinates: see knn.hip)
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
#else
    lane_zero = 0;
#endif
    float qf[kRQ][D];
    double qd[kRQ][D];
    int64_t pos[kRQ];
    int32_t oq[kRQ], found[kRQ];
#pragma unroll
    for (int u = 0; u < kRQ; ++u) {
        const int64_t r = q0 + (u < nq ? u : nq - 1);
        const float *__restrict__ row = xs + r * D + lane_zero;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            qf[u][d] = row[d];
            qd[u][d] = (double)qf[u][d];
        }
        oq[u] = sidx[r];
        pos[u] = FILL ? off[oq[u]] : 0;
        found[u] = 0;
    }
    const float r2m = (float)r2 * 1.00001f + 1e-30f;
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int cc = c0 + lane < n_chunks ? c0 + lane : n_chunks - 1;
        float lo[D], hi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            lo[d] = box[(int64_t)cc * 2 * D + d];
            hi[d] = box[(int64_t)cc * 2 * D + D + d];
        }
        int qmask = 0;
#pragma unroll
        for (int u = 0; u < kRQ; ++u) {
            float lb = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float gd = fmaxf(fmaxf(lo[d] - qf[u][d], qf[u][d] - hi[d]), 0.f);
                lb = lb + gd * gd;
            }
            if (c0 + lane < n_chunks && u < nq && lb <= r2m) qmask |= 1 << u;
        }
        unsigned long long near = __ballot(qmask != 0);
        while (near != 0ull) {
            const int i = __ffsll(near) - 1;
            near &= near - 1ull;
#ifdef __HIP_DEVICE_COMPILE__
            const int qm = __builtin_amdgcn_readlane(qmask, i);
#else
            const int qm = __shfl(qmask, i);
#endif
            const int64_t p2 = (int64_t)(c0 + i) * 64 + lane;
            double xc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xc[d] = (double)xs[p2 * D + d];
            const int32_t id = sidx[p2];
#pragma unroll
            for (int u = 0; u < kRQ; ++u) {
                if (((qm >> u) & 1) == 0) continue;
                double s2 = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const double t = qd[u][d] - xc[d];
                    s2 = s2 + t * t;  // (no FMA: as the reference computes)
                }
                const bool ok = id >= 0 && s2 <= r2;
                const unsigned long long m = __ballot(ok);
                if (m == 0ull) continue;
                if (FILL) {
                    if (ok) {
                        const int64_t at = pos[u] + __popcll(m & ((1ull << lane) - 1ull));
                        nbr[at] = id;
                        dist[at] = sqrt(s2);
                    }
                    pos[u] += __popcll(m);
                } else {
                    found[u] += __popcll(m);
                }
            }
        }
    }
    if (!FILL && lane == 0) {
#pragma unroll
        for (int u = 0; u < kRQ; ++u)
            if (u < nq) cnt[oq[u]] = found[u];
    }
}

// one wave per query: its list from the staging arrays into the output, ascending neighbour index
// (rank of an entry = number of smaller ids in the list; ids are distinct)
__global__ __launch_bounds__(kRTpb) void radius_order_kernel(const int64_t *__restrict__ off, int64_t n,
                                                             const int32_t *__restrict__ nbr_in,
                                                             const double *__restrict__ dist_in,
                                                             int32_t *__restrict__ nbr, double *__restrict__ dist) {
    constexpr int kCap = 1024;  // list lengths served from LDS
    __shared__ int s_v[kRWaves][kCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * kRWaves + wv;
    if (q >= n) return;
    const int64_t o = off[q];
    const int64_t len = off[q + 1] - o;
    const bool in_lds = len <= kCap;
    if (in_lds) {
        for (int64_t i = lane; i < len; i += 64) s_v[wv][i] = nbr_in[o + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int64_t i = lane; i < len; i += 64) {
        const int v = nbr_in[o + i];
        int64_t rank = 0;
        if (in_lds) {
            for (int64_t j = 0; j < len; ++j) rank += s_v[wv][j] < v ? 1 : 0;
        } else {
            for (int64_t j = 0; j < len; ++j) rank += nbr_in[o + j] < v ? 1 : 0;
        }
        nbr[o + rank] = v;
        dist[o + rank] = dist_in[o + i];
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
    scan_counts_launch(cnt, 0x7fffffff, n, offsets, stream);
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

// ---- pruned graph: workspace and launchers ---------------------------------------------------
// ws_points: [xs | sidx | box | build scratch] - filled by the count pass, read by the fill pass;
// ws_edges (fill pass): staging of the unordered lists, m_edges * 12 bytes
struct RadiusWs {
    size_t xs, sidx, box, scratch, total;
    int n_chunks, dp;
};
static RadiusWs radius_ws_layout(int64_t n, int dim) {
    RadiusWs w{};
    w.dp = spatial_dp(dim);
    w.n_chunks = spatial_n_chunks(n);
    const size_t rows = (size_t)w.n_chunks * 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    w.xs = take(rows * w.dp * 4);
    w.sidx = take(rows * 4);
    w.box = take((size_t)w.n_chunks * 2 * w.dp * 4);
    w.scratch = take(spatial_scratch_bytes(n));
    w.total = o;
    return w;
}
constexpr int64_t kRadiusPrunedMinRows = 4096;
constexpr int64_t kRadiusPrunedMaxDegree = 256;  // denser graphs: nothing to prune, ordering would dominate

size_t radius_points_ws_bytes(int64_t n, int dim) {
    if (n < 1 || n > 0x7fffffff || dim < 1 || dim > 16) return 0;
    return radius_ws_layout(n, dim).total;
}
size_t radius_edges_ws_bytes(int64_t m_edges) {
    return align_up((size_t)(m_edges > 0 ? m_edges : 1) * 4, 256) + align_up((size_t)(m_edges > 0 ? m_edges : 1) * 8, 256);
}

int radius_count_ws_launch(const float *x, int64_t n, int dim, int stride, double radius, int32_t *cnt,
                           int64_t *offsets, void *ws_points, size_t ws_bytes, int flags, hipStream_t stream) {
    const bool pruned = ws_points && dim <= 16 && !(flags & 2) && ((flags & 1) || n >= kRadiusPrunedMinRows) && n > 0;
    if (!pruned) return radius_count_launch(x, n, dim, stride, radius, cnt, offsets, stream);
    int rc = check_points(x, n, dim, stride, radius, "radius_count");
    if (rc) return rc;
    if (!offsets || !cnt) return fail(GNNTRK_EINVAL, "radius_count: NULL argument");
    const RadiusWs w = radius_ws_layout(n, dim);
    if (ws_bytes < w.total) return fail(GNNTRK_EINVAL, "radius_count: workspace too small");
    char *b = static_cast<char *>(ws_points);
    float *xs = reinterpret_cast<float *>(b + w.xs);
    int32_t *sidx = reinterpret_cast<int32_t *>(b + w.sidx);
    float *box = reinterpret_cast<float *>(b + w.box);
    rc = spatial_chunks_build(x, n, dim, stride, nullptr, 0, xs, sidx, box, b + w.scratch, w.total - w.scratch, stream);
    if (rc) return rc;
    const double r2 = radius * radius;
    const unsigned grid = (unsigned)ceil_div(n, (int64_t)kRWaves * radius_queries_per_wave(w.dp));
#define GNNTRK_RP_COUNT(D_)                                                                                         \
    hipLaunchKernelGGL((radius_pruned_kernel<D_, false>), dim3(grid), dim3(kRTpb), 0, stream, (const float *)xs,     \
                       (const int32_t *)sidx, (const float *)box, n, w.n_chunks, r2, cnt, (const int64_t *)nullptr, \
                       (int32_t *)nullptr, (double *)nullptr)
    if (w.dp == 4) GNNTRK_RP_COUNT(4);
    else if (w.dp == 8) GNNTRK_RP_COUNT(8);
    else GNNTRK_RP_COUNT(16);
#undef GNNTRK_RP_COUNT
    scan_counts_launch(cnt, 0x7fffffff, n, offsets, stream);
    return check_launch("radius_count(pruned)");
}

int radius_fill_ws_launch(const float *x, int64_t n, int dim, int stride, double radius, const int64_t *off,
                          int64_t m_edges, int32_t *nbr, double *dist, void *ws_points, size_t ws_bytes,
                          void *ws_edges, size_t ws_edges_bytes, int flags, hipStream_t stream) {
    const bool pruned = ws_points && ws_edges && dim <= 16 && !(flags & 2) && n > 0 &&
                        ((flags & 1) || (n >= kRadiusPrunedMinRows && m_edges <= kRadiusPrunedMaxDegree * n));
    if (!pruned) return radius_fill_launch(x, n, dim, stride, radius, off, nbr, dist, stream);
    int rc = check_points(x, n, dim, stride, radius, "radius_fill");
    if (rc) return rc;
    if (!off || !nbr || !dist || m_edges < 0) return fail(GNNTRK_EINVAL, "radius_fill: bad argument");
    const RadiusWs w = radius_ws_layout(n, dim);
    if (ws_bytes < w.total || ws_edges_bytes < radius_edges_ws_bytes(m_edges))
        return fail(GNNTRK_EINVAL, "radius_fill: workspace too small");
    char *b = static_cast<char *>(ws_points);
    const float *xs = reinterpret_cast<const float *>(b + w.xs);
    const int32_t *sidx = reinterpret_cast<const int32_t *>(b + w.sidx);
    const float *box = reinterpret_cast<const float *>(b + w.box);
    int32_t *t_nbr = reinterpret_cast<int32_t *>(ws_edges);
    double *t_dist = reinterpret_cast<double *>(static_cast<char *>(ws_edges) +
                                                align_up((size_t)(m_edges > 0 ? m_edges : 1) * 4, 256));
    const double r2 = radius * radius;
    const unsigned grid = (unsigned)ceil_div(n, (int64_t)kRWaves * radius_queries_per_wave(w.dp));
#define GNNTRK_RP_FILL(D_)                                                                                     \
    hipLaunchKernelGGL((radius_pruned_kernel<D_, true>), dim3(grid), dim3(kRTpb), 0, stream, xs, sidx, box, n, \
                       w.n_chunks, r2, (int32_t *)nullptr, off, t_nbr, t_dist)
    if (w.dp == 4) GNNTRK_RP_FILL(4);
    else if (w.dp == 8) GNNTRK_RP_FILL(8);
    else GNNTRK_RP_FILL(16);
#undef GNNTRK_RP_FILL
    hipLaunchKernelGGL(radius_order_kernel, dim3((unsigned)ceil_div(n, kRWaves)), dim3(kRTpb), 0, stream, off, n,
                       (const int32_t *)t_nbr, (const double *)t_dist, nbr, dist);
    return check_launch("radius_fill(pruned)");
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
