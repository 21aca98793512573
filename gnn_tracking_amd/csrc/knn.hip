// Brute-force kNN / radius graph construction (reference: models/graph_construction.py:222-237
// = torch_cluster.knn_graph + max-radius filter; metrics/losses/oc.py:115-117 radius_graph).
//
// Exact arithmetic contract (oracle/knn_ref.c): d2(q,c) is the fmaf chain over the
// dimensions in order, neighbours are the k smallest (d2, index) pairs, ties -> lower
// index, self excluded by index; radius filter sqrtf(d2) < r (strict).  Indices are
// therefore bit-exact against the CPU oracle.
//
// One wave owns QW consecutive queries and streams ALL candidates 64 at a time
// (lane = candidate, coordinates in registers, coalesced row loads).  For every query the
// wave keeps a buffer of candidate keys (d2 bits << 32 | index) in LDS; a lane appends its
// candidate only if the key beats the query's current threshold (ballot + prefix rank); when
// a buffer cannot take another full chunk it is bitonic-sorted in LDS by the wave and cut to
// the k best, which tightens the threshold.  With a radius the threshold starts at ~r^2, so
// in a clustered embedding almost every (query, chunk) step is 2*D VALU + compare + ballot.
// Bound: fp32 VALU (N^2 * D fma), candidates stay L2 resident.
#include "host_util.h"

namespace gnntrk {

typedef unsigned long long u64;
constexpr int kKnnBlock = 256;
constexpr int kKnnWaves = 4;
constexpr int kKnnLdsPerWave = 8 * 1024;  // key buffers of one wave (bytes): 20 waves per CU
constexpr u64 kKeyMax = ~0ull;
constexpr int kKnnGroup = 4;  // queries per step (loads of a group overlap)

__device__ __forceinline__ void knn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ascending bitonic sort of cap (power of two) 64-bit keys in LDS by one wave
__device__ inline void wave_bitonic_sort(u64 *buf, int cap, int lane) {
    for (int size = 2; size <= cap; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = lane; p < (cap >> 1); p += 64) {
                const int i = ((p / stride) * (stride << 1)) + (p % stride);
                const int j = i + stride;
                const bool up = (i & size) == 0;
                const u64 a = buf[i], b = buf[j];
                if ((a > b) == up) {
                    buf[i] = b;
                    buf[j] = a;
                }
            }
            knn_wave_sync();
        }
    }
}

// segment (event) of row q: the last s with seg_ptr[s] <= q (seg_ptr ascending, seg_ptr[0] = 0)
__device__ __forceinline__ int knn_segment_of(const int64_t *__restrict__ seg_ptr, int n_seg, int64_t q) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_ptr[mid] <= q) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// FULL: dim == DP (no per-feature guard on the scalar loads).  BATCH: rows are grouped into
// segments (the events of a collated batch, `batch` of torch_cluster's knn_graph / radius_graph:
// metrics/losses/metric_learning.py:97); a query only sees candidates of its own segment.
// QW > 0: the wave's QW queries are compile-time unrolled and their coordinates are loaded ONCE
// (QW * DP <= 64 registers held for the whole candidate stream); QW = 0: the
// query count is a run-time value and the coordinates are re-read (constant cache) per chunk.
template <int DP, bool FULL, bool BATCH, int QW>
__global__ __launch_bounds__(kKnnBlock) void knn_kernel(const float *__restrict__ x, int64_t n,
                                                        int dim, int stride, int k, int cap,
                                                        int qw, float max_radius,
                                                        const int64_t *__restrict__ seg_ptr, int n_seg,
                                                        int32_t *__restrict__ nbr,
                                                        int32_t *__restrict__ cnt_out) {
    __shared__ __attribute__((aligned(16))) u64 s_keys[kKnnWaves][kKnnLdsPerWave / 8];
    __shared__ u64 s_tau[kKnnWaves][32];
    __shared__ int s_cnt[kKnnWaves][32];
    __shared__ int s_lo[kKnnWaves][32], s_hi[kKnnWaves][32];  // candidate range of each query (BATCH)
    // the wave index through readfirstlane: q0 and everything derived from it are scalars, so
    // the query coordinates below are SCALAR loads (constant cache) feeding the VALU as SGPR
    // operands - no LDS traffic for them in the (query, chunk) step
    const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u64 *keys = s_keys[wv];
    u64 *tau = s_tau[wv];
    int *cnt = s_cnt[wv];
    const int64_t q0 = ((int64_t)blockIdx.x * kKnnWaves + wv) * qw;
    if (q0 >= n) return;
    const int nq = (int)((n - q0 < qw) ? (n - q0) : qw);

    // threshold: with a radius only candidates that can pass sqrtf(d2) < r are buffered
    u64 tau0 = kKeyMax;
    if (max_radius > 0.f) {
        const float r2 = max_radius * max_radius * 1.000001f + 1e-30f;
        tau0 = ((u64)__float_as_uint(r2) << 32) | 0xffffffffull;
    }
    if (lane < 32) {
        tau[lane] = tau0;
        cnt[lane] = 0;
    }
    // candidate range of the wave = the segments its queries live in
    int64_t c_begin = 0, c_end = n;
    int *qlo = s_lo[wv], *qhi = s_hi[wv];
    if (BATCH) {
        if (lane < 32) {
            const int64_t q = q0 + (lane < nq ? lane : nq - 1);
            const int sg = knn_segment_of(seg_ptr, n_seg, q);
            qlo[lane] = (int)seg_ptr[sg];
            qhi[lane] = (int)seg_ptr[sg + 1];
        }
        knn_wave_sync();
        c_begin = qlo[0];
        c_end = qhi[nq - 1];
    }
    knn_wave_sync();

    // candidate rows: every lane loads unconditionally (row index clamped, `d < dim` is a
    // uniform condition) - a load inside a divergent branch forces s_waitcnt vmcnt(0) on each
    // of them - and the NEXT chunk is fetched while the queries run over the current one
    auto load_chunk = [&](int64_t c0, float (&v)[DP]) {
        const int64_t j = c0 + lane;
        const float *__restrict__ row = x + (j < n ? j : n - 1) * stride;
#pragma unroll
        for (int d = 0; d < DP; ++d) v[d] = (FULL || d < dim) ? row[d] : 0.f;
    };
    // (QW > 0) query coordinates, loaded once: wave-uniform addresses -> scalar registers
    constexpr int kQ = QW > 0 ? QW : 1;
    float qv[kQ][DP];
    if (QW > 0) {
        // (through a lane offset the optimiser cannot see through: as uniform values the 64
        // coordinates would be scalar registers, more than a wave has - they would be spilled and
        // fetched back with v_readlane in the inner loop; as VGPRs they cost occupancy, not time)
        int lane_zero;
#ifdef __HIP_DEVICE_COMPILE__
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
#else
        lane_zero = 0;
#endif
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const float *__restrict__ xq = x + (q0 + (q < nq ? q : nq - 1)) * stride + lane_zero;
#pragma unroll
            for (int d = 0; d < DP; ++d) qv[q][d] = (FULL || d < dim) ? xq[d] : 0.f;
        }
    }
    // (QW > 0) the distance part of every query's threshold, in registers too: the common step
    // then touches neither LDS nor memory; refreshed after the (rare) sorts
    uint32_t tau_hi[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q) tau_hi[q] = (uint32_t)(tau0 >> 32);
    float xc[DP], xn[DP];
    load_chunk(c_begin, xc);
    for (int64_t c0 = c_begin; c0 < c_end; c0 += 64) {
        const int64_t j = c0 + lane;
        load_chunk(c0 + 64 < c_end ? c0 + 64 : c0, xn);
        // queries in groups of kKnnGroup: the scalar loads of the coordinates and the LDS
        // reads of the thresholds of the whole group are in flight together (a scalar load
        // can only be waited for with lgkmcnt(0): one query per step means one full memory
        // latency per step), then the distances, then the (rare) appends
        const int q_stop = QW > 0 ? QW : nq;
#pragma unroll
        for (int qb = 0; qb < q_stop; qb += kKnnGroup) {
            u64 tq[kKnnGroup];
            float d2q[kKnnGroup];
            if (QW == 0) {
#pragma unroll
                for (int u = 0; u < kKnnGroup; ++u) tq[u] = tau[qb + u < nq ? qb + u : nq - 1];
            }
#pragma unroll
            for (int u = 0; u < kKnnGroup; ++u) {
                const int q = qb + u < nq ? qb + u : nq - 1;
                const float *__restrict__ xq = x + (q0 + q) * stride;  // wave-uniform address
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const float qd = QW > 0 ? qv[(qb + u) < kQ ? (qb + u) : kQ - 1][d] : ((FULL || d < dim) ? xq[d] : 0.f);
                    const float t = __fsub_rn(qd, xc[d]);
                    d2 = __fmaf_rn(t, t, d2);
                }
                d2q[u] = d2;
            }
#pragma unroll
            for (int u = 0; u < kKnnGroup; ++u) {
                // cheap prefilter on the distance bits alone (d2 >= 0: its bit pattern orders like the
                // value); the exact (d2, index) key, the self / range / event exclusions and the append
                // only run for the rare (query, chunk) steps in which some lane gets past it
                const uint32_t th = QW > 0 ? tau_hi[(qb + u) < kQ ? (qb + u) : kQ - 1] : (uint32_t)(tq[u] >> 32);
                if (__ballot(__float_as_uint(d2q[u]) <= th) == 0ull) continue;
                const int q = qb + u < nq ? qb + u : nq - 1;
                if (QW > 0) tq[u] = tau[q];
                u64 key = ((u64)__float_as_uint(d2q[u]) << 32) | (u64)(uint32_t)j;
                if (j >= c_end || j == q0 + q || qb + u >= nq) key = kKeyMax;
                if (BATCH && (j < qlo[q] || j >= qhi[q])) key = kKeyMax;  // another event's hit
                const bool pass = key < tq[u];
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {
                    const int base = cnt[q];
                    if (pass) keys[q * cap + base + __popcll(mask & ((1ull << lane) - 1ull))] = key;
                    int nc = base + __popcll(mask);
                    knn_wave_sync();
                    if (nc > cap - 64) {  // no room for another full chunk: keep the k best
                        u64 *b = keys + q * cap;
                        for (int i = nc + lane; i < cap; i += 64) b[i] = kKeyMax;
                        knn_wave_sync();
                        wave_bitonic_sort(b, cap, lane);
                        nc = k;
                        if (QW > 0) tau_hi[(qb + u) < kQ ? (qb + u) : kQ - 1] = (uint32_t)(b[k - 1] >> 32);
                        if (lane == 0) tau[q] = b[k - 1];
                    }
                    if (lane == 0) cnt[q] = nc;
                    knn_wave_sync();
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DP; ++d) xc[d] = xn[d];
    }

    // final: sort every buffer, apply the radius filter (a prefix: keys ascend), emit
    for (int q = 0; q < nq; ++q) {
        u64 *b = keys + q * cap;
        const int nc = cnt[q];
        for (int i = nc + lane; i < cap; i += 64) b[i] = kKeyMax;
        knn_wave_sync();
        wave_bitonic_sort(b, cap, lane);
        const int m = nc < k ? nc : k;
        int out = 0;
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            bool ok = false;
            u64 key = kKeyMax;
            if (i < m) {
                key = b[i];
                ok = key != kKeyMax;
                if (ok && max_radius > 0.f)
                    ok = __fsqrt_rn(__uint_as_float((uint32_t)(key >> 32))) < max_radius;
            }
            if (ok) nbr[(q0 + q) * k + i] = (int32_t)(uint32_t)(key & 0xffffffffull);
            out += __popcll(__ballot(ok));
        }
        if (lane == 0) cnt_out[q0 + q] = out;
        knn_wave_sync();
    }
}

// edge_index[0][off[q]+i] = nbr[q][i] (neighbour = source j), edge_index[1][..] = q (target i)
__global__ __launch_bounds__(256) void knn_emit_kernel(const int32_t *__restrict__ nbr,
                                                       const int32_t *__restrict__ cnt,
                                                       const int64_t *__restrict__ off, int64_t n,
                                                       int k_stride, int k, int64_t m_total,
                                                       int64_t *__restrict__ ei) {
    // the first min(cnt[q], k) neighbours of rows that are k_stride wide
    const int64_t total = n * k;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * 256) {
        const int64_t q = t / k;
        const int i = (int)(t - q * k);
        if (i < cnt[q]) {
            const int64_t o = off[q] + i;
            ei[o] = nbr[q * k_stride + i];
            ei[m_total + o] = q;
        }
    }
}

// serial-per-block inclusive scan is plenty for n <= a few million counts (HBM-trivial)
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t *__restrict__ cnt_raw, int k_take,
                                                           int64_t n, int64_t *__restrict__ off) {
    auto cnt = [&](int64_t i) { const int32_t c = cnt_raw[i]; return c < k_take ? c : k_take; };
    __shared__ long long s_part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = t * per, e = (b + per < n) ? b + per : n;
    long long s = 0;
    for (int64_t i = b; i < e; ++i) s += cnt(i);
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
        run += cnt(i);
    }
}

// MLGraphConstruction.forward (models/graph_construction.py:365-367, :386-393)
__global__ __launch_bounds__(256) void edge_features_kernel(const float *__restrict__ x, int dim,
                                                            int stride,
                                                            const int64_t *__restrict__ ei,
                                                            int64_t m, float *__restrict__ out) {
    const int64_t total = m * dim;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * 256) {
        const int64_t e = t / dim;
        const int f = (int)(t - e * dim);
        const float a = x[ei[e] * stride + f], b = x[ei[m + e] * stride + f];
        out[e * 2 * dim + f] = a - b;
        out[e * 2 * dim + dim + f] = a + b;
    }
}
__global__ __launch_bounds__(256) void edge_labels_kernel(const int64_t *__restrict__ pid,
                                                          const int64_t *__restrict__ ei, int64_t m,
                                                          int64_t *__restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < m;
         e += (int64_t)gridDim.x * 256) {
        const int64_t a = pid[ei[e]], b = pid[ei[m + e]];
        y[e] = (a == b && a > 0) ? 1 : 0;
    }
}

static int stream_grid(int64_t n) {
    int64_t g = ceil_div(n, 256);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

int knn_search_launch(const float *x, int64_t n, int dim, int stride, int k, float max_radius,
                      const int64_t *seg_ptr, int n_seg, int32_t *nbr, int32_t *cnt, hipStream_t stream) {
    if (!x || !nbr || !cnt || n < 0 || dim < 1 || stride < dim || k < 1)
        return fail(GNNTRK_EINVAL, "knn_search: bad argument");
    if (seg_ptr && n_seg < 1) return fail(GNNTRK_EINVAL, "knn_search: seg_ptr needs n_seg >= 1");
    if (dim > 32) return fail(GNNTRK_EUNSUPPORTED, "knn_search: dim > 32 not supported");
    if (k > 448) return fail(GNNTRK_EUNSUPPORTED, "knn_search: k > 448 not supported");
    if (n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "knn_search: n must fit int32");
    if (n == 0) return GNNTRK_OK;
    int cap = 128;
    while (cap < k + 64) cap <<= 1;
    int qw = kKnnLdsPerWave / (cap * 8);
    if (qw > 32) qw = 32;
    const int64_t grid = ceil_div(n, (int64_t)qw * kKnnWaves);
#define KNN_LAUNCH_Q(DP, FULL_, BATCH_, QW_)                                                        \
    hipLaunchKernelGGL((knn_kernel<DP, FULL_, BATCH_, QW_>), dim3((unsigned)grid), dim3(kKnnBlock), 0, stream, x, n, \
                       dim, stride, k, cap, qw, max_radius, seg_ptr, n_seg, nbr, cnt)
#define KNN_LAUNCH(DP, FULL_, BATCH_)                                                               \
    if (DP <= 8 && qw == 8) KNN_LAUNCH_Q(DP, FULL_, BATCH_, (DP <= 8 ? 8 : 0));                     \
    else if (DP <= 8 && qw == 4) KNN_LAUNCH_Q(DP, FULL_, BATCH_, (DP <= 8 ? 4 : 0));                \
    else KNN_LAUNCH_Q(DP, FULL_, BATCH_, 0)
#define KNN_CALL(DP)                                                                               \
    if (seg_ptr) {                                                                                 \
        if (dim == DP) KNN_LAUNCH(DP, true, true); else KNN_LAUNCH(DP, false, true);               \
    } else {                                                                                       \
        if (dim == DP) KNN_LAUNCH(DP, true, false); else KNN_LAUNCH(DP, false, false);             \
    }
    if (dim <= 4) {
        KNN_CALL(4);
    } else if (dim <= 8) {
        KNN_CALL(8);
    } else if (dim <= 16) {
        KNN_CALL(16);
    } else {
        KNN_CALL(32);
    }
#undef KNN_CALL
#undef KNN_LAUNCH
#undef KNN_LAUNCH_Q
    return check_launch("knn_search");
}

int knn_emit_launch(const int32_t *nbr, const int32_t *cnt, int64_t n, int k_stride, int k, int64_t *offsets,
                    int64_t *edge_index, int64_t m_total, hipStream_t stream) {
    if (!nbr || !cnt || !offsets || n < 0 || k < 1 || k > k_stride)
        return fail(GNNTRK_EINVAL, "knn_emit: bad argument");
    if (n == 0) return GNNTRK_OK;
    if (!edge_index) {  // phase 1: offsets only (offsets[n] = total edge count)
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, cnt, k, n, offsets);
        return check_launch("knn_emit(scan)");
    }
    if (m_total > 0)
        hipLaunchKernelGGL(knn_emit_kernel, dim3(stream_grid(n * k)), dim3(256), 0, stream, nbr, cnt,
                           (const int64_t *)offsets, n, k_stride, k, m_total, edge_index);
    return check_launch("knn_emit");
}

int edge_features_launch(const float *x, int dim, int stride, const int64_t *ei, int64_t m,
                         float *out, hipStream_t stream) {
    if (!x || dim < 1 || stride < dim || m < 0) return fail(GNNTRK_EINVAL, "edge_features: bad argument");
    if (m == 0) return GNNTRK_OK;
    if (!ei || !out) return fail(GNNTRK_EINVAL, "edge_features: NULL pointer");
    hipLaunchKernelGGL(edge_features_kernel, dim3(stream_grid(m * dim)), dim3(256), 0, stream, x, dim,
                       stride, ei, m, out);
    return check_launch("edge_features");
}

int edge_labels_launch(const int64_t *pid, const int64_t *ei, int64_t m, int64_t *y,
                       hipStream_t stream) {
    if (m < 0) return fail(GNNTRK_EINVAL, "edge_labels: bad argument");
    if (m == 0) return GNNTRK_OK;
    if (!pid || !ei || !y) return fail(GNNTRK_EINVAL, "edge_labels: NULL pointer");
    hipLaunchKernelGGL(edge_labels_kernel, dim3(stream_grid(m)), dim3(256), 0, stream, pid, ei, m, y);
    return check_launch("edge_labels");
}

}  // namespace gnntrk
