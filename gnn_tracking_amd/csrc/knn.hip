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

// value of lane `src` (wave-uniform index)
__device__ __forceinline__ int knn_read_lane(int v, int src) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_readlane(v, src);
#else
    return __shfl(v, src);
#endif
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

// ------------------------------------------------------------------------------------------
// Pruned search: the SAME result as knn_kernel (bit for bit), without visiting most candidates.
//
// The points are sorted by (event, Morton code of the quantised coordinates) and cut into chunks
// of 64 consecutive sorted points, each with its axis-aligned bounding box.  A wave owns QW
// consecutive SORTED queries (their coordinates in registers, as above).  Candidate chunks are
// taken 64 at a time (lane = chunk): for every query the lower bound
//     LB = fmaf chain over the dimensions of  g_d = max(lo_d - q_d, q_d - hi_d, 0)
// is compared with the query's current threshold; a chunk is only streamed for the queries whose
// bound passes.  LB <= d2(q, c) holds for every candidate c of the chunk IN THE KERNEL'S OWN
// ARITHMETIC: fp32 subtraction and fma round monotonically, |fl(q_d - c_d)| >= g_d for
// lo_d <= c_d <= hi_d, so by induction over the chain LB_d <= d2_d.  A skipped (query, chunk)
// pair is therefore one in which every candidate fails the distance prefilter of the brute-force
// step - nothing that could have been appended is lost, the k smallest (d2, index) keys are
// the same.  Batches of chunks are visited outwards from the wave's own position in the sorted
// order and every query's buffer is cut to its k best after a batch that added to it, so the
// thresholds tighten early (without a radius they start at infinity).
constexpr int kKnnBoxParts = 64;
constexpr int kSpMaxDim = 16;  // dimensions the sorted-chunk structure covers

// per-dimension min / max of all points: partials of up to kKnnBoxParts blocks (no atomics)
__global__ __launch_bounds__(256) void knn_bbox_partial_kernel(const float *__restrict__ x, int64_t n, int dim,
                                                               int stride, float *__restrict__ part) {
    __shared__ float s_lo[4][kSpMaxDim], s_hi[4][kSpMaxDim];
    float lo[kSpMaxDim], hi[kSpMaxDim];
#pragma unroll
    for (int d = 0; d < kSpMaxDim; ++d) {
        lo[d] = 3.402823466e38f;
        hi[d] = -3.402823466e38f;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int d = 0; d < kSpMaxDim; ++d)
            if (d < dim) {
                const float v = x[i * stride + d];
                lo[d] = fminf(lo[d], v);
                hi[d] = fmaxf(hi[d], v);
            }
    }
#pragma unroll
    for (int d = 0; d < kSpMaxDim; ++d)
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < kSpMaxDim; ++d) {
            s_lo[wv][d] = lo[d];
            s_hi[wv][d] = hi[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < kSpMaxDim) {
        const int d = threadIdx.x;
        part[blockIdx.x * 2 * kSpMaxDim + d] = fminf(fminf(s_lo[0][d], s_lo[1][d]), fminf(s_lo[2][d], s_lo[3][d]));
        part[blockIdx.x * 2 * kSpMaxDim + kSpMaxDim + d] = fmaxf(fmaxf(s_hi[0][d], s_hi[1][d]), fmaxf(s_hi[2][d], s_hi[3][d]));
    }
}

// sort key of every point: [event | Morton code of the coordinates quantised inside the bounding
// box] (the ordering only steers the pruning - any order gives the same neighbours)
__global__ __launch_bounds__(256) void knn_morton_kernel(const float *__restrict__ x, int64_t n, int dim,
                                                         int stride, const float *__restrict__ part, int n_part,
                                                         const int64_t *__restrict__ seg_ptr, int n_seg,
                                                         int seg_bits, u64 *__restrict__ keys,
                                                         uint32_t *__restrict__ vals) {
    __shared__ float s_lo[kSpMaxDim], s_scale[kSpMaxDim];
    const int bits = (64 - seg_bits) / dim < 16 ? (64 - seg_bits) / dim : 16;
    if (threadIdx.x < kSpMaxDim) {
        const int d = threadIdx.x;
        float lo = 3.402823466e38f, hi = -3.402823466e38f;
        for (int b = 0; b < n_part; ++b) {
            lo = fminf(lo, part[b * 2 * kSpMaxDim + d]);
            hi = fmaxf(hi, part[b * 2 * kSpMaxDim + kSpMaxDim + d]);
        }
        const float w = hi - lo;
        s_lo[d] = lo;
        s_scale[d] = (d < dim && w > 0.f && w < 3.0e38f) ? (float)(1u << bits) / w : 0.f;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t q[kSpMaxDim];
    const uint32_t qmax = (1u << bits) - 1u;
#pragma unroll
    for (int d = 0; d < kSpMaxDim; ++d) {
        q[d] = 0;
        if (d < dim) {
            const float t = (x[i * stride + d] - s_lo[d]) * s_scale[d];
            q[d] = t >= (float)qmax ? qmax : (t > 0.f ? (uint32_t)t : 0u);  // (NaN -> 0)
        }
    }
    u64 code = 0;
    for (int b = bits - 1; b >= 0; --b)
#pragma unroll
        for (int d = 0; d < kSpMaxDim; ++d)
            if (d < dim) code = (code << 1) | (u64)((q[d] >> b) & 1u);
    if (seg_bits > 0) {
        const u64 sg = (u64)knn_segment_of(seg_ptr, n_seg, i);
        code |= sg << (64 - seg_bits);
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// one wave per chunk of 64 sorted points: gather the rows into the sorted array (rows of DP
// floats, the tail of the last chunk repeats the last point), remember their original indices
// (-1 in the tail) and write the chunk's bounding box [lo[DP] | hi[DP]]
template <int DP>
__global__ __launch_bounds__(256) void knn_gather_box_kernel(const float *__restrict__ x, int64_t n, int dim,
                                                             int stride, const uint32_t *__restrict__ order,
                                                             int n_chunks, float *__restrict__ xs,
                                                             int32_t *__restrict__ sidx,
                                                             float *__restrict__ box) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int64_t p = (int64_t)c * 64 + lane;
    const int64_t i = order[p < n ? p : n - 1];
    float v[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) v[d] = d < dim ? x[i * stride + d] : 0.f;
#pragma unroll
    for (int d = 0; d < DP; ++d) xs[p * DP + d] = v[d];
    sidx[p] = p < n ? (int32_t)i : -1;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        float lo = v[d], hi = v[d];
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o));
            hi = fmaxf(hi, __shfl_xor(hi, o));
        }
        if (lane == 0) {
            box[(int64_t)c * 2 * DP + d] = lo;
            box[(int64_t)c * 2 * DP + DP + d] = hi;
        }
    }
}

template <int DP, bool BATCH, int QW>
__global__ __launch_bounds__(kKnnBlock) void knn_pruned_kernel(const float *__restrict__ xs,
                                                               const int32_t *__restrict__ sidx,
                                                               const float *__restrict__ box, int64_t n,
                                                               int n_chunks, int k, int cap, float max_radius,
                                                               const int64_t *__restrict__ seg_ptr, int n_seg,
                                                               int32_t *__restrict__ nbr,
                                                               int32_t *__restrict__ cnt_out) {
    __shared__ __attribute__((aligned(16))) u64 s_keys[kKnnWaves][kKnnLdsPerWave / 8];
    __shared__ u64 s_tau[kKnnWaves][QW];
    __shared__ int s_cnt[kKnnWaves][QW], s_oq[kKnnWaves][QW];
    __shared__ int s_lo[kKnnWaves][QW], s_hi[kKnnWaves][QW];  // original-index range of each query's event
    __shared__ float s_gbox[kKnnWaves][2 * DP];                // bounding box of the wave's queries
    const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u64 *keys = s_keys[wv];
    u64 *tau = s_tau[wv];
    int *cnt = s_cnt[wv], *oq = s_oq[wv], *qlo = s_lo[wv], *qhi = s_hi[wv];
    const int64_t q0 = ((int64_t)blockIdx.x * kKnnWaves + wv) * QW;  // position in the SORTED order
    if (q0 >= n) return;
    const int nq = (int)((n - q0 < QW) ? (n - q0) : QW);

    u64 tau0 = kKeyMax;
    if (max_radius > 0.f) {
        const float r2 = max_radius * max_radius * 1.000001f + 1e-30f;
        tau0 = ((u64)__float_as_uint(r2) << 32) | 0xffffffffull;
    }
    if (lane < QW) {
        tau[lane] = tau0;
        cnt[lane] = 0;
        const int o = sidx[q0 + (lane < nq ? lane : nq - 1)];
        oq[lane] = o;
        if (BATCH) {
            const int sg = knn_segment_of(seg_ptr, n_seg, o);
            qlo[lane] = (int)seg_ptr[sg];
            qhi[lane] = (int)seg_ptr[sg + 1];
        }
    }
    knn_wave_sync();
    // chunks the wave has to look at: the events of its queries occupy the same positions in the
    // sorted order as in the original one (the event is the top of the sort key)
    int c_lo = 0, c_hi = n_chunks;
    if (BATCH) {
        c_lo = qlo[0] >> 6;
        c_hi = (qhi[nq - 1] + 63) >> 6;
    }
    c_lo = (int)__builtin_amdgcn_readfirstlane(c_lo);
    c_hi = (int)__builtin_amdgcn_readfirstlane(c_hi);

    int lane_zero;
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
#else
    lane_zero = 0;
#endif
    float qv[QW][DP];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const float *__restrict__ xq = xs + (q0 + (q < nq ? q : nq - 1)) * DP + lane_zero;
#pragma unroll
        for (int d = 0; d < DP; ++d) qv[q][d] = xq[d];
    }
    uint32_t tau_hi[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) tau_hi[q] = (uint32_t)(tau0 >> 32);
    // the queries' own bounding box: a batch of chunks none of which comes closer to THIS box than
    // the largest threshold cannot pass any query's test (box-to-box gap <= point-to-box gap in
    // every dimension, same monotone chain) - one cheap test instead of QW
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            float lo = qv[0][d], hi = qv[0][d];
#pragma unroll
            for (int q = 1; q < QW; ++q) {
                lo = fminf(lo, qv[q][d]);
                hi = fmaxf(hi, qv[q][d]);
            }
            s_gbox[wv][d] = lo;
            s_gbox[wv][DP + d] = hi;
        }
    }
    knn_wave_sync();

    auto load_chunk = [&](int c, float (&v)[DP], int &id) {
        const int64_t p = (int64_t)c * 64 + lane;
        const float *__restrict__ row = xs + p * DP;
#pragma unroll
        for (int d = 0; d < DP; ++d) v[d] = row[d];
        id = sidx[p];
    };

    const int b_lo = c_lo >> 6, b_hi = (c_hi + 63) >> 6;
    const int b_own = (int)((q0 >> 6) >> 6);
    const int span = (b_own - b_lo) > (b_hi - 1 - b_own) ? (b_own - b_lo) : (b_hi - 1 - b_own);
    for (int t = 0; t <= 2 * span; ++t) {
        const int off = (t + 1) >> 1;
        const int b = (t & 1) ? b_own + off : b_own - off;
        if (b < b_lo || b >= b_hi) continue;
        // lane = chunk: lower bound of every query's distance to the chunk's box
        const int c = b * 64 + lane;
        const bool cvalid = c >= c_lo && c < c_hi;
        const float *__restrict__ bx = box + (int64_t)(cvalid ? c : c_lo) * 2 * DP;
        float lo[DP], hi[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            lo[d] = bx[d];
            hi[d] = bx[DP + d];
        }
        {
            uint32_t tau_max = tau_hi[0];
#pragma unroll
            for (int u = 1; u < QW; ++u) tau_max = tau_hi[u] > tau_max ? tau_hi[u] : tau_max;
            float lbg = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const float g = fmaxf(fmaxf(__fsub_rn(lo[d], s_gbox[wv][DP + d]), __fsub_rn(s_gbox[wv][d], hi[d])), 0.f);
                lbg = __fmaf_rn(g, g, lbg);
            }
            if (__ballot(cvalid && __float_as_uint(lbg) <= tau_max) == 0ull) continue;
        }
        int qmask = 0;  // bit u: query u has to look at this lane's chunk
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            float lb = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const float g = fmaxf(fmaxf(__fsub_rn(lo[d], qv[u][d]), __fsub_rn(qv[u][d], hi[d])), 0.f);
                lb = __fmaf_rn(g, g, lb);
            }
            if (cvalid && u < nq && __float_as_uint(lb) <= tau_hi[u]) qmask |= 1 << u;
        }
        u64 any = __ballot(qmask != 0);
        int dirty = 0;
        if (any != 0ull) {
            float xc[DP], xn[DP];
            int ic, in;
            int ci = __ffsll(any) - 1;
            load_chunk(b * 64 + ci, xc, ic);
            while (any != 0ull) {
                any &= any - 1ull;
                const int cn = any != 0ull ? __ffsll(any) - 1 : ci;
                load_chunk(b * 64 + cn, xn, in);  // the next surviving chunk is in flight
                const int64_t j = (int64_t)(b * 64 + ci) * 64 + lane;  // sorted position of the candidate
                const int qm = knn_read_lane(qmask, ci);
#pragma unroll
                for (int u = 0; u < QW; ++u) {
                    if (((qm >> u) & 1) == 0) continue;
                    float d2 = 0.f;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        const float tt = __fsub_rn(qv[u][d], xc[d]);
                        d2 = __fmaf_rn(tt, tt, d2);
                    }
                    if (__ballot(__float_as_uint(d2) <= tau_hi[u]) == 0ull) continue;
                    const u64 tq = tau[u];
                    u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)(uint32_t)ic;
                    if (ic < 0 || j == q0 + u) key = kKeyMax;  // tail of the last chunk / the query itself
                    if (BATCH && (ic < qlo[u] || ic >= qhi[u])) key = kKeyMax;  // another event's hit
                    const bool pass = key < tq;
                    const u64 mask = __ballot(pass);
                    if (mask != 0ull) {
                        const int base = cnt[u];
                        if (pass) keys[u * cap + base + __popcll(mask & ((1ull << lane) - 1ull))] = key;
                        int nc = base + __popcll(mask);
                        dirty |= 1 << u;
                        knn_wave_sync();
                        if (nc > cap - 64) {  // no room for another full chunk: keep the k best
                            u64 *bq = keys + u * cap;
                            for (int i = nc + lane; i < cap; i += 64) bq[i] = kKeyMax;
                            knn_wave_sync();
                            wave_bitonic_sort(bq, cap, lane);
                            nc = k;
                            tau_hi[u] = (uint32_t)(bq[k - 1] >> 32);
                            if (lane == 0) tau[u] = bq[k - 1];
                            dirty &= ~(1 << u);
                        }
                        if (lane == 0) cnt[u] = nc;
                        knn_wave_sync();
                    }
                }
#pragma unroll
                for (int d = 0; d < DP; ++d) xc[d] = xn[d];
                ic = in;
                ci = cn;
            }
        }
        // a buffer that grew to k or more entries is cut to its k best now: the threshold of the
        // NEXT batch of boxes is the k-th key found so far
        // (only while the thresholds still move much: after the wave's own batch and its two
        // neighbours; a sort costs ~600 VALU instructions per query, a later one tightens little -
        // the buffers are still cut whenever they fill up)
        dirty = (int)__builtin_amdgcn_readfirstlane(dirty);
        if (dirty != 0 && t <= 2) {
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                if (((dirty >> u) & 1) == 0) continue;
                const int nc = cnt[u];
                if (nc < k) continue;
                u64 *bq = keys + u * cap;
                for (int i = nc + lane; i < cap; i += 64) bq[i] = kKeyMax;
                knn_wave_sync();
                wave_bitonic_sort(bq, cap, lane);
                tau_hi[u] = (uint32_t)(bq[k - 1] >> 32);
                if (lane == 0) {
                    tau[u] = bq[k - 1];
                    cnt[u] = k;
                }
                knn_wave_sync();
            }
        }
    }

    // final: sort every buffer, apply the radius filter (a prefix: keys ascend), emit into the
    // rows of the ORIGINAL query indices
    for (int q = 0; q < nq; ++q) {
        u64 *bq = keys + q * cap;
        const int nc = cnt[q];
        const int64_t row = oq[q];
        for (int i = nc + lane; i < cap; i += 64) bq[i] = kKeyMax;
        knn_wave_sync();
        wave_bitonic_sort(bq, cap, lane);
        const int mm = nc < k ? nc : k;
        int out = 0;
        for (int i0 = 0; i0 < mm; i0 += 64) {
            const int i = i0 + lane;
            bool ok = false;
            u64 key = kKeyMax;
            if (i < mm) {
                key = bq[i];
                ok = key != kKeyMax;
                if (ok && max_radius > 0.f)
                    ok = __fsqrt_rn(__uint_as_float((uint32_t)(key >> 32))) < max_radius;
            }
            if (ok) nbr[row * k + i] = (int32_t)(uint32_t)(key & 0xffffffffull);
            out += __popcll(__ballot(ok));
        }
        if (lane == 0) cnt_out[row] = out;
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

// exclusive scan of min(cnt, k_take) by ONE workgroup (n <= a few million counts): every WAVE owns
// a contiguous segment and walks it in chunks of 256 counts (one 16-byte load per lane, the next
// chunk in flight), first for the segment totals, then - after the totals of the waves are scanned
// - again for the offsets: lane-local prefix, shuffle scan across the lanes, four 8-byte stores per
// lane into consecutive addresses.  (A thread-per-range walk wrote 64 cache lines per store
// instruction and took 0.1 ms for 200 k counts, the serial first version 0.36 ms.)
template <bool VEC>
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t *__restrict__ cnt_raw, int k_take,
                                                           int64_t n, int64_t *__restrict__ off) {
    auto clip = [&](int32_t c) { return c < k_take ? c : k_take; };
    __shared__ long long s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nw = blockDim.x >> 6;  // 4 or 16 waves
    int64_t per = (n + nw - 1) / nw;
    per = (per + 255) & ~(int64_t)255;  // segments of whole chunks (16-byte aligned lane loads)
    const int64_t b = wv * per < n ? wv * per : n, e = (b + per < n) ? b + per : n;
    auto load4 = [&](int64_t at, int (&v)[4]) {  // counts at..at+3 of this lane (0 beyond e)
        if (VEC && at + 3 < e) {
            const int4 q = *reinterpret_cast<const int4 *>(cnt_raw + at);
            v[0] = clip(q.x); v[1] = clip(q.y); v[2] = clip(q.z); v[3] = clip(q.w);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = at + u < e ? clip(cnt_raw[at + u]) : 0;
        }
    };
    constexpr int kU = 8;  // chunks per step: their loads are in flight together (the walk is latency bound)
    long long s = 0;
    for (int64_t c0 = b; c0 < e; c0 += 256 * kU) {
        int v[kU][4];
#pragma unroll
        for (int u = 0; u < kU; ++u) load4(c0 + 256 * u + 4 * lane, v[u]);
#pragma unroll
        for (int u = 0; u < kU; ++u) s += (long long)v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) s_wave[wv] = s;
    __syncthreads();
    long long run = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
        const long long v = s_wave[w];
        if (w < wv) run += v;
        total += v;
    }
    if (t == 0) off[n] = total;
    for (int64_t c0 = b; c0 < e; c0 += 256 * kU) {
        int v[kU][4];
#pragma unroll
        for (int u = 0; u < kU; ++u) load4(c0 + 256 * u + 4 * lane, v[u]);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (c0 + 256 * u >= e) break;  // (wave-uniform)
            const long long mine = (long long)v[u][0] + v[u][1] + v[u][2] + v[u][3];
            long long inc = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const long long o = __shfl(inc, lane >= d ? lane - d : lane);
                if (lane >= d) inc += o;
            }
            long long at = run + inc - mine;
            const int64_t i0 = c0 + 256 * u + 4 * lane;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (i0 + w < e) off[i0 + w] = at;
                at += v[u][w];
            }
            run += __shfl(inc, 63);
        }
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

// ---- sorted chunks + boxes: shared by the pruned search and the condensation losses (oc.hip) ---
// Layout of the three outputs for n points of dimension dim (<= 8): rows of DP = 4 or 8 floats.
int spatial_dp(int dim) { return dim <= 4 ? 4 : dim <= 8 ? 8 : 16; }
int spatial_n_chunks(int64_t n) { return (int)ceil_div(n, 64); }
// scratch of the build: [box partials | keys a | keys b | vals a | vals b | radix-sort temp]
struct SpatialScratch {
    size_t part, keys_a, keys_b, vals_a, vals_b, temp, total;
};
static SpatialScratch spatial_scratch_layout(int64_t n) {
    SpatialScratch w{};
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    w.part = take((size_t)kKnnBoxParts * 2 * kSpMaxDim * sizeof(float));
    w.keys_a = take((size_t)n * 8);
    w.keys_b = take((size_t)n * 8);
    w.vals_a = take((size_t)n * 4);
    w.vals_b = take((size_t)n * 4);
    w.temp = take(sort_pairs_u64_temp_bytes(n));
    w.total = o;
    return w;
}
size_t spatial_scratch_bytes(int64_t n) { return spatial_scratch_layout(n).total; }

// xs[n_chunks * 64][DP]: the points in (event, Morton) order, the tail of the last chunk repeats the
// last point; sidx[n_chunks * 64]: their original indices (-1 in the tail); box[n_chunks][2 * DP]
int spatial_chunks_build(const float *x, int64_t n, int dim, int stride, const int64_t *seg_ptr, int n_seg,
                         float *xs, int32_t *sidx, float *box, void *scratch, size_t scratch_bytes,
                         hipStream_t stream) {
    if (!x || n < 1 || dim < 1 || dim > kSpMaxDim || stride < dim || !xs || !sidx || !box || !scratch)
        return fail(GNNTRK_EINVAL, "spatial_chunks: bad argument");
    const SpatialScratch w = spatial_scratch_layout(n);
    if (scratch_bytes < w.total) return fail(GNNTRK_EINVAL, "spatial_chunks: scratch too small");
    char *base = static_cast<char *>(scratch);
    float *part = reinterpret_cast<float *>(base + w.part);
    u64 *keys_a = reinterpret_cast<u64 *>(base + w.keys_a), *keys_b = reinterpret_cast<u64 *>(base + w.keys_b);
    uint32_t *vals_a = reinterpret_cast<uint32_t *>(base + w.vals_a), *vals_b = reinterpret_cast<uint32_t *>(base + w.vals_b);
    int seg_bits = 0;
    if (seg_ptr)
        while ((1 << seg_bits) < n_seg) ++seg_bits;
    const int n_part = (int)(ceil_div(n, 1024) < kKnnBoxParts ? ceil_div(n, 1024) : kKnnBoxParts);
    hipLaunchKernelGGL(knn_bbox_partial_kernel, dim3(n_part), dim3(256), 0, stream, x, n, dim, stride, part);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, x, n, dim, stride,
                       (const float *)part, n_part, seg_ptr, n_seg, seg_bits, keys_a, vals_a);
    int rc = check_launch("spatial_chunks(sort keys)");
    if (rc != GNNTRK_OK) return rc;
    rc = sort_pairs_u64(keys_a, keys_b, vals_a, vals_b, n, base + w.temp, sort_pairs_u64_temp_bytes(n), stream);
    if (rc != GNNTRK_OK) return rc;
    const int n_chunks = spatial_n_chunks(n);
    const unsigned gb = (unsigned)ceil_div(n_chunks, 4);
    if (spatial_dp(dim) == 4)
        hipLaunchKernelGGL((knn_gather_box_kernel<4>), dim3(gb), dim3(256), 0, stream, x, n, dim, stride,
                           (const uint32_t *)vals_b, n_chunks, xs, sidx, box);
    else if (spatial_dp(dim) == 8)
        hipLaunchKernelGGL((knn_gather_box_kernel<8>), dim3(gb), dim3(256), 0, stream, x, n, dim, stride,
                           (const uint32_t *)vals_b, n_chunks, xs, sidx, box);
    else
        hipLaunchKernelGGL((knn_gather_box_kernel<16>), dim3(gb), dim3(256), 0, stream, x, n, dim, stride,
                           (const uint32_t *)vals_b, n_chunks, xs, sidx, box);
    return check_launch("spatial_chunks(gather)");
}

// ---- pruned search: workspace layout and launch -------------------------------------------
struct KnnWs {
    size_t xs, sidx, box, scratch, total;
    int n_chunks, dp;
};
static KnnWs knn_ws_layout(int64_t n, int dim) {
    KnnWs w{};
    w.dp = spatial_dp(dim);
    w.n_chunks = spatial_n_chunks(n);
    const size_t rows = (size_t)w.n_chunks * 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    w.xs = take(rows * w.dp * sizeof(float));
    w.sidx = take(rows * 4);
    w.box = take((size_t)w.n_chunks * 2 * w.dp * sizeof(float));
    w.scratch = take(spatial_scratch_bytes(n));
    w.total = o;
    return w;
}

// queries per wave of the pruned kernel for this k (0: not covered, use the brute-force kernel)
static int knn_pruned_qw(int dim, int k, int *cap_out) {
    int cap = 128;
    while (cap < k + 64) cap <<= 1;
    *cap_out = cap;
    const int qw = kKnnLdsPerWave / (cap * 8);
    if (dim > kSpMaxDim || !(qw == 8 || qw == 4 || qw == 2)) return 0;
    return (dim > 8 && qw == 8) ? 4 : qw;  // 16 coordinates per query: four queries fill the registers
}

size_t knn_workspace_bytes(int64_t n, int dim, int k) {
    int cap;
    if (n <= 0 || dim < 1 || k < 1 || n > 0x7fffffff || knn_pruned_qw(dim, k, &cap) == 0) return 0;
    return knn_ws_layout(n, dim).total;
}

constexpr int64_t kKnnPrunedMinRows = 8192;  // below this the sort + boxes cost more than they save

int knn_search_ws_launch(const float *x, int64_t n, int dim, int stride, int k, float max_radius,
                         const int64_t *seg_ptr, int n_seg, int32_t *nbr, int32_t *cnt, void *ws,
                         size_t ws_bytes, int flags, hipStream_t stream) {
    int cap = 0;
    const int qw = (n > 0 && dim >= 1 && k >= 1 && n <= 0x7fffffff) ? knn_pruned_qw(dim, k, &cap) : 0;
    const bool force_brute = (flags & 2) != 0, force_pruned = (flags & 1) != 0;
    bool pruned = qw != 0 && ws != nullptr && !force_brute && (force_pruned || n >= kKnnPrunedMinRows);
    if (pruned && seg_ptr && n_seg > 65536) pruned = false;
    if (!pruned) return knn_search_launch(x, n, dim, stride, k, max_radius, seg_ptr, n_seg, nbr, cnt, stream);
    if (!x || !nbr || !cnt || stride < dim) return fail(GNNTRK_EINVAL, "knn_search: bad argument");
    if (seg_ptr && n_seg < 1) return fail(GNNTRK_EINVAL, "knn_search: seg_ptr needs n_seg >= 1");
    const KnnWs w = knn_ws_layout(n, dim);
    if (ws_bytes < w.total) return fail(GNNTRK_EINVAL, "knn_search: workspace too small (gnntrk_knn_workspace_bytes)");
    char *base = static_cast<char *>(ws);
    float *xs = reinterpret_cast<float *>(base + w.xs);
    int32_t *sidx = reinterpret_cast<int32_t *>(base + w.sidx);
    float *box = reinterpret_cast<float *>(base + w.box);
    const int rc = spatial_chunks_build(x, n, dim, stride, seg_ptr, n_seg, xs, sidx, box, base + w.scratch,
                                        w.total - w.scratch, stream);
    if (rc != GNNTRK_OK) return rc;
    const unsigned grid = (unsigned)ceil_div(n, (int64_t)qw * kKnnWaves);
#define KNN_PRUNED(DP, QW_)                                                                                  \
    if (seg_ptr)                                                                                             \
        hipLaunchKernelGGL((knn_pruned_kernel<DP, true, QW_>), dim3(grid), dim3(kKnnBlock), 0, stream,       \
                           (const float *)xs, (const int32_t *)sidx, (const float *)box, n, w.n_chunks, k,   \
                           cap, max_radius, seg_ptr, n_seg, nbr, cnt);                                       \
    else                                                                                                     \
        hipLaunchKernelGGL((knn_pruned_kernel<DP, false, QW_>), dim3(grid), dim3(kKnnBlock), 0, stream,      \
                           (const float *)xs, (const int32_t *)sidx, (const float *)box, n, w.n_chunks, k,   \
                           cap, max_radius, seg_ptr, n_seg, nbr, cnt)
    if (w.dp == 4) {
        if (qw == 8) { KNN_PRUNED(4, 8); } else if (qw == 4) { KNN_PRUNED(4, 4); } else { KNN_PRUNED(4, 2); }
    } else if (w.dp == 8) {
        if (qw == 8) { KNN_PRUNED(8, 8); } else if (qw == 4) { KNN_PRUNED(8, 4); } else { KNN_PRUNED(8, 2); }
    } else {
        if (qw == 4) { KNN_PRUNED(16, 4); } else { KNN_PRUNED(16, 2); }
    }
#undef KNN_PRUNED
    return check_launch("knn_search(pruned)");
}

// off[0..n] = exclusive scan of min(cnt[i], k_take) (also the condensation-point selection's scan)
void scan_counts_launch(const int32_t *cnt, int k_take, int64_t n, int64_t *off, hipStream_t stream) {
    const int nt = n < 65536 ? 256 : 1024;
    if ((reinterpret_cast<uintptr_t>(cnt) & 15u) == 0)
        hipLaunchKernelGGL(scan_counts_kernel<true>, dim3(1), dim3(nt), 0, stream, cnt, k_take, n, off);
    else
        hipLaunchKernelGGL(scan_counts_kernel<false>, dim3(1), dim3(nt), 0, stream, cnt, k_take, n, off);
}

int knn_emit_launch(const int32_t *nbr, const int32_t *cnt, int64_t n, int k_stride, int k, int64_t *offsets,
                    int64_t *edge_index, int64_t m_total, hipStream_t stream) {
    if (!nbr || !cnt || !offsets || n < 0 || k < 1 || k > k_stride)
        return fail(GNNTRK_EINVAL, "knn_emit: bad argument");
    if (n == 0) return GNNTRK_OK;
    if (!edge_index) {  // phase 1: offsets only (offsets[n] = total edge count)
        scan_counts_launch(cnt, k, n, offsets, stream);
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
