// Stream compaction for the ModularGraphTCN glue (models/track_condensation_networks.py:
// 251-262): the threshold cut on the edge weights (`data.edge_subgraph(W > ec_threshold)`)
// and the orphan-node masking (`edge_index.flatten().unique()` -> `index_to_mask` ->
// `data.subgraph(connected)`: new node ids + relabelled edge_index).
//
// HBM-bound integer work.  One deterministic three-pass compaction (no atomics, output in
// ascending index order, as boolean indexing gives it):
//   1. per-tile flag counts        (coalesced read of the flag source)
//   2. exclusive scan of the counts (one workgroup)
//   3. per-tile ranks -> index list (+ mask / new-id side outputs)
// The connected-node variant replaces the reference's sort-based `unique` over 2E ids by a
// mark pass (idempotent byte stores) + the same compaction over N, then relabels the 2E ids.
#include "host_util.h"

namespace gnntrk {

constexpr int kCTpb = 256;                 // 4 waves
constexpr int kCItems = 8;                 // consecutive items per thread (wide loads)
constexpr int kCTile = kCTpb * kCItems;    // 2048 items per workgroup
constexpr int kCWaves = kCTpb / 64;
constexpr int kScanTpb = 1024;

// Flag sources.  A thread owns kCItems = 8 CONSECUTIVE items and fetches them with wide
// loads (2 x 16 bytes of weights / one 8-byte word of flag bytes) when the tile is full and the
// base pointer is aligned (`vec`); the ragged last tile and unaligned views go item by item.
struct FlagThreshold {
    const float *w;
    float thr;
    bool vec;  // w is 16-byte aligned
    __device__ __forceinline__ void load(int64_t i0, int64_t n, bool (&f)[8]) const {
        if (vec && i0 + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4 *>(w + i0);
            const float4 b = *reinterpret_cast<const float4 *>(w + i0 + 4);
            f[0] = a.x > thr; f[1] = a.y > thr; f[2] = a.z > thr; f[3] = a.w > thr;  // NaN -> false
            f[4] = b.x > thr; f[5] = b.y > thr; f[6] = b.z > thr; f[7] = b.w > thr;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = i0 + k < n && w[i0 + k] > thr;
        }
    }
};
struct FlagByte {
    const uint8_t *b;
    bool vec;  // b is 8-byte aligned
    __device__ __forceinline__ void load(int64_t i0, int64_t n, bool (&f)[8]) const {
        if (vec && i0 + 8 <= n) {
            const unsigned long long v = *reinterpret_cast<const unsigned long long *>(b + i0);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = ((v >> (8 * k)) & 0xffull) != 0;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = i0 + k < n && b[i0 + k] != 0;
        }
    }
};

template <class F>
__global__ __launch_bounds__(kCTpb) void compact_count_kernel(F flag, int64_t n, int32_t *__restrict__ counts) {
    __shared__ int32_t s_cnt[kCWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kCTile + (int64_t)tid * kCItems;
    bool f[kCItems];
    flag.load(i0, n, f);
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < kCItems; ++k) c += __popcll(__ballot(f[k]));  // wave-uniform count
    if (lane == 0) s_cnt[wv] = c;
    __syncthreads();
    if (tid == 0) {
        int32_t t = 0;
        for (int w = 0; w < kCWaves; ++w) t += s_cnt[w];
        counts[blockIdx.x] = t;
    }
}

// exclusive scan of counts[0..nb) in place; total -> n_out[0].  One workgroup: every thread
// owns a contiguous chunk, the chunk sums are scanned through LDS.
__global__ __launch_bounds__(kScanTpb) void compact_scan_kernel(int32_t *__restrict__ counts, int32_t nb,
                                                               int64_t *__restrict__ n_out) {
    __shared__ int64_t s_sum[kScanTpb];
    const int tid = threadIdx.x;
    const int per = (nb + kScanTpb - 1) / kScanTpb;
    const int lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    int64_t mine = 0;
    for (int i = lo; i < hi; ++i) mine += counts[i];
    s_sum[tid] = mine;
    __syncthreads();
    for (int d = 1; d < kScanTpb; d <<= 1) {  // Hillis-Steele inclusive scan
        const int64_t v = tid >= d ? s_sum[tid - d] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    int64_t run = s_sum[tid] - mine;
    for (int i = lo; i < hi; ++i) {
        const int32_t c = counts[i];
        counts[i] = (int32_t)run;
        run += c;
    }
    if (tid == kScanTpb - 1) n_out[0] = s_sum[tid];
}

// idx[rank] = i for every flagged i (ascending); optional side outputs:
//   mask[i] = flag (uint8),  newid[i] = rank or -1 (int32)
// Rank of item k of lane l in wave v = tile offset + items of waves < v + items of lanes < l
// (one popcount per item slot over the ballots) + flagged items k' < k of the own thread.
template <class F>
__global__ __launch_bounds__(kCTpb) void compact_write_kernel(F flag, int64_t n, const int32_t *__restrict__ offsets,
                                                              int32_t *__restrict__ idx, uint8_t *__restrict__ mask,
                                                              bool mask_vec, int32_t *__restrict__ newid) {
    __shared__ int32_t s_cnt[kCWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kCTile + (int64_t)tid * kCItems;
    bool f[kCItems];
    flag.load(i0, n, f);
    const unsigned long long below = (1ull << lane) - 1ull;
    int32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kCItems; ++k) {
        const unsigned long long b = __ballot(f[k]);
        before += __popcll(b & below);
        total += __popcll(b);
    }
    if (lane == 0) s_cnt[wv] = total;
    __syncthreads();
    int32_t rank = offsets[blockIdx.x] + before;
#pragma unroll
    for (int w = 0; w < kCWaves; ++w) rank += w < wv ? s_cnt[w] : 0;
    if (mask) {
        if (mask_vec && i0 + 8 <= n) {
            unsigned long long v = 0;
#pragma unroll
            for (int k = 0; k < kCItems; ++k) v |= (unsigned long long)(f[k] ? 1 : 0) << (8 * k);
            *reinterpret_cast<unsigned long long *>(mask + i0) = v;
        } else {
#pragma unroll
            for (int k = 0; k < kCItems; ++k)
                if (i0 + k < n) mask[i0 + k] = f[k] ? 1 : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kCItems; ++k) {
        const int64_t i = i0 + k;
        if (f[k]) idx[rank] = (int32_t)i;
        if (newid && i < n) newid[i] = f[k] ? rank : -1;
        rank += f[k] ? 1 : 0;
    }
}

// hit[id] = 1 for both endpoints of every edge (ids[0..m): the flattened [2, E] edge_index)
__global__ __launch_bounds__(kCTpb) void mark_nodes_kernel(const int64_t *__restrict__ ids, int64_t m, int64_t n_nodes,
                                                           uint8_t *__restrict__ hit, int64_t *__restrict__ bad) {
    for (int64_t k = (int64_t)blockIdx.x * kCTpb + threadIdx.x; k < m; k += (int64_t)gridDim.x * kCTpb) {
        const int64_t v = ids[k];
        if (v < 0 || v >= n_nodes)
            bad[0] = 1;  // (reported through n_out[1]; any writer stores the same value)
        else
            hit[v] = 1;
    }
}

__global__ __launch_bounds__(kCTpb) void relabel_kernel(const int64_t *__restrict__ ids, int64_t m, int64_t n_nodes,
                                                        const int32_t *__restrict__ newid,
                                                        int64_t *__restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * kCTpb + threadIdx.x; k < m; k += (int64_t)gridDim.x * kCTpb) {
        const int64_t v = ids[k];
        out[k] = (v >= 0 && v < n_nodes) ? (int64_t)newid[v] : -1;
    }
}

static int tiles_of(int64_t n) { return (int)ceil_div(n > 0 ? n : 1, kCTile); }
static int stream_blocks(int64_t n) {
    int64_t g = ceil_div(n > 0 ? n : 1, kCTpb);
    const int64_t cap = (int64_t)cu_count() * 8;
    return (int)(g > cap ? cap : g);
}

size_t compact_ws_bytes(int64_t n) { return align_up((size_t)tiles_of(n) * sizeof(int32_t), 256); }

template <class F>
static int compact_run(F flag, int64_t n, int32_t *idx, uint8_t *mask, int32_t *newid, int64_t *n_out, void *ws,
                       size_t ws_bytes, hipStream_t stream, const char *who) {
    if (n < 0 || n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "compact: n must fit int32");
    if (!n_out) return fail(GNNTRK_EINVAL, "compact: NULL count output");
    if (n == 0) return check_hip(hipMemsetAsync(n_out, 0, sizeof(int64_t), stream), who);
    if (!idx) return fail(GNNTRK_EINVAL, "compact: NULL index output");
    if (!ws || ws_bytes < compact_ws_bytes(n)) return fail(GNNTRK_EINVAL, "compact: workspace too small");
    int32_t *counts = reinterpret_cast<int32_t *>(ws);
    const int nb = tiles_of(n);
    hipLaunchKernelGGL((compact_count_kernel<F>), dim3(nb), dim3(kCTpb), 0, stream, flag, n, counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(kScanTpb), 0, stream, counts, nb, n_out);
    hipLaunchKernelGGL((compact_write_kernel<F>), dim3(nb), dim3(kCTpb), 0, stream, flag, n, counts, idx, mask,
                       ((uintptr_t)mask & 7) == 0, newid);
    return check_launch(who);
}

int threshold_compact_launch(const float *w, int64_t n, float threshold, uint8_t *mask, int32_t *idx, int64_t *n_out,
                             void *ws, size_t ws_bytes, hipStream_t stream) {
    if (n > 0 && (!w || !mask)) return fail(GNNTRK_EINVAL, "threshold_compact: NULL argument");
    return compact_run(FlagThreshold{w, threshold, ((uintptr_t)w & 15) == 0}, n, idx, mask, nullptr, n_out, ws, ws_bytes, stream,
                       "threshold_compact");
}

// idx = ascending positions of the non-zero bytes, newid = their ranks (or -1), n_out[0] = count
int compact_bytes_launch(const uint8_t *flags, int64_t n, int32_t *idx, int32_t *newid, int64_t *n_out, void *ws,
                         size_t ws_bytes, hipStream_t stream) {
    if (n > 0 && !flags) return fail(GNNTRK_EINVAL, "compact_bytes: NULL flags");
    return compact_run(FlagByte{flags, ((uintptr_t)flags & 7) == 0}, n, idx, nullptr, newid, n_out, ws, ws_bytes,
                       stream, "compact_bytes");
}

int connected_nodes_launch(const int64_t *edge_index, int64_t n_edges, int64_t n_nodes, uint8_t *hit,
                           int32_t *node_idx, int32_t *newid, int64_t *n_out, int64_t *edge_index_out, void *ws,
                           size_t ws_bytes, hipStream_t stream) {
    if (n_edges < 0 || n_nodes < 0 || n_nodes > 0x7fffffff || n_edges > 0x3fffffff)
        return fail(GNNTRK_EUNSUPPORTED, "connected_nodes: sizes must fit int32");
    if (!n_out) return fail(GNNTRK_EINVAL, "connected_nodes: NULL count output");
    int rc = check_hip(hipMemsetAsync(n_out, 0, 2 * sizeof(int64_t), stream), "connected_nodes(memset)");
    if (rc || n_nodes == 0) return rc;
    if (!hit || !node_idx || !newid || (n_edges > 0 && (!edge_index || !edge_index_out)))
        return fail(GNNTRK_EINVAL, "connected_nodes: NULL argument");
    rc = check_hip(hipMemsetAsync(hit, 0, (size_t)n_nodes, stream), "connected_nodes(memset)");
    if (rc) return rc;
    const int64_t m = 2 * n_edges;
    if (m > 0)
        hipLaunchKernelGGL(mark_nodes_kernel, dim3(stream_blocks(m)), dim3(kCTpb), 0, stream, edge_index, m, n_nodes,
                           hit, n_out + 1);
    rc = compact_run(FlagByte{hit, ((uintptr_t)hit & 7) == 0}, n_nodes, node_idx, nullptr, newid, n_out, ws, ws_bytes, stream,
                     "connected_nodes");
    if (rc) return rc;
    if (m > 0)
        hipLaunchKernelGGL(relabel_kernel, dim3(stream_blocks(m)), dim3(kCTpb), 0, stream, edge_index, m, n_nodes,
                           newid, edge_index_out);
    return check_launch("connected_nodes");
}

}  // namespace gnntrk
