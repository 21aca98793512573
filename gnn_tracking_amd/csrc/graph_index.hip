// Graph index: stable target-sorted (CSR) and source-sorted views of a COO edge list.
// See gnntrk_graph_index in include/gnntrk.h.  HBM-bound integer work: coalesced
// streams over E, one gather (src[perm[k]]), two stable device radix sorts.
#include "host_util.h"

namespace gnntrk {

constexpr int kTpb = 256;

__global__ __launch_bounds__(kTpb) void gi_keys_kernel(const int64_t *__restrict__ ids, int64_t E,
                                                       int64_t N, uint32_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals,
                                                       int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < E;
         e += (int64_t)gridDim.x * kTpb) {
        const int64_t v = ids[e];
        if (v < 0 || v >= N) atomicAdd(bad, 1);
        keys[e] = (uint32_t)v;
        vals[e] = (uint32_t)e;
    }
}

// src_sorted[k] = src[perm[k]]; also emits the keys/vals of the second (by-source) sort
__global__ __launch_bounds__(kTpb) void gi_gather_src_kernel(const int64_t *__restrict__ src,
                                                             const uint32_t *__restrict__ perm,
                                                             int64_t E, int64_t N, int32_t *__restrict__ src_s,
                                                             uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals,
                                                             int *__restrict__ bad) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < E;
         k += (int64_t)gridDim.x * kTpb) {
        const int64_t v = src[perm[k]];
        if (v < 0 || v >= N) atomicAdd(bad, 1);
        const uint32_t s = (uint32_t)v;
        src_s[k] = (int32_t)s;
        keys[k] = s;
        vals[k] = (uint32_t)k;
    }
}

// rowptr[n] = first position k with sorted[k] >= n  (sorted ascending, E entries)
__global__ __launch_bounds__(kTpb) void gi_rowptr_kernel(const uint32_t *__restrict__ sorted,
                                                         int64_t E, int64_t N,
                                                         int32_t *__restrict__ rowptr) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k <= E;
         k += (int64_t)gridDim.x * kTpb) {
        const int64_t prev = (k == 0) ? -1 : (int64_t)sorted[k - 1];
        const int64_t cur = (k == E) ? N : (int64_t)sorted[k];
        for (int64_t n = prev + 1; n <= cur && n <= N; ++n) rowptr[n] = (int32_t)k;
    }
}

// inv[spos[k]] = k
__global__ __launch_bounds__(kTpb) void gi_invert_kernel(const int32_t *__restrict__ spos, int64_t E,
                                                         int32_t *__restrict__ inv) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < E; k += (int64_t)gridDim.x * kTpb)
        inv[spos[k]] = (int32_t)k;
}

static int stream_grid(int64_t n) {
    int64_t g = ceil_div(n, kTpb);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

static int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n) ++b;
    return b;
}

size_t graph_index_ws_bytes(int64_t N, int64_t E) {
    (void)N;
    const size_t arr = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint32_t), 256);
    return 256 /* flags */ + 3 * arr + align_up(sort_pairs_temp_bytes(E), 256);
}

int graph_index_build(const int64_t *edge_index, const gnntrk_graph_index *o, void *ws,
                      size_t ws_bytes, hipStream_t stream) {
    if (!o) return fail(GNNTRK_EINVAL, "graph_index_build: NULL output descriptor");
    const int64_t E = o->n_edges, N = o->n_nodes;
    if (E < 0 || N < 0 || E > 0x7fffffff || N > 0x7fffffff)
        return fail(GNNTRK_EUNSUPPORTED, "graph_index_build: sizes must fit int32");
    if (!o->rowptr_t || !o->rowptr_s || (E > 0 && (!o->perm || !o->tgt || !o->src || !o->spos)))
        return fail(GNNTRK_EINVAL, "graph_index_build: NULL output array");
    if (E > 0 && !edge_index) return fail(GNNTRK_EINVAL, "graph_index_build: NULL edge_index");
    if (!ws || ws_bytes < graph_index_ws_bytes(N, E))
        return fail(GNNTRK_EINVAL, "graph_index_build: workspace too small");

    char *p = reinterpret_cast<char *>(ws);
    int *bad = reinterpret_cast<int *>(p);
    p += 256;
    const size_t arr = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint32_t), 256);
    uint32_t *keys_a = reinterpret_cast<uint32_t *>(p);
    p += arr;
    uint32_t *vals_a = reinterpret_cast<uint32_t *>(p);
    p += arr;
    uint32_t *keys_b = reinterpret_cast<uint32_t *>(p);
    p += arr;
    void *temp = p;
    const size_t temp_bytes = sort_pairs_temp_bytes(E);

    int rc = check_hip(hipMemsetAsync(bad, 0, 256, stream), "graph_index_build(memset)");
    if (rc) return rc;
    const int bits = bits_for(N > 1 ? N : 2);
    const int grid = stream_grid(E + 1);
    if (E > 0) {
        const int64_t *src = edge_index, *tgt = edge_index + E;
        hipLaunchKernelGGL(gi_keys_kernel, dim3(grid), dim3(kTpb), 0, stream, tgt, E, N, keys_a,
                           vals_a, bad);
        // the sorted keys ARE the CSR targets: sort straight into the output array
        uint32_t *tgt_sorted = reinterpret_cast<uint32_t *>(o->tgt);
        rc = sort_pairs_u32(keys_a, tgt_sorted, vals_a, reinterpret_cast<uint32_t *>(o->perm), E, bits,
                            temp, temp_bytes, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(gi_rowptr_kernel, dim3(grid), dim3(kTpb), 0, stream, tgt_sorted, E, N,
                           o->rowptr_t);
        hipLaunchKernelGGL(gi_gather_src_kernel, dim3(grid), dim3(kTpb), 0, stream, src,
                           reinterpret_cast<const uint32_t *>(o->perm), E, N, o->src, keys_a, vals_a, bad);
        rc = sort_pairs_u32(keys_a, keys_b, vals_a, reinterpret_cast<uint32_t *>(o->spos), E, bits,
                            temp, temp_bytes, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(gi_rowptr_kernel, dim3(grid), dim3(kTpb), 0, stream, keys_b, E, N,
                           o->rowptr_s);
        if (o->spos_inv)
            hipLaunchKernelGGL(gi_invert_kernel, dim3(grid), dim3(kTpb), 0, stream, o->spos, E, o->spos_inv);
    } else {
        rc = check_hip(hipMemsetAsync(o->rowptr_t, 0, (size_t)(N + 1) * 4, stream), "memset");
        if (rc) return rc;
        rc = check_hip(hipMemsetAsync(o->rowptr_s, 0, (size_t)(N + 1) * 4, stream), "memset");
        if (rc) return rc;
    }
    return check_launch("graph_index_build");
}

}  // namespace gnntrk
