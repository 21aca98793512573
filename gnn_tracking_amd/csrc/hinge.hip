// Hinge embedding loss of the metric-learning stage (metrics/losses/metric_learning.py:14-55,
// edge selection :88-110): the two edge-list reductions as fused kernels - see include/gnntrk.h.
// HBM / latency bound: per edge two random row gathers of 4 dim bytes and 16 bytes of ids.
//
// Forward: thread = edge; selection (node mask of the first endpoint, different particle ids) applied in
// place; per-block fp64 partial sums of (term, count) in a fixed tree, one finishing block adds the
// partials in block order and forms sum / denom: deterministic.
// Backward: eight lanes = node, over the graph index of the same edge list (CSR by edges[1], source-sorted view
// of edges[0]): every incident edge's term is recomputed from x and added in list order - no per-edge
// intermediate, no atomics, bit-reproducible.
#include "host_util.h"

namespace gnntrk {
namespace {

constexpr int kHingeTpb = 256;
constexpr int kHingeMaxDim = 32;
constexpr int kHingeMaxBlocks = 2048;

struct HingeTerm {
    float term;   // contribution to the sum
    float coef;   // d(term) / d(diff) = coef * diff
    bool on;      // passes the selection
};

// d = ||x[a] - x[b]||_2 (fp32 sum in feature order, as torch.linalg.norm on a contiguous row), term and
// derivative coefficient.  p == 1 and p == 2 avoid powf (exact d and d * d).
__device__ __forceinline__ HingeTerm hinge_term(const gnntrk_hinge_args &h, int64_t a, int64_t b) {
    HingeTerm t;
    t.term = 0.f;
    t.coef = 0.f;
    t.on = true;
    if (h.node_mask != nullptr && h.node_mask[a] == 0) t.on = false;
    if (h.particle_id != nullptr && h.particle_id[a] == h.particle_id[b]) t.on = false;
    if (!t.on) return t;
    const float *xa = h.x + a * h.x_stride, *xb = h.x + b * h.x_stride;
    float d2 = 0.f;
    for (int f = 0; f < h.dim; ++f) {
        const float df = xa[f] - xb[f];
        d2 += df * df;
    }
    const float d = sqrtf(d2);
    float dp, ddp;   // d^p and d(d^p)/dd / d  (the factor that multiplies diff)
    if (h.p == 1.f) {
        dp = d;
        ddp = d > 0.f ? 1.f / d : 0.f;
    } else if (h.p == 2.f) {
        dp = d * d;
        ddp = 2.f;
    } else {
        dp = powf(d, h.p);
        ddp = d > 0.f ? h.p * powf(d, h.p - 2.f) : 0.f;
    }
    if (h.repulsive) {
        const float v = h.r_emb - dp;
        t.term = v > 0.f ? v : 0.f;
        t.coef = v > 0.f ? -ddp : 0.f;
    } else {
        t.term = dp;
        t.coef = ddp;
    }
    return t;
}

__device__ __forceinline__ void block_sum2(double &s, double &n, double *sh) {
    for (int m = 1; m < 64; m <<= 1) {
        s += __shfl_xor(s, m);
        n += __shfl_xor(n, m);
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
        sh[2 * w] = s;
        sh[2 * w + 1] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0;
        n = 0.0;
        for (int i = 0; i < kHingeTpb / 64; ++i) {
            s += sh[2 * i];
            n += sh[2 * i + 1];
        }
    }
}

__global__ __launch_bounds__(kHingeTpb) void hinge_fwd_kernel(const gnntrk_hinge_args h, const int64_t *edges, int64_t n_edges,
                                                           int64_t edge_stride, double *part) {
    __shared__ double sh[2 * kHingeTpb / 64];
    double s = 0.0, n = 0.0;
    // every block takes one contiguous share of the edges: the sum does not depend on the grid's timing
    const int64_t per = (n_edges + gridDim.x - 1) / gridDim.x;
    const int64_t e0 = per * blockIdx.x, e1 = e0 + per < n_edges ? e0 + per : n_edges;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kHingeTpb) {
        const HingeTerm t = hinge_term(h, edges[e], edges[edge_stride + e]);
        if (t.on) {
            s += (double)t.term;
            n += 1.0;
        }
    }
    block_sum2(s, n, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s;
        part[2 * blockIdx.x + 1] = n;
    }
}

__global__ __launch_bounds__(kHingeTpb) void hinge_finish_kernel(const double *part, int n_part, const float *norm, float *out) {
    __shared__ double sh[2 * kHingeTpb / 64];
    double s = 0.0, n = 0.0;
    // (thread t adds partials t, t + 256, ... in order; the tree below is fixed)
    for (int i = threadIdx.x; i < n_part; i += kHingeTpb) {
        s += part[2 * i];
        n += part[2 * i + 1];
    }
    block_sum2(s, n, sh);
    if (threadIdx.x == 0) {
        const float cnt = (float)n;
        // the reference's `count + eps` / `n_hits_oi + eps` meets an fp32 tensor: an fp32 denominator
        const float denom = (float)((double)(norm != nullptr ? norm[0] : cnt) + 1e-9);
        out[0] = (float)s / denom;
        out[1] = cnt;
        out[2] = denom;
    }
}

// kHingeLpn lanes share a node: lane j takes the edges j, j + kHingeLpn, ... of both of the node's lists (the
// radius graph's degrees range from 0 to max_num_neighbors = 256: one thread per node left most of a wave
// waiting for its longest list), the lanes' sums are added in a fixed shuffle tree.
constexpr int kHingeLpn = 8;

template <int DP>   // padded dim (registers)
__global__ __launch_bounds__(kHingeTpb) void hinge_bwd_kernel(const gnntrk_hinge_args h, const gnntrk_graph_index gi, const float *g,
                                                           const float *denom, float *gx, int gx_stride, int accumulate) {
    const int64_t n = ((int64_t)blockIdx.x * kHingeTpb + threadIdx.x) / kHingeLpn;
    const int j = threadIdx.x % kHingeLpn;
    const bool live = n < h.n_nodes;
    const int64_t nn = live ? n : 0;   // (idle lanes of the last block walk node 0's lists and discard the result)
    const float scale = g[0] / denom[0];
    float acc[DP];
#pragma unroll
    for (int f = 0; f < DP; ++f) acc[f] = 0.f;
    const float *xn = h.x + nn * h.x_stride;
    // edges whose second endpoint (edges[1], the CSR target) is n: gradient -coef * (x[a] - x[n])
    for (int k = gi.rowptr_t[nn] + j; k < gi.rowptr_t[nn + 1]; k += kHingeLpn) {
        const int64_t a = gi.src[k];
        const HingeTerm t = hinge_term(h, a, nn);
        if (t.on && t.coef != 0.f) {
            const float *xa = h.x + a * h.x_stride;
#pragma unroll
            for (int f = 0; f < DP; ++f)
                if (f < h.dim) acc[f] -= t.coef * (xa[f] - xn[f]);
        }
    }
    // edges whose first endpoint is n (source-sorted view): gradient +coef * (x[n] - x[b])
    for (int m = gi.rowptr_s[nn] + j; m < gi.rowptr_s[nn + 1]; m += kHingeLpn) {
        const int64_t b = gi.tgt[gi.spos[m]];
        const HingeTerm t = hinge_term(h, nn, b);
        if (t.on && t.coef != 0.f) {
            const float *xb = h.x + b * h.x_stride;
#pragma unroll
            for (int f = 0; f < DP; ++f)
                if (f < h.dim) acc[f] += t.coef * (xn[f] - xb[f]);
        }
    }
#pragma unroll
    for (int f = 0; f < DP; ++f) {
        float v = acc[f];
        for (int m = 1; m < kHingeLpn; m <<= 1) v += __shfl_xor(v, m);
        acc[f] = v;
    }
    if (live && j == 0) {
        float *o = gx + n * gx_stride;
#pragma unroll
        for (int f = 0; f < DP; ++f)
            if (f < h.dim) o[f] = (accumulate ? o[f] : 0.f) + scale * acc[f];
    }
}

int hinge_check(const gnntrk_hinge_args *a, const char *who) {
    if (!a) return fail(GNNTRK_EINVAL, "hinge: NULL args");
    if (a->dim < 1 || a->dim > kHingeMaxDim || a->x_stride < a->dim || a->n_nodes < 0 || (a->n_nodes > 0 && !a->x))
        return fail(GNNTRK_EINVAL, "hinge: bad embedding (dim must be in [1, 32])");
    if (!(a->p > 0.f)) return fail(GNNTRK_EINVAL, "hinge: p must be positive");
    (void)who;
    return GNNTRK_OK;
}

int hinge_blocks(int64_t n_edges) {
    int64_t b = (n_edges + 4 * kHingeTpb - 1) / (4 * kHingeTpb);
    if (b > kHingeMaxBlocks) b = kHingeMaxBlocks;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace gnntrk

using namespace gnntrk;

extern "C" {

size_t gnntrk_hinge_workspace_bytes(int64_t n_edges) {
    (void)n_edges;
    return (size_t)kHingeMaxBlocks * 2 * sizeof(double);
}

int gnntrk_hinge_forward(const gnntrk_hinge_args *args, const int64_t *edges, int64_t n_edges, int64_t edge_stride,
                         const float *norm, float *out, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hinge_check(args, "hinge_forward");
    if (rc) return rc;
    if (n_edges < 0 || (n_edges > 0 && !edges) || !out || edge_stride < n_edges)
        return fail(GNNTRK_EINVAL, "hinge_forward: bad edge list");
    if (!workspace || workspace_bytes < gnntrk_hinge_workspace_bytes(n_edges) || ((uintptr_t)workspace & 7))
        return fail(GNNTRK_EINVAL, "hinge_forward: workspace too small or misaligned");
    double *part = reinterpret_cast<double *>(workspace);
    const int nb = n_edges > 0 ? hinge_blocks(n_edges) : 0;
    if (nb > 0) {
        hipLaunchKernelGGL(hinge_fwd_kernel, dim3(nb), dim3(kHingeTpb), 0, stream, *args, edges, n_edges, edge_stride, part);
        rc = check_launch("hinge_forward");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(hinge_finish_kernel, dim3(1), dim3(kHingeTpb), 0, stream, part, nb, norm, out);
    return check_launch("hinge_forward");
}

int gnntrk_hinge_backward(const gnntrk_hinge_args *args, const gnntrk_graph_index *index, const float *g, const float *denom,
                          float *gx, int32_t gx_stride, int32_t accumulate, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hinge_check(args, "hinge_backward");
    if (rc) return rc;
    if (!index || !g || !denom || !gx || gx_stride < args->dim) return fail(GNNTRK_EINVAL, "hinge_backward: bad argument");
    if (index->n_nodes != args->n_nodes) return fail(GNNTRK_EINVAL, "hinge_backward: the index is over another node set");
    if (args->n_nodes == 0) return GNNTRK_OK;
    const int nb = (int)((args->n_nodes * kHingeLpn + kHingeTpb - 1) / kHingeTpb);
#define GNNTRK_HINGE_BWD(DP_)                                                                                  \
    hipLaunchKernelGGL((hinge_bwd_kernel<DP_>), dim3(nb), dim3(kHingeTpb), 0, stream, *args, *index, g, denom, gx, \
                       (int)gx_stride, (int)accumulate)
    if (args->dim <= 4) GNNTRK_HINGE_BWD(4);
    else if (args->dim <= 8) GNNTRK_HINGE_BWD(8);
    else if (args->dim <= 16) GNNTRK_HINGE_BWD(16);
    else GNNTRK_HINGE_BWD(32);
#undef GNNTRK_HINGE_BWD
    return check_launch("hinge_backward");
}

}  // extern "C"
