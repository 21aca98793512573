// HBM-bound helper kernels: deterministic CSR segment sums (aggregation and all
// gradient scatters), row permutations, axpby, and the BCE loss reductions.
#include "host_util.h"

namespace gnntrk {

constexpr int kTpb = 256;

static int stream_grid(int64_t n) {
    int64_t g = ceil_div(n, kTpb);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

// thread <-> (segment n, feature f); consecutive threads read consecutive floats of
// the same CSR row, consecutive segments read adjacent row ranges: coalesced when
// pos == NULL, L2-resident gathers otherwise.  Summation in CSR order.
__global__ __launch_bounds__(kTpb) void segment_sum_kernel(
    const float *__restrict__ rows, int dim, int row_stride, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ pos, int64_t n_seg, float *__restrict__ out, int out_stride,
    int accumulate) {
    const int64_t total = n_seg * dim;
    for (int64_t t = (int64_t)blockIdx.x * kTpb + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * kTpb) {
        const int64_t n = t / dim;
        const int f = (int)(t - n * dim);
        const int32_t k0 = rowptr[n], k1 = rowptr[n + 1];
        float s = 0.f;
        if (pos) {
            for (int32_t k = k0; k < k1; ++k) s += rows[(int64_t)pos[k] * row_stride + f];
        } else {
            for (int32_t k = k0; k < k1; ++k) s += rows[(int64_t)k * row_stride + f];
        }
        float *o = out + n * out_stride + f;
        *o = accumulate ? *o + s : s;
    }
}

// dim % 4 == 0 with 16-byte aligned rows (the 40-wide stacks of the graph-construction models): thread <->
// (segment, four features), 16-byte loads, four rows in flight per thread; the sums are taken in the same CSR
// order as above, so the two kernels agree bit for bit
__global__ __launch_bounds__(kTpb) void segment_sum_v4_kernel(
    const float4 *__restrict__ rows, int quads, int row_stride4, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ pos, int64_t n_seg, float4 *__restrict__ out, int out_stride4,
    int accumulate) {
    const int64_t total = n_seg * quads;
    for (int64_t t = (int64_t)blockIdx.x * kTpb + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * kTpb) {
        const int64_t n = t / quads;
        const int q = (int)(t - n * quads);
        const int32_t k0 = rowptr[n], k1 = rowptr[n + 1];
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int32_t k = k0;
        for (; k + 4 <= k1; k += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = rows[(int64_t)(pos ? pos[k + u] : k + u) * row_stride4 + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
                s.z += v[u].z;
                s.w += v[u].w;
            }
        }
        for (; k < k1; ++k) {
            const float4 v = rows[(int64_t)(pos ? pos[k] : k) * row_stride4 + q];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        float4 *o = out + n * out_stride4 + q;
        if (accumulate) {
            const float4 c = *o;
            s.x = c.x + s.x;
            s.y = c.y + s.y;
            s.z = c.z + s.z;
            s.w = c.w + s.w;
        }
        *o = s;
    }
}

// dim == 4 (the default edge width): one thread per segment, 16-B row loads
__global__ __launch_bounds__(kTpb) void segment_sum4_kernel(
    const float *__restrict__ rows, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ pos, int64_t n_seg, float *__restrict__ out, int out_stride,
    int accumulate) {
    const float4 *r4 = reinterpret_cast<const float4 *>(rows);
    for (int64_t n = (int64_t)blockIdx.x * kTpb + threadIdx.x; n < n_seg;
         n += (int64_t)gridDim.x * kTpb) {
        const int32_t k0 = rowptr[n], k1 = rowptr[n + 1];
        float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
        for (int32_t k = k0; k < k1; ++k) {
            const float4 v = r4[pos ? (int64_t)pos[k] : (int64_t)k];
            sx += v.x;
            sy += v.y;
            sz += v.z;
            sw += v.w;
        }
        float *o = out + n * out_stride;
        if (accumulate) {
            sx += o[0];
            sy += o[1];
            sz += o[2];
            sw += o[3];
        }
        o[0] = sx;
        o[1] = sy;
        o[2] = sz;
        o[3] = sw;
    }
}

__global__ __launch_bounds__(kTpb) void permute_rows_kernel(const float *__restrict__ in, int dim,
                                                            int in_stride,
                                                            const int32_t *__restrict__ idx,
                                                            int64_t n_rows, float *__restrict__ out,
                                                            int out_stride, int scatter) {
    const int64_t total = n_rows * dim;
    for (int64_t t = (int64_t)blockIdx.x * kTpb + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * kTpb) {
        const int64_t m = t / dim;
        const int f = (int)(t - m * dim);
        const int64_t j = idx[m];
        if (scatter)
            out[j * out_stride + f] = in[m * in_stride + f];
        else
            out[m * out_stride + f] = in[j * in_stride + f];
    }
}

__global__ __launch_bounds__(kTpb) void axpby_kernel(float a, const float *__restrict__ x, float b,
                                                     const float *__restrict__ y,
                                                     const float *__restrict__ mask,
                                                     float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kTpb) {
        float v = a * x[i];
        if (y) v += b * y[i];
        if (mask && !(mask[i] > 0.f)) v = 0.f;
        out[i] = v;
    }
}

// ------------------------------------------------------------------------ BCE
__device__ __forceinline__ float bce_target(const float *y, const int64_t *src_node,
                                            const float *pt, float thld, int64_t e) {
    float t = y[e];
    if (thld > 0.f) t = (t != 0.f && pt[src_node[e]] > thld) ? 1.f : 0.f;
    return t;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kTpb / 64; ++i) s += sh[i];
    return s;  // valid on thread 0
}

__global__ __launch_bounds__(kTpb) void bce_partial_kernel(const float *__restrict__ w,
                                                           const float *__restrict__ y,
                                                           const int64_t *__restrict__ src_node,
                                                           const float *__restrict__ pt, float thld,
                                                           int64_t n, double *__restrict__ part) {
    __shared__ double sh[kTpb / 64];
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * kTpb) {
        const float t = bce_target(y, src_node, pt, thld, e);
        const float wi = w[e];
        const float lw = fmaxf(logf(wi), -100.f);
        const float l1w = fmaxf(log1pf(-wi), -100.f);
        acc += (double)(-(t * lw + (1.f - t) * l1w));
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(kTpb) void bce_final_kernel(const double *__restrict__ part, int n_part,
                                                         int64_t n, float *__restrict__ loss) {
    __shared__ double sh[kTpb / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kTpb) acc += part[i];
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)n);
}

__global__ __launch_bounds__(kTpb) void bce_bwd_kernel(const float *__restrict__ w,
                                                       const float *__restrict__ y,
                                                       const int64_t *__restrict__ src_node,
                                                       const float *__restrict__ pt, float thld,
                                                       int64_t n, const float *__restrict__ gscale,
                                                       float *__restrict__ gw) {
    const float gs = gscale[0] / (float)n;
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * kTpb) {
        const float t = bce_target(y, src_node, pt, thld, e);
        const float wi = w[e];
        gw[e] = gs * (wi - t) / fmaxf((1.f - wi) * wi, 1e-12f);
    }
}

// BCE of CSR-ordered edge weights against the 1-byte CSR-ordered labels the graph-index build
// carried along (gnntrk_graph_index_carry), forward AND the gradient for a unit upstream value in
// ONE pass: t = label != 0 [&& pt[src_csr[k]] > thld]; loss partial as above;
// gw_unit[k] = (1 / n) (w - t) / max((1 - w) w, 1e-12)  (= bce_bwd_kernel with gscale = 1).
__device__ __forceinline__ float bce_csr_one(float wi, float t, float gs, double &acc) {
    const float lw = fmaxf(logf(wi), -100.f);
    const float l1w = fmaxf(log1pf(-wi), -100.f);
    acc += (double)(-(t * lw + (1.f - t) * l1w));
    return gs * (wi - t) / fmaxf((1.f - wi) * wi, 1e-12f);
}
// four consecutive edges per thread and step (16-byte weight / gradient accesses, one 4-byte label
// word); `vec`: the three arrays are 16 / 4 / 16-byte aligned.  The loss partial of a thread sums its
// edges in ascending order whatever the access width, so both forms give the same bits.
__global__ __launch_bounds__(kTpb) void bce_csr_kernel(const float *__restrict__ w, const uint8_t *__restrict__ label,
                                                       const int32_t *__restrict__ src_csr,
                                                       const float *__restrict__ pt, float thld, int64_t n, bool vec,
                                                       double *__restrict__ part, float *__restrict__ gw_unit) {
    __shared__ double sh[kTpb / 64];
    double acc = 0.0;
    const float gs = 1.f / (float)n;
    for (int64_t e0 = ((int64_t)blockIdx.x * kTpb + threadIdx.x) * 4; e0 < n; e0 += (int64_t)gridDim.x * kTpb * 4) {
        float wi[4], t[4], g[4];
        const bool full = vec && e0 + 4 <= n;
        if (full) {
            const float4 wv = *reinterpret_cast<const float4 *>(w + e0);
            const uint32_t lv = *reinterpret_cast<const uint32_t *>(label + e0);
            wi[0] = wv.x; wi[1] = wv.y; wi[2] = wv.z; wi[3] = wv.w;
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = ((lv >> (8 * q)) & 0xffu) ? 1.f : 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = e0 + q < n;
                wi[q] = in ? w[e0 + q] : 0.5f;
                t[q] = (in && label[e0 + q]) ? 1.f : 0.f;
            }
        }
        if (thld > 0.f) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e0 + q < n) t[q] = (t[q] != 0.f && pt[src_csr[e0 + q]] > thld) ? 1.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a = 0.0;
            g[q] = bce_csr_one(wi[q], t[q], gs, a);
            if (e0 + q < n) acc += a;
        }
        if (gw_unit) {
            if (full) {
                *reinterpret_cast<float4 *>(gw_unit + e0) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (e0 + q < n) gw_unit[e0 + q] = g[q];
            }
        }
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// Labels of the edges in CSR (target-sorted) order, pt-falsified: the once-per-batch gather that
// lets the loss run on the CSR-ordered edge weights the classification head produces.
//   out[k] = y[perm[k]]                                          (thld <= 0)
//   out[k] = y[perm[k]] != 0 && pt[src_csr[k]] > thld ? 1 : 0    (falsify_low_pt_edges, ec.py:71-92)
template <class Y>  // float labels (what the losses take) or the dataset's 1-byte bool labels
__global__ __launch_bounds__(kTpb) void edge_targets_csr_kernel(const Y *__restrict__ y,
                                                                const int32_t *__restrict__ perm,
                                                                const int32_t *__restrict__ src_csr,
                                                                const float *__restrict__ pt, float thld, int64_t n,
                                                                float *__restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < n; k += (int64_t)gridDim.x * kTpb) {
        float t = (float)y[perm[k]];
        if (thld > 0.f) t = (t != 0.f && pt[src_csr[k]] > thld) ? 1.f : 0.f;
        out[k] = t;
    }
}

// ------------------------------------------------------------------ focal loss
// metrics/losses/ec.py:13-31: mean over edges of
//   -alpha * pw * (1-w)^gamma * t * log(w)  -  (1-alpha) * w^gamma * (1-t) * log(1-w)
// (no log clamp, as the reference).  haughty = 0 (EdgeWeightFocalLoss): t = falsified label,
// pw = the scalar pos_weight; haughty = 1 (HaughtyFocalLoss :153-183): t = the label as given,
// pw = the falsified label of the edge.
__device__ __forceinline__ float focal_pow(float x, float gamma) {
    if (gamma == 0.f) return 1.f;  // torch: pow(x, 0) = 1
    if (gamma == 1.f) return x;
    if (gamma == 2.f) return x * x;
    return powf(x, gamma);
}
// d/dx x^gamma
__device__ __forceinline__ float focal_dpow(float x, float gamma) {
    if (gamma == 0.f) return 0.f;
    if (gamma == 1.f) return 1.f;
    if (gamma == 2.f) return 2.f * x;
    return gamma * powf(x, gamma - 1.f);
}
__device__ __forceinline__ void focal_terms(const float *y, const int64_t *src_node, const float *pt, float thld,
                                            float pos_weight, int haughty, int64_t e, float &t, float &pw) {
    const float yf = bce_target(y, src_node, pt, thld, e);
    t = haughty ? y[e] : yf;
    pw = haughty ? yf : pos_weight;
}

__global__ __launch_bounds__(kTpb) void focal_partial_kernel(const float *__restrict__ w, const float *__restrict__ y,
                                                             const int64_t *__restrict__ src_node,
                                                             const float *__restrict__ pt, float thld, float alpha,
                                                             float gamma, float pos_weight, int haughty, int64_t n,
                                                             double *__restrict__ part) {
    __shared__ double sh[kTpb / 64];
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kTpb) {
        float t, pw;
        focal_terms(y, src_node, pt, thld, pos_weight, haughty, e, t, pw);
        const float p = w[e], q = 1.f - p;
        const float pos = -alpha * pw * focal_pow(q, gamma) * t * logf(p);
        const float neg = -(1.f - alpha) * focal_pow(p, gamma) * (1.f - t) * logf(q);
        acc += (double)(pos + neg);
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(kTpb) void focal_bwd_kernel(const float *__restrict__ w, const float *__restrict__ y,
                                                         const int64_t *__restrict__ src_node,
                                                         const float *__restrict__ pt, float thld, float alpha,
                                                         float gamma, float pos_weight, int haughty, int64_t n,
                                                         const float *__restrict__ gscale, float *__restrict__ gw) {
    const float gs = gscale[0] / (float)n;
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kTpb) {
        float t, pw;
        focal_terms(y, src_node, pt, thld, pos_weight, haughty, e, t, pw);
        const float p = w[e], q = 1.f - p;
        // d/dp [(1-p)^g log p] = -g (1-p)^(g-1) log p + (1-p)^g / p ;  d/dp [p^g log(1-p)] = g p^(g-1) log(1-p) - p^g / (1-p)
        const float dpos = -focal_dpow(q, gamma) * logf(p) + focal_pow(q, gamma) / p;
        const float dneg = focal_dpow(p, gamma) * logf(q) - focal_pow(p, gamma) / q;
        gw[e] = gs * (-alpha * pw * t * dpos - (1.f - alpha) * (1.f - t) * dneg);
    }
}

// ------------------------------------------------------------------ launchers
int segment_sum_launch(const float *rows, int dim, int row_stride, const int32_t *rowptr,
                       const int32_t *pos, int64_t n_seg, float *out, int out_stride,
                       int accumulate, hipStream_t stream) {
    if (n_seg == 0) return GNNTRK_OK;
    // rows may be NULL when there are no rows at all (every segment empty: zeros are written)
    if (!rowptr || !out || dim < 1 || row_stride < dim || out_stride < dim || n_seg < 0)
        return fail(GNNTRK_EINVAL, "segment_sum: bad argument");
    if (rows && dim == 4 && row_stride == 4 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {
        hipLaunchKernelGGL(segment_sum4_kernel, dim3(stream_grid(n_seg)), dim3(kTpb), 0, stream, rows,
                           rowptr, pos, n_seg, out, out_stride, accumulate);
        return check_launch("segment_sum");
    }
    if (rows && dim % 4 == 0 && row_stride % 4 == 0 && out_stride % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        hipLaunchKernelGGL(segment_sum_v4_kernel, dim3(stream_grid(n_seg * (dim / 4))), dim3(kTpb), 0, stream,
                           reinterpret_cast<const float4 *>(rows), dim / 4, row_stride / 4, rowptr, pos, n_seg,
                           reinterpret_cast<float4 *>(out), out_stride / 4, accumulate);
        return check_launch("segment_sum");
    }
    hipLaunchKernelGGL(segment_sum_kernel, dim3(stream_grid(n_seg * dim)), dim3(kTpb), 0, stream,
                       rows, dim, row_stride, rowptr, pos, n_seg, out, out_stride, accumulate);
    return check_launch("segment_sum");
}

int permute_rows_launch(const float *in, int dim, int in_stride, const int32_t *idx, int64_t n_rows,
                        float *out, int out_stride, int scatter, hipStream_t stream) {
    if (n_rows == 0) return GNNTRK_OK;  // empty tensors may carry NULL pointers
    if (!in || !idx || !out || dim < 1 || in_stride < dim || out_stride < dim || n_rows < 0)
        return fail(GNNTRK_EINVAL, "permute_rows: bad argument");
    hipLaunchKernelGGL(permute_rows_kernel, dim3(stream_grid(n_rows * dim)), dim3(kTpb), 0, stream,
                       in, dim, in_stride, idx, n_rows, out, out_stride, scatter);
    return check_launch("permute_rows");
}

int axpby_launch(float a, const float *x, float b, const float *y, const float *mask, float *out,
                 int64_t n, hipStream_t stream) {
    if (n == 0) return GNNTRK_OK;
    if (!x || !out || n < 0) return fail(GNNTRK_EINVAL, "axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(n)), dim3(kTpb), 0, stream, a, x, b, y, mask,
                       out, n);
    return check_launch("axpby");
}

static int bce_grid(int64_t n) {
    int g = stream_grid(n);
    return g > 1024 ? 1024 : g;
}

size_t bce_ws_bytes(int64_t n) {
    (void)n;
    return 1024 * sizeof(double);
}

int bce_forward_launch(const float *w, const float *y, const int64_t *src_node, const float *pt,
                       float thld, int64_t n, float *loss, void *ws, size_t ws_bytes,
                       hipStream_t stream) {
    if (!w || !y || !loss || n < 1) return fail(GNNTRK_EINVAL, "bce_forward: bad argument");
    if (thld > 0.f && (!src_node || !pt))
        return fail(GNNTRK_EINVAL, "bce_forward: pt threshold needs edge_index and pt");
    if (!ws || ws_bytes < bce_ws_bytes(n)) return fail(GNNTRK_EINVAL, "bce_forward: workspace too small");
    const int g = bce_grid(n);
    double *part = reinterpret_cast<double *>(ws);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(g), dim3(kTpb), 0, stream, w, y, src_node, pt, thld,
                       n, part);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(kTpb), 0, stream,
                       reinterpret_cast<const double *>(part), g, n, loss);
    return check_launch("bce_forward");
}

int bce_backward_launch(const float *w, const float *y, const int64_t *src_node, const float *pt,
                        float thld, int64_t n, const float *gscale, float *gw, hipStream_t stream) {
    if (!w || !y || !gscale || !gw || n < 1) return fail(GNNTRK_EINVAL, "bce_backward: bad argument");
    if (thld > 0.f && (!src_node || !pt))
        return fail(GNNTRK_EINVAL, "bce_backward: pt threshold needs edge_index and pt");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(stream_grid(n)), dim3(kTpb), 0, stream, w, y, src_node,
                       pt, thld, n, gscale, gw);
    return check_launch("bce_backward");
}

int bce_csr_launch(const float *w, const uint8_t *label, const int32_t *src_csr, const float *pt, float thld, int64_t n,
                   float *loss, float *gw_unit, void *ws, size_t ws_bytes, hipStream_t stream) {
    if (!w || !label || !loss || n < 1) return fail(GNNTRK_EINVAL, "bce_csr: bad argument");
    if (thld > 0.f && (!src_csr || !pt)) return fail(GNNTRK_EINVAL, "bce_csr: pt threshold needs the CSR source ids and pt");
    if (!ws || ws_bytes < bce_ws_bytes(n)) return fail(GNNTRK_EINVAL, "bce_csr: workspace too small");
    const int g = bce_grid(n);
    double *part = reinterpret_cast<double *>(ws);
    const bool vec = (((uintptr_t)w | (uintptr_t)gw_unit) & 15) == 0 && ((uintptr_t)label & 3) == 0;
    hipLaunchKernelGGL(bce_csr_kernel, dim3(g), dim3(kTpb), 0, stream, w, label, src_csr, pt, thld, n, vec, part,
                       gw_unit);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(kTpb), 0, stream, reinterpret_cast<const double *>(part), g, n,
                       loss);
    return check_launch("bce_csr");
}

int edge_targets_csr_launch(const void *y, int y_is_u8, const int32_t *perm, const int32_t *src_csr, const float *pt,
                            float thld, int64_t n, float *out, hipStream_t stream) {
    if (n == 0) return GNNTRK_OK;
    if (!y || !perm || !out || n < 0) return fail(GNNTRK_EINVAL, "edge_targets_csr: bad argument");
    if (thld > 0.f && (!src_csr || !pt))
        return fail(GNNTRK_EINVAL, "edge_targets_csr: pt threshold needs the CSR source ids and pt");
    if (y_is_u8)
        hipLaunchKernelGGL(edge_targets_csr_kernel<uint8_t>, dim3(stream_grid(n)), dim3(kTpb), 0, stream,
                           reinterpret_cast<const uint8_t *>(y), perm, src_csr, pt, thld, n, out);
    else
        hipLaunchKernelGGL(edge_targets_csr_kernel<float>, dim3(stream_grid(n)), dim3(kTpb), 0, stream,
                           reinterpret_cast<const float *>(y), perm, src_csr, pt, thld, n, out);
    return check_launch("edge_targets_csr");
}

int focal_forward_launch(const float *w, const float *y, const int64_t *src_node, const float *pt, float thld,
                         float alpha, float gamma, float pos_weight, int haughty, int64_t n, float *loss, void *ws,
                         size_t ws_bytes, hipStream_t stream) {
    if (!w || !y || !loss || n < 1) return fail(GNNTRK_EINVAL, "focal_forward: bad argument");
    if (!(gamma >= 0.f) || !(alpha >= 0.f && alpha <= 1.f)) return fail(GNNTRK_EINVAL, "focal_forward: bad alpha / gamma");
    if (thld > 0.f && (!src_node || !pt))
        return fail(GNNTRK_EINVAL, "focal_forward: pt threshold needs edge_index and pt");
    if (!ws || ws_bytes < bce_ws_bytes(n)) return fail(GNNTRK_EINVAL, "focal_forward: workspace too small");
    const int g = bce_grid(n);
    double *part = reinterpret_cast<double *>(ws);
    hipLaunchKernelGGL(focal_partial_kernel, dim3(g), dim3(kTpb), 0, stream, w, y, src_node, pt, thld, alpha, gamma,
                       pos_weight, haughty, n, part);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(kTpb), 0, stream, reinterpret_cast<const double *>(part), g, n,
                       loss);
    return check_launch("focal_forward");
}

int focal_backward_launch(const float *w, const float *y, const int64_t *src_node, const float *pt, float thld,
                          float alpha, float gamma, float pos_weight, int haughty, int64_t n, const float *gscale,
                          float *gw, hipStream_t stream) {
    if (!w || !y || !gscale || !gw || n < 1) return fail(GNNTRK_EINVAL, "focal_backward: bad argument");
    if (thld > 0.f && (!src_node || !pt))
        return fail(GNNTRK_EINVAL, "focal_backward: pt threshold needs edge_index and pt");
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(stream_grid(n)), dim3(kTpb), 0, stream, w, y, src_node, pt, thld, alpha,
                       gamma, pos_weight, haughty, n, gscale, gw);
    return check_launch("focal_backward");
}

}  // namespace gnntrk
