// HBM-bound helpers of the bf16-storage path: fp32 -> padded bf16 row conversion (with
// optional row gather), deterministic CSR segment sums over bf16 rows (fp32 accumulate),
// row permutations of padded bf16 rows.  Rows are handled in 8-byte chunks of four bf16
// (the padding convention of gnntrk_mlp_forward_bf16).
#include "host_util.h"
#include "tile_bf16.h"

namespace gnntrk {
namespace {

constexpr int kTpb16 = 256;

int grid_for_threads(int64_t n) {
    int64_t g = ceil_div(n, kTpb16);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

// thread <-> (output row m, chunk ch)
__global__ __launch_bounds__(kTpb16) void rows_to_bf16_kernel(const float *__restrict__ in, int dim,
                                                               int in_stride,
                                                               const int32_t *__restrict__ idx,
                                                               int64_t n_rows, uint16_t *__restrict__ out,
                                                               int out_stride) {
    const int nch = (dim + 3) >> 2;
    const int64_t total = n_rows * nch;
    for (int64_t t = (int64_t)blockIdx.x * kTpb16 + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * kTpb16) {
        const int64_t m = t / nch;
        const int ch = (int)(t - m * nch);
        const float *src = in + (idx ? (int64_t)idx[m] : m) * in_stride + 4 * ch;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (4 * ch + r < dim) ? src[r] : 0.f;
        u32x2 o;
        o[0] = bf16x2_pack(v[0], v[1]);
        o[1] = bf16x2_pack(v[2], v[3]);
        *reinterpret_cast<u32x2 *>(out + m * out_stride + 4 * ch) = o;
    }
}

// dim == 4 with 16-byte aligned fp32 rows (the edge features: edge_classifier.py:97 reads
// `edge_attr[E, 4]`, gathered here into target-sorted order): ONE 16-byte load and one 8-byte
// store per row.  The general kernel above issues four predicated dword loads per row, which
// for a random gather quadruples the request count of a sector-bound kernel.
__global__ __launch_bounds__(kTpb16) void rows4_to_bf16_kernel(const float *__restrict__ in, int in_stride,
                                                                const int32_t *__restrict__ idx,
                                                                int64_t n_rows, uint16_t *__restrict__ out,
                                                                int out_stride) {
    for (int64_t m = (int64_t)blockIdx.x * kTpb16 + threadIdx.x; m < n_rows;
         m += (int64_t)gridDim.x * kTpb16) {
        const int64_t r = idx ? (int64_t)idx[m] : m;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(in + r * in_stride);
        u32x2 o;
        o[0] = bf16x2_pack(v[0], v[1]);
        o[1] = bf16x2_pack(v[2], v[3]);
        *reinterpret_cast<u32x2 *>(out + m * out_stride) = o;
    }
}

// Four lanes per segment: lane j of the group sums rows k0 + j, k0 + j + 4, ... (whole
// rows: 8 or 16 bytes per load), the group combines with two xor-shuffles, lane 0 rounds
// and stores.  Neighbouring lanes read neighbouring rows, so a CSR-ordered input streams
// through full sectors; with `pos` (source-sorted folds) the rows are random 8/16-byte
// gathers and the kernel is sector-bound.  The order of the fp32 additions is fixed.
template <int NCH, bool WIDE>  // WIDE: rows are 16-byte aligned -> one 16-byte load per chunk pair
__global__ __launch_bounds__(kTpb16) void segment_sum_bf16_kernel(
    const uint16_t *__restrict__ rows, int dim, int row_stride, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ pos, int64_t n_seg, uint16_t *__restrict__ out, int out_stride,
    const uint16_t *__restrict__ addend, int addend_stride) {
    const int j = threadIdx.x & 3;
    for (int64_t n = ((int64_t)blockIdx.x * kTpb16 + threadIdx.x) >> 2; n < ((n_seg + 63) & ~(int64_t)63);
         n += ((int64_t)gridDim.x * kTpb16) >> 2) {
        float s[4 * NCH];
#pragma unroll
        for (int i = 0; i < 4 * NCH; ++i) s[i] = 0.f;
        const bool on = n < n_seg;
        const int32_t k0 = on ? rowptr[n] : 0, k1 = on ? rowptr[n + 1] : 0;
        int32_t k = k0 + j;
        if (WIDE && NCH == 2) {
            // two rows per step: both 16-byte loads are in flight before the first is used (the walk
            // is latency bound: ~13 rows per segment over four lanes); same order of the additions
            // four rows per step first (16 rows of a segment across the four lanes: the mean degree of the
            // default graphs in one round of loads), then two, then one
            for (; k + 12 < k1; k += 16) {
                int64_t r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = pos ? (int64_t)pos[k + 4 * u] : (int64_t)(k + 4 * u);
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4 *>(rows + r[u] * row_stride);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s[2 * w + 0] += bf16_lo(v[u][w]);
                        s[2 * w + 1] += bf16_hi(v[u][w]);
                    }
                }
            }
            for (; k + 4 < k1; k += 8) {
                const int64_t ra = pos ? (int64_t)pos[k] : (int64_t)k, rb = pos ? (int64_t)pos[k + 4] : (int64_t)(k + 4);
                const u32x4 va = *reinterpret_cast<const u32x4 *>(rows + ra * row_stride);
                const u32x4 vb = *reinterpret_cast<const u32x4 *>(rows + rb * row_stride);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[2 * w + 0] += bf16_lo(va[w]);
                    s[2 * w + 1] += bf16_hi(va[w]);
                }
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[2 * w + 0] += bf16_lo(vb[w]);
                    s[2 * w + 1] += bf16_hi(vb[w]);
                }
            }
        }
        for (; k < k1; k += 4) {
            const int64_t r = pos ? (int64_t)pos[k] : (int64_t)k;
            const uint16_t *p = rows + r * row_stride;
            if (WIDE) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch += 2) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(p + 4 * ch);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s[4 * ch + 2 * w + 0] += bf16_lo(v[w]);
                        s[4 * ch + 2 * w + 1] += bf16_hi(v[w]);
                    }
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const u32x2 v = *reinterpret_cast<const u32x2 *>(p + 4 * ch);
                    s[4 * ch + 0] += bf16_lo(v[0]);
                    s[4 * ch + 1] += bf16_hi(v[0]);
                    s[4 * ch + 2] += bf16_lo(v[1]);
                    s[4 * ch + 3] += bf16_hi(v[1]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * NCH; ++i) {
            s[i] += __shfl_xor(s[i], 1);
            s[i] += __shfl_xor(s[i], 2);
        }
        if (on && j == 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int d = dim - 4 * ch;
                if (addend) {   // (one more fp32 term before the single rounding)
                    const u32x2 a = *reinterpret_cast<const u32x2 *>(addend + n * addend_stride + 4 * ch);
                    s[4 * ch + 0] += bf16_lo(a[0]);
                    s[4 * ch + 1] += bf16_hi(a[0]);
                    s[4 * ch + 2] += bf16_lo(a[1]);
                    s[4 * ch + 3] += bf16_hi(a[1]);
                }
                u32x2 o;
                o[0] = bf16x2_pack(d >= 1 ? s[4 * ch] : 0.f, d >= 2 ? s[4 * ch + 1] : 0.f);
                o[1] = bf16x2_pack(d >= 3 ? s[4 * ch + 2] : 0.f, d >= 4 ? s[4 * ch + 3] : 0.f);
                *reinterpret_cast<u32x2 *>(out + n * out_stride + 4 * ch) = o;
            }
        }
    }
}

// 16-byte rows in CSR order (the folds of the per-edge gradients of gathered node rows - target side, and
// source side after the backward kernel's permuted store): the same four-lanes-per-segment sums as above IN THE
// SAME ORDER (bit-identical), but the rows reach the lanes through a wave-private LDS image that the whole wave
// fills with coalesced 16-byte loads, four per lane in flight - the walk above waits for one dependent round
// trip to HBM per 4-16 rows of a segment.  A wave owns 16 consecutive segments at a time and stages their rows
// in chunks of 256.
#ifndef GNNTRK_PAIR8_NT
#define GNNTRK_PAIR8_NT 0   // (measured, round 6: 125 -> 166 us per 64 M rows - the pairs shared by neighbouring segments must stay cached)
#endif
#ifndef GNNTRK_SEGSUM_NT
#define GNNTRK_SEGSUM_NT 1   // (non-temporal loads of the once-read gradient rows: 231-235 -> 225 us per 64 M rows, round 6)
#endif
#ifndef GNNTRK_SEGSUM_STREAM
#define GNNTRK_SEGSUM_STREAM 1
#endif
constexpr int kStreamChunk = 256;   // rows per staged chunk (4 KB per wave)
__global__ __launch_bounds__(kTpb16) void segment_sum_bf16_stream16_kernel(
    const uint16_t *__restrict__ rows, int dim, const int32_t *__restrict__ rowptr, int64_t n_seg,
    uint16_t *__restrict__ out, int out_stride, const uint16_t *__restrict__ addend, int addend_stride) {
    __shared__ __attribute__((aligned(16))) u32x4 s_buf[kTpb16 / 64][kStreamChunk];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 3, sg = lane >> 2;
    const int64_t n_groups = (n_seg + 15) >> 4;
    const u32x4 *r16 = reinterpret_cast<const u32x4 *>(rows);
    for (int64_t grp = (int64_t)blockIdx.x * (kTpb16 / 64) + wv; grp < n_groups;
         grp += (int64_t)gridDim.x * (kTpb16 / 64)) {
        const int64_t n = grp * 16 + sg, n_last = grp * 16 + 16 < n_seg ? grp * 16 + 16 : n_seg;
        const bool on = n < n_seg;
        const int32_t k0 = on ? rowptr[n] : 0, k1 = on ? rowptr[n + 1] : 0;
        const int32_t r0 = rowptr[grp * 16], r1 = rowptr[n_last];   // (wave-uniform)
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.f;
        int32_t k = k0 + j;
        for (int32_t c0 = r0; c0 < r1; c0 += kStreamChunk) {
            u32x4 v[kStreamChunk / 64];
#pragma unroll
            for (int u = 0; u < kStreamChunk / 64; ++u) {
                const int32_t r = c0 + lane + 64 * u;
#if GNNTRK_SEGSUM_NT   // (A/B: non-temporal loads of the once-read rows - measured, round 6: see DESIGN.md 4.4)
                v[u] = r < r1 ? __builtin_nontemporal_load(r16 + r) : u32x4{0u, 0u, 0u, 0u};
#else
                v[u] = r < r1 ? r16[r] : u32x4{0u, 0u, 0u, 0u};
#endif
            }
            lds_wave_order();   // (the previous chunk's reads are issued before these writes)
#pragma unroll
            for (int u = 0; u < kStreamChunk / 64; ++u) s_buf[wv][lane + 64 * u] = v[u];
            lds_wave_order();
            const int32_t lim = k1 < c0 + kStreamChunk ? k1 : c0 + kStreamChunk;
            for (; k + 4 < lim; k += 8) {
                const u32x4 va = s_buf[wv][k - c0], vb = s_buf[wv][k + 4 - c0];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[2 * w + 0] += bf16_lo(va[w]);
                    s[2 * w + 1] += bf16_hi(va[w]);
                }
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[2 * w + 0] += bf16_lo(vb[w]);
                    s[2 * w + 1] += bf16_hi(vb[w]);
                }
            }
            for (; k < lim; k += 4) {
                const u32x4 va = s_buf[wv][k - c0];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[2 * w + 0] += bf16_lo(va[w]);
                    s[2 * w + 1] += bf16_hi(va[w]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s[i] += __shfl_xor(s[i], 1);
            s[i] += __shfl_xor(s[i], 2);
        }
        if (on && j == 0) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int d = dim - 4 * ch;
                if (addend) {
                    const u32x2 a = *reinterpret_cast<const u32x2 *>(addend + n * addend_stride + 4 * ch);
                    s[4 * ch + 0] += bf16_lo(a[0]);
                    s[4 * ch + 1] += bf16_hi(a[0]);
                    s[4 * ch + 2] += bf16_lo(a[1]);
                    s[4 * ch + 3] += bf16_hi(a[1]);
                }
                u32x2 o;
                o[0] = bf16x2_pack(d >= 1 ? s[4 * ch] : 0.f, d >= 2 ? s[4 * ch + 1] : 0.f);
                o[1] = bf16x2_pack(d >= 3 ? s[4 * ch + 2] : 0.f, d >= 4 ? s[4 * ch + 3] : 0.f);
                *reinterpret_cast<u32x2 *>(out + n * out_stride + 4 * ch) = o;
            }
        }
    }
}

// The aggregation of the interaction network (interaction_network.py:36, aggr="add": the edge
// embeddings e~ [E, 4] summed per target node) - 8-byte rows, contiguous, CSR order.  8-byte
// loads stream at ~3.2 TB/s on this chip, 16-byte ones at 4+: every lane loads an ALIGNED PAIR
// of rows (2p, 2p + 1) and keeps the rows inside its segment [k0, k1) (the pairs at the two ends
// of a segment are shared with the neighbouring segments and come from cache).  Four lanes per
// segment as above: lane j takes pairs p0 + j, p0 + j + 4, ...; fixed order of the additions.
__global__ __launch_bounds__(kTpb16) void segment_sum_bf16_pair8_kernel(
    const uint16_t *__restrict__ rows, int dim, const int32_t *__restrict__ rowptr, int64_t n_seg,
    uint16_t *__restrict__ out, int out_stride) {
    const int j = threadIdx.x & 3;
    const int32_t n_rows = rowptr[n_seg];
    for (int64_t n = ((int64_t)blockIdx.x * kTpb16 + threadIdx.x) >> 2; n < ((n_seg + 63) & ~(int64_t)63);
         n += ((int64_t)gridDim.x * kTpb16) >> 2) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const bool on = n < n_seg;
        const int32_t k0 = on ? rowptr[n] : 0, k1 = on ? rowptr[n + 1] : 0;
        if (k1 > k0) {
            const int32_t p1 = (k1 - 1) >> 1;
            auto load_pair = [&](int32_t p) {
                u32x4 v;
                if (2 * p + 1 < n_rows) {
#if GNNTRK_PAIR8_NT
                    v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + (int64_t)p * 8));
#else
                    v = *reinterpret_cast<const u32x4 *>(rows + (int64_t)p * 8);
#endif
                } else {  // the odd last row of the whole tensor: nothing may be read behind it
                    const u32x2 h = *reinterpret_cast<const u32x2 *>(rows + (int64_t)p * 8);
                    v[0] = h[0], v[1] = h[1], v[2] = 0u, v[3] = 0u;
                }
                return v;
            };
            auto add_pair = [&](int32_t p, const u32x4 &v) {
                const int32_t r = 2 * p;
                if (r >= k0) {
                    s[0] += bf16_lo(v[0]);
                    s[1] += bf16_hi(v[0]);
                    s[2] += bf16_lo(v[1]);
                    s[3] += bf16_hi(v[1]);
                }
                if (r + 1 < k1) {
                    s[0] += bf16_lo(v[2]);
                    s[1] += bf16_hi(v[2]);
                    s[2] += bf16_lo(v[3]);
                    s[3] += bf16_hi(v[3]);
                }
            };
            int32_t p = (k0 >> 1) + j;
            for (; p + 4 <= p1; p += 8) {   // two pairs per lane in flight (a segment of the default graphs: one round)
                const u32x4 va = load_pair(p), vb = load_pair(p + 4);
                add_pair(p, va);
                add_pair(p + 4, vb);
            }
            for (; p <= p1; p += 4) add_pair(p, load_pair(p));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] += __shfl_xor(s[i], 1);
            s[i] += __shfl_xor(s[i], 2);
        }
        if (on && j == 0) {
            u32x2 o;
            o[0] = bf16x2_pack(dim >= 1 ? s[0] : 0.f, dim >= 2 ? s[1] : 0.f);
            o[1] = bf16x2_pack(dim >= 3 ? s[2] : 0.f, dim >= 4 ? s[3] : 0.f);
            *reinterpret_cast<u32x2 *>(out + n * out_stride) = o;
        }
    }
}

// gather: out[m] = in[idx[m]];  scatter: out[idx[m]] = in[m]   (whole chunks are copied)
__global__ __launch_bounds__(kTpb16) void permute_rows_bf16_kernel(
    const uint16_t *__restrict__ in, int nch, int in_stride, const int32_t *__restrict__ idx,
    int64_t n_rows, uint16_t *__restrict__ out, int out_stride, int scatter) {
    const int64_t total = n_rows * nch;
    for (int64_t t = (int64_t)blockIdx.x * kTpb16 + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * kTpb16) {
        const int64_t m = t / nch;
        const int ch = (int)(t - m * nch);
        const int64_t j = idx[m];
        const int64_t src = scatter ? m : j, dst = scatter ? j : m;
        *reinterpret_cast<u32x2 *>(out + dst * out_stride + 4 * ch) =
            *reinterpret_cast<const u32x2 *>(in + src * in_stride + 4 * ch);
    }
}

bool rows_ok(const void *p, int dim, int stride) {
    return p && dim >= 1 && stride % 4 == 0 && stride >= (dim + 3) / 4 * 4 && ((uintptr_t)p & 7) == 0;
}

}  // namespace

int rows_to_bf16_launch(const float *in, int dim, int in_stride, const int32_t *idx, int64_t n_rows,
                        uint16_t *out, int out_stride, hipStream_t stream) {
    if (n_rows == 0) return GNNTRK_OK;  // empty tensors may carry NULL pointers
    if (!in || dim < 1 || in_stride < dim || n_rows < 0 || !rows_ok(out, dim, out_stride))
        return fail(GNNTRK_EINVAL, "rows_to_bf16: bad argument");
    const int nch = (dim + 3) / 4;
    if (dim == 4 && in_stride % 4 == 0 && ((uintptr_t)in & 15) == 0)
        hipLaunchKernelGGL(rows4_to_bf16_kernel, dim3(grid_for_threads(n_rows)), dim3(kTpb16), 0, stream, in,
                           in_stride, idx, n_rows, out, out_stride);
    else
        hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid_for_threads(n_rows * nch)), dim3(kTpb16), 0, stream,
                           in, dim, in_stride, idx, n_rows, out, out_stride);
    return check_launch("rows_to_bf16");
}

int segment_sum_bf16_launch(const uint16_t *rows, int dim, int row_stride, const int32_t *rowptr,
                            const int32_t *pos, int64_t n_seg, uint16_t *out, int out_stride,
                            const uint16_t *addend, int addend_stride, hipStream_t stream) {
    if (n_seg == 0) return GNNTRK_OK;
    if (addend && !rows_ok(addend, dim, addend_stride)) return fail(GNNTRK_EINVAL, "segment_sum_bf16: bad addend rows");
    // rows may be NULL when there are no rows at all (every segment empty)
    if (!rowptr || n_seg < 0 || (rows && !rows_ok(rows, dim, row_stride)) || !rows_ok(out, dim, out_stride))
        return fail(GNNTRK_EINVAL, "segment_sum_bf16: bad argument");
    if (dim > 16) {
        // rows wider than 16 features (the 40-wide edge embeddings of GraphConstructionResIN): one pass per
        // block of 16 columns - same strides, shifted base pointers (32 bytes: the alignment classes stay)
        for (int c0 = 0; c0 < dim; c0 += 16) {
            const int rc = segment_sum_bf16_launch(rows ? rows + c0 : rows, dim - c0 < 16 ? dim - c0 : 16, row_stride, rowptr,
                                                   pos, n_seg, out + c0, out_stride, addend ? addend + c0 : addend,
                                                   addend_stride, stream);
            if (rc) return rc;
        }
        return GNNTRK_OK;
    }
    const int nch = (dim + 3) / 4;
    const int grid = grid_for_threads(n_seg * 4);
    if (nch == 1 && row_stride == 4 && !pos && rows && !addend && ((uintptr_t)rows & 15) == 0) {
        // contiguous 8-byte rows in CSR order (the message aggregation): aligned 16-byte pair loads
        hipLaunchKernelGGL(segment_sum_bf16_pair8_kernel, dim3(grid), dim3(kTpb16), 0, stream, rows, dim, rowptr,
                           n_seg, out, out_stride);
        return check_launch("segment_sum_bf16");
    }
    const bool wide = nch % 2 == 0 && row_stride % 8 == 0 && ((uintptr_t)rows & 15) == 0;
    if (GNNTRK_SEGSUM_STREAM && nch == 2 && wide && row_stride == 8 && !pos && rows) {
        // contiguous 16-byte rows in CSR order (the gradient folds): rows staged through LDS, same sums
        hipLaunchKernelGGL(segment_sum_bf16_stream16_kernel, dim3(grid), dim3(kTpb16), 0, stream, rows, dim, rowptr,
                           n_seg, out, out_stride, addend, addend_stride);
        return check_launch("segment_sum_bf16");
    }
#define GNNTRK_SEGSUM16(N)                                                                              \
    if (nch == N) {                                                                                     \
        if (wide && N % 2 == 0)                                                                         \
            hipLaunchKernelGGL((segment_sum_bf16_kernel<N, (N % 2 == 0)>), dim3(grid), dim3(kTpb16), 0, stream, \
                               rows, dim, row_stride, rowptr, pos, n_seg, out, out_stride, addend, addend_stride); \
        else                                                                                            \
            hipLaunchKernelGGL((segment_sum_bf16_kernel<N, false>), dim3(grid), dim3(kTpb16), 0, stream, rows, dim, \
                               row_stride, rowptr, pos, n_seg, out, out_stride, addend, addend_stride); \
    }
    GNNTRK_SEGSUM16(1) GNNTRK_SEGSUM16(2) GNNTRK_SEGSUM16(3) GNNTRK_SEGSUM16(4)
#undef GNNTRK_SEGSUM16
    return check_launch("segment_sum_bf16");
}

int permute_rows_bf16_launch(const uint16_t *in, int dim, int in_stride, const int32_t *idx, int64_t n_rows,
                             uint16_t *out, int out_stride, int scatter, hipStream_t stream) {
    if (n_rows == 0) return GNNTRK_OK;
    if (!idx || n_rows < 0 || !rows_ok(in, dim, in_stride) || !rows_ok(out, dim, out_stride))
        return fail(GNNTRK_EINVAL, "permute_rows_bf16: bad argument");
    const int nch = (dim + 3) / 4;
    hipLaunchKernelGGL(permute_rows_bf16_kernel, dim3(grid_for_threads(n_rows * nch)), dim3(kTpb16), 0, stream,
                       in, nch, in_stride, idx, n_rows, out, out_stride, scatter);
    return check_launch("permute_rows_bf16");
}

// ------------------------------------------------------------------ carries + gate of the in-kernel target fold
// gnntrk_gfold (include/gnntrk.h): a unit of the backward kernel (32 consecutive rows) writes the part of a run of
// equal target ids that began in an EARLIER unit to carry[unit] (8 bf16).  A node's rows are [rowptr[n], rowptr[n + 1]),
// so the units it reaches after its first are known from the row pointers: one thread per node adds their carry rows
// in unit order onto the node's row (fixed order, one rounding) and applies the relu' gate of the folded segment
// (x: the segment's own input rows) - once per node instead of once per edge.
__global__ __launch_bounds__(kTpb16) void fold_finish_bf16_kernel(uint16_t *__restrict__ out, int out_stride, int64_t n_nodes,
                                                                  const int32_t *__restrict__ rowptr,
                                                                  const uint16_t *__restrict__ carry, int64_t n_units,
                                                                  const uint16_t *__restrict__ x, int x_stride) {
    for (int64_t n = (int64_t)blockIdx.x * kTpb16 + threadIdx.x; n < n_nodes; n += (int64_t)gridDim.x * kTpb16) {
        const int32_t r0 = rowptr[n], r1 = rowptr[n + 1];
        if (r1 <= r0) continue;   // (no row: the caller's zero stays)
        const int64_t u0 = r0 >> 5, u1 = (int64_t)(r1 - 1) >> 5;
        if (u1 == u0 && !x) continue;
        uint16_t *row = out + n * out_stride;
        const uint4 rv = *reinterpret_cast<const uint4 *>(row);
        uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
        if (u1 > u0) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = bf16_lo(rw[i]);
                acc[2 * i + 1] = bf16_hi(rw[i]);
            }
            for (int64_t u = u0 + 1; u <= u1 && u < n_units; ++u) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(carry + 8 * u);
                const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += bf16_lo(cw[i]);
                    acc[2 * i + 1] += bf16_hi(cw[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) rw[i] = bf16x2_pack(acc[2 * i], acc[2 * i + 1]);
        }
        if (x) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(x + n * x_stride);
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // keep a half where the input's half is a positive bf16
                const uint32_t lo = ((int16_t)(xw[i] & 0xffffu) > 0) ? 0x0000ffffu : 0u;
                const uint32_t hi = ((int16_t)(xw[i] >> 16) > 0) ? 0xffff0000u : 0u;
                rw[i] &= lo | hi;
            }
        }
        *reinterpret_cast<uint4 *>(row) = uint4{rw[0], rw[1], rw[2], rw[3]};
    }
}

int fold_finish_bf16_launch(uint16_t *out, int out_stride, int64_t n_nodes, const int32_t *rowptr, const uint16_t *carry,
                            int64_t n_units, const uint16_t *x, int x_stride, hipStream_t stream) {
    if (n_units == 0 || n_nodes == 0) return GNNTRK_OK;
    if (!out || !carry || !rowptr || n_units < 0 || n_nodes < 0 || out_stride != 8 || ((uintptr_t)out & 15) != 0 ||
        ((uintptr_t)carry & 15) != 0 || (x && (x_stride != 8 || ((uintptr_t)x & 15) != 0)))
        return fail(GNNTRK_EINVAL, "fold_finish_bf16: bad argument (rows of 8 bf16, 16-byte aligned)");
    hipLaunchKernelGGL(fold_finish_bf16_kernel, dim3(grid_for_threads(n_nodes)), dim3(kTpb16), 0, stream, out, out_stride,
                       n_nodes, rowptr, carry, n_units, x, x_stride);
    return check_launch("fold_finish_bf16");
}

}  // namespace gnntrk
