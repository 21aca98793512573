// fp32 fused gather-concat-MLP for shapes beyond the register-resident kernels of mlp.hip:
// in <= 128, hidden <= 128, out <= 48, two or three layers (reference models/mlp.py:18-62 applied to the
// concatenation of gathered segments, e.g. GraphConstructionResIN(hidden_dim=40) - a 120 -> 40 -> 40 -> 40
// relational model, models/graph_construction.py:136-219 - in the reference's own precision).  Same
// operator and argument structs as gnntrk_mlp_forward / _backward (include/gnntrk.h); epilogues NONE,
// RELU, RESIDUAL.
//
// Structure of resfcnn.hip (orientation, fragment packing, LDS staging of one layer's weights at a time,
// K = rows weight-gradient contractions through wave-private [feature][row] images, per-block partial
// sums reduced in a fixed order - see there), with three differences: the first layer's input is the
// gathered concatenation of up to GNNTRK_MAX_SEGS segments (a feature -> (segment, column) table in LDS,
// row ids staged per wave), the last layer has up to three output tiles, and the backward writes the
// per-row input-gradient slices of the segments that want one (the caller folds gathered rows with its
// deterministic segment sums, as for the narrow kernels).
//
// Trade-off against the narrow kernels: the forward leaves the hidden layers' pre-activations in `acts`
// ([n_layers - 1][n_rows][hidden_pad] floats) instead of the backward recomputing them - with 128-wide
// weight gradients one layer's accumulators fill the register file, so the backward is layer-outer and
// would have to recompute the chain once per layer.  N- or E-sized fp32 traffic of 4 hidden_pad bytes per
// row and layer: the price of the reference's precision at these widths (bf16 storage has its own fused
// kernels for them).
#include <cmath>
#include <cstring>

#include "host_util.h"
#include "tile_mlp.h"

namespace gnntrk {
namespace {

constexpr int kWMaxIn = 128, kWMaxHidden = 128, kWMaxOut = 48;
constexpr int kWMaxOT = kWMaxOut / 16;
constexpr int kWLd = 20;   // leading dim of a [feature][row] staging image

__host__ __device__ inline int w_tiles(int d) { return (d + 15) / 16; }

// ---- fragment packing (as resfcnn.hip): dst[(to * KS + ks) * 64 + lane] = Mat[16 to + c][16 (ks >> 2) + 4 g + (ks & 3)]
struct WPackJob {
    const float *W;
    float *dst;
    int32_t rows, cols, rt, kt, ld, transposed;
};
struct WPackArgs {
    WPackJob job[6];
    int32_t n_jobs;
};
__global__ __launch_bounds__(256) void mlpw_pack_kernel(const WPackArgs a) {
    if ((int)blockIdx.y >= a.n_jobs) return;
    const WPackJob j = a.job[blockIdx.y];
    const int KS = 4 * j.kt, n = j.rt * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int fr = i >> 6, l = i & 63, to = fr / KS, ks = fr - to * KS, g = l >> 4, c = l & 15;
        const int r = 16 * to + c, k = 16 * (ks >> 2) + 4 * g + (ks & 3);
        float v = 0.f;
        if (r < j.rows && k < j.cols) v = j.transposed ? j.W[(int64_t)k * j.ld + r] : j.W[(int64_t)r * j.ld + k];
        j.dst[i] = v;
    }
}

struct WArgs {
    gnntrk_seg seg[GNNTRK_MAX_SEGS];
    gnntrk_gseg gseg[GNNTRK_MAX_SEGS];
    gnntrk_gterm gout[3];
    const float *frag[3], *fragT[3], *bias[3];   // layer 0 .. n_layers - 1
    const float *res;
    float *out;
    float *acts;
    const float *fwd_out;
    float *gstream, *part;
    int64_t n_rows;
    int32_t n_seg, n_layers, epilogue, n_gout;
    int32_t in_dim, hidden, out_dim;
    int32_t res_stride, out_stride, part_total, want_dx;
    int32_t vec4;    // every 4-feature group of the input lies in one segment, 16-byte aligned: float4 I/O
    int32_t vec4o;   // output side likewise (out / res / upstream terms: widths and strides multiples of 4, aligned)
    float ca, cb;
};

// feature -> (segment, column) of the concatenated input; staged once per workgroup
struct WFeatTab {
    const float *ptr[kWMaxIn];   // segment base + column
    float *gptr[kWMaxIn];        // gradient slice base + column, or NULL
    int32_t stride[kWMaxIn], gstride[kWMaxIn];
    int8_t seg[kWMaxIn], relu[kWMaxIn];
    const int32_t *idx[GNNTRK_MAX_SEGS];
};
__device__ inline void w_build_tab(WFeatTab &tb, const WArgs &a, int tid, int nthreads) {
    for (int f = tid; f < kWMaxIn; f += nthreads) {
        int j = 0, f0 = 0;
        while (j < a.n_seg && f >= f0 + a.seg[j].dim) {
            f0 += a.seg[j].dim;
            ++j;
        }
        const bool on = j < a.n_seg && f < a.in_dim;
        tb.ptr[f] = on ? a.seg[j].ptr + (f - f0) : nullptr;
        tb.stride[f] = on ? a.seg[j].stride : 0;
        tb.seg[f] = (int8_t)(on ? j : 0);
        tb.relu[f] = (int8_t)(on ? (a.seg[j].relu != 0) : 0);
        tb.gptr[f] = (on && a.gseg[j].ptr) ? a.gseg[j].ptr + (f - f0) : nullptr;
        tb.gstride[f] = on ? a.gseg[j].stride : 0;
    }
    for (int j = tid; j < GNNTRK_MAX_SEGS; j += nthreads) tb.idx[j] = j < a.n_seg ? a.seg[j].idx : nullptr;
}

__device__ __forceinline__ void w_stage(float *s_frag, const float *src, int n_floats, float *s_bias, const float *bias,
                                        int n_bias, int n_bias_pad, int tid) {
    __syncthreads();
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(s_frag);
    for (int i = tid; i < n_floats / 4; i += kBlock) d4[i] = s4[i];
    for (int i = tid; i < n_bias_pad; i += kBlock) s_bias[i] = (bias != nullptr && i < n_bias) ? bias[i] : 0.f;
    __syncthreads();
}

// the gathered, concatenated (and ReLU'd) input rows of one tile in accumulator layout.  `s_rid`: this
// wave's [segment][16] row-id scratch.
template <int KT>
__device__ __forceinline__ void w_load_input(const WArgs &a, const WFeatTab &tb, int32_t *s_rid, int64_t row, bool valid,
                                             int g, int c, int kti, f32x4 (&xin)[KT], int t0 = 0) {   // tiles t0 .. t0 + kti - 1
    lds_wave_order();   // (the previous tile's readers are done with the ids)
    if (g == 0) {
        for (int j = 0; j < a.n_seg; ++j) {
            const int32_t *ix = tb.idx[j];
            s_rid[j * 16 + c] = ix != nullptr ? ix[row] : (int32_t)row;
        }
    }
    lds_wave_order();
    if (a.vec4) {   // one 16-byte load per tile and lane (the four features 16 t + 4 g .. + 3 of row c)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < kti) {
                const int f = 16 * (t0 + t) + 4 * g;
                const float *p = tb.ptr[f];
                if (valid && p != nullptr) {
                    v = *reinterpret_cast<const f32x4 *>(p + (int64_t)s_rid[tb.seg[f] * 16 + c] * tb.stride[f]);
                    if (tb.relu[f]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                }
            }
            xin[t] = v;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < kti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * (t0 + t) + 4 * g + r;
                const float *p = tb.ptr[f];
                if (valid && p != nullptr) {
                    const float x = p[(int64_t)s_rid[tb.seg[f] * 16 + c] * tb.stride[f]];
                    v[r] = tb.relu[f] ? fmaxf(x, 0.f) : x;
                }
            }
        }
        xin[t] = v;
    }
}

// ================================================================================ forward
template <int HT, int KT, int T>
__global__ __launch_bounds__(kBlock) void mlpw_fwd_kernel(const WArgs a) {
    constexpr int KSH = 4 * HT, KSI = 4 * KT;
    constexpr int kKsMax = KSH > KSI ? KSH : KSI;
    constexpr int kRtMax = HT > kWMaxOT ? HT : kWMaxOT;
    __shared__ __attribute__((aligned(16))) float s_frag[kRtMax * kKsMax * 64];
    __shared__ __attribute__((aligned(16))) float s_bias[16 * kRtMax];
    __shared__ WFeatTab s_tab;
    __shared__ int32_t s_rid[kWaves][GNNTRK_MAX_SEGS * 16];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wv = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int kti = w_tiles(a.in_dim), ksi = 4 * kti, ot = w_tiles(a.out_dim);
    const int HP = 16 * HT;
    const bool three = a.n_layers == 3;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    const int64_t n_batches = (n_tiles + kWaves * T - 1) / (kWaves * T);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    w_build_tab(s_tab, a, tid, kBlock);
    __syncthreads();

    for (int64_t b = blockIdx.x; b < n_batches; b += gridDim.x) {
        f32x4 h[T][HT];
        int64_t row[T];
        bool valid[T];
        {   // ---- layer 1: z1 = W1 x + b1
            f32x4 xin[T][KT];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t r0 = ((b * kWaves + wv) * T + t) * 16 + c;
                valid[t] = r0 < a.n_rows;
                row[t] = valid[t] ? r0 : a.n_rows - 1;
                w_load_input<KT>(a, s_tab, s_rid[wv], row[t], valid[t], g, c, kti, xin[t]);
            }
            w_stage(s_frag, a.frag[0], HT * ksi * 64, s_bias, a.bias[0], a.hidden, HP, tid);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int to = 0; to < HT; ++to) h[t][to] = *reinterpret_cast<const f32x4 *>(s_bias + 16 * to + 4 * g);
#pragma unroll
            for (int ti = 0; ti < KT; ++ti)
                if (ti < kti) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int to = 0; to < HT; ++to) {
                            const float fa = s_frag[(to * ksi + 4 * ti + r) * 64 + lane];
#pragma unroll
                            for (int t = 0; t < T; ++t) h[t][to] = mfma4(fa, xin[t][ti][r], h[t][to]);
                        }
                }
        }
        auto save = [&](int l) {
            if (a.acts == nullptr) return;
            float *dst = a.acts + (int64_t)l * a.n_rows * HP;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (valid[t]) {
#pragma unroll
                    for (int to = 0; to < HT; ++to) *reinterpret_cast<f32x4 *>(dst + row[t] * HP + 16 * to + 4 * g) = h[t][to];
                }
        };
        save(0);
        if (three) {   // ---- layer 2: z2 = W2 relu(z1) + b2
            w_stage(s_frag, a.frag[1], HT * KSH * 64, s_bias, a.bias[1], a.hidden, HP, tid);
            f32x4 p[T][HT], acc[T][HT];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int to = 0; to < HT; ++to) {
                    acc[t][to] = *reinterpret_cast<const f32x4 *>(s_bias + 16 * to + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[t][to][r] = fmaxf(h[t][to][r], 0.f);
                }
#pragma unroll
            for (int ti = 0; ti < HT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int to = 0; to < HT; ++to) {
                        const float fa = s_frag[(to * KSH + 4 * ti + r) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < T; ++t) acc[t][to] = mfma4(fa, p[t][ti][r], acc[t][to]);
                    }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int to = 0; to < HT; ++to) h[t][to] = acc[t][to];
            save(1);
        }
        // ---- last layer + epilogue
        const int L = a.n_layers - 1;
        w_stage(s_frag, a.frag[L], ot * KSH * 64, s_bias, a.bias[L], a.out_dim, 16 * kWMaxOT, tid);
        f32x4 y[T][kWMaxOT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int to = 0; to < kWMaxOT; ++to)
                y[t][to] = to < ot ? *reinterpret_cast<const f32x4 *>(s_bias + 16 * to + 4 * g) : zero;
#pragma unroll
        for (int ti = 0; ti < HT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv[T];
#pragma unroll
                for (int t = 0; t < T; ++t) pv[t] = fmaxf(h[t][ti][r], 0.f);
#pragma unroll
                for (int to = 0; to < kWMaxOT; ++to)
                    if (to < ot) {
                        const float fa = s_frag[(to * KSH + 4 * ti + r) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < T; ++t) y[t][to] = mfma4(fa, pv[t], y[t][to]);
                    }
            }
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (valid[t]) {
                if (a.vec4o) {   // 16 bytes per lane and output tile
#pragma unroll
                    for (int to = 0; to < kWMaxOT; ++to) {
                        const int f = 16 * to + 4 * g;
                        if (to < ot && f < a.out_dim) {
                            f32x4 v = y[t][to];
                            if (a.epilogue == GNNTRK_EPI_RELU) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                            }
                            if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
                                const f32x4 rv = *reinterpret_cast<const f32x4 *>(a.res + row[t] * a.res_stride + f);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = a.ca * rv[r] + a.cb * v[r];
                            }
                            *reinterpret_cast<f32x4 *>(a.out + row[t] * a.out_stride + f) = v;
                        }
                    }
                } else {
#pragma unroll
                    for (int to = 0; to < kWMaxOT; ++to)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * to + 4 * g + r;
                            if (to < ot && f < a.out_dim) {
                                float v = y[t][to][r];
                                if (a.epilogue == GNNTRK_EPI_RELU) v = fmaxf(v, 0.f);
                                if (a.epilogue == GNNTRK_EPI_RESIDUAL) v = a.ca * a.res[row[t] * a.res_stride + f] + a.cb * v;
                                a.out[row[t] * a.out_stride + f] = v;
                            }
                        }
                }
            }
    }
}

// ================================================================================ backward
template <int NT>
__device__ __forceinline__ void w_stage_tiles(float *img, const f32x4 (&v)[NT], int nt, int g, int c) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * t + 4 * g + r) * kWLd + c] = v[t][r];
        }
}
__device__ __forceinline__ f32x4 w_read_k(const float *img, int t, int g, int c) {
    return *reinterpret_cast<const f32x4 *>(img + (16 * t + c) * kWLd + 4 * g);
}
// (`K` columns of the accumulator tiles go to columns i0 .. i0 + K - 1 of a destination with leading dimension ldk)
template <int NO, int NI>
__device__ __forceinline__ void w_emit_dw(float *s_red, float *dst, const f32x4 (&acc)[NO][NI], int no, int ni, int O, int K,
                                          int wv, int tid, int g, int c, int ldk = -1, int i0 = 0) {
    if (ldk < 0) ldk = K;
    __syncthreads();
    for (int w = 0; w < kWaves; ++w) {
        if (wv == w) {
#pragma unroll
            for (int to = 0; to < NO; ++to)
#pragma unroll
                for (int ti = 0; ti < NI; ++ti)
                    if (to < no && ti < ni) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = 16 * to + 4 * g + r, i = 16 * ti + c;
                            if (o < O && i < K) {
                                float *p = s_red + o * K + i;
                                *p = (w == 0) ? acc[to][ti][r] : *p + acc[to][ti][r];
                            }
                        }
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < O * K; i += kBlock) dst[(i / K) * ldk + i0 + (i % K)] = s_red[i];
}
template <int NO>
__device__ __forceinline__ void w_emit_db(float *s_redb, float *dst, const f32x4 (&dbacc)[NO], int no, int O, int wv, int tid,
                                          int g, int c) {
    for (int w = 0; w < kWaves; ++w) {
        if (wv == w) {
#pragma unroll
            for (int to = 0; to < NO; ++to)
                if (to < no) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = dbacc[to][r];
                        v += __shfl_xor(v, 1);
                        v += __shfl_xor(v, 2);
                        v += __shfl_xor(v, 4);
                        v += __shfl_xor(v, 8);
                        const int o = 16 * to + 4 * g + r;
                        if (c == 0 && o < O) s_redb[o] = (w == 0) ? v : s_redb[o] + v;
                    }
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < O; i += kBlock) dst[i] = s_redb[i];
}

// KT = 4: the first-layer stage takes the input tiles FOUR at a time (one pass over the wave's rows per group of 64
// input features: at 128 inputs the 24 + 8 + 8 tiles of dW1 / gx / xin of a single pass cost the second wave
// per SIMD - 369 registers - and with it the cover for every load).
template <int HT>
__global__ __launch_bounds__(kBlock, HT > 3 ? 1 : 2) void mlpw_bwd_kernel(const WArgs a) {
    constexpr int KSH = 4 * HT, KT = 4;
    // W^T fragment sets: last layer HT x (4 OT), middle HT x KSH, first (per pass) KT x KSH; the same buffer takes
    // the block's weight-gradient sums (at most 16 HT x 16 KT floats per pass = KT x KSH x 64)
    constexpr int kKsLast = 4 * kWMaxOT;
    constexpr int kFragSteps = HT * (KSH > kKsLast ? KSH : kKsLast) > KT * KSH ? HT * (KSH > kKsLast ? KSH : kKsLast) : KT * KSH;
    constexpr int kRtMax = HT > KT ? HT : KT;
    constexpr int kImgGTiles = HT > kWMaxOT ? HT : kWMaxOT;   // gradient side: hidden or output tiles
    constexpr int kImgPTiles = kRtMax;                        // activation side: hidden or input tiles
    constexpr int kImgG = 16 * kImgGTiles * kWLd, kImgP = 16 * kImgPTiles * kWLd;
    __shared__ __attribute__((aligned(16))) float s_frag[kFragSteps * 64];
    __shared__ __attribute__((aligned(16))) float s_imgG[kWaves][kImgG];
    __shared__ __attribute__((aligned(16))) float s_imgP[kWaves][kImgP];
    __shared__ __attribute__((aligned(16))) float s_redb[16 * (HT > kWMaxOT ? HT : kWMaxOT)];
    __shared__ WFeatTab s_tab;
    __shared__ int32_t s_rid[kWaves][GNNTRK_MAX_SEGS * 16];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wv = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int kti = w_tiles(a.in_dim), ot = w_tiles(a.out_dim), kso = 4 * ot;
    const int HP = 16 * HT, H = a.hidden;
    const bool three = a.n_layers == 3;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    const int64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t tb0 = per * blockIdx.x, tb1 = (tb0 + per < n_tiles) ? tb0 + per : n_tiles;
    float *imgG = s_imgG[wv], *imgP = s_imgP[wv];
    float *part = a.part + (int64_t)blockIdx.x * a.part_total;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int poff = 0;   // partial block: last layer (W, b), middle layer (W, b), first layer (W, b)
    w_build_tab(s_tab, a, tid, kBlock);

    {   // ---------------------------------------------------------------- last layer
        const int L = a.n_layers - 1;
        w_stage(s_frag, a.fragT[L], HT * kso * 64, s_redb, nullptr, 0, 0, tid);
        const float *xl = a.acts + (int64_t)(L - 1) * a.n_rows * HP;
        f32x4 dW[kWMaxOT][HT], dbacc[kWMaxOT];
#pragma unroll
        for (int to = 0; to < kWMaxOT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) dW[to][ti] = zero;
        }
        // (one wave per SIMD at the wide shapes: nobody covers a tile's load latency, so the next tile's rows are
        //  requested before the current one is worked on)
        auto load_last = [&](int64_t tile, f32x4 (&go)[kWMaxOT], f32x4 (&xv)[HT]) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            int64_t grow[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) grow[t] = (t < a.n_gout && a.gout[t].idx != nullptr) ? a.gout[t].idx[row] : row;
#pragma unroll
            for (int to = 0; to < kWMaxOT; ++to) {
                go[to] = zero;
                const int f0 = 16 * to + 4 * g;
                if (to < ot && a.vec4o) {
                    if (valid && f0 < a.out_dim) {
                        f32x4 v = zero;
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            if (t < a.n_gout) {
                                const f32x4 u = *reinterpret_cast<const f32x4 *>(a.gout[t].ptr + grow[t] * a.gout[t].stride + f0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += u[r];
                            }
                        if (a.epilogue == GNNTRK_EPI_RELU) {
                            const f32x4 o = *reinterpret_cast<const f32x4 *>(a.fwd_out + row * a.out_stride + f0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = o[r] > 0.f ? v[r] : 0.f;
                        }
                        if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= a.cb;
                        }
                        go[to] = v;
                    }
                } else if (to < ot) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = f0 + r;
                        if (valid && f < a.out_dim) {
                            float v = 0.f;
#pragma unroll
                            for (int t = 0; t < 3; ++t)
                                if (t < a.n_gout) v += a.gout[t].ptr[grow[t] * a.gout[t].stride + f];
                            if (a.epilogue == GNNTRK_EPI_RELU) v = a.fwd_out[row * a.out_stride + f] > 0.f ? v : 0.f;
                            if (a.epilogue == GNNTRK_EPI_RESIDUAL) v *= a.cb;
                            go[to][r] = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) xv[t] = *reinterpret_cast<const f32x4 *>(xl + row * HP + 16 * t + 4 * g);
        };
        f32x4 goN[kWMaxOT], xvN[HT];
        if (tb0 + wv < tb1) load_last(tb0 + wv, goN, xvN);
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            f32x4 go[kWMaxOT], p[HT];
#pragma unroll
            for (int to = 0; to < kWMaxOT; ++to) go[to] = goN[to];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p[t][r] = fmaxf(xvN[t][r], 0.f);
            }
            if (tile + kWaves < tb1) load_last(tile + kWaves, goN, xvN);
            f32x4 gh[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gh[t] = zero;
#pragma unroll
            for (int to = 0; to < kWMaxOT; ++to)
                if (to < ot) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gv = go[to][r];
#pragma unroll
                        for (int t = 0; t < HT; ++t) gh[t] = mfma4(s_frag[(t * kso + 4 * to + r) * 64 + lane], gv, gh[t]);
                    }
                }
            if (valid) {
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = p[t][r] > 0.f ? gh[t][r] : 0.f;
                    *reinterpret_cast<f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) = v;
                }
            }
            w_stage_tiles<kWMaxOT>(imgG, go, ot, g, c);
            w_stage_tiles<HT>(imgP, p, HT, g, c);
            lds_wave_order();
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) {
                const f32x4 b4 = w_read_k(imgP, ti, g, c);
#pragma unroll
                for (int to = 0; to < kWMaxOT; ++to)
                    if (to < ot) {
                        const f32x4 a4 = w_read_k(imgG, to, g, c);
#pragma unroll
                        for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                    }
            }
#pragma unroll
            for (int to = 0; to < kWMaxOT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[to][r] += go[to][r];
            lds_wave_order();
        }
        w_emit_dw<kWMaxOT, HT>(s_frag, part + poff, dW, ot, HT, a.out_dim, H, wv, tid, g, c);
        poff += a.out_dim * H;
        w_emit_db<kWMaxOT>(s_redb, part + poff, dbacc, ot, a.out_dim, wv, tid, g, c);
        poff += a.out_dim;
    }

    if (three) {   // ---------------------------------------------------------------- middle layer
        w_stage(s_frag, a.fragT[1], HT * KSH * 64, s_redb, nullptr, 0, 0, tid);
        const float *xl = a.acts;   // z1
        f32x4 dW[HT][HT], dbacc[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) dW[to][ti] = zero;
        }
        auto load_mid = [&](int64_t tile, f32x4 (&gy)[HT], f32x4 (&xv)[HT]) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                gy[t] = valid ? *reinterpret_cast<const f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) : zero;
                xv[t] = *reinterpret_cast<const f32x4 *>(xl + row * HP + 16 * t + 4 * g);
            }
        };
        f32x4 gyN[HT], xvN[HT];
        if (tb0 + wv < tb1) load_mid(tb0 + wv, gyN, xvN);
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            f32x4 gy[HT], p[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gy[t] = gyN[t];
            f32x4 xvC[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) xvC[t] = xvN[t];
            if (tile + kWaves < tb1) load_mid(tile + kWaves, gyN, xvN);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const f32x4 xv = xvC[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[t][r] = fmaxf(xv[r], 0.f);
                    imgG[(16 * t + 4 * g + r) * kWLd + c] = gy[t][r];
                    imgP[(16 * t + 4 * g + r) * kWLd + c] = p[t][r];
                }
            }
            f32x4 gp[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gp[t] = zero;
            lds_wave_order();
#pragma unroll 4
            for (int ks = 0; ks < KSH; ++ks) {
                const float gzv = imgG[(16 * (ks >> 2) + 4 * g + (ks & 3)) * kWLd + c];
#pragma unroll
                for (int t = 0; t < HT; ++t) gp[t] = mfma4(s_frag[(t * KSH + ks) * 64 + lane], gzv, gp[t]);
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = p[t][r] > 0.f ? gp[t][r] : 0.f;
                    *reinterpret_cast<f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) = v;
                }
            }
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[to][r] += gy[to][r];
            if constexpr (HT <= 4) {
                // gradient-side operands read once; consecutive MFMAs go to different accumulators
                f32x4 aG[HT];
#pragma unroll
                for (int to = 0; to < HT; ++to) aG[to] = w_read_k(imgG, to, g, c);
#pragma unroll
                for (int ti = 0; ti < HT; ++ti) {
                    const f32x4 b4 = w_read_k(imgP, ti, g, c);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int to = 0; to < HT; ++to) dW[to][ti] = mfma4(aG[to][s], b4[s], dW[to][ti]);
                }
            } else {
#pragma unroll
                for (int ti = 0; ti < HT; ++ti) {
                    const f32x4 b4 = w_read_k(imgP, ti, g, c);
#pragma unroll
                    for (int to = 0; to < HT; ++to) {
                        const f32x4 a4 = w_read_k(imgG, to, g, c);
#pragma unroll
                        for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            lds_wave_order();
        }
        w_emit_dw<HT, HT>(s_frag, part + poff, dW, HT, HT, H, H, wv, tid, g, c);
        poff += H * H;
        w_emit_db<HT>(s_redb, part + poff, dbacc, HT, H, wv, tid, g, c);
        poff += H;
    }

    const int npass = (kti + KT - 1) / KT;
    for (int pass = 0; pass < npass; ++pass) {   // ------------------------------------------------ first layer
        const int t0 = KT * pass, ktp = kti - t0 < KT ? kti - t0 : KT;
        if (a.want_dx) w_stage(s_frag, a.fragT[0] + (size_t)t0 * KSH * 64, ktp * KSH * 64, s_redb, nullptr, 0, 0, tid);
        else __syncthreads();
        f32x4 dW[HT][KT], dbacc[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < KT; ++ti) dW[to][ti] = zero;
        }
        auto load_gy = [&](int64_t tile, f32x4 (&gy)[HT]) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
#pragma unroll
            for (int t = 0; t < HT; ++t)
                gy[t] = valid ? *reinterpret_cast<const f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) : zero;
        };
        f32x4 gyN[HT];
        if (tb0 + wv < tb1) load_gy(tb0 + wv, gyN);
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            f32x4 gy[HT], xin[KT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gy[t] = gyN[t];
            if (tile + kWaves < tb1) load_gy(tile + kWaves, gyN);
            w_load_input<KT>(a, s_tab, s_rid[wv], row, valid, g, c, ktp, xin, t0);
            w_stage_tiles<HT>(imgG, gy, HT, g, c);
            w_stage_tiles<KT>(imgP, xin, ktp, g, c);
            lds_wave_order();
            if (a.want_dx) {
                // gradient at the concatenated input (through the ReLU on load where a segment has one),
                // written to the per-row slices of the segments that want it
                f32x4 gx[KT];
#pragma unroll
                for (int ti = 0; ti < KT; ++ti) gx[ti] = zero;
#pragma unroll 4
                for (int ks = 0; ks < KSH; ++ks) {
                    const float gv = imgG[(16 * (ks >> 2) + 4 * g + (ks & 3)) * kWLd + c];
#pragma unroll
                    for (int ti = 0; ti < KT; ++ti)
                        if (ti < ktp) gx[ti] = mfma4(s_frag[(ti * KSH + ks) * 64 + lane], gv, gx[ti]);
                }
                if (valid && a.vec4) {
#pragma unroll
                    for (int ti = 0; ti < KT; ++ti) {
                        const int f = 16 * (t0 + ti) + 4 * g;
                        float *gp = (ti < ktp && f < a.in_dim) ? s_tab.gptr[f] : nullptr;
                        if (gp != nullptr) {
                            f32x4 v = gx[ti];
                            if (s_tab.relu[f]) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = xin[ti][r] > 0.f ? v[r] : 0.f;
                            }
                            *reinterpret_cast<f32x4 *>(gp + r0 * s_tab.gstride[f]) = v;
                        }
                    }
                } else if (valid) {
#pragma unroll
                    for (int ti = 0; ti < KT; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * (t0 + ti) + 4 * g + r;
                            if (ti < ktp && f < a.in_dim) {
                                float *gp = s_tab.gptr[f];
                                if (gp != nullptr) {
                                    float v = gx[ti][r];
                                    if (s_tab.relu[f]) v = xin[ti][r] > 0.f ? v : 0.f;
                                    gp[r0 * s_tab.gstride[f]] = v;
                                }
                            }
                        }
                }
            }
            if constexpr (HT <= 4) {
                f32x4 aG[HT];
#pragma unroll
                for (int to = 0; to < HT; ++to) aG[to] = w_read_k(imgG, to, g, c);
#pragma unroll
                for (int ti = 0; ti < KT; ++ti)
                    if (ti < ktp) {
                        const f32x4 b4 = w_read_k(imgP, ti, g, c);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int to = 0; to < HT; ++to) dW[to][ti] = mfma4(aG[to][s], b4[s], dW[to][ti]);
                    }
            } else {
#pragma unroll
                for (int ti = 0; ti < KT; ++ti)
                    if (ti < ktp) {
                        const f32x4 b4 = w_read_k(imgP, ti, g, c);
#pragma unroll
                        for (int to = 0; to < HT; ++to) {
                            const f32x4 a4 = w_read_k(imgG, to, g, c);
#pragma unroll
                            for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            if (pass == 0) {
#pragma unroll
                for (int to = 0; to < HT; ++to)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dbacc[to][r] += gy[to][r];
            }
            lds_wave_order();
        }
        // this pass's columns 16 t0 .. of dW1 (leading dimension in_dim)
        const int kcols = a.in_dim - 16 * t0 < 16 * KT ? a.in_dim - 16 * t0 : 16 * KT;
        w_emit_dw<HT, KT>(s_frag, part + poff, dW, HT, ktp, H, kcols, wv, tid, g, c, a.in_dim, 16 * t0);
        if (pass == 0) w_emit_db<HT>(s_redb, part + poff + H * a.in_dim, dbacc, HT, H, wv, tid, g, c);
    }
}

// ---- final reduction of the per-block partials into gW / gb (fixed order)
struct WReduceArgs {
    const float *part;
    int32_t n_part, part_total, n_seg, accumulate;
    int32_t off[7];
    float *dst[6];
};
// 32 parameters x 8 slices of the partial blocks per workgroup: a slice adds its blocks in order, the eight slice
// sums are added in slice order - a fixed association, and 256 instead of 32 workgroups in flight (the one-thread-
// per-parameter walk over 256 blocks was a 90 us latency chain for 32 KB of sums)
__global__ __launch_bounds__(256) void mlpw_reduce_kernel(const WReduceArgs a) {
    __shared__ float s_part[8][32];
    const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + pl;
    const int per = (a.n_part + 7) / 8, b0 = sl * per, b1 = b0 + per < a.n_part ? b0 + per : a.n_part;
    float acc = 0.f;
    if (i < a.part_total)
        for (int b = b0; b < b1; ++b) acc += a.part[(int64_t)b * a.part_total + i];
    s_part[sl][pl] = acc;
    __syncthreads();
    if (sl != 0 || i >= a.part_total) return;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += s_part[q][pl];
    int j = 0;
    while (j + 1 < a.n_seg && i >= a.off[j + 1]) ++j;
    if (a.dst[j] == nullptr) return;
    float *p = a.dst[j] + (i - a.off[j]);
    *p = a.accumulate ? *p + s : s;
}

// ---- host
int w_ht(int hidden) {
    const int t = w_tiles(hidden);
    return t <= 4 ? t : t <= 6 ? 6 : 8;
}
int w_kt(int in_dim) { return w_tiles(in_dim) <= 4 ? 4 : 8; }

int w_check(const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg, int epilogue, const char *who) {
    if (m.n_layers != 2 && m.n_layers != 3) return fail(GNNTRK_EUNSUPPORTED, "mlp_wide: n_layers must be 2 or 3");
    if (m.in_dim < 1 || m.in_dim > kWMaxIn || m.hidden < 1 || m.hidden > kWMaxHidden || m.out_dim < 1 || m.out_dim > kWMaxOut)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_wide: limits are in <= 128, hidden <= 128, out <= 48");
    if (epilogue != GNNTRK_EPI_NONE && epilogue != GNNTRK_EPI_RELU && epilogue != GNNTRK_EPI_RESIDUAL)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_wide: epilogues NONE, RELU, RESIDUAL");
    if (n_seg < 1 || n_seg > GNNTRK_MAX_SEGS) return fail(GNNTRK_EINVAL, "mlp_wide: bad segment count");
    int tot = 0;
    for (int j = 0; j < n_seg; ++j) {
        if (!seg[j].ptr || seg[j].dim < 1 || seg[j].stride < seg[j].dim) return fail(GNNTRK_EINVAL, "mlp_wide: bad segment");
        tot += seg[j].dim;
    }
    if (tot != m.in_dim) return fail(GNNTRK_EINVAL, "mlp_wide: segment dims do not sum to in_dim");
    for (int i = 0; i < m.n_layers; ++i)
        if (!m.W[i]) return fail(GNNTRK_EINVAL, "mlp_wide: NULL weight pointer");
    (void)who;
    return GNNTRK_OK;
}

struct WLayout {
    size_t fwd[3], bwd[3], total;
    int rt_f[3], kt_f[3], rt_b[3], kt_b[3];
};
WLayout w_layout(const gnntrk_mlp &m, bool with_bwd) {
    WLayout L;
    const int HT = w_ht(m.hidden), KT = w_kt(m.in_dim), KTI = w_tiles(m.in_dim), OT = w_tiles(m.out_dim);
    (void)KT;
    size_t off = 0;
    auto take = [&](size_t n) {
        const size_t o = off;
        off += (n + 63) / 64 * 64;
        return o;
    };
    const int nl = m.n_layers;
    for (int l = 0; l < nl; ++l) {
        L.rt_f[l] = l == nl - 1 ? OT : HT;
        L.kt_f[l] = l == 0 ? KTI : HT;
        L.fwd[l] = take((size_t)L.rt_f[l] * 4 * L.kt_f[l] * 64);
    }
    for (int l = 0; l < nl; ++l) {
        L.rt_b[l] = l == 0 ? KTI : HT;
        L.kt_b[l] = l == nl - 1 ? OT : HT;
        L.bwd[l] = with_bwd ? take((size_t)L.rt_b[l] * 4 * L.kt_b[l] * 64) : 0;
    }
    L.total = off;
    return L;
}
int w_part_total(const gnntrk_mlp &m) {
    return m.out_dim * m.hidden + m.out_dim + (m.n_layers == 3 ? m.hidden * m.hidden + m.hidden : 0) + m.hidden * m.in_dim +
           m.hidden;
}
int w_bwd_grid(int64_t n_rows, int hidden) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t g = (tiles + kWaves - 1) / kWaves;
    const int64_t cap = (int64_t)cu_count() * (w_ht(hidden) > 3 ? 1 : 2);   // (resident workgroups per CU)
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}
int w_pack(const gnntrk_mlp &m, float *base, const WLayout &L, bool with_bwd, hipStream_t stream) {
    WPackArgs pa;
    memset(&pa, 0, sizeof(pa));
    const int nl = m.n_layers;
    int n = 0;
    for (int pass = 0; pass < (with_bwd ? 2 : 1); ++pass)
        for (int l = 0; l < nl; ++l) {
            const int out = l == nl - 1 ? m.out_dim : m.hidden, in = l == 0 ? m.in_dim : m.hidden;
            WPackJob &j = pa.job[n++];
            j.W = m.W[l];
            j.ld = in;
            j.transposed = pass;
            j.rows = pass ? in : out;
            j.cols = pass ? out : in;
            j.rt = pass ? L.rt_b[l] : L.rt_f[l];
            j.kt = pass ? L.kt_b[l] : L.kt_f[l];
            j.dst = base + (pass ? L.bwd[l] : L.fwd[l]);
        }
    pa.n_jobs = n;
    hipLaunchKernelGGL(mlpw_pack_kernel, dim3(8, n), dim3(256), 0, stream, pa);
    return check_launch("mlp_wide_pack");
}
void w_common(WArgs &w, const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg, int epilogue, float ca, float cb,
              int64_t n_rows, const float *base, const WLayout &L, bool with_bwd) {
    memset(&w, 0, sizeof(w));
    for (int j = 0; j < n_seg; ++j) w.seg[j] = seg[j];
    for (int l = 0; l < m.n_layers; ++l) {
        w.frag[l] = base + L.fwd[l];
        w.fragT[l] = with_bwd ? base + L.bwd[l] : nullptr;
        w.bias[l] = m.b[l];
    }
    w.n_seg = n_seg;
    w.n_layers = m.n_layers;
    w.epilogue = epilogue;
    w.in_dim = m.in_dim;
    w.hidden = m.hidden;
    w.out_dim = m.out_dim;
    w.ca = ca;
    w.cb = cb;
    w.n_rows = n_rows;
    bool v4 = true;
    for (int j = 0; j < n_seg; ++j)
        if (seg[j].dim % 4 != 0 || seg[j].stride % 4 != 0 || ((uintptr_t)seg[j].ptr & 15) != 0) v4 = false;
    w.vec4 = v4 ? 1 : 0;
}

}  // namespace
}  // namespace gnntrk

using namespace gnntrk;

extern "C" {

int32_t gnntrk_mlp_wide_hidden_pad(int32_t hidden) { return 16 * w_ht(hidden); }

size_t gnntrk_mlp_wide_forward_workspace_bytes(const gnntrk_mlp *m) {
    if (!m || m->n_layers < 2 || m->n_layers > 3) return 0;
    return w_layout(*m, false).total * sizeof(float);
}

int gnntrk_mlp_forward_wide(const gnntrk_mlp_fwd_args *a, float *acts, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!a) return fail(GNNTRK_EINVAL, "mlp_forward_wide: NULL args");
    int rc = w_check(a->mlp, a->n_seg, a->seg, a->epilogue, "mlp_forward_wide");
    if (rc) return rc;
    if (a->out_idx) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_wide: out_idx is not supported");
    if (a->n_rows < 0 || a->n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "mlp_forward_wide: bad n_rows");
    if (a->n_rows == 0) return GNNTRK_OK;
    if (!a->out || a->out_stride < a->mlp.out_dim) return fail(GNNTRK_EINVAL, "mlp_forward_wide: bad output");
    if (a->epilogue == GNNTRK_EPI_RESIDUAL && (!a->res || a->res_stride < a->mlp.out_dim))
        return fail(GNNTRK_EINVAL, "mlp_forward_wide: RESIDUAL needs res rows");
    const WLayout L = w_layout(a->mlp, false);
    if (!workspace || workspace_bytes < L.total * sizeof(float) || ((uintptr_t)workspace & 15))
        return fail(GNNTRK_EINVAL, "mlp_forward_wide: workspace too small or misaligned");
    if (acts && ((uintptr_t)acts & 15)) return fail(GNNTRK_EINVAL, "mlp_forward_wide: acts must be 16-byte aligned");
    float *base = reinterpret_cast<float *>(workspace);
    rc = w_pack(a->mlp, base, L, false, stream);
    if (rc) return rc;
    WArgs w;
    w_common(w, a->mlp, a->n_seg, a->seg, a->epilogue, a->ca, a->cb, a->n_rows, base, L, false);
    w.res = a->res;
    w.res_stride = a->res_stride;
    w.out = a->out;
    w.out_stride = a->out_stride;
    w.acts = acts;
    w.vec4o = (a->mlp.out_dim % 4 == 0 && a->out_stride % 4 == 0 && ((uintptr_t)a->out & 15) == 0 &&
               (a->epilogue != GNNTRK_EPI_RESIDUAL || (a->res_stride % 4 == 0 && ((uintptr_t)a->res & 15) == 0)))
                  ? 1 : 0;
    const int HT = w_ht(a->mlp.hidden), KT = w_kt(a->mlp.in_dim);
    const int64_t tiles = (a->n_rows + 15) / 16;
    bool launched = false;
#define GNNTRK_MW_FWD(HT_, KT_, T_)                                                                       \
    if (!launched && HT == HT_ && KT == KT_) {                                                            \
        int64_t grid = (tiles + kWaves * T_ - 1) / (kWaves * T_);                                         \
        const int64_t cap = (int64_t)cu_count() * 2;                                                      \
        if (grid > cap) grid = cap;                                                                       \
        hipLaunchKernelGGL((mlpw_fwd_kernel<HT_, KT_, T_>), dim3((int)grid), dim3(kBlock), 0, stream, w); \
        launched = true;                                                                                  \
    }
    GNNTRK_MW_FWD(1, 4, 2) GNNTRK_MW_FWD(2, 4, 2) GNNTRK_MW_FWD(3, 4, 2) GNNTRK_MW_FWD(4, 4, 2) GNNTRK_MW_FWD(6, 4, 1)
    GNNTRK_MW_FWD(8, 4, 1) GNNTRK_MW_FWD(1, 8, 2) GNNTRK_MW_FWD(2, 8, 2) GNNTRK_MW_FWD(3, 8, 2) GNNTRK_MW_FWD(4, 8, 1)
    GNNTRK_MW_FWD(6, 8, 1) GNNTRK_MW_FWD(8, 8, 1)
#undef GNNTRK_MW_FWD
    return check_launch("mlp_forward_wide");
}

size_t gnntrk_mlp_wide_backward_workspace_bytes(const gnntrk_mlp *m, int64_t n_rows) {
    if (!m || m->n_layers < 2 || m->n_layers > 3 || n_rows < 0) return 0;
    const size_t frag = w_layout(*m, true).total * sizeof(float);
    const size_t gs = align_up((size_t)n_rows * 16 * w_ht(m->hidden) * sizeof(float), 256);
    const size_t part = (size_t)w_bwd_grid(n_rows, m->hidden) * w_part_total(*m) * sizeof(float);
    return align_up(frag, 256) + gs + align_up(part, 256);
}

int gnntrk_mlp_backward_wide(const gnntrk_mlp_bwd_args *a, const float *acts, const float *out, int32_t out_stride,
                             void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!a) return fail(GNNTRK_EINVAL, "mlp_backward_wide: NULL args");
    int rc = w_check(a->mlp, a->n_seg, a->seg, a->epilogue, "mlp_backward_wide");
    if (rc) return rc;
    const gnntrk_mlp &m = a->mlp;
    if (a->n_rows < 0 || a->n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "mlp_backward_wide: bad n_rows");
    if (a->n_gout < 1 || a->n_gout > 3) return fail(GNNTRK_EINVAL, "mlp_backward_wide: bad upstream gradient terms");
    for (int t = 0; t < a->n_gout && a->n_rows > 0; ++t)
        if (!a->gout[t].ptr || a->gout[t].stride < m.out_dim) return fail(GNNTRK_EINVAL, "mlp_backward_wide: bad upstream gradient terms");
    if (a->n_rows > 0 && (!acts || ((uintptr_t)acts & 15))) return fail(GNNTRK_EINVAL, "mlp_backward_wide: bad acts");
    if (a->epilogue == GNNTRK_EPI_RELU && a->n_rows > 0 && (!out || out_stride < m.out_dim))
        return fail(GNNTRK_EINVAL, "mlp_backward_wide: RELU needs the forward's output");
    bool want_dx = false;
    for (int j = 0; j < a->n_seg; ++j)
        if (a->gseg[j].ptr) {
            if (a->gseg[j].idx || a->gseg[j].accumulate || a->gseg[j].stride < a->seg[j].dim)
                return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_wide: gradient slices are plain per-row tensors");
            want_dx = true;
        }
    if (!workspace || workspace_bytes < gnntrk_mlp_wide_backward_workspace_bytes(&m, a->n_rows) || ((uintptr_t)workspace & 15))
        return fail(GNNTRK_EINVAL, "mlp_backward_wide: workspace too small or misaligned");
    const bool want_dw = a->gW[0] != nullptr;
    if (want_dw)
        for (int i = 0; i < m.n_layers; ++i)
            if (!a->gW[i]) return fail(GNNTRK_EINVAL, "mlp_backward_wide: gW must be all set or all NULL");
    const WLayout L = w_layout(m, true);
    float *base = reinterpret_cast<float *>(workspace);
    uint8_t *bytes = reinterpret_cast<uint8_t *>(workspace);
    float *gstream = reinterpret_cast<float *>(bytes + align_up(L.total * sizeof(float), 256));
    const int HT = w_ht(m.hidden);
    float *part = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(gstream) +
                                            align_up((size_t)a->n_rows * 16 * HT * sizeof(float), 256));
    const int PT = w_part_total(m);
    int grid = 0;
    if (a->n_rows > 0) {
        rc = w_pack(m, base, L, true, stream);
        if (rc) return rc;
        WArgs w;
        w_common(w, m, a->n_seg, a->seg, a->epilogue, a->ca, a->cb, a->n_rows, base, L, true);
        for (int j = 0; j < a->n_seg; ++j) {
            w.gseg[j] = a->gseg[j];
            if (a->gseg[j].ptr && (a->gseg[j].stride % 4 != 0 || ((uintptr_t)a->gseg[j].ptr & 15) != 0)) w.vec4 = 0;
        }
        for (int t = 0; t < a->n_gout; ++t) w.gout[t] = a->gout[t];
        w.n_gout = a->n_gout;
        bool vo = m.out_dim % 4 == 0;
        for (int t = 0; t < a->n_gout; ++t)
            if (a->gout[t].stride % 4 != 0 || ((uintptr_t)a->gout[t].ptr & 15) != 0) vo = false;
        if (a->epilogue == GNNTRK_EPI_RELU && (out_stride % 4 != 0 || ((uintptr_t)out & 15) != 0)) vo = false;
        w.vec4o = vo ? 1 : 0;
        w.acts = const_cast<float *>(acts);
        w.fwd_out = out;
        w.out_stride = out_stride;
        w.gstream = gstream;
        w.part = part;
        w.part_total = PT;
        w.want_dx = want_dx ? 1 : 0;
        grid = w_bwd_grid(a->n_rows, m.hidden);
        bool launched = false;
#define GNNTRK_MW_BWD(HT_)                                                                               \
    if (!launched && HT == HT_) {                                                                         \
        hipLaunchKernelGGL((mlpw_bwd_kernel<HT_>), dim3(grid), dim3(kBlock), 0, stream, w);               \
        launched = true;                                                                                  \
    }
        GNNTRK_MW_BWD(1) GNNTRK_MW_BWD(2) GNNTRK_MW_BWD(3) GNNTRK_MW_BWD(4) GNNTRK_MW_BWD(6) GNNTRK_MW_BWD(8)
#undef GNNTRK_MW_BWD
        rc = check_launch("mlp_backward_wide");
        if (rc) return rc;
    }
    if (!want_dw) return GNNTRK_OK;
    WReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.part = part;
    ra.n_part = grid;
    ra.part_total = PT;
    ra.accumulate = a->accumulate_params;
    int n = 0, off = 0;
    auto seg = [&](float *dst, int len) {
        ra.off[n] = off;
        ra.dst[n] = dst;
        off += len;
        ++n;
    };
    const int Ln = m.n_layers - 1;
    seg(a->gW[Ln], m.out_dim * m.hidden);
    seg(m.b[Ln] ? a->gb[Ln] : nullptr, m.out_dim);
    if (m.n_layers == 3) {
        seg(a->gW[1], m.hidden * m.hidden);
        seg(m.b[1] ? a->gb[1] : nullptr, m.hidden);
    }
    seg(a->gW[0], m.hidden * m.in_dim);
    seg(m.b[0] ? a->gb[0] : nullptr, m.hidden);
    ra.off[n] = off;
    ra.n_seg = n;
    hipLaunchKernelGGL(mlpw_reduce_kernel, dim3((PT + 31) / 32), dim3(256), 0, stream, ra);
    return check_launch("mlp_wide_reduce");
}

}  // extern "C"
