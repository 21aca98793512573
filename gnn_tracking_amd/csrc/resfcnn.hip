// Residual fully connected network (models/mlp.py:65-120: ResFCNN; the embedding networks of
// models/graph_construction.py:25-132 are built on it) as ONE forward and ONE backward launch,
// fp32 on v_mfma_f32_16x16x4_f32 (see include/gnntrk.h for the operator and the C ABI).
//
// Orientation as in tile_mlp.h: activations TRANSPOSED, features x rows, in the accumulator layout
//     lane l = 16 g + c, register r:   D[4g + r][c]  <->  feature 16 t + 4g + r of row (tile base + c)
// so the D registers of one layer are the B operand of the next (k-step (t, r) contracts the features
// {16t + 4g + r : g = 0..3}; the A fragments are packed with the same permutation of k).
//
// Forward: a wave owns T tiles of 16 rows and carries them through ALL layers in registers; the weights
// of one layer at a time sit in LDS as A fragments (a straight 16-byte copy of what the pack kernel left
// in the workspace), two workgroup barriers per layer.
//
// Backward: the weight gradients of ONE hidden layer are up to 64 accumulator tiles (256 registers at
// hidden 128) - they cannot live next to those of the other layers.  So the backward is layer-outer:
// every wave walks its own rows once per layer, from the decoder down, with that layer's W^T fragments in
// LDS, its weight-gradient tiles in registers (K = rows contractions: both operands through wave-private
// [feature][row] LDS images, read back 16 bytes = four k-steps at a time), the gradient of the residual
// stream in a workspace buffer between layers (written and read back by the same lane: no
// synchronisation), the forward's residual stream from `acts`.  Per-block partial sums, fixed-order
// reduction: deterministic, no atomics.
#include <cmath>
#include <cstring>

#include "host_util.h"
#include "tile_mlp.h"

namespace gnntrk {
namespace {

constexpr int kRfMaxL = GNNTRK_RESFCNN_MAX_HIDDEN + 2;   // encoder, hidden layers, decoder
constexpr int kRfMaxKTI = GNNTRK_RESFCNN_MAX_IN / 16;     // input tiles
constexpr int kRfMaxOT = GNNTRK_RESFCNN_MAX_OUT / 16;     // output tiles
constexpr int kRfLd = 20;                                 // leading dim of a [feature][row] staging image

__host__ __device__ inline int rf_tiles(int d) { return (d + 15) / 16; }

// ---- fragment packing --------------------------------------------------------------------------
// dst[(to * KS + ks) * 64 + lane] = Mat[16 to + c][16 (ks >> 2) + 4 g + (ks & 3)]   (0 outside the matrix)
// Mat = W (rows = out features, k = in features) or W^T, W in nn.Linear storage [out][in].
struct RfPackJob {
    const float *W;
    float *dst;
    int32_t rows, cols;   // of Mat
    int32_t rt, kt;       // row tiles / k tiles of the fragment image (the kernel's padded counts)
    int32_t ld;           // leading dim of W
    int32_t transposed;
};
struct RfPackArgs {
    RfPackJob job[2 * kRfMaxL + 1];
    int32_t n_jobs;
};

__global__ __launch_bounds__(256) void resfcnn_pack_kernel(const RfPackArgs a) {
    const RfPackJob j = a.job[blockIdx.y];
    if ((int)blockIdx.y >= a.n_jobs) return;
    const int RT = j.rt, KS = 4 * j.kt;
    const int n = RT * KS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int fr = i >> 6, l = i & 63;
        const int to = fr / KS, ks = fr - to * KS;
        const int g = l >> 4, c = l & 15;
        const int r = 16 * to + c, k = 16 * (ks >> 2) + 4 * g + (ks & 3);
        float v = 0.f;
        if (r < j.rows && k < j.cols) v = j.transposed ? j.W[(int64_t)k * j.ld + r] : j.W[(int64_t)r * j.ld + k];
        j.dst[i] = v;
    }
}

// ---- kernel arguments ----------------------------------------------------------------------------
struct RfArgs {
    const float *x;
    int64_t n_rows;
    const float *frag[kRfMaxL];    // forward fragments: 0 encoder, 1 .. n_hidden hidden, n_hidden + 1 decoder
    const float *fragT[kRfMaxL];   // transposed fragments (backward)
    const float *bias[kRfMaxL];    // or NULL
    const float *out_scale;
    float *out;
    float *acts;                   // forward: written (or NULL); backward: read
    // backward only
    const float *gout;
    const float *fwd_out;
    float *gstream;                // [n_rows][HP] gradient of the residual stream
    float *gx;
    float *part;                   // [grid][part_total]
    int32_t x_stride, out_stride, gout_stride, gx_stride;
    int32_t in_dim, hidden, out_dim, n_hidden;
    int32_t normalize, out_relu, want_enc_dx, part_total;
    float sa, sb;
};

__device__ __forceinline__ void rf_stage(float *s_frag, const float *src, int n_floats, float *s_bias, const float *bias,
                                         int n_bias, int n_bias_pad, int tid) {
    __syncthreads();   // every wave is done with the previous layer's fragments
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(src);
    f32x4 *d4 = reinterpret_cast<f32x4 *>(s_frag);
    for (int i = tid; i < n_floats / 4; i += kBlock) d4[i] = s4[i];
    for (int i = tid; i < n_bias_pad; i += kBlock) s_bias[i] = (bias != nullptr && i < n_bias) ? bias[i] : 0.f;
    __syncthreads();
}

// x rows of one tile in accumulator layout (+ the L2 norm of every row for the normalised form)
template <int KTI>
__device__ __forceinline__ void rf_load_input(const RfArgs &a, int64_t row, bool valid, int g, int kti, f32x4 (&xin)[KTI],
                                              float &nrm) {
    const float *xr = a.x + row * a.x_stride;
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < KTI; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < kti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * t + 4 * g + r;
                v[r] = (valid && f < a.in_dim) ? xr[f] : 0.f;
                ss += v[r] * v[r];
            }
        }
        xin[t] = v;
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    nrm = 1.f;
    if (a.normalize) {
        nrm = fmaxf(sqrtf(ss), 1e-12f);   // torch.nn.functional.normalize: x / max(||x||, eps)
#pragma unroll
        for (int t = 0; t < KTI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) xin[t][r] = xin[t][r] / nrm;
    }
}

// ================================================================================ forward
template <int HT, int T>
__global__ __launch_bounds__(kBlock) void resfcnn_fwd_kernel(const RfArgs a) {
    constexpr int KSH = 4 * HT;
    constexpr int kFragFloats = HT * (KSH > 4 * kRfMaxKTI ? KSH : 4 * kRfMaxKTI) * 64;
    __shared__ __attribute__((aligned(16))) float s_frag[kFragFloats];
    __shared__ __attribute__((aligned(16))) float s_bias[16 * (HT > kRfMaxOT ? HT : kRfMaxOT)];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wv = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int kti = rf_tiles(a.in_dim), ksi = 4 * kti, ot = rf_tiles(a.out_dim);
    const int HP = 16 * HT;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    const int64_t n_batches = (n_tiles + kWaves * T - 1) / (kWaves * T);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float scale = 1.f;
    if (a.out_scale != nullptr) scale = a.out_scale[0];

    for (int64_t b = blockIdx.x; b < n_batches; b += gridDim.x) {
        f32x4 h[T][HT];
        int64_t row[T];
        bool valid[T];
        {   // ---- encoder: h = W_enc xn + b_enc
            f32x4 xin[T][kRfMaxKTI];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t r0 = ((b * kWaves + wv) * T + t) * 16 + c;
                valid[t] = r0 < a.n_rows;
                row[t] = valid[t] ? r0 : a.n_rows - 1;
                float nrm;
                rf_load_input<kRfMaxKTI>(a, row[t], valid[t], g, kti, xin[t], nrm);
            }
            rf_stage(s_frag, a.frag[0], HT * ksi * 64, s_bias, a.bias[0], a.hidden, HP, tid);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int to = 0; to < HT; ++to) h[t][to] = *reinterpret_cast<const f32x4 *>(s_bias + 16 * to + 4 * g);
#pragma unroll
            for (int ti = 0; ti < kRfMaxKTI; ++ti)
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
        if (a.acts != nullptr) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (valid[t]) {
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        *reinterpret_cast<f32x4 *>(a.acts + row[t] * HP + 16 * to + 4 * g) = h[t][to];
                }
        }
        // ---- hidden layers: h = sa h + sb (W relu(h) + b)
        for (int l = 1; l <= a.n_hidden; ++l) {
            rf_stage(s_frag, a.frag[l], HT * KSH * 64, s_bias, a.bias[l], a.hidden, HP, tid);
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
                for (int to = 0; to < HT; ++to)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[t][to][r] = a.sa * h[t][to][r] + a.sb * acc[t][to][r];
            if (a.acts != nullptr) {
                float *dst = a.acts + (int64_t)l * a.n_rows * HP;
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (valid[t]) {
#pragma unroll
                        for (int to = 0; to < HT; ++to)
                            *reinterpret_cast<f32x4 *>(dst + row[t] * HP + 16 * to + 4 * g) = h[t][to];
                    }
            }
        }
        // ---- decoder: y = W_dec relu(h) + b_dec (, * scale) (, relu)
        rf_stage(s_frag, a.frag[a.n_hidden + 1], ot * KSH * 64, s_bias, a.bias[a.n_hidden + 1], a.out_dim, 16 * kRfMaxOT,
                 tid);
        f32x4 y[T][kRfMaxOT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int to = 0; to < kRfMaxOT; ++to)
                y[t][to] = to < ot ? *reinterpret_cast<const f32x4 *>(s_bias + 16 * to + 4 * g) : zero;
#pragma unroll
        for (int ti = 0; ti < HT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv[T];
#pragma unroll
                for (int t = 0; t < T; ++t) pv[t] = fmaxf(h[t][ti][r], 0.f);
#pragma unroll
                for (int to = 0; to < kRfMaxOT; ++to)
                    if (to < ot) {
                        const float fa = s_frag[(to * KSH + 4 * ti + r) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < T; ++t) y[t][to] = mfma4(fa, pv[t], y[t][to]);
                    }
            }
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (valid[t]) {
#pragma unroll
                for (int to = 0; to < kRfMaxOT; ++to)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * to + 4 * g + r;
                        if (to < ot && f < a.out_dim) {
                            float v = y[t][to][r];
                            if (a.out_scale != nullptr) v = v * scale;
                            if (a.out_relu) v = fmaxf(v, 0.f);
                            a.out[row[t] * a.out_stride + f] = v;
                        }
                    }
            }
    }
}

// ================================================================================ backward
// wave-private staging images [feature][row] (leading dim kRfLd): operands of the K = rows contractions
template <int NT>
__device__ __forceinline__ void rf_stage_tiles(float *img, const f32x4 (&v)[NT], int nt, int g, int c) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * t + 4 * g + r) * kRfLd + c] = v[t][r];
        }
}
// k-step s of the K = rows contraction pairs lane group g with row 4g + s: both operands are 16-byte
// reads of feature (16 t + c), rows 4g .. 4g + 3
__device__ __forceinline__ f32x4 rf_read_k(const float *img, int t, int g, int c) {
    return *reinterpret_cast<const f32x4 *>(img + (16 * t + c) * kRfLd + 4 * g);
}

// sums the four waves' accumulator tiles in wave order through LDS and writes the block's partial:
// acc[to][ti] register r of lane (g, c) = dW[16 to + 4g + r][16 ti + c]
template <int NO, int NI>
__device__ __forceinline__ void rf_emit_dw(float *s_red, float *dst, const f32x4 (&acc)[NO][NI], int no, int ni, int O,
                                           int K, int wv, int tid, int g, int c) {
    __syncthreads();   // the fragments are no longer needed
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
    for (int i = tid; i < O * K; i += kBlock) dst[i] = s_red[i];
}
// bias gradients: dbacc[to] register r of lane (g, c) = sum over this wave's tiles of g[16 to + 4g + r][row c]
template <int NO>
__device__ __forceinline__ void rf_emit_db(float *s_redb, float *dst, const f32x4 (&dbacc)[NO], int no, int O, int wv,
                                           int tid, int g, int c) {
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
    // (the caller's next rf_stage starts with a barrier)
}

template <int HT>
__global__ __launch_bounds__(kBlock, HT > 4 ? 1 : 2) void resfcnn_bwd_kernel(const RfArgs a) {
    constexpr int KSH = 4 * HT;
    constexpr int kFragFloats = HT * (KSH > 4 * kRfMaxKTI ? KSH : 4 * kRfMaxKTI) * 64;   // >= hidden^2, hidden * in, out * hidden
    // (an image holds hidden, input or output tiles: as many rows as the widest of the three can have)
    constexpr int kImgTiles = HT > kRfMaxKTI ? HT : kRfMaxKTI;
    constexpr int kImg = 16 * kImgTiles * kRfLd;
    __shared__ __attribute__((aligned(16))) float s_frag[kFragFloats];
    __shared__ __attribute__((aligned(16))) float s_img[kWaves][2][kImg];
    __shared__ __attribute__((aligned(16))) float s_redb[16 * (HT > kRfMaxOT ? HT : kRfMaxOT)];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wv = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int kti = rf_tiles(a.in_dim), ot = rf_tiles(a.out_dim), kso = 4 * ot;
    const int HP = 16 * HT, H = a.hidden;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    // the block's tiles: one contiguous range (the gradient stream of a row is written and read by the same lane)
    const int64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t tb0 = per * blockIdx.x, tb1 = (tb0 + per < n_tiles) ? tb0 + per : n_tiles;
    float *imgG = s_img[wv][0], *imgP = s_img[wv][1];
    float *part = a.part + (int64_t)blockIdx.x * a.part_total;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float scale = 1.f;
    if (a.out_scale != nullptr) scale = a.out_scale[0];
    int poff = 0;   // running offset inside the partial block: dec W, dec b, hidden n_hidden .. 1 (W, b), enc W, enc b

    {   // ---------------------------------------------------------------- decoder
        const int L = a.n_hidden + 1;
        rf_stage(s_frag, a.fragT[L], HT * kso * 64, s_redb, nullptr, 0, 0, tid);
        const float *xl = a.acts + (int64_t)a.n_hidden * a.n_rows * HP;
        f32x4 dW[kRfMaxOT][HT], dbacc[kRfMaxOT];
#pragma unroll
        for (int to = 0; to < kRfMaxOT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) dW[to][ti] = zero;
        }
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            f32x4 go[kRfMaxOT], p[HT];
#pragma unroll
            for (int to = 0; to < kRfMaxOT; ++to) {
                go[to] = zero;
                if (to < ot) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * to + 4 * g + r;
                        if (valid && f < a.out_dim) {
                            float v = a.gout[row * a.gout_stride + f];
                            if (a.out_relu) v = a.fwd_out[row * a.out_stride + f] > 0.f ? v : 0.f;
                            go[to][r] = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xl + row * HP + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[t][r] = fmaxf(xv[r], 0.f);
            }
            // gradient at the decoder's input, through relu: the gradient of the residual stream
            f32x4 gh[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gh[t] = zero;
#pragma unroll
            for (int to = 0; to < kRfMaxOT; ++to)
                if (to < ot) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gv = go[to][r] * scale;
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
            // dW_dec (without the output scale: applied in the final reduction, which also derives the
            // scale's own gradient from these sums), db_dec
            rf_stage_tiles<kRfMaxOT>(imgG, go, ot, g, c);
            rf_stage_tiles<HT>(imgP, p, HT, g, c);
            lds_wave_order();
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) {
                const f32x4 b4 = rf_read_k(imgP, ti, g, c);
#pragma unroll
                for (int to = 0; to < kRfMaxOT; ++to)
                    if (to < ot) {
                        const f32x4 a4 = rf_read_k(imgG, to, g, c);
#pragma unroll
                        for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                    }
            }
#pragma unroll
            for (int to = 0; to < kRfMaxOT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[to][r] += go[to][r];
            lds_wave_order();
        }
        rf_emit_dw<kRfMaxOT, HT>(s_frag, part + poff, dW, ot, HT, a.out_dim, H, wv, tid, g, c);
        poff += a.out_dim * H;
        rf_emit_db<kRfMaxOT>(s_redb, part + poff, dbacc, ot, a.out_dim, wv, tid, g, c);
        poff += a.out_dim;
    }

    // ---------------------------------------------------------------- hidden layers, last to first
    for (int l = a.n_hidden; l >= 1; --l) {
        rf_stage(s_frag, a.fragT[l], HT * KSH * 64, s_redb, nullptr, 0, 0, tid);
        const float *xl = a.acts + (int64_t)(l - 1) * a.n_rows * HP;
        f32x4 dW[HT][HT], dbacc[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) dW[to][ti] = zero;
        }
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            // (gz = sb gy, the gradient at the layer's linear output, is never held: it is staged straight into its
            //  image and re-derived per use - at hidden 128 the 256 weight-gradient registers leave room for three
            //  32-register tiles of state, not four)
            f32x4 gy[HT], p[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                gy[t] = valid ? *reinterpret_cast<const f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) : zero;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xl + row * HP + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[t][r] = fmaxf(xv[r], 0.f);   // (x > 0 <=> relu(x) > 0: the mask is read off p)
                    imgG[(16 * t + 4 * g + r) * kRfLd + c] = a.sb * gy[t][r];
                    imgP[(16 * t + 4 * g + r) * kRfLd + c] = p[t][r];
                }
            }
            f32x4 gp[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) gp[t] = zero;
            lds_wave_order();
            // W^T gz: the B operand of k-step ks is read back from the staged image (a run-time loop: fully
            // unrolled, the scheduler hoists all 4 HT^2 fragment reads and spills at eight tiles)
#pragma unroll 4
            for (int ks = 0; ks < KSH; ++ks) {
                const float gzv = imgG[(16 * (ks >> 2) + 4 * g + (ks & 3)) * kRfLd + c];
#pragma unroll
                for (int t = 0; t < HT; ++t) gp[t] = mfma4(s_frag[(t * KSH + ks) * 64 + lane], gzv, gp[t]);
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = a.sa * gy[t][r] + (p[t][r] > 0.f ? gp[t][r] : 0.f);
                    *reinterpret_cast<f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) = v;
                }
            }
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[to][r] += a.sb * gy[to][r];
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) {
                const f32x4 b4 = rf_read_k(imgP, ti, g, c);
#pragma unroll
                for (int to = 0; to < HT; ++to) {
                    const f32x4 a4 = rf_read_k(imgG, to, g, c);
#pragma unroll
                    for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                }
                __builtin_amdgcn_sched_barrier(0);   // (keeps the operand reads of the next column tile from being hoisted)
            }
            lds_wave_order();
        }
        rf_emit_dw<HT, HT>(s_frag, part + poff, dW, HT, HT, H, H, wv, tid, g, c);
        poff += H * H;
        rf_emit_db<HT>(s_redb, part + poff, dbacc, HT, H, wv, tid, g, c);
        poff += H;
    }

    {   // ---------------------------------------------------------------- encoder
        const int ksh_used = KSH;
        if (a.want_enc_dx) rf_stage(s_frag, a.fragT[0], kti * ksh_used * 64, s_redb, nullptr, 0, 0, tid);
        f32x4 dW[HT][kRfMaxKTI], dbacc[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            dbacc[to] = zero;
#pragma unroll
            for (int ti = 0; ti < kRfMaxKTI; ++ti) dW[to][ti] = zero;
        }
        for (int64_t tile = tb0 + wv; tile < tb1; tile += kWaves) {
            const int64_t r0 = tile * 16 + c;
            const bool valid = r0 < a.n_rows;
            const int64_t row = valid ? r0 : a.n_rows - 1;
            f32x4 gy[HT], xin[kRfMaxKTI];
            float nrm;
#pragma unroll
            for (int t = 0; t < HT; ++t)
                gy[t] = valid ? *reinterpret_cast<const f32x4 *>(a.gstream + row * HP + 16 * t + 4 * g) : zero;
            rf_load_input<kRfMaxKTI>(a, row, valid, g, kti, xin, nrm);
            if (a.want_enc_dx) {
                // gradient at the normalised input, then through x / max(||x||, eps)
                f32x4 gxn[kRfMaxKTI];
#pragma unroll
                for (int ti = 0; ti < kRfMaxKTI; ++ti) gxn[ti] = zero;
#pragma unroll
                for (int to = 0; to < HT; ++to)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ti = 0; ti < kRfMaxKTI; ++ti)
                            if (ti < kti) gxn[ti] = mfma4(s_frag[(ti * KSH + 4 * to + r) * 64 + lane], gy[to][r], gxn[ti]);
                float dot = 0.f;
#pragma unroll
                for (int ti = 0; ti < kRfMaxKTI; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot += xin[ti][r] * gxn[ti][r];
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                // (below eps the denominator is the constant eps: no projection term)
                const bool proj = a.normalize && nrm > 1e-12f;
                if (valid) {
#pragma unroll
                    for (int ti = 0; ti < kRfMaxKTI; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = 16 * ti + 4 * g + r;
                            if (ti < kti && f < a.in_dim) {
                                float v = gxn[ti][r];
                                if (a.normalize) v = (proj ? v - xin[ti][r] * dot : v) / nrm;
                                a.gx[row * a.gx_stride + f] = v;
                            }
                        }
                }
            }
            rf_stage_tiles<HT>(imgG, gy, HT, g, c);
            rf_stage_tiles<kRfMaxKTI>(imgP, xin, kti, g, c);
            lds_wave_order();
#pragma unroll
            for (int ti = 0; ti < kRfMaxKTI; ++ti)
                if (ti < kti) {
                    const f32x4 b4 = rf_read_k(imgP, ti, g, c);
#pragma unroll
                    for (int to = 0; to < HT; ++to) {
                        const f32x4 a4 = rf_read_k(imgG, to, g, c);
#pragma unroll
                        for (int s = 0; s < 4; ++s) dW[to][ti] = mfma4(a4[s], b4[s], dW[to][ti]);
                    }
                }
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dbacc[to][r] += gy[to][r];
            lds_wave_order();
        }
        rf_emit_dw<HT, kRfMaxKTI>(s_frag, part + poff, dW, HT, kti, H, a.in_dim, wv, tid, g, c);
        poff += H * a.in_dim;
        rf_emit_db<HT>(s_redb, part + poff, dbacc, HT, H, wv, tid, g, c);
    }
}

// ---- final reduction of the per-block partials ---------------------------------------------------
struct RfReduceArgs {
    const float *part;
    int32_t n_part, part_total, n_seg, accumulate;
    int32_t off[2 * kRfMaxL + 1];   // first float of segment j inside a partial block (+ the end)
    float *dst[2 * kRfMaxL];        // destination or NULL
    int32_t scaled[2 * kRfMaxL];    // segment is multiplied by out_scale (decoder W / b)
    const float *out_scale;
    float *raw;                     // [n_raw] unscaled sums of the first n_raw entries (decoder W, b), or NULL
    int32_t n_raw, _pad;
};

// 32 parameters x 8 slices of the partial blocks per workgroup: a slice adds its blocks in order, the eight slice
// sums are added in slice order - a fixed association, and 256 instead of 32 workgroups in flight (the one-thread-
// per-parameter walk over 256 blocks was a 90 us latency chain for 32 KB of sums)
__global__ __launch_bounds__(256) void resfcnn_reduce_kernel(const RfReduceArgs a) {
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
    if (a.raw != nullptr && i < a.n_raw) a.raw[i] = s;
    int j = 0;
    while (j + 1 < a.n_seg && i >= a.off[j + 1]) ++j;
    if (a.dst[j] == nullptr) return;
    if (a.scaled[j] && a.out_scale != nullptr) s *= a.out_scale[0];
    float *p = a.dst[j] + (i - a.off[j]);
    *p = a.accumulate ? *p + s : s;
}

// gradient of the output scale: sum over the decoder's parameters of  value * (unscaled gradient)
//   d/ds sum_rows g . (s (W p + b)) = sum_{o,i} W[o][i] dWraw[o][i] + sum_o b[o] dbraw[o]
// (`raw`: the reduced unscaled sums the reduction kernel left behind)
__global__ __launch_bounds__(256) void resfcnn_scale_grad_kernel(const float *raw, const float *W, const float *b, int nW, int nb,
                                                                float *dst, int accumulate) {
    __shared__ double s_sum[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nW + nb; i += 256) {
        const float v = i < nW ? W[i] : (b != nullptr ? b[i - nW] : 0.f);
        acc += (double)v * (double)raw[i];
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s_sum[threadIdx.x] += s_sum[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[0] = (accumulate ? dst[0] : 0.f) + (float)s_sum[0];
}

// ---- host -----------------------------------------------------------------------------------------
int rf_check(const gnntrk_resfcnn *m, const char *who) {
    if (!m) return fail(GNNTRK_EINVAL, "resfcnn: NULL model");
    if (m->in_dim < 1 || m->in_dim > GNNTRK_RESFCNN_MAX_IN || m->hidden < 1 || m->hidden > GNNTRK_RESFCNN_MAX_WIDTH ||
        m->out_dim < 1 || m->out_dim > GNNTRK_RESFCNN_MAX_OUT || m->n_hidden < 0 || m->n_hidden > GNNTRK_RESFCNN_MAX_HIDDEN)
        return fail(GNNTRK_EUNSUPPORTED, "resfcnn: limits are in <= 64, hidden <= 128, out <= 32, depth - 1 <= 16");
    if (!m->W_enc || !m->W_dec) return fail(GNNTRK_EINVAL, "resfcnn: NULL weight pointer");
    for (int l = 0; l < m->n_hidden; ++l)
        if (!m->W_hid[l]) return fail(GNNTRK_EINVAL, "resfcnn: NULL weight pointer");
    if (!(m->alpha >= 0.f && m->alpha <= 1.f)) return fail(GNNTRK_EINVAL, "resfcnn: alpha must be in [0, 1]");
    (void)who;
    return GNNTRK_OK;
}

int rf_ht(int hidden) {
    const int t = rf_tiles(hidden);
    return t <= 4 ? t : t <= 6 ? 6 : 8;
}

// floats of the packed fragments: forward set, then transposed set, per layer (tile counts as the kernels
// index them: hidden widths are padded to the instantiated tile count)
struct RfLayout {
    size_t fwd[kRfMaxL], bwd[kRfMaxL], total;
    int rt_f[kRfMaxL], kt_f[kRfMaxL], rt_b[kRfMaxL], kt_b[kRfMaxL];
};
RfLayout rf_layout(const gnntrk_resfcnn *m, bool with_bwd) {
    RfLayout L;
    const int HT = rf_ht(m->hidden), KTI = rf_tiles(m->in_dim), OT = rf_tiles(m->out_dim);
    size_t off = 0;
    auto take = [&](size_t n) {
        const size_t o = off;
        off += (n + 63) / 64 * 64;
        return o;
    };
    const int nl = m->n_hidden + 2;
    for (int l = 0; l < nl; ++l) {
        const int RT = l == nl - 1 ? OT : HT, KT = l == 0 ? KTI : HT;
        L.rt_f[l] = RT;
        L.kt_f[l] = KT;
        L.fwd[l] = take((size_t)RT * 4 * KT * 64);
    }
    for (int l = 0; l < nl; ++l) {
        const int RT = l == 0 ? KTI : HT, KT = l == nl - 1 ? OT : HT;   // of W^T
        L.rt_b[l] = RT;
        L.kt_b[l] = KT;
        L.bwd[l] = with_bwd ? take((size_t)RT * 4 * KT * 64) : 0;
    }
    L.total = off;
    return L;
}

int rf_part_total(const gnntrk_resfcnn *m) {
    return m->out_dim * m->hidden + m->out_dim + m->n_hidden * (m->hidden * m->hidden + m->hidden) +
           m->hidden * m->in_dim + m->hidden;
}

int rf_bwd_grid(int64_t n_rows) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t g = (tiles + kWaves - 1) / kWaves;
    if (g > cu_count()) g = cu_count();
    return (int)(g < 1 ? 1 : g);
}

void rf_fill_args(RfArgs &a, const gnntrk_resfcnn *m, const float *frag_base, const RfLayout &L, bool with_bwd) {
    memset(&a, 0, sizeof(a));
    const int nl = m->n_hidden + 2;
    for (int l = 0; l < nl; ++l) {
        a.frag[l] = frag_base + L.fwd[l];
        a.fragT[l] = with_bwd ? frag_base + L.bwd[l] : nullptr;
        a.bias[l] = l == 0 ? m->b_enc : l == nl - 1 ? m->b_dec : m->b_hid[l - 1];
    }
    a.out_scale = m->out_scale;
    a.in_dim = m->in_dim;
    a.hidden = m->hidden;
    a.out_dim = m->out_dim;
    a.n_hidden = m->n_hidden;
    a.normalize = m->normalize;
    a.out_relu = m->out_relu;
    a.sa = (float)sqrt((double)m->alpha);          // np.sqrt(alpha) * x: a double, rounded when it meets the fp32 tensor
    a.sb = (float)sqrt(1.0 - (double)m->alpha);
}

int rf_pack(const gnntrk_resfcnn *m, float *frag_base, const RfLayout &L, bool with_bwd, hipStream_t stream) {
    RfPackArgs pa;
    memset(&pa, 0, sizeof(pa));
    const int nl = m->n_hidden + 2;
    int n = 0;
    for (int pass = 0; pass < (with_bwd ? 2 : 1); ++pass)
        for (int l = 0; l < nl; ++l) {
            const float *W = l == 0 ? m->W_enc : l == nl - 1 ? m->W_dec : m->W_hid[l - 1];
            const int out = l == nl - 1 ? m->out_dim : m->hidden, in = l == 0 ? m->in_dim : m->hidden;
            RfPackJob &j = pa.job[n++];
            j.W = W;
            j.ld = in;
            j.transposed = pass;
            j.rows = pass ? in : out;
            j.cols = pass ? out : in;
            j.rt = pass ? L.rt_b[l] : L.rt_f[l];
            j.kt = pass ? L.kt_b[l] : L.kt_f[l];
            j.dst = frag_base + (pass ? L.bwd[l] : L.fwd[l]);
        }
    pa.n_jobs = n;
    hipLaunchKernelGGL(resfcnn_pack_kernel, dim3(8, n), dim3(256), 0, stream, pa);
    return check_launch("resfcnn_pack");
}

}  // namespace
}  // namespace gnntrk

using namespace gnntrk;

extern "C" {

int32_t gnntrk_resfcnn_hidden_pad(int32_t hidden) { return 16 * rf_ht(hidden); }

size_t gnntrk_resfcnn_forward_workspace_bytes(const gnntrk_resfcnn *m) {
    if (!m || rf_check(m, "resfcnn")) return 0;
    return rf_layout(m, false).total * sizeof(float);
}

int gnntrk_resfcnn_forward(const gnntrk_resfcnn *m, const float *x, int32_t x_stride, int64_t n_rows, float *out,
                           int32_t out_stride, float *acts, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = rf_check(m, "resfcnn_forward");
    if (rc) return rc;
    if (n_rows < 0 || n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "resfcnn_forward: bad n_rows");
    if (n_rows == 0) return GNNTRK_OK;
    if (!x || !out || x_stride < m->in_dim || out_stride < m->out_dim) return fail(GNNTRK_EINVAL, "resfcnn_forward: bad rows");
    const RfLayout L = rf_layout(m, false);
    if (!workspace || workspace_bytes < L.total * sizeof(float) || ((uintptr_t)workspace & 15))
        return fail(GNNTRK_EINVAL, "resfcnn_forward: workspace too small or misaligned");
    if (acts && ((uintptr_t)acts & 15)) return fail(GNNTRK_EINVAL, "resfcnn_forward: acts must be 16-byte aligned");
    float *frag = reinterpret_cast<float *>(workspace);
    rc = rf_pack(m, frag, L, false, stream);
    if (rc) return rc;
    RfArgs a;
    rf_fill_args(a, m, frag, L, false);
    a.x = x;
    a.x_stride = x_stride;
    a.n_rows = n_rows;
    a.out = out;
    a.out_stride = out_stride;
    a.acts = acts;
    const int HT = rf_ht(m->hidden);
    const int64_t tiles = (n_rows + 15) / 16;
#define GNNTRK_RF_FWD(HT_, T_)                                                                  \
    if (HT == HT_) {                                                                            \
        int64_t grid = (tiles + kWaves * T_ - 1) / (kWaves * T_);                               \
        const int64_t cap = (int64_t)cu_count() * (HT_ > 4 ? 1 : 2);                            \
        if (grid > cap) grid = cap;                                                             \
        hipLaunchKernelGGL((resfcnn_fwd_kernel<HT_, T_>), dim3((int)grid), dim3(kBlock), 0, stream, a); \
    }
    GNNTRK_RF_FWD(1, 2) GNNTRK_RF_FWD(2, 2) GNNTRK_RF_FWD(3, 2) GNNTRK_RF_FWD(4, 2) GNNTRK_RF_FWD(6, 1) GNNTRK_RF_FWD(8, 1)
#undef GNNTRK_RF_FWD
    return check_launch("resfcnn_forward");
}

size_t gnntrk_resfcnn_backward_workspace_bytes(const gnntrk_resfcnn *m, int64_t n_rows) {
    if (!m || rf_check(m, "resfcnn") || n_rows < 0) return 0;
    const size_t frag = rf_layout(m, true).total * sizeof(float);
    const size_t gs = align_up((size_t)n_rows * 16 * rf_ht(m->hidden) * sizeof(float), 256);
    const size_t part = (size_t)rf_bwd_grid(n_rows) * rf_part_total(m) * sizeof(float);
    const size_t raw = (size_t)(m->out_dim * m->hidden + m->out_dim) * sizeof(float);
    return align_up(frag, 256) + gs + align_up(part, 256) + align_up(raw, 256);
}

int gnntrk_resfcnn_backward(const gnntrk_resfcnn *m, const float *x, int32_t x_stride, int64_t n_rows, const float *acts,
                            const float *out, int32_t out_stride, const float *gout, int32_t gout_stride, float *gx,
                            int32_t gx_stride, const gnntrk_resfcnn_grads *grads, int32_t accumulate, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = rf_check(m, "resfcnn_backward");
    if (rc) return rc;
    if (!grads) return fail(GNNTRK_EINVAL, "resfcnn_backward: NULL grads");
    if (n_rows < 0 || n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "resfcnn_backward: bad n_rows");
    if (n_rows > 0 && (!x || !acts || !gout || x_stride < m->in_dim || gout_stride < m->out_dim || ((uintptr_t)acts & 15)))
        return fail(GNNTRK_EINVAL, "resfcnn_backward: bad rows");
    if (m->out_relu && n_rows > 0 && (!out || out_stride < m->out_dim))
        return fail(GNNTRK_EINVAL, "resfcnn_backward: out_relu needs the forward's output");
    if (gx && gx_stride < m->in_dim) return fail(GNNTRK_EINVAL, "resfcnn_backward: bad gx stride");
    if (!workspace || workspace_bytes < gnntrk_resfcnn_backward_workspace_bytes(m, n_rows) || ((uintptr_t)workspace & 15))
        return fail(GNNTRK_EINVAL, "resfcnn_backward: workspace too small or misaligned");
    const RfLayout L = rf_layout(m, true);
    float *frag = reinterpret_cast<float *>(workspace);
    uint8_t *base = reinterpret_cast<uint8_t *>(workspace);
    float *gstream = reinterpret_cast<float *>(base + align_up(L.total * sizeof(float), 256));
    const int HT = rf_ht(m->hidden);
    float *part = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(gstream) +
                                            align_up((size_t)n_rows * 16 * HT * sizeof(float), 256));
    const int PT = rf_part_total(m);
    float *raw = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(part) +
                                           align_up((size_t)rf_bwd_grid(n_rows) * PT * sizeof(float), 256));
    int grid = 0;
    if (n_rows > 0) {
        rc = rf_pack(m, frag, L, true, stream);
        if (rc) return rc;
        RfArgs a;
        rf_fill_args(a, m, frag, L, true);
        a.x = x;
        a.x_stride = x_stride;
        a.n_rows = n_rows;
        a.acts = const_cast<float *>(acts);
        a.fwd_out = out;
        a.out_stride = out_stride;
        a.gout = gout;
        a.gout_stride = gout_stride;
        a.gstream = gstream;
        a.gx = gx;
        a.gx_stride = gx_stride;
        a.want_enc_dx = gx != nullptr;
        a.part = part;
        a.part_total = PT;
        grid = rf_bwd_grid(n_rows);
#define GNNTRK_RF_BWD(HT_) \
    if (HT == HT_) hipLaunchKernelGGL((resfcnn_bwd_kernel<HT_>), dim3(grid), dim3(kBlock), 0, stream, a);
        GNNTRK_RF_BWD(1) GNNTRK_RF_BWD(2) GNNTRK_RF_BWD(3) GNNTRK_RF_BWD(4) GNNTRK_RF_BWD(6) GNNTRK_RF_BWD(8)
#undef GNNTRK_RF_BWD
        rc = check_launch("resfcnn_backward");
        if (rc) return rc;
    }
    // segments of a partial block in the kernel's order: dec W, dec b, hidden n_hidden .. 1 (W, b), enc W, enc b
    RfReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.part = part;
    ra.n_part = grid;
    ra.part_total = PT;
    ra.accumulate = accumulate;
    ra.out_scale = m->out_scale;
    int n = 0, off = 0;
    auto seg = [&](float *dst, int len, int scaled) {
        ra.off[n] = off;
        ra.dst[n] = dst;
        ra.scaled[n] = scaled;
        off += len;
        ++n;
    };
    seg(grads->W_dec, m->out_dim * m->hidden, 1);
    seg(m->b_dec ? grads->b_dec : nullptr, m->out_dim, 1);
    for (int l = m->n_hidden; l >= 1; --l) {
        seg(grads->W_hid[l - 1], m->hidden * m->hidden, 0);
        seg(m->b_hid[l - 1] ? grads->b_hid[l - 1] : nullptr, m->hidden, 0);
    }
    seg(grads->W_enc, m->hidden * m->in_dim, 0);
    seg(m->b_enc ? grads->b_enc : nullptr, m->hidden, 0);
    ra.off[n] = off;
    ra.n_seg = n;
    const bool want_scale = m->out_scale && grads->out_scale;
    ra.raw = want_scale ? raw : nullptr;
    ra.n_raw = m->out_dim * m->hidden + m->out_dim;
    hipLaunchKernelGGL(resfcnn_reduce_kernel, dim3((PT + 31) / 32), dim3(256), 0, stream, ra);
    rc = check_launch("resfcnn_reduce");
    if (rc) return rc;
    if (want_scale) {
        hipLaunchKernelGGL(resfcnn_scale_grad_kernel, dim3(1), dim3(256), 0, stream, raw, m->W_dec, m->b_dec,
                           m->out_dim * m->hidden, m->out_dim, grads->out_scale, accumulate);
        rc = check_launch("resfcnn_scale_grad");
    }
    return rc;
}

}  // extern "C"
