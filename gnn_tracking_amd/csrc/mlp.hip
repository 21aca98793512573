// Fused gather-concat -> MLP -> epilogue kernels, forward and backward, for gfx950.
//
// One persistent workgroup = 4 waves (one per SIMD).  Every wave walks 16-row tiles
// of the op (rows = edges or nodes), keeps all activations in MFMA accumulator
// layout (tile_mlp.h) and never writes a hidden activation to memory.
//
//   forward  : gather/concat inputs -> fp32 MFMA chain -> epilogue -> store.
//   backward : recompute the forward from the op inputs, propagate the upstream
//              gradient through W^T with the same MFMA chain, write per-row input
//              gradient slices, and accumulate weight gradients with MFMAs whose
//              k dimension is the 16 rows of the tile.  The operands of those need
//              rows on the k axis, i.e. the transpose of the accumulator layout:
//              they go through a 5 KB wave-private LDS buffer (4 ds_write_b32 +
//              1 ds_read_b128 per 16x16 tile).
//
// Weights are packed once per workgroup into LDS as ready-to-use A fragments.
// Bound: fp32 matrix pipe (157 TF) for hidden width 40; see DESIGN.md.
#include "tile_mlp.h"

#include "host_util.h"

namespace gnntrk {

// ------------------------------------------------------------------- forward
template <int KT, int HT>
struct FwdSmem {
    float w1[HT * 4 * KT * 64];
    float w2[HT * 4 * HT * 64];
    float w3[4 * HT * 64];
    f32x4 b1[HT * 64];
    f32x4 b2[HT * 64];
    f32x4 b3[64];
    SegTable segs;
};

template <int KT, int HT>
__global__ __launch_bounds__(kBlock) void mlp_fwd_kernel(const gnntrk_mlp_fwd_args a) {
    __shared__ __attribute__((aligned(16))) FwdSmem<KT, HT> sm;
    Maps mp;
    mp.in = make_dimmap(a.mlp.in_dim);
    mp.hid = make_dimmap(a.mlp.hidden);
    mp.out = make_dimmap(a.mlp.out_dim);
    mp.three = a.mlp.n_layers == 3;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int last = a.mlp.n_layers - 1;

    fill_frags(sm.w1, a.mlp.W[0], a.mlp.in_dim, mp.hid, mp.in, false, tid, kBlock);
    if (mp.three) fill_frags(sm.w2, a.mlp.W[1], a.mlp.hidden, mp.hid, mp.hid, false, tid, kBlock);
    fill_frags(sm.w3, a.mlp.W[last], a.mlp.hidden, mp.out, mp.hid, false, tid, kBlock);
    fill_bias(sm.b1, a.mlp.b[0], mp.hid, tid, kBlock);
    if (mp.three) fill_bias(sm.b2, a.mlp.b[1], mp.hid, tid, kBlock);
    fill_bias(sm.b3, a.mlp.b[last], mp.out, tid, kBlock);
    stage_segs(sm.segs, a.seg, nullptr, a.n_seg, tid);
    __syncthreads();

    InSlot slot[KT * 4];
    unsigned relu_bits;
    setup_in_slots<KT>(sm.segs, a.n_seg, mp.in, g, slot, relu_bits);

    int fo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) fo[r] = feat_of(mp.out, 0, g, r);

    const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
    TileSched sch = make_sched(n_tiles);
    for (int64_t tile = sch.cur; tile < sch.end; tile += sch.step) {
        const int64_t row = tile * kTileRows + c;
        const bool valid = row < a.n_rows;
        f32x4 bin[KT];
        load_inputs<KT>(slot, relu_bits, mp.in, row, valid, bin);
        f32x4 a1[HT], a2[HT], y;
        mlp_tile_forward<KT, HT>(mp, sm.w1, sm.w2, sm.w3, sm.b1, sm.b2, sm.b3, lane, bin, a1, a2, y,
                                 true);
        if (valid) {
            const int64_t orow = a.out_idx ? (int64_t)a.out_idx[row] : row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (fo[r] >= 0) {
                    float v = y[r];
                    if (a.epilogue == GNNTRK_EPI_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
                        v = a.ca * a.res[row * a.res_stride + fo[r]] + a.cb * v;
                    } else if (a.epilogue == GNNTRK_EPI_SIGMOID) {
                        v = a.ca + a.cb * sigmoidf_(v);
                    }
                    a.out[orow * a.out_stride + fo[r]] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ backward
template <int KT, int HT>
struct BwdSmem {
    float w1[HT * 4 * KT * 64];   // W1   rows=hid, k=in
    float w2[HT * 4 * HT * 64];   // W2   rows=hid, k=hid
    float w3[4 * HT * 64];        // Wout rows=out, k=hid
    float w1t[KT * 4 * HT * 64];  // W1^T rows=in,  k=hid
    float w2t[HT * 4 * HT * 64];  // W2^T rows=hid, k=hid
    float w3t[HT * 4 * 64];       // Wout^T rows=hid, k=out
    f32x4 b1[HT * 64];
    f32x4 b2[HT * 64];
    f32x4 b3[64];
    float tb[kWaves][kTbRows * kTbLd];  // wave-private transpose buffers
    SegTable segs;
};

// accumulator layout -> "rows on k" layout: xT[t][s] = x[feature 16t + c][row 4g + s]
template <int NT>
__device__ __forceinline__ void transpose_tiles(float *tb, const DimMap &map, int g, int c,
                                                const f32x4 (&x)[NT], f32x4 (&xT)[NT]) {
    lds_wave_sync();  // previous readers of tb are done
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < map.nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = feat_of(map, t, g, r);
                if (f >= 0) tb[f * kTbLd + c] = x[t][r];
            }
        }
    lds_wave_sync();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < map.nt)
            xT[t] = *reinterpret_cast<const f32x4 *>(&tb[(16 * t + c) * kTbLd + 4 * g]);
        else
            xT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

struct BwdPartLayout {  // float offsets inside one wave's partial block
    int w[3], b[3], total;
};
__host__ __device__ inline BwdPartLayout part_layout(const gnntrk_mlp &m) {
    BwdPartLayout p;
    int off = 0;
    for (int i = 0; i < 3; ++i) {
        p.w[i] = p.b[i] = -1;
        if (i < m.n_layers) {
            const int rows = (i == m.n_layers - 1) ? m.out_dim : m.hidden;
            const int cols = (i == 0) ? m.in_dim : m.hidden;
            p.w[i] = off;
            off += rows * cols;
            p.b[i] = off;
            off += rows;
        }
    }
    p.total = off;
    return p;
}

// dW tile (rows 16to.., cols 16ti..) in accumulator layout -> partial block
__device__ __forceinline__ void store_dw_tile(float *dst, int rows, int cols, int to, int ti, int g,
                                              int c, const f32x4 &v) {
    const int i = 16 * ti + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * to + 4 * g + r;
        if (o < rows && i < cols) dst[o * cols + i] = v[r];
    }
}

template <int KT, int HT>
__global__ __launch_bounds__(kBlock) void mlp_bwd_kernel(const gnntrk_mlp_bwd_args a,
                                                         float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) BwdSmem<KT, HT> sm;
    Maps mp;
    mp.in = make_dimmap(a.mlp.in_dim);
    mp.hid = make_dimmap(a.mlp.hidden);
    mp.out = make_dimmap(a.mlp.out_dim);
    mp.three = a.mlp.n_layers == 3;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, wv = tid >> 6;
    const int last = a.mlp.n_layers - 1;

    fill_frags(sm.w1, a.mlp.W[0], a.mlp.in_dim, mp.hid, mp.in, false, tid, kBlock);
    fill_frags(sm.w1t, a.mlp.W[0], a.mlp.in_dim, mp.in, mp.hid, true, tid, kBlock);
    if (mp.three) {
        fill_frags(sm.w2, a.mlp.W[1], a.mlp.hidden, mp.hid, mp.hid, false, tid, kBlock);
        fill_frags(sm.w2t, a.mlp.W[1], a.mlp.hidden, mp.hid, mp.hid, true, tid, kBlock);
    }
    fill_frags(sm.w3, a.mlp.W[last], a.mlp.hidden, mp.out, mp.hid, false, tid, kBlock);
    fill_frags(sm.w3t, a.mlp.W[last], a.mlp.hidden, mp.hid, mp.out, true, tid, kBlock);
    fill_bias(sm.b1, a.mlp.b[0], mp.hid, tid, kBlock);
    if (mp.three) fill_bias(sm.b2, a.mlp.b[1], mp.hid, tid, kBlock);
    fill_bias(sm.b3, a.mlp.b[last], mp.out, tid, kBlock);
    for (int i = tid; i < kWaves * kTbRows * kTbLd; i += kBlock) (&sm.tb[0][0])[i] = 0.f;
    stage_segs(sm.segs, a.seg, a.gseg, a.n_seg, tid);
    __syncthreads();
    float *tb = sm.tb[wv];

    InSlot slot[KT * 4];
    unsigned relu_bits;
    setup_in_slots<KT>(sm.segs, a.n_seg, mp.in, g, slot, relu_bits);

    // per-lane gradient slots (same feature <-> (tile,reg) map as the loader)
    float *gbase[KT * 4];
    const int32_t *gidx[KT * 4];
    int32_t gstride[KT * 4];
    unsigned gacc_bits = 0;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gbase[t * 4 + r] = nullptr;
            gidx[t * 4 + r] = nullptr;
            gstride[t * 4 + r] = 0;
            const int f = feat_of(mp.in, t, g, r);
            int off = 0;
            for (int j = 0; j < a.n_seg; ++j) {
                const int d = sm.segs.dim[j];
                if (f >= off && f < off + d && sm.segs.gptr[j] != nullptr) {
                    gbase[t * 4 + r] = sm.segs.gptr[j] + (f - off);
                    gidx[t * 4 + r] = sm.segs.gidx[j];
                    gstride[t * 4 + r] = sm.segs.gstride[j];
                    if (sm.segs.gacc[j]) gacc_bits |= 1u << (t * 4 + r);
                }
                off += d;
            }
        }

    int fo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) fo[r] = feat_of(mp.out, 0, g, r);
    const bool need_y = a.epilogue == GNNTRK_EPI_RELU || a.epilogue == GNNTRK_EPI_SIGMOID;
    const bool want_dw = a.gW[0] != nullptr;

    // weight-gradient accumulators (accumulator layout: rows = out feature, cols = in feature)
    f32x4 dW1[HT][KT], dW2[HT][HT], dW3[HT], db1[HT], db2[HT], db3;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HT; ++i) {
#pragma unroll
        for (int j = 0; j < KT; ++j) dW1[i][j] = zero4;
#pragma unroll
        for (int j = 0; j < HT; ++j) dW2[i][j] = zero4;
        dW3[i] = zero4;
        db1[i] = zero4;
        db2[i] = zero4;
    }
    db3 = zero4;

    const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
    TileSched sch = make_sched(n_tiles);
    for (int64_t tile = sch.cur; tile < sch.end; tile += sch.step) {
        const int64_t row = tile * kTileRows + c;
        const bool valid = row < a.n_rows;
        f32x4 bin[KT];
        load_inputs<KT>(slot, relu_bits, mp.in, row, valid, bin);
        f32x4 a1[HT], a2[HT], y;
        mlp_tile_forward<KT, HT>(mp, sm.w1, sm.w2, sm.w3, sm.b1, sm.b2, sm.b3, lane, bin, a1, a2, y,
                                 need_y);

        // upstream gradient in B layout (k = out feature), epilogue differentiated
        f32x4 gy = zero4;
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fo[r] >= 0) {
                    float v = 0.f;
                    {
                        const int64_t rr = a.gout[0].idx ? (int64_t)a.gout[0].idx[row] : row;
                        v = a.gout[0].ptr[rr * a.gout[0].stride + fo[r]];
                    }
                    if (a.n_gout > 1) {
                        const int64_t rr = a.gout[1].idx ? (int64_t)a.gout[1].idx[row] : row;
                        v += a.gout[1].ptr[rr * a.gout[1].stride + fo[r]];
                    }
                    if (a.epilogue == GNNTRK_EPI_RELU) {
                        v = y[r] > 0.f ? v : 0.f;
                    } else if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
                        v *= a.cb;
                    } else if (a.epilogue == GNNTRK_EPI_SIGMOID) {
                        const float s = sigmoidf_(y[r]);
                        v *= a.cb * s * (1.f - s);
                    }
                    gy[r] = v;
                }
        }

        // delta at the last hidden layer: (Wout^T gy) * relu'
        f32x4 dl[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) dl[to] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (kvalid(mp.out, 0, r)) {
                const int ks = kindex(mp.out, 0, r);
#pragma unroll
                for (int to = 0; to < HT; ++to)
                    if (to < mp.hid.nt)
                        dl[to] = mfma4(sm.w3t[(to * mp.out.ks + ks) * 64 + lane], gy[r], dl[to]);
            }
        f32x4 d1[HT];
        if (mp.three) {
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) dl[to][r] = a2[to][r] > 0.f ? dl[to][r] : 0.f;
#pragma unroll
            for (int to = 0; to < HT; ++to) d1[to] = zero4;
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kvalid(mp.hid, t, r)) {
                        const int ks = kindex(mp.hid, t, r);
#pragma unroll
                        for (int to = 0; to < HT; ++to)
                            if (to < mp.hid.nt)
                                d1[to] = mfma4(sm.w2t[(to * mp.hid.ks + ks) * 64 + lane], dl[t][r],
                                               d1[to]);
                    }
        } else {
#pragma unroll
            for (int to = 0; to < HT; ++to) d1[to] = dl[to];
        }
#pragma unroll
        for (int to = 0; to < HT; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[to][r] = a1[to][r] > 0.f ? d1[to][r] : 0.f;

        // input gradient: W1^T d1 (rows = concatenated input features)
        f32x4 gin[KT];
#pragma unroll
        for (int ti = 0; ti < KT; ++ti) gin[ti] = zero4;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kvalid(mp.hid, t, r)) {
                    const int ks = kindex(mp.hid, t, r);
#pragma unroll
                    for (int ti = 0; ti < KT; ++ti)
                        if (ti < mp.in.nt)
                            gin[ti] = mfma4(sm.w1t[(ti * mp.hid.ks + ks) * 64 + lane], d1[t][r],
                                            gin[ti]);
                }
        if (valid) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *gb = gbase[t * 4 + r];
                    if (gb != nullptr) {
                        float v = gin[t][r];
                        if (((relu_bits >> (t * 4 + r)) & 1u) && !(bin[t][r] > 0.f)) v = 0.f;
                        const int64_t rr = gidx[t * 4 + r] ? (int64_t)gidx[t * 4 + r][row] : row;
                        float *p = gb + rr * gstride[t * 4 + r];
                        if ((gacc_bits >> (t * 4 + r)) & 1u) v += *p;
                        *p = v;
                    }
                }
        }

        // parameter gradients: MFMA with k = the 16 rows of this tile
        if (want_dw) {
#pragma unroll
            for (int to = 0; to < HT; ++to) {
                db1[to] += d1[to];
                if (mp.three) db2[to] += dl[to];
            }
            db3 += gy;

            f32x4 gyv[1] = {gy}, gyT[1];
            transpose_tiles<1>(tb, mp.out, g, c, gyv, gyT);
            f32x4 hT[HT];  // last hidden activation, rows on k
            if (mp.three)
                transpose_tiles<HT>(tb, mp.hid, g, c, a2, hT);
            else
                transpose_tiles<HT>(tb, mp.hid, g, c, a1, hT);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ti = 0; ti < HT; ++ti)
                    if (ti < mp.hid.nt) dW3[ti] = mfma4(gyT[0][s], hT[ti][s], dW3[ti]);

            if (mp.three) {
                f32x4 dT[HT];
                transpose_tiles<HT>(tb, mp.hid, g, c, dl, dT);
                transpose_tiles<HT>(tb, mp.hid, g, c, a1, hT);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        if (to < mp.hid.nt) {
#pragma unroll
                            for (int ti = 0; ti < HT; ++ti)
                                if (ti < mp.hid.nt)
                                    dW2[to][ti] = mfma4(dT[to][s], hT[ti][s], dW2[to][ti]);
                        }
            }
            {
                f32x4 dT[HT], mT[KT];
                transpose_tiles<HT>(tb, mp.hid, g, c, d1, dT);
                transpose_tiles<KT>(tb, mp.in, g, c, bin, mT);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        if (to < mp.hid.nt) {
#pragma unroll
                            for (int ti = 0; ti < KT; ++ti)
                                if (ti < mp.in.nt)
                                    dW1[to][ti] = mfma4(dT[to][s], mT[ti][s], dW1[to][ti]);
                        }
            }
        }
    }

    if (want_dw) {
        const BwdPartLayout pl = part_layout(a.mlp);
        float *dst = part + (int64_t)(blockIdx.x * kWaves + wv) * pl.total;
        // weights
#pragma unroll
        for (int to = 0; to < HT; ++to) {
#pragma unroll
            for (int ti = 0; ti < KT; ++ti)
                store_dw_tile(dst + pl.w[0], a.mlp.hidden, a.mlp.in_dim, to, ti, g, c, dW1[to][ti]);
            if (mp.three) {
#pragma unroll
                for (int ti = 0; ti < HT; ++ti)
                    store_dw_tile(dst + pl.w[1], a.mlp.hidden, a.mlp.hidden, to, ti, g, c,
                                  dW2[to][ti]);
            }
            store_dw_tile(dst + pl.w[last], a.mlp.out_dim, a.mlp.hidden, 0, to, g, c, dW3[to]);
        }
        // biases: reduce the per-lane partial sums over the 16 row-lanes
#pragma unroll
        for (int to = 0; to < HT; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v1 = db1[to][r], v2 = db2[to][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    v1 += __shfl_xor(v1, m);
                    v2 += __shfl_xor(v2, m);
                }
                const int f = feat_of(mp.hid, to, g, r);
                if (c == 0 && f >= 0) {
                    dst[pl.b[0] + f] = v1;
                    if (mp.three) dst[pl.b[1] + f] = v2;
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = db3[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
            if (c == 0 && fo[r] >= 0) dst[pl.b[last] + fo[r]] = v;
        }
    }
}

// fixed-order reduction of the per-wave partial blocks into the gradient tensors
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float *__restrict__ part,
                                                                 int n_part, int total,
                                                                 gnntrk_mlp mlp, float *gW0,
                                                                 float *gW1, float *gW2, float *gb0,
                                                                 float *gb1, float *gb2,
                                                                 int accumulate) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= total) return;
    float s = 0.f;
    for (int w = 0; w < n_part; ++w) s += part[(int64_t)w * total + p];
    const BwdPartLayout pl = part_layout(mlp);
    float *gW[3] = {gW0, gW1, gW2};
    float *gb[3] = {gb0, gb1, gb2};
    for (int i = 0; i < 3; ++i) {
        if (pl.w[i] < 0) continue;
        const int wn = pl.b[i] - pl.w[i];
        const int bn = ((i + 1 < 3 && pl.w[i + 1] >= 0) ? pl.w[i + 1] : pl.total) - pl.b[i];
        if (p >= pl.w[i] && p < pl.w[i] + wn) {
            if (gW[i]) gW[i][p - pl.w[i]] = accumulate ? gW[i][p - pl.w[i]] + s : s;
        } else if (p >= pl.b[i] && p < pl.b[i] + bn) {
            if (gb[i]) gb[i][p - pl.b[i]] = accumulate ? gb[i][p - pl.b[i]] + s : s;
        }
    }
}

// ------------------------------------------------------------------ launchers
static int check_mlp(const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg) {
    if (m.n_layers != 2 && m.n_layers != 3) return fail(GNNTRK_EUNSUPPORTED, "mlp: n_layers must be 2 or 3");
    if (m.in_dim < 1 || m.in_dim > GNNTRK_MAX_IN) return fail(GNNTRK_EUNSUPPORTED, "mlp: in_dim out of range [1,48]");
    if (m.hidden < 1 || m.hidden > GNNTRK_MAX_HIDDEN) return fail(GNNTRK_EUNSUPPORTED, "mlp: hidden out of range [1,64]");
    if (m.out_dim < 1 || m.out_dim > GNNTRK_MAX_OUT) return fail(GNNTRK_EUNSUPPORTED, "mlp: out_dim out of range [1,16]");
    if (n_seg < 1 || n_seg > GNNTRK_MAX_SEGS) return fail(GNNTRK_EINVAL, "mlp: bad segment count");
    int tot = 0;
    for (int j = 0; j < n_seg; ++j) {
        if (!seg[j].ptr || seg[j].dim < 1 || seg[j].stride < seg[j].dim)
            return fail(GNNTRK_EINVAL, "mlp: bad segment descriptor");
        tot += seg[j].dim;
    }
    if (tot != m.in_dim) return fail(GNNTRK_EINVAL, "mlp: segment dims do not sum to in_dim");
    for (int i = 0; i < m.n_layers; ++i)
        if (!m.W[i]) return fail(GNNTRK_EINVAL, "mlp: NULL weight pointer");
    return GNNTRK_OK;
}

static int grid_for(int64_t n_rows, int blocks_per_cu) {
    const int64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    int64_t g = (tiles + kWaves - 1) / kWaves;
    const int64_t cap = (int64_t)cu_count() * blocks_per_cu;
    if (g > cap) g = cap;
    if (g >= 8) g -= g % 8;  // XCD-aware schedule wants a multiple of 8
    if (g < 1) g = 1;
    return (int)g;
}

constexpr int kBwdBlocksPerCu = 2;
constexpr int kFwdBlocksPerCu = 4;

#define GNNTRK_DISPATCH(KTn, HTn, CALL)          \
    if (kt <= 1 && ht <= 1) { CALL(1, 1); }      \
    else if (kt <= 1 && ht <= 3) { CALL(1, 3); } \
    else if (kt <= 2 && ht <= 2) { CALL(2, 2); } \
    else if (kt <= 2 && ht <= 3) { CALL(2, 3); } \
    else { CALL(3, 4); }

int mlp_forward_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_forward: NULL args");
    int rc = check_mlp(a->mlp, a->n_seg, a->seg);
    if (rc) return rc;
    if (!a->out || a->out_stride < a->mlp.out_dim) return fail(GNNTRK_EINVAL, "mlp_forward: bad output");
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_forward: bad epilogue");
    if (a->epilogue == GNNTRK_EPI_RESIDUAL && (!a->res || a->res_stride < a->mlp.out_dim))
        return fail(GNNTRK_EINVAL, "mlp_forward: residual epilogue needs res");
    if (a->n_rows < 0) return fail(GNNTRK_EINVAL, "mlp_forward: negative n_rows");
    if (a->n_rows == 0) return GNNTRK_OK;
    const int kt = (a->mlp.in_dim + 15) / 16, ht = (a->mlp.hidden + 15) / 16;
    const int grid = grid_for(a->n_rows, kFwdBlocksPerCu);
#define CALL_FWD(K, H)                                                                  \
    {                                                                                   \
        auto kfn = mlp_fwd_kernel<K, H>;                                                \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);               \
    }
    GNNTRK_DISPATCH(kt, ht, CALL_FWD)
#undef CALL_FWD
    return check_launch("mlp_forward");
}

size_t mlp_backward_ws_bytes(const gnntrk_mlp *m) {
    if (!m) return 0;
    const BwdPartLayout pl = part_layout(*m);
    return (size_t)cu_count() * kBwdBlocksPerCu * kWaves * (size_t)pl.total * sizeof(float);
}

int mlp_backward_launch(const gnntrk_mlp_bwd_args *a, void *ws, size_t ws_bytes,
                        hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_backward: NULL args");
    int rc = check_mlp(a->mlp, a->n_seg, a->seg);
    if (rc) return rc;
    if (a->n_gout < 1 || a->n_gout > 2 || !a->gout[0].ptr || (a->n_gout == 2 && !a->gout[1].ptr))
        return fail(GNNTRK_EINVAL, "mlp_backward: bad upstream gradient terms");
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_backward: bad epilogue");
    if (a->n_rows < 0) return fail(GNNTRK_EINVAL, "mlp_backward: negative n_rows");
    const bool want_dw = a->gW[0] != nullptr;
    if (want_dw) {
        for (int i = 0; i < a->mlp.n_layers; ++i)
            if (!a->gW[i]) return fail(GNNTRK_EINVAL, "mlp_backward: gW must be all set or all NULL");
        if (!ws || ws_bytes < mlp_backward_ws_bytes(&a->mlp))
            return fail(GNNTRK_EINVAL, "mlp_backward: workspace too small");
    }
    const BwdPartLayout pl = part_layout(a->mlp);
    int grid = 0;
    if (a->n_rows > 0) {
        const int kt = (a->mlp.in_dim + 15) / 16, ht = (a->mlp.hidden + 15) / 16;
        grid = grid_for(a->n_rows, kBwdBlocksPerCu);
        float *part = reinterpret_cast<float *>(ws);
#define CALL_BWD(K, H)                                                                  \
    {                                                                                   \
        auto kfn = mlp_bwd_kernel<K, H>;                                                \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part);         \
    }
        GNNTRK_DISPATCH(kt, ht, CALL_BWD)
#undef CALL_BWD
        rc = check_launch("mlp_backward");
        if (rc) return rc;
    }
    if (want_dw) {
        const int n_part = grid * kWaves;
        const int rgrid = (pl.total + kBlock - 1) / kBlock;
        auto rfn = reduce_partials_kernel;
        hipLaunchKernelGGL(rfn, dim3(rgrid), dim3(kBlock), 0, stream,
                           reinterpret_cast<const float *>(ws), n_part, pl.total, a->mlp, a->gW[0],
                           a->gW[1], a->gW[2], a->gb[0], a->gb[1], a->gb[2], a->accumulate_params);
        rc = check_launch("mlp_backward(reduce)");
    }
    return rc;
}

}  // namespace gnntrk
