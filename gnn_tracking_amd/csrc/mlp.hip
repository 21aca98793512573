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
#include <atomic>
#include "tile_mlp.h"

#include "host_util.h"

namespace gnntrk {

// Weight/bias operand policy: static shapes keep every fragment in registers,
// generic shapes read them from LDS at each use.
template <class Dims>
struct OperandPolicy {  // DynDims
    template <int N>
    using Frags = LdsFrags<N>;
    template <int N>
    using Bias = LdsBias<N>;
};
template <int KSI, int KSH, int KSO, bool THREE, int NIT>
struct OperandPolicy<StaticDims<KSI, KSH, KSO, THREE, NIT>> {
    template <int N>
    using Frags = RegFrags<N>;
    template <int N>
    using Bias = RegBias<N>;
};

// exact k-step counts for LDS sizing: static shapes know them, generic ones take the tile max
template <class Dims, int KT, int HT>
struct KsOf {
    static constexpr int in = 4 * KT, hid = 4 * HT, out = 4;
};
template <int KSI, int KSH, int KSO, bool THREE, int NIT, int KT, int HT>
struct KsOf<StaticDims<KSI, KSH, KSO, THREE, NIT>, KT, HT> {
    static constexpr int in = KSI, hid = KSH, out = KSO;
};

// ------------------------------------------------------------------- forward
template <int KT, int HT>
struct FwdSmem {
    float w1[HT * 4 * KT * 64];
    float w2[HT * 4 * HT * 64];
    float w3[4 * HT * 64];
    f32x4 b1[HT * 64];
    f32x4 b2[HT * 64];
    f32x4 b3[64];
    float sin[kWaves][(16 * KT + 2) * kTbLd];  // input staging per wave ([feature][row])
    SegTable segs;
    LoadList ll;
};

template <int KT, int HT, class Dims>
__global__ __launch_bounds__(kBlock) void mlp_fwd_kernel(const gnntrk_mlp_fwd_args a) {
    __shared__ __attribute__((aligned(16))) FwdSmem<KT, HT> sm;
    using OP = OperandPolicy<DynDims>;  // operands from LDS: thread-level parallelism (4-5 waves
                                        // per SIMD) hides the gather latency of the short tiles
    constexpr int NI = Dims::kItems > 0 ? Dims::kItems : 4 * KT + 4;
    const Dims dm = make_dims<Dims>(a.mlp);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, wv = tid >> 6;
    const int part = g;
    const int last = a.mlp.n_layers - 1;

    fill_frags(sm.w1, a.mlp.W[0], a.mlp.in_dim, dm.hid, dm.in, false, tid, kBlock);
    if (dm.three()) fill_frags(sm.w2, a.mlp.W[1], a.mlp.hidden, dm.hid, dm.hid, false, tid, kBlock);
    fill_frags(sm.w3, a.mlp.W[last], a.mlp.hidden, dm.out, dm.hid, false, tid, kBlock);
    fill_bias(sm.b1, a.mlp.b[0], dm.hid, tid, kBlock);
    if (dm.three()) fill_bias(sm.b2, a.mlp.b[1], dm.hid, tid, kBlock);
    fill_bias(sm.b3, a.mlp.b[last], dm.out, tid, kBlock);
    stage_segs(sm.segs, a.seg, nullptr, a.n_seg, tid);
    for (int i = tid; i < kWaves * (16 * KT + 2) * kTbLd; i += kBlock) (&sm.sin[0][0])[i] = 0.f;
    __syncthreads();
    if (tid == 0) build_load_list(sm.ll, sm.segs, a.n_seg);
    __syncthreads();

    const int nti = (dm.in_ks() + 3) >> 2;
    const int nth = (dm.hid_ks() + 3) >> 2;
    typename OP::template Frags<HT * KT * 4> w1;
    typename OP::template Frags<HT * HT * 4> w2;
    typename OP::template Frags<HT * 4> w3;
    typename OP::template Bias<HT> b1, b2;
    w1.load(sm.w1, nth * dm.in_ks(), lane);
    w2.load(sm.w2, dm.three() ? nth * dm.hid_ks() : 0, lane);
    w3.load(sm.w3, dm.hid_ks(), lane);
    b1.load(sm.b1, nth, lane);
    b2.load(sm.b2, dm.three() ? nth : 0, lane);
    const f32x4 b3 = sm.b3[lane];

    Items<NI, Dims::kStatic> it;
    it.load(sm.ll);
    int boff[KT * 4];
    operand_offsets<KT>(dm.in, g, c, 16 * KT + 1, boff);

    int fo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) fo[r] = feat_of(dm.out, 0, g, r);
    const gci_ptr out_idx = (gci_ptr)a.out_idx;
    const gf_ptr outp = (gf_ptr)a.out;
    const gcf_ptr resp = (gcf_ptr)a.res;

    const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
    const TileSched sch = make_sched(n_tiles);
    int64_t tile = sch.cur;
    if (tile >= sch.end) return;
    const int64_t last_row = a.n_rows - 1;
    const bool ld_on = !(a.debug_flags & 2);
    auto clamp_row = [&](int64_t t) {
        const int64_t r = t * kTileRows + c;
        return r < last_row ? r : last_row;
    };
    auto row_valid = [&](int64_t t) { return t < sch.end && t * kTileRows + c < a.n_rows; };

    // software pipeline: values of tile n+1 (registers) and row ids of tile n+2 are in flight
    // while tile n computes; the values are staged into LDS at the end of tile n
    int32_t rid_n[NI];
    float pv[NI];
    float *sc = sm.sin[wv];
    {
        int32_t rid_c[NI];
        item_row_ids<NI>(it, clamp_row(tile), rid_c);
        item_values<NI>(it, rid_c, part, row_valid(tile) && ld_on, pv);
        stage_items<NI>(it, sc, part, c, pv);
        item_row_ids<NI>(it, clamp_row(tile + sch.step < sch.end ? tile + sch.step : tile), rid_n);
    }
    // counted loop on a scalar trip count: a plain do-while for the compiler (loop-carried
    // MFMA accumulators are then updated in place instead of being copied every iteration)
    const int n_iter = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tile < sch.end ? (sch.end - tile + sch.step - 1) / sch.step : 0));
    for (int iter = 0; iter < n_iter; ++iter, tile += sch.step) {
        const int64_t row = tile * kTileRows + c;
        const bool valid = row < a.n_rows;
        item_values<NI>(it, rid_n, part, row_valid(tile + sch.step) && ld_on, pv);
        {
            const int64_t t2 = tile + 2 * sch.step;
            item_row_ids<NI>(it, clamp_row(t2 < sch.end ? t2 : tile), rid_n);
        }
        int64_t orow = row;
        if (valid && out_idx) orow = out_idx[row];

        lds_wave_order();
        f32x4 bin[KT];
        read_operand<KT>(sc, boff, nti, bin);
        lds_wave_order();  // all lanes hold their operands: the buffer may be restaged
        f32x4 a1[HT], a2[HT];
        mlp_layer1<KT, HT>(dm, w1, b1, lane, bin, a1);
        mlp_layer2<HT>(dm, w2, b2, lane, a1, a2);
        const f32x4 y = mlp_layer3<HT>(dm, w3, b3, lane, a2);
        if (valid && !((a.debug_flags & 1) && y[0] != 12345.678f)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (fo[r] >= 0) {
                    float v = y[r];
                    if (a.epilogue == GNNTRK_EPI_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
                        v = a.ca * resp[row * a.res_stride + fo[r]] + a.cb * v;
                    } else if (a.epilogue == GNNTRK_EPI_SIGMOID) {
                        v = a.ca + a.cb * sigmoidf_(v);
                    }
                    outp[orow * a.out_stride + fo[r]] = v;
                }
            }
        }
        stage_items<NI>(it, sc, part, c, pv);
    }
}

// ------------------------------------------------------------------ backward
template <int KT, int HT, int WPB, class KS>
struct BwdSmem {
    float w1[HT * KS::in * 64];    // W1   rows=hid, k=in
    float w2[HT * KS::hid * 64];   // W2   rows=hid, k=hid
    float w3[KS::hid * 64];        // Wout rows=out, k=hid
    float w1t[KT * KS::hid * 64];  // W1^T rows=in,  k=hid
    float w2t[HT * KS::hid * 64];  // W2^T rows=hid, k=hid
    float w3t[HT * KS::out * 64];  // Wout^T rows=hid, k=out
    f32x4 b1[HT * 64];
    f32x4 b2[HT * 64];
    f32x4 b3[64];
    // wave-private buffers, all [feature][row] with leading dim kTbLd; each has two extra rows:
    // a write-only garbage row (padding registers) and a never-written zero row
    float sin[WPB][(16 * KT + 2) * kTbLd];  // staged inputs = m (B operand) and m^T (dW1)
    float sgy[WPB][(4 * KS::out + 2) * kTbLd];  // staged upstream gradient
    float tb0[WPB][(16 * KT + 2) * kTbLd];  // gy^T, then the outgoing input gradient
    float tb[WPB][2][(16 * HT + 2) * kTbLd];  // a1 / dl and a2 / d1 transposes
    SegTable segs;
    LoadList ll;
};

// accumulator layout -> LDS [feature][row]; padding registers go to the garbage row
template <int NT>
__device__ __forceinline__ void tr_offsets(const DimMap &map, int g, int c, int garbage_row,
                                           int (&off)[NT * 4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(map, t, g, r);
            off[t * 4 + r] = (f >= 0 ? f : garbage_row) * kTbLd + c;
        }
}
template <int NT>
__device__ __forceinline__ void tr_write(float *tb, const int (&off)[NT * 4], int nt, int ks,
                                         const f32x4 (&x)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (t < nt && 4 * t + r < ks) tb[off[t * 4 + r]] = x[t][r];
}
// "rows on k" operand: xT[t][s] = buffer[feature 16t + c][row 4g + s]
template <int NT>
__device__ __forceinline__ void tr_read(const float *tb, int nt, int g, int c, f32x4 (&xT)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt)
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

// dW tile (rows 16to.., cols 16ti..) in accumulator layout -> partial block.  With the
// constant-one row trick the column `cols` of the tile is the bias gradient.
__device__ __forceinline__ void store_dw_tile(float *dst, float *dstb, int rows, int cols, int to,
                                              int ti, int g, int c, const f32x4 &v) {
    const int i = 16 * ti + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * to + 4 * g + r;
        if (o < rows) {
            if (i < cols)
                dst[o * cols + i] = v[r];
            else if (i == cols && dstb != nullptr)
                dstb[o] = v[r];
        }
    }
}

// WPB waves per workgroup: the static shapes run 8 (two per SIMD, 256 registers each,
// operands in LDS): the second wave of a SIMD covers the first one's memory and LDS
// waits; the generic shapes keep 4 (their LDS footprint does not allow more).
template <int KT, int HT, class Dims, int WPB>
__global__ __launch_bounds__(WPB * 64) void mlp_bwd_kernel(const gnntrk_mlp_bwd_args a,
                                                           float *__restrict__ part_out) {
    using KS = KsOf<Dims, KT, HT>;
    __shared__ __attribute__((aligned(16))) BwdSmem<KT, HT, WPB, KS> sm;
    using OP = OperandPolicy<DynDims>;
    constexpr int kGyRows = 4 * KS::out + 2;
    constexpr int kBwdBlock = WPB * 64;
    constexpr int NI = Dims::kItems > 0 ? Dims::kItems : 4 * KT + 4;
    constexpr int NGC = KsOf<Dims, KT, HT>::out;  // 4-feature chunks of the upstream gradient
    const Dims dm = make_dims<Dims>(a.mlp);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15, wv = tid >> 6;
    const int part = g;
    const int last = a.mlp.n_layers - 1;

    fill_frags(sm.w1, a.mlp.W[0], a.mlp.in_dim, dm.hid, dm.in, false, tid, kBwdBlock);
    fill_frags(sm.w1t, a.mlp.W[0], a.mlp.in_dim, dm.in, dm.hid, true, tid, kBwdBlock);
    if (dm.three()) {
        fill_frags(sm.w2, a.mlp.W[1], a.mlp.hidden, dm.hid, dm.hid, false, tid, kBwdBlock);
        fill_frags(sm.w2t, a.mlp.W[1], a.mlp.hidden, dm.hid, dm.hid, true, tid, kBwdBlock);
    }
    fill_frags(sm.w3, a.mlp.W[last], a.mlp.hidden, dm.out, dm.hid, false, tid, kBwdBlock);
    fill_frags(sm.w3t, a.mlp.W[last], a.mlp.hidden, dm.hid, dm.out, true, tid, kBwdBlock);
    fill_bias(sm.b1, a.mlp.b[0], dm.hid, tid, kBwdBlock);
    if (dm.three()) fill_bias(sm.b2, a.mlp.b[1], dm.hid, tid, kBwdBlock);
    fill_bias(sm.b3, a.mlp.b[last], dm.out, tid, kBwdBlock);
    stage_segs(sm.segs, a.seg, a.gseg, a.n_seg, tid);
    // constant-one rows: feature row `D` of a buffer whose features stop short of a tile
    // boundary makes column D of the matching weight-gradient tile the bias gradient
    const bool ones_i = Dims::kStatic || (a.mlp.in_dim % 16) != 0,
               ones_h = Dims::kStatic || (a.mlp.hidden % 16) != 0;
    constexpr int kSinN = (16 * KT + 2) * kTbLd, kTbN = (16 * HT + 2) * kTbLd;
    for (int i = tid; i < WPB * kSinN; i += kBwdBlock) {
        const int f = (i % kSinN) / kTbLd, col = i % kTbLd;
        (&sm.sin[0][0])[i] = (ones_i && f == a.mlp.in_dim && col < 16) ? 1.f : 0.f;
        (&sm.tb0[0][0])[i] = 0.f;
    }
    for (int i = tid; i < WPB * kGyRows * kTbLd; i += kBwdBlock) (&sm.sgy[0][0])[i] = 0.f;
    for (int i = tid; i < WPB * 2 * kTbN; i += kBwdBlock) {
        const int f = (i % kTbN) / kTbLd, col = i % kTbLd;
        (&sm.tb[0][0][0])[i] = (ones_h && f == a.mlp.hidden && col < 16) ? 1.f : 0.f;
    }
    __syncthreads();
    if (tid == 0) build_load_list(sm.ll, sm.segs, a.n_seg);
    __syncthreads();
    float *tb0 = sm.tb0[wv], *tb1 = sm.tb[wv][0], *tb2 = sm.tb[wv][1];
    float *sc = sm.sin[wv], *gc = sm.sgy[wv];

    const int nti = (dm.in_ks() + 3) >> 2;
    const int nth = (dm.hid_ks() + 3) >> 2;
    const bool need_y = a.epilogue == GNNTRK_EPI_RELU || a.epilogue == GNNTRK_EPI_SIGMOID;
    const bool want_dw = a.gW[0] != nullptr;
    const bool tr_on = !(a.debug_flags & 8);
    const bool ld_on = !(a.debug_flags & 2);

    typename OP::template Frags<HT * KT * 4> w1;
    typename OP::template Frags<HT * HT * 4> w2;
    typename OP::template Frags<HT * 4> w3;
    typename OP::template Frags<KT * HT * 4> w1t;
    typename OP::template Frags<HT * HT * 4> w2t;
    typename OP::template Frags<HT * 4> w3t;
    typename OP::template Bias<HT> b1, b2;
    w1.load(sm.w1, nth * dm.in_ks(), lane);
    w2.load(sm.w2, dm.three() ? nth * dm.hid_ks() : 0, lane);
    w3.load(sm.w3, need_y ? dm.hid_ks() : 0, lane);
    w1t.load(sm.w1t, nti * dm.hid_ks(), lane);
    w2t.load(sm.w2t, dm.three() ? nth * dm.hid_ks() : 0, lane);
    w3t.load(sm.w3t, nth * dm.out_ks(), lane);
    b1.load(sm.b1, nth, lane);
    b2.load(sm.b2, dm.three() ? nth : 0, lane);
    const f32x4 b3 = sm.b3[lane];

    Items<NI, Dims::kStatic> it;
    it.load(sm.ll);
    int boff[KT * 4], ooff[4], off_i[KT * 4], off_h[HT * 4], off_o[4];
    operand_offsets<KT>(dm.in, g, c, 16 * KT + 1, boff);
    operand_offsets<1>(dm.out, g, c, kGyRows - 1, ooff);
    tr_offsets<KT>(dm.in, g, c, 16 * KT, off_i);   // -> tb0
    tr_offsets<HT>(dm.hid, g, c, 16 * HT, off_h);  // -> tb1 / tb2
    tr_offsets<1>(dm.out, g, c, 16 * KT, off_o);   // -> tb0

    // upstream-gradient terms (uniform descriptors)
    const gcf_ptr go_ptr0 = (gcf_ptr)a.gout[0].ptr, go_ptr1 = (gcf_ptr)a.gout[1].ptr;
    const gci_ptr go_idx0 = (gci_ptr)a.gout[0].idx, go_idx1 = (gci_ptr)a.gout[1].idx;
    const int n_och = (a.mlp.out_dim + 3) >> 2;  // 4-feature chunks of the output

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
    const TileSched sch = make_sched(n_tiles, WPB);
    int64_t tile = sch.cur;
    const int64_t last_row = a.n_rows - 1;
    auto clamp_row = [&](int64_t t) {
        const int64_t r = t * kTileRows + c;
        return r < last_row ? r : last_row;
    };
    auto row_valid = [&](int64_t t) { return t < sch.end && t * kTileRows + c < a.n_rows; };
    // upstream gradient of (row, chunk ch, element part): sum of the gout terms
    auto gout_rows = [&](int64_t r, int32_t &r0, int32_t &r1) {
        r0 = r1 = (int32_t)r;
        if (go_idx0) r0 = go_idx0[r];
        if (a.n_gout > 1 && go_idx1) r1 = go_idx1[r];
    };
    auto gout_vals = [&](int32_t r0, int32_t r1, bool valid, float (&gv)[NGC]) {
#pragma unroll
        for (int ch = 0; ch < NGC; ++ch) {
            float v = 0.f;
            const int f = 4 * ch + part;
            if (ch < n_och && valid && f < a.mlp.out_dim) {
                v = go_ptr0[(int64_t)r0 * a.gout[0].stride + f];
                if (a.n_gout > 1) v += go_ptr1[(int64_t)r1 * a.gout[1].stride + f];
            }
            gv[ch] = v;
        }
    };
    auto stage_gout = [&](float *buf, const float (&gv)[NGC]) {
#pragma unroll
        for (int ch = 0; ch < NGC; ++ch)
            if (ch < n_och) buf[(4 * ch + part) * kTbLd + c] = gv[ch];
    };

    // software pipeline: values of tile n+1 and row ids of tile n+2 in flight
    int32_t rid_n[NI], gr0_n = 0, gr1_n = 0;
    float pv[NI], gv[NGC];
    if (tile < sch.end) {
        int32_t rid_c[NI], r0, r1;
        item_row_ids<NI>(it, clamp_row(tile), rid_c);
        gout_rows(clamp_row(tile), r0, r1);
        item_values<NI>(it, rid_c, part, row_valid(tile) && ld_on, pv);
        gout_vals(r0, r1, row_valid(tile) && ld_on, gv);
        stage_items<NI>(it, sc, part, c, pv);
        stage_gout(gc, gv);
        const int64_t t1 = tile + sch.step < sch.end ? tile + sch.step : tile;
        item_row_ids<NI>(it, clamp_row(t1), rid_n);
        gout_rows(clamp_row(t1), gr0_n, gr1_n);
    }
    // counted loop on a scalar trip count: a plain do-while for the compiler (loop-carried
    // MFMA accumulators are then updated in place instead of being copied every iteration)
    const int n_iter = (int)__builtin_amdgcn_readfirstlane((uint32_t)(tile < sch.end ? (sch.end - tile + sch.step - 1) / sch.step : 0));
    for (int iter = 0; iter < n_iter; ++iter, tile += sch.step) {
        const int64_t row = tile * kTileRows + c;
        const bool valid = row < a.n_rows;
        item_values<NI>(it, rid_n, part, row_valid(tile + sch.step) && ld_on, pv);
        gout_vals(gr0_n, gr1_n, row_valid(tile + sch.step) && ld_on, gv);
        {
            const int64_t t2 = tile + 2 * sch.step < sch.end ? tile + 2 * sch.step : tile;
            item_row_ids<NI>(it, clamp_row(t2), rid_n);
            gout_rows(clamp_row(t2), gr0_n, gr1_n);
        }
        lds_wave_order();  // staged inputs of this tile are visible
        f32x4 bin[KT], gyv[1];
        read_operand<KT>(sc, boff, nti, bin);
        read_operand<1>(gc, ooff, 1, gyv);
        f32x4 gy = gyv[0];

        // ---- S0: layer 1; a1 goes to LDS for the weight-gradient MFMAs ----------------
        f32x4 a1[HT], a2[HT];
        mlp_layer1<KT, HT>(dm, w1, b1, lane, bin, a1);
        if (want_dw && tr_on) tr_write<HT>(tb1, off_h, nth, dm.hid_ks(), a1);
        // ---- S1: layer 2 (covers the LDS write latency) ---------------------------
        mlp_layer2<HT>(dm, w2, b2, lane, a1, a2);
        f32x4 mT[KT], a1T[HT];
#pragma unroll
        for (int t = 0; t < KT; ++t) mT[t] = bin[t];
#pragma unroll
        for (int t = 0; t < HT; ++t) a1T[t] = a1[t];
        if (want_dw && tr_on) {
            lds_wave_order();
            tr_read<KT>(sc, nti, g, c, mT);  // the staging buffer IS m^T
            tr_read<HT>(tb1, nth, g, c, a1T);
        }
        if (need_y) {
            const f32x4 y = mlp_layer3<HT>(dm, w3, b3, lane, a2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (a.epilogue == GNNTRK_EPI_RELU) {
                    gy[r] = y[r] > 0.f ? gy[r] : 0.f;
                } else {
                    const float s = sigmoidf_(y[r]);
                    gy[r] *= a.cb * s * (1.f - s);
                }
            }
        } else if (a.epilogue == GNNTRK_EPI_RESIDUAL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gy[r] *= a.cb;
        }
        if (want_dw && tr_on) {
            tr_write<HT>(tb2, off_h, nth, dm.hid_ks(), a2);
            f32x4 gyw[1] = {gy};
            tr_write<1>(tb0, off_o, 1, dm.out_ks(), gyw);
        }
        // ---- S2: delta at the last hidden layer: (Wout^T gy) * relu' ---------------
        f32x4 dl[HT];
#pragma unroll
        for (int to = 0; to < HT; ++to) dl[to] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < dm.out_ks()) {
#pragma unroll
                for (int to = 0; to < HT; ++to)
                    if (to < nth) dl[to] = mfma4(w3t.get(to * dm.out_ks() + r, lane), gy[r], dl[to]);
            }
#pragma unroll
        for (int to = 0; to < HT; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) dl[to][r] = a2[to][r] > 0.f ? dl[to][r] : 0.f;
        if (want_dw) {
            if (!ones_h) {
#pragma unroll
                for (int to = 0; to < HT; ++to) {
                    if (dm.three())
                        db2[to] += dl[to];
                    else
                        db1[to] += dl[to];
                }
                db3 += gy;
            }
            f32x4 a2T[HT], gyT[1];
#pragma unroll
            for (int t = 0; t < HT; ++t) a2T[t] = a2[t];
            gyT[0] = gy;
            if (tr_on) {
                lds_wave_order();
                tr_read<HT>(tb2, nth, g, c, a2T);
                tr_read<1>(tb0, 1, g, c, gyT);
                tr_write<HT>(tb1, off_h, nth, dm.hid_ks(), dl);  // a1T was read a stage ago
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ti = 0; ti < HT; ++ti)
                    if (ti < nth) dW3[ti] = mfma4(gyT[0][s], a2T[ti][s], dW3[ti]);
        }
        // ---- S3: delta at the first hidden layer ---------------------------------
        f32x4 d1[HT];
        if (dm.three()) {
#pragma unroll
            for (int to = 0; to < HT; ++to) d1[to] = zero4;
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < dm.hid_ks()) {
#pragma unroll
                        for (int to = 0; to < HT; ++to)
                            if (to < nth)
                                d1[to] = mfma4(w2t.get(to * dm.hid_ks() + 4 * t + r, lane), dl[t][r],
                                               d1[to]);
                    }
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[to][r] = a1[to][r] > 0.f ? d1[to][r] : 0.f;
            if (want_dw) {
                if (!ones_i) {
#pragma unroll
                    for (int to = 0; to < HT; ++to) db1[to] += d1[to];
                }
                f32x4 dlT[HT];
#pragma unroll
                for (int t = 0; t < HT; ++t) dlT[t] = dl[t];
                if (tr_on) {
                    lds_wave_order();
                    tr_read<HT>(tb1, nth, g, c, dlT);
                    tr_write<HT>(tb2, off_h, nth, dm.hid_ks(), d1);  // a2T was read a stage ago
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        if (to < nth) {
#pragma unroll
                            for (int ti = 0; ti < HT; ++ti)
                                if (ti < nth) dW2[to][ti] = mfma4(dlT[to][s], a1T[ti][s], dW2[to][ti]);
                        }
            }
        } else {
#pragma unroll
            for (int to = 0; to < HT; ++to) d1[to] = dl[to];  // a2 == a1: mask already applied
            if (want_dw && !ones_i) {
                // two layers: db1 was accumulated from dl only when the hidden ones-row is
                // absent; with it present but no input ones-row, take it here
                if (ones_h) {
#pragma unroll
                    for (int to = 0; to < HT; ++to) db1[to] += d1[to];
                }
            }
        }
        // ---- S4: input gradient W1^T d1 (rows = concatenated input features) -------
        f32x4 gin[KT];
#pragma unroll
        for (int ti = 0; ti < KT; ++ti) gin[ti] = zero4;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * t + r < dm.hid_ks()) {
#pragma unroll
                    for (int ti = 0; ti < KT; ++ti)
                        if (ti < nti)
                            gin[ti] = mfma4(w1t.get(ti * dm.hid_ks() + 4 * t + r, lane), d1[t][r],
                                            gin[ti]);
                }
        if (want_dw) {
            f32x4 d1T[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) d1T[t] = d1[t];
            if (tr_on) {
                lds_wave_order();
                // three layers: d1 went to tb2 in S3; two layers: d1 == dl sits in tb1
                tr_read<HT>(dm.three() ? tb2 : tb1, nth, g, c, d1T);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int to = 0; to < HT; ++to)
                    if (to < nth) {
#pragma unroll
                        for (int ti = 0; ti < KT; ++ti)
                            if (ti < nti) dW1[to][ti] = mfma4(d1T[to][s], mT[ti][s], dW1[to][ti]);
                    }
        }
        // ---- output: input-gradient slices leave through tb0, row-wise per item ------
        lds_wave_order();  // tb0 (gy^T) reads are done
        tr_write<KT>(tb0, off_i, nti, dm.in_ks(), gin);
        lds_wave_order();
        if (!((a.debug_flags & 1) && gin[0][0] != 12345.678f)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (i < it.n) {
                    const gf_ptr gp = it.gptr(i);
                    if (gp != nullptr) {
                        const int rm = it.rem(i), fr = it.frow(i), gst = it.gstride(i);
                        const bool rl = it.relu(i);
                        if (valid && part < rm) {
                            const int o = (fr + part) * kTbLd + c;
                            float v = tb0[o];
                            if (rl && !(sc[o] > 0.f)) v = 0.f;
                            gp[row * gst + part] = v;
                        }
                    }
                }
            }
        }
        // stage the next tile (every read of this tile's buffers is behind the syncs above)
        stage_items<NI>(it, sc, part, c, pv);
        stage_gout(gc, gv);
    }

    if (want_dw && !(a.debug_flags & 16)) {
        const BwdPartLayout pl = part_layout(a.mlp);
        float *dst = part_out + (int64_t)(blockIdx.x * WPB + wv) * pl.total;
        float *b_in = ones_i ? dst + pl.b[0] : nullptr;                  // db1 from dW1's ones column
        float *b_mid = (ones_h && dm.three()) ? dst + pl.b[1] : nullptr;  // db2 from dW2's
        float *b_out = ones_h ? dst + pl.b[last] : nullptr;               // db3 from dW3's
#pragma unroll
        for (int to = 0; to < HT; ++to) {
#pragma unroll
            for (int ti = 0; ti < KT; ++ti)
                store_dw_tile(dst + pl.w[0], b_in, a.mlp.hidden, a.mlp.in_dim, to, ti, g, c, dW1[to][ti]);
            if (dm.three()) {
#pragma unroll
                for (int ti = 0; ti < HT; ++ti)
                    store_dw_tile(dst + pl.w[1], b_mid, a.mlp.hidden, a.mlp.hidden, to, ti, g, c,
                                  dW2[to][ti]);
            }
            store_dw_tile(dst + pl.w[last], b_out, a.mlp.out_dim, a.mlp.hidden, 0, to, g, c, dW3[to]);
        }
        // biases without a ones row: reduce the per-lane partial sums over the 16 row-lanes
        const bool db1_valu = !ones_i, db2_valu = dm.three() && !ones_h, db3_valu = !ones_h;
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
                const int f = feat_of(dm.hid, to, g, r);
                if (c == 0 && f >= 0) {
                    if (db1_valu) dst[pl.b[0] + f] = v1;
                    if (db2_valu) dst[pl.b[1] + f] = v2;
                }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = db3[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
            const int fo = feat_of(dm.out, 0, g, r);
            if (c == 0 && fo >= 0 && db3_valu) dst[pl.b[last] + fo] = v;
        }
    }
}

// fixed-order reduction of the per-wave partial blocks into the gradient tensors.
// One workgroup per 64 parameters: wave w sums the partial blocks w, w+4, w+8, ... (lane =
// parameter, coalesced 256 B rows), then the four wave sums are added in wave order.
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float *__restrict__ part,
                                                                 int n_part, int total,
                                                                 gnntrk_mlp mlp, float *gW0,
                                                                 float *gW1, float *gW2, float *gb0,
                                                                 float *gb1, float *gb2,
                                                                 int accumulate) {
    __shared__ float s_sum[kWaves][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (p < total) {
        // (latency bound: eight loads in flight, added in the same fixed order)
        int w = wv;
        for (; w + 7 * kWaves < n_part; w += 8 * kWaves) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(w + u * kWaves) * total + p];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; w < n_part; w += kWaves) acc += part[(int64_t)w * total + p];
    }
    s_sum[wv][lane] = acc;
    __syncthreads();
    if (wv != 0 || p >= total) return;
    const float s = ((s_sum[0][lane] + s_sum[1][lane]) + s_sum[2][lane]) + s_sum[3][lane];
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
// n_rows == 0 is a valid no-op: row pointers may then be NULL (what an empty tensor hands over)
static int check_mlp(const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg, int64_t n_rows) {
    if (m.n_layers != 2 && m.n_layers != 3) return fail(GNNTRK_EUNSUPPORTED, "mlp: n_layers must be 2 or 3");
    if (m.in_dim < 1 || m.in_dim > GNNTRK_MAX_IN) return fail(GNNTRK_EUNSUPPORTED, "mlp: in_dim out of range [1,48]");
    if (m.hidden < 1 || m.hidden > GNNTRK_MAX_HIDDEN) return fail(GNNTRK_EUNSUPPORTED, "mlp: hidden out of range [1,64]");
    if (m.out_dim < 1 || m.out_dim > GNNTRK_MAX_OUT) return fail(GNNTRK_EUNSUPPORTED, "mlp: out_dim out of range [1,16]");
    if (n_seg < 1 || n_seg > GNNTRK_MAX_SEGS) return fail(GNNTRK_EINVAL, "mlp: bad segment count");
    int tot = 0;
    for (int j = 0; j < n_seg; ++j) {
        if ((!seg[j].ptr && n_rows != 0) || seg[j].dim < 1 || seg[j].stride < seg[j].dim)
            return fail(GNNTRK_EINVAL, "mlp: bad segment descriptor");
        tot += seg[j].dim;
    }
    if (tot != m.in_dim) return fail(GNNTRK_EINVAL, "mlp: segment dims do not sum to in_dim");
    for (int i = 0; i < m.n_layers; ++i)
        if (!m.W[i]) return fail(GNNTRK_EINVAL, "mlp: NULL weight pointer");
    return GNNTRK_OK;
}

static int count_items(int n_seg, const gnntrk_seg *seg) {
    int n = 0;
    for (int j = 0; j < n_seg; ++j) n += (seg[j].dim + 3) / 4;
    return n;
}

static int grid_for(int64_t n_rows, int blocks_per_cu, int waves = kWaves) {
    const int64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    int64_t g = (tiles + waves - 1) / waves;
    const int64_t cap = (int64_t)cu_count() * blocks_per_cu;
    if (g > cap) g = cap;
    if (g >= 8) g -= g % 8;  // XCD-aware schedule wants a multiple of 8
    if (g < 1) g = 1;
    return (int)g;
}

constexpr int kBwdWavesPerCu = 8;  // static: 1 block of 8 waves; generic: 2 blocks of 4
constexpr int kFwdBlocksPerCu = 4;

// Static instantiations: every loop bound known at compile time (straight-line MFMA
// code) for the shapes of the reference's default configuration (hidden width 37..40,
// models/edge_classifier.py + tests/test_configs): key = (k-steps in, k-steps hidden,
// k-steps out, 3 layers).  Everything else takes the generic run-time-bound kernels.
// last column: load-list capacity (4-feature chunks over all segments)
#define GNNTRK_STATIC_SHAPES(X) \
    X(4, 10, 2, false, 4) /* node encoder   14 -> 40 -> 5       x[14]                   */ \
    X(1, 10, 1, false, 1) /* edge encoder    4 -> 40 -> 4       edge_attr[4]            */ \
    X(4, 10, 1, true, 5)  /* relational     14 -> 40 -> 40 -> 4  h[5], h[5], e[4]        */ \
    X(3, 10, 2, true, 3)  /* object          9 -> 40 -> 40 -> 5  h[5], aggr[4]           */ \
    X(7, 10, 1, true, 8)  /* W head         26 -> 40 -> 40 -> 1  h[5], h[5], 4 x e[4]    */ \
    X(7, 10, 1, false, 7) /* edge encoder   28 -> 40 -> 4       MLGraphConstruction's 2 x 14 edge features */ \
    X(2, 10, 1, true, 2)  /* beta / cluster heads  5 -> 40 -> 40 -> 1 (or 2..4)   h[5]  */ \
    X(2, 10, 2, true, 2)  /* cluster head          5 -> 40 -> 40 -> 5..8          h[5]  */

static bool static_shape(int ksi, int ksh, int kso, bool three, int n_items) {
    bool hit = false;
#define GNNTRK_MATCH(KSI, KSH, KSO, THREE, NIT) \
    hit = hit || (ksi == KSI && ksh == KSH && kso == KSO && three == THREE && n_items <= NIT);
    GNNTRK_STATIC_SHAPES(GNNTRK_MATCH)
#undef GNNTRK_MATCH
    return hit;
}

#define GNNTRK_DISPATCH(CALL_STATIC, CALL_DYN)                                  \
    {                                                                           \
        bool done_ = false;                                                     \
        GNNTRK_STATIC_SHAPES(CALL_STATIC)                                       \
        if (!done_) {                                                           \
            if (kt <= 1 && ht <= 1) { CALL_DYN(1, 1); }                         \
            else if (kt <= 1 && ht <= 3) { CALL_DYN(1, 3); }                    \
            else if (kt <= 2 && ht <= 2) { CALL_DYN(2, 2); }                    \
            else if (kt <= 2 && ht <= 3) { CALL_DYN(2, 3); }                    \
            else { CALL_DYN(3, 4); }                                            \
        }                                                                       \
    }

// Name of the instantiation the launchers below pick (as rocprofv3 prints it), so that
// host-side timers and profiles can be matched kernel by kernel.
int mlp_kernel_name(const gnntrk_mlp *m, int n_seg, const gnntrk_seg *seg, int backward,
                    char *buf, size_t len) {
    if (!m || !seg || !buf || len == 0) return fail(GNNTRK_EINVAL, "mlp_kernel_name: bad argument");
    const int n_items = count_items(n_seg, seg);
    int kt = (m->in_dim + 15) / 16;
    const int ht = (m->hidden + 15) / 16;
    const bool items_fit = n_items <= 4 * kt + 4;
    while (4 * kt + 4 < n_items) ++kt;
    const bool ones_ok = (m->in_dim % 16) != 0 && (m->hidden % 16) != 0;
    const int ksi = (items_fit && (!backward || ones_ok)) ? make_dimmap(m->in_dim).ks : -1;
    const int ksh = make_dimmap(m->hidden).ks, kso = make_dimmap(m->out_dim).ks;
    const bool three = m->n_layers == 3;
    const char *dir = backward ? "bwd" : "fwd";
#define NAME_S(KSI, KSH, KSO, THREE, NIT)                                                       \
    if (!done_ && ksi == KSI && ksh == KSH && kso == KSO && three == THREE && n_items <= NIT) { \
        snprintf(buf, len, "mlp_%s_kernel<%d, %d, StaticDims<%d, %d, %d, %s, %d>%s>", dir,      \
                 (KSI + 3) / 4, (KSH + 3) / 4, KSI, KSH, KSO, THREE ? "true" : "false", NIT,    \
                 backward ? ", 8" : " ");                                                       \
        done_ = true;                                                                           \
    }
#define NAME_D(K, H) \
    { snprintf(buf, len, "mlp_%s_kernel<%d, %d, DynDims%s>", dir, K, H, backward ? ", 4" : ""); }
    GNNTRK_DISPATCH(NAME_S, NAME_D)
#undef NAME_S
#undef NAME_D
    return GNNTRK_OK;
}

int mlp_forward_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_forward: NULL args");
    int rc = check_mlp(a->mlp, a->n_seg, a->seg, a->n_rows);
    if (rc) return rc;
    if (a->n_rows == 0) return GNNTRK_OK;  // nothing to write: NULL row pointers are fine
    if (!a->out || a->out_stride < a->mlp.out_dim) return fail(GNNTRK_EINVAL, "mlp_forward: bad output");
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_forward: bad epilogue");
    if (a->epilogue == GNNTRK_EPI_RESIDUAL && (!a->res || a->res_stride < a->mlp.out_dim))
        return fail(GNNTRK_EINVAL, "mlp_forward: residual epilogue needs res");
    if (a->n_rows < 0) return fail(GNNTRK_EINVAL, "mlp_forward: negative n_rows");
    if (a->n_rows == 0) return GNNTRK_OK;
    const int n_items = count_items(a->n_seg, a->seg);
    if (n_items > kMaxItems) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward: too many input segments/chunks (max 16 4-feature chunks)");
    int kt = (a->mlp.in_dim + 15) / 16;
    const int ht = (a->mlp.hidden + 15) / 16;
    while (4 * kt + 4 < n_items) ++kt;  // the load list of an instantiation holds 4*KT+4 items
    int grid = grid_for(a->n_rows, kFwdBlocksPerCu);
    // (the persistent grid of an instantiation = its resident workgroups, asked of the runtime once per instantiation:
    //  the static shapes hold 124-156 registers = three, not four, workgroups per CU - see mlp_bf16_fwd.hip)
#define GNNTRK_FWD32_GRID(kfn_)  /* (occupancy cached per DEVICE ordinal, atomically: launch threads race, devices differ) */ \
    {                                                                                   \
        static std::atomic<int> occ_dev_[16];                                           \
        int dev_ = 0;                                                                   \
        (void)hipGetDevice(&dev_);                                                      \
        std::atomic<int>& slot_ = occ_dev_[dev_ & 15];                                  \
        int occ_ = slot_.load(std::memory_order_relaxed);                               \
        if (occ_ <= 0) {                                                                \
            int o_ = 0;                                                                 \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_, kfn_, kBlock, 0) != hipSuccess || o_ < 1) \
                o_ = kFwdBlocksPerCu;                                                 \
            occ_ = o_ > 8 ? 8 : o_;                                                     \
            slot_.store(occ_, std::memory_order_relaxed);                               \
        }                                                                               \
        grid = grid_for(a->n_rows, occ_);                                                       \
    }
    const int ksh = make_dimmap(a->mlp.hidden).ks, kso = make_dimmap(a->mlp.out_dim).ks;
    const bool three = a->mlp.n_layers == 3;
    // static instantiations need their own load-list capacity
    const int ksi = (n_items <= 4 * ((a->mlp.in_dim + 15) / 16) + 4) ? make_dimmap(a->mlp.in_dim).ks : -1;
#define CALL_FWD_S(KSI, KSH, KSO, THREE, NIT)                                                  \
    if (!done_ && ksi == KSI && ksh == KSH && kso == KSO && three == THREE && n_items <= NIT) { \
        auto kfn = mlp_fwd_kernel<(KSI + 3) / 4, (KSH + 3) / 4, StaticDims<KSI, KSH, KSO, THREE, NIT>>; \
        GNNTRK_FWD32_GRID(kfn)                                                                  \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);                       \
        done_ = true;                                                                           \
    }
#define CALL_FWD_D(K, H)                                                                \
    {                                                                                   \
        auto kfn = mlp_fwd_kernel<K, H, DynDims>;                                       \
        GNNTRK_FWD32_GRID(kfn)                                                          \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);               \
    }
    GNNTRK_DISPATCH(CALL_FWD_S, CALL_FWD_D)
#undef CALL_FWD_S
#undef CALL_FWD_D
#undef GNNTRK_FWD32_GRID
    return check_launch("mlp_forward");
}

size_t mlp_backward_ws_bytes(const gnntrk_mlp *m) {
    if (!m) return 0;
    const BwdPartLayout pl = part_layout(*m);
    return (size_t)cu_count() * kBwdWavesPerCu * (size_t)pl.total * sizeof(float);
}

int mlp_backward_launch(const gnntrk_mlp_bwd_args *a, void *ws, size_t ws_bytes,
                        hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_backward: NULL args");
    int rc = check_mlp(a->mlp, a->n_seg, a->seg, a->n_rows);
    if (rc) return rc;
    const bool empty = a->n_rows == 0;  // no rows: only the parameter gradients are written (zeros)
    if (a->n_gout < 1 || a->n_gout > 2 || (!empty && (!a->gout[0].ptr || (a->n_gout == 2 && !a->gout[1].ptr))))
        return fail(GNNTRK_EINVAL, "mlp_backward: bad upstream gradient terms");
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_backward: bad epilogue");
    if (a->n_rows < 0) return fail(GNNTRK_EINVAL, "mlp_backward: negative n_rows");
    for (int j = 0; j < a->n_seg; ++j)
        if (a->gseg[j].ptr && (a->gseg[j].idx || a->gseg[j].accumulate))
            return fail(GNNTRK_EUNSUPPORTED,
                        "mlp_backward: gseg.idx / gseg.accumulate are reserved (row-aligned '=' only)");
    const bool want_dw = a->gW[0] != nullptr;
    if (want_dw) {
        for (int i = 0; i < a->mlp.n_layers; ++i)
            if (!a->gW[i]) return fail(GNNTRK_EINVAL, "mlp_backward: gW must be all set or all NULL");
        if (!ws || ws_bytes < mlp_backward_ws_bytes(&a->mlp))
            return fail(GNNTRK_EINVAL, "mlp_backward: workspace too small");
    }
    const BwdPartLayout pl = part_layout(a->mlp);
    int grid = 0, wpb = 4;
    const int n_items = count_items(a->n_seg, a->seg);
    if (n_items > kMaxItems) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward: too many input segments/chunks (max 16 4-feature chunks)");
    const bool items_fit = n_items <= 4 * ((a->mlp.in_dim + 15) / 16) + 4;
    // the static kernels also rely on the constant-one rows for the bias gradients
    const bool ones_ok = (a->mlp.in_dim % 16) != 0 && (a->mlp.hidden % 16) != 0;
    const int ksi0 = (items_fit && ones_ok) ? make_dimmap(a->mlp.in_dim).ks : -1,
              ksh0 = make_dimmap(a->mlp.hidden).ks, kso0 = make_dimmap(a->mlp.out_dim).ks;
    if (a->n_rows > 0) {
        int kt = (a->mlp.in_dim + 15) / 16;
        const int ht = (a->mlp.hidden + 15) / 16;
        while (4 * kt + 4 < n_items) ++kt;
        const bool is_static = static_shape(ksi0, ksh0, kso0, a->mlp.n_layers == 3, n_items);
        wpb = is_static ? 8 : 4;
        grid = grid_for(a->n_rows, is_static ? 1 : 2, wpb);
        float *part = reinterpret_cast<float *>(ws);
        const int ksi = ksi0, ksh = ksh0, kso = kso0;
        const bool three = a->mlp.n_layers == 3;
#define CALL_BWD_S(KSI, KSH, KSO, THREE, NIT)                                                  \
    if (!done_ && ksi == KSI && ksh == KSH && kso == KSO && three == THREE && n_items <= NIT) { \
        auto kfn = mlp_bwd_kernel<(KSI + 3) / 4, (KSH + 3) / 4, StaticDims<KSI, KSH, KSO, THREE, NIT>, 8>; \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), 0, stream, *a, part);                    \
        done_ = true;                                                                           \
    }
#define CALL_BWD_D(K, H)                                                                \
    {                                                                                   \
        auto kfn = mlp_bwd_kernel<K, H, DynDims, 4>;                                    \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 0, stream, *a, part);            \
    }
        GNNTRK_DISPATCH(CALL_BWD_S, CALL_BWD_D)
#undef CALL_BWD_S
#undef CALL_BWD_D
        rc = check_launch("mlp_backward");
        if (rc) return rc;
    }
    if (want_dw && !(a->debug_flags & 32))
        rc = reduce_partials_launch(reinterpret_cast<const float *>(ws), grid * wpb, &a->mlp, a->gW, a->gb,
                                    a->accumulate_params, stream);
    return rc;
}

// Fixed-order sum of n_part partial blocks (parameter layout of part_layout()) into the
// gradient tensors; n_part = 0 writes zeros.  Shared with the bf16 kernels (mlp_bf16.hip).
int reduce_partials_launch(const float *part, int n_part, const gnntrk_mlp *mlp, float *const gW[3],
                           float *const gb[3], int accumulate, hipStream_t stream) {
    const BwdPartLayout pl = part_layout(*mlp);
    const int rgrid = (pl.total + 63) / 64;
    auto rfn = reduce_partials_kernel;
    hipLaunchKernelGGL(rfn, dim3(rgrid), dim3(kBlock), 0, stream, part, n_part, pl.total, *mlp, gW[0], gW[1],
                       gW[2], gb[0], gb[1], gb[2], accumulate);
    return check_launch("mlp_backward(reduce)");
}

}  // namespace gnntrk
