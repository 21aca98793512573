// Fused gather-concat-MLP kernels, bf16 storage / bf16 MFMA inputs / fp32 accumulate
// (BASELINE configs 3 and 4).  Same operator as mlp.hip - reference models/mlp.py:18-62
// applied to the concatenation the reference materialises with index_select + cat
// (interaction_network.py:75-103, edge_classifier.py:103-116) - with every activation
// tensor stored as bf16 and the contractions on v_mfma_f32_16x16x{32,16}_bf16.
// Parameters, parameter gradients and the EPI_SIGMOID output stay fp32.
//
// Rounding points (what oracle/ref_cpu.py:mlp_bf16 restates): inputs are bf16 as stored;
// weights and biases are rounded to bf16 (RNE) when the fragments are packed; every layer
// accumulates in fp32; hidden activations are rounded to bf16 after the ReLU (the same
// value as rounding before it); the output is rounded to bf16 after the epilogue.
//
// No LDS in the forward tile loop: a lane loads its own B-operand chunks straight from
// HBM/L2 (8 bytes per chunk), the layer chain runs register to register (tile_bf16.h), the
// weights are read-only LDS fragments.  Latency is covered by a one-tile software prefetch
// plus the 4-6 waves per SIMD the small register footprint allows.
#include <hip/hip_runtime.h>

#include "host_util.h"
#include "tile_bf16.h"

namespace gnntrk {
namespace {

constexpr uint32_t kBf16One = 0x3f80u;
constexpr int kFwdDepth = 4;  // tiles per wave and pipeline stage (forward)

__device__ __forceinline__ void stage_seg_args(gnntrk_seg *dst, const gnntrk_seg (&seg)[GNNTRK_MAX_SEGS],
                                               int tid) {
#pragma unroll
    for (int j = 0; j < GNNTRK_MAX_SEGS; ++j)
        if (tid == j) dst[j] = seg[j];
}

// per-lane description of the chunks this lane loads: lane (g, c) owns chunks 8kk + 2g + h
template <int KI>
struct LaneChunks {
    gch_ptr base[KI][2];
    gci_ptr idx[KI][2];
    int32_t stride[KI][2];
    uint32_t keep[KI][2][2], ones[KI][2][2], rmin[KI][2];
    bool on[KI][2];

    __device__ __forceinline__ void init(const SlotPlan &P, const gnntrk_seg *seg, int g) {
#pragma unroll
        for (int kk = 0; kk < KI; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = 8 * kk + 2 * g + h;
                const int j = (p < P.n_chunks) ? P.seg[p] : -1;
                on[kk][h] = j >= 0;
                int d = 0;
                base[kk][h] = nullptr;
                idx[kk][h] = nullptr;
                stride[kk][h] = 0;
                rmin[kk][h] = 0x80008000u;
                if (j >= 0) {
                    d = seg[j].dim - 4 * P.first[p];
                    d = d > 4 ? 4 : d;
                    base[kk][h] = (gch_ptr)(reinterpret_cast<const uint16_t *>(seg[j].ptr) + 4 * P.first[p]);
                    idx[kk][h] = (gci_ptr)seg[j].idx;
                    stride[kk][h] = seg[j].stride;
                    if (seg[j].relu) rmin[kk][h] = 0u;
                }
                keep[kk][h][0] = (d >= 1 ? 0x0000ffffu : 0u) | (d >= 2 ? 0xffff0000u : 0u);
                keep[kk][h][1] = (d >= 3 ? 0x0000ffffu : 0u) | (d >= 4 ? 0xffff0000u : 0u);
                ones[kk][h][0] = ones[kk][h][1] = 0u;
                if (P.ones_slot >= 0 && (P.ones_slot >> 2) == p) {
                    const int r = P.ones_slot & 3;
                    ones[kk][h][0] = r < 2 ? kBf16One << (16 * (r & 1)) : 0u;
                    ones[kk][h][1] = r >= 2 ? kBf16One << (16 * (r & 1)) : 0u;
                }
            }
    }
};

template <int KI>
struct RowIds {
    int32_t v[KI][2];
};
template <int KI>
struct RawTile {
    u32x2 v[KI][2];
};

template <int KI>
__device__ __forceinline__ void load_row_ids(const LaneChunks<KI> &L, int32_t row, RowIds<KI> &r) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) r.v[kk][h] = L.idx[kk][h] ? L.idx[kk][h][row] : row;
}
template <int KI>
__device__ __forceinline__ void load_raw(const LaneChunks<KI> &L, const RowIds<KI> &r, RawTile<KI> &t) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x2 v = {0u, 0u};
            if (L.on[kk][h])
                v = *reinterpret_cast<const u32x2 GNNTRK_GLOBAL *>(
                    L.base[kk][h] + (int64_t)r.v[kk][h] * L.stride[kk][h]);
            t.v[kk][h] = v;
        }
}
// pads -> 0, ones slot -> 1.0, optional ReLU: the B operand of layer 1
template <int KI>
__device__ __forceinline__ void finish_inputs(const LaneChunks<KI> &L, const RawTile<KI> &t,
                                              u32x4 (&B)[KI]) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) {
        u32x2 h2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < 2; ++w)
                h2[h][w] = i16x2_max((t.v[kk][h][w] & L.keep[kk][h][w]) | L.ones[kk][h][w], L.rmin[kk][h]);
        B[kk] = join(h2[0], h2[1]);
    }
}

// ---- weight fragment images -----------------------------------------------------------
// forward:  A1 [HT][KI] K32 | A2 [HT] x hid_k | A3 [1] x hid_k          (dwords)
template <int KI, int HT>
struct FwdImg {
    static constexpr int kA1 = 0;
    static constexpr int kA2 = kA1 + HT * KI * 256;
    static constexpr int kA3 = kA2 + HT * hid_k_dwords(HT);
    static constexpr int kTotal = kA3 + hid_k_dwords(HT);
};

// image of `rows x hidden-k` fragments: element(o, f) for fragment row o (absolute) and
// hidden feature f
template <int HT, class F>
__device__ __forceinline__ void pack_hidden_k(uint32_t *dst, int row0, F element, int tid, int nthreads) {
#pragma unroll
    for (int u = 0; u < HT / 2; ++u)
        pack_frag_k32(dst + 256 * u,
                      [&](int i, int g, int e) { return element(row0 + i, hid_feat_k32(u, g, e)); }, tid,
                      nthreads);
    if (HT % 2)  // odd last tile: K = 32 fragment with a zero upper half (see contract_hidden)
        pack_frag_k32(dst + 256 * (HT / 2),
                      [&](int i, int g, int e) {
                          return e < 4 ? element(row0 + i, hid_feat_k16(HT - 1, g, e)) : 0.f;
                      },
                      tid, nthreads);
}

template <int KI, int HT, bool THREE>
__device__ __forceinline__ void pack_forward_weights(uint32_t *img, const AugWeights &w, const SlotPlan &P,
                                                     const gnntrk_seg *seg, int tid, int nthreads) {
    using I = FwdImg<KI, HT>;
    for (int t = 0; t < HT; ++t)
        for (int kk = 0; kk < KI; ++kk)
            pack_frag_k32(img + I::kA1 + (t * KI + kk) * 256,
                          [&](int i, int g, int e) {
                              return w.w1(16 * t + i, slot_col(P, seg, 32 * kk + 8 * g + e));
                          },
                          tid, nthreads);
    if (THREE)
        for (int t = 0; t < HT; ++t)
            pack_hidden_k<HT>(img + I::kA2 + t * hid_k_dwords(HT), 16 * t,
                              [&](int o, int f) { return w.wmid(o, f); }, tid, nthreads);
    pack_hidden_k<HT>(img + I::kA3, 0, [&](int o, int f) { return w.wlast(o, f); }, tid, nthreads);
}

// layers 1..(last-1): inputs B -> packed hidden activations feeding the last layer
template <int KI, int HT, bool THREE>
__device__ __forceinline__ void hidden_chain(const uint32_t *img, const u32x4 (&B)[KI], int lane,
                                             u32x2 (&P1)[HT], u32x2 (&P2)[HT]) {
    using I = FwdImg<KI, HT>;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        f32x4 acc = zero;
#pragma unroll
        for (int kk = 0; kk < KI; ++kk)
            acc = mfma_bf16_k32(frag_k32(img + I::kA1 + (t * KI + kk) * 256, lane), B[kk], acc);
        P1[t] = pack_tile_relu(acc);
    }
    if (THREE) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
            P2[t] = pack_tile_relu(contract_hidden<HT>(img + I::kA2 + t * hid_k_dwords(HT), P1, lane, zero));
    }
}

template <int KI, int HT, bool THREE>
__global__ __launch_bounds__(kBlock) void mlp16_fwd_kernel(const gnntrk_mlp_fwd_args a) {
    using I = FwdImg<KI, HT>;
    __shared__ __attribute__((aligned(16))) uint32_t s_img[I::kTotal];
    __shared__ SlotPlan s_plan;
    __shared__ gnntrk_seg s_seg[GNNTRK_MAX_SEGS];  // kernel arguments cannot be indexed dynamically
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    stage_seg_args(s_seg, a.seg, tid);
    __syncthreads();
    if (tid == 0) make_slot_plan(s_plan, a.mlp, a.n_seg, s_seg, nullptr);
    __syncthreads();
    {
        const AugWeights w = make_aug(a.mlp, s_plan);
        pack_forward_weights<KI, HT, THREE>(s_img, w, s_plan, s_seg, tid, kBlock);
    }
    LaneChunks<KI> L;
    L.init(s_plan, s_seg, g);
    __syncthreads();

    const int out_dim = a.mlp.out_dim;
    const bool out_lane = 4 * g < out_dim;  // this lane holds real output features
    const gci_ptr out_idx = (gci_ptr)a.out_idx;
    const int epi = a.epilogue;
    const gch_ptr resp = (gch_ptr) reinterpret_cast<const uint16_t *>(a.res);
    uint32_t okeep[2];
    {
        const int d = out_dim - 4 * g;
        okeep[0] = (d >= 1 ? 0x0000ffffu : 0u) | (d >= 2 ? 0xffff0000u : 0u);
        okeep[1] = (d >= 3 ? 0x0000ffffu : 0u) | (d >= 4 ? 0xffff0000u : 0u);
    }

    const int64_t n_tiles = (a.n_rows + kTileRows - 1) / kTileRows;
    const TileSched sch = make_sched(n_tiles);
    if (sch.cur >= sch.end) return;
    const int32_t last_row = (int32_t)(a.n_rows - 1);
    auto clamp_row = [&](int64_t t) {
        const int64_t r = t * kTileRows + c;
        return (int32_t)(r < last_row ? r : last_row);
    };

    // Software pipeline over groups of kDepth tiles: the raw chunks of group n+1 and the row
    // ids of group n+2 are in flight while group n computes - 16 * kDepth rows of loads per
    // wave cover the gather latency (one tile per wave in flight left the kernel latency
    // bound at 1/3 of the rate).  Tiles past the end of the schedule load clamped rows and
    // are not computed.
    constexpr int D = kFwdDepth;
    RowIds<KI> rid[D];
    RawTile<KI> cur[D], nxt[D];
    int32_t orow_c[D], orow_n[D], orow_nn[D];
    auto tile_of = [&](int64_t grp, int d) { return sch.cur + (grp * D + d) * sch.step; };
    auto ids_of = [&](int64_t grp, int d, RowIds<KI> &r, int32_t &orow) {
        const int32_t row = clamp_row(tile_of(grp, d));
        load_row_ids<KI>(L, row, r);
        orow = row;
        if (out_lane && out_idx) orow = out_idx[row];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ids_of(0, d, rid[d], orow_c[d]);
#pragma unroll
    for (int d = 0; d < D; ++d) load_raw<KI>(L, rid[d], cur[d]);
#pragma unroll
    for (int d = 0; d < D; ++d) ids_of(1, d, rid[d], orow_n[d]);

    for (int64_t grp = 0; tile_of(grp, 0) < sch.end; ++grp) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_raw<KI>(L, rid[d], nxt[d]);
#pragma unroll
        for (int d = 0; d < D; ++d) ids_of(grp + 2, d, rid[d], orow_nn[d]);

#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t tile = tile_of(grp, d);
            if (tile >= sch.end) break;
            u32x4 B[KI];
            finish_inputs<KI>(L, cur[d], B);
            u32x2 P1[HT], P2[HT];
            hidden_chain<KI, HT, THREE>(s_img, B, lane, P1, P2);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 y = contract_hidden<HT>(s_img + I::kA3, THREE ? P2 : P1, lane, zero);

            const int64_t row = tile * kTileRows + c;
            if (out_lane && row < a.n_rows && !(a.debug_flags & 1)) {
                if (epi == GNNTRK_EPI_SIGMOID) {
                    float *outp = a.out + (int64_t)orow_c[d] * a.out_stride + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < out_dim) outp[r] = a.ca + a.cb * sigmoidf_(y[r]);
                } else {
                    if (epi == GNNTRK_EPI_RESIDUAL) {
                        const u32x2 rv =
                            *reinterpret_cast<const u32x2 GNNTRK_GLOBAL *>(resp + row * a.res_stride + 4 * g);
                        y[0] = a.ca * bf16_lo(rv[0]) + a.cb * y[0];
                        y[1] = a.ca * bf16_hi(rv[0]) + a.cb * y[1];
                        y[2] = a.ca * bf16_lo(rv[1]) + a.cb * y[2];
                        y[3] = a.ca * bf16_hi(rv[1]) + a.cb * y[3];
                    }
                    u32x2 o = (epi == GNNTRK_EPI_RELU) ? pack_tile_relu(y) : pack_tile(y);
                    o[0] &= okeep[0];
                    o[1] &= okeep[1];
                    uint16_t *outp =
                        reinterpret_cast<uint16_t *>(a.out) + (int64_t)orow_c[d] * a.out_stride + 4 * g;
                    *reinterpret_cast<u32x2 GNNTRK_GLOBAL *>((gh_ptr)outp) = o;
                }
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            cur[d] = nxt[d];
            orow_c[d] = orow_n[d];
            orow_n[d] = orow_nn[d];
        }
    }
}

// ------------------------------------------------------------------ launchers
int check_bf16_mlp(const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg, const char *who) {
    if (m.n_layers != 2 && m.n_layers != 3) return fail(GNNTRK_EUNSUPPORTED, "mlp(bf16): n_layers must be 2 or 3");
    if (m.in_dim < 1 || m.hidden < 1 || m.hidden > 63 || m.out_dim < 1 || m.out_dim > 16)
        return fail(GNNTRK_EUNSUPPORTED, "mlp(bf16): hidden must be in [1,63], out in [1,16]");
    if (n_seg < 1 || n_seg > GNNTRK_MAX_SEGS) return fail(GNNTRK_EINVAL, "mlp(bf16): bad segment count");
    int tot = 0;
    for (int j = 0; j < n_seg; ++j) {
        const int padded = (seg[j].dim + 3) / 4 * 4;
        if (!seg[j].ptr || seg[j].dim < 1 || seg[j].stride < padded || seg[j].stride % 4 != 0 ||
            ((uintptr_t)seg[j].ptr & 7) != 0)
            return fail(GNNTRK_EINVAL,
                        "mlp(bf16): segment rows must be 8-byte aligned bf16 with stride a multiple of 4 "
                        "elements >= dim rounded up to 4");
        tot += seg[j].dim;
    }
    if (tot != m.in_dim) return fail(GNNTRK_EINVAL, "mlp(bf16): segment dims do not sum to in_dim");
    for (int i = 0; i < m.n_layers; ++i)
        if (!m.W[i]) return fail(GNNTRK_EINVAL, "mlp(bf16): NULL weight pointer");
    (void)who;
    return GNNTRK_OK;
}

int grid16(int64_t n_rows, int blocks_per_cu, int waves) {
    const int64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    int64_t g = (tiles + waves - 1) / waves;
    const int64_t cap = (int64_t)cu_count() * blocks_per_cu;
    if (g > cap) g = cap;
    if (g >= 8) g -= g % 8;
    if (g < 1) g = 1;
    return (int)g;
}

constexpr int kFwd16BlocksPerCu = 5;

}  // namespace

#define GNNTRK_FWD16_CASE(KI_, HT_)                                                         \
    if (P.KI == KI_ && P.HT == HT_) {                                                       \
        if (three) {                                                                        \
            auto kfn = mlp16_fwd_kernel<KI_, HT_, true>;                                    \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);               \
        } else {                                                                            \
            auto kfn = mlp16_fwd_kernel<KI_, HT_, false>;                                   \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a);               \
        }                                                                                   \
        launched = true;                                                                    \
    }

int mlp16_kernel_name(const gnntrk_mlp *m, int n_seg, const gnntrk_seg *seg, int backward, char *buf,
                      size_t len) {
    if (!m || !seg || !buf || len == 0) return fail(GNNTRK_EINVAL, "mlp_kernel_name: bad argument");
    SlotPlan P;
    make_slot_plan(P, *m, n_seg, seg, nullptr);
    snprintf(buf, len, "mlp16_%s_kernel<%d, %d, %s>", backward ? "bwd" : "fwd", P.KI, P.HT,
             m->n_layers == 3 ? "true" : "false");
    return GNNTRK_OK;
}

int mlp_forward_bf16_launch(const gnntrk_mlp_fwd_args *a, hipStream_t stream) {
    if (!a) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: NULL args");
    int rc = check_bf16_mlp(a->mlp, a->n_seg, a->seg, "mlp_forward_bf16");
    if (rc) return rc;
    if (a->epilogue < 0 || a->epilogue > 3) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad epilogue");
    const int out_pad = (a->mlp.out_dim + 3) / 4 * 4;
    if (a->epilogue == GNNTRK_EPI_SIGMOID) {
        if (!a->out || a->out_stride < a->mlp.out_dim)
            return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad (fp32) output");
    } else if (!a->out || a->out_stride < out_pad || a->out_stride % 4 != 0 || ((uintptr_t)a->out & 7) != 0) {
        return fail(GNNTRK_EINVAL,
                    "mlp_forward_bf16: output rows must be 8-byte aligned bf16, stride a multiple of 4 >= "
                    "out_dim rounded up to 4");
    }
    if (a->epilogue == GNNTRK_EPI_RESIDUAL &&
        (!a->res || a->res_stride < out_pad || a->res_stride % 4 != 0 || ((uintptr_t)a->res & 7) != 0))
        return fail(GNNTRK_EINVAL, "mlp_forward_bf16: residual epilogue needs padded bf16 res rows");
    if (a->n_rows < 0 || a->n_rows > 0x7fffffff) return fail(GNNTRK_EINVAL, "mlp_forward_bf16: bad n_rows");
    if (a->n_rows == 0) return GNNTRK_OK;
    SlotPlan P;
    make_slot_plan(P, a->mlp, a->n_seg, a->seg, nullptr);
    if (!P.ok || P.KI > 2)
        return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: more than 16 input chunks / 4 hidden tiles");
    const bool three = a->mlp.n_layers == 3;
    const int grid = grid16(a->n_rows, kFwd16BlocksPerCu, kWaves);
    bool launched = false;
    GNNTRK_FWD16_CASE(1, 1)
    GNNTRK_FWD16_CASE(1, 2)
    GNNTRK_FWD16_CASE(1, 3)
    GNNTRK_FWD16_CASE(1, 4)
    GNNTRK_FWD16_CASE(2, 1)
    GNNTRK_FWD16_CASE(2, 2)
    GNNTRK_FWD16_CASE(2, 3)
    GNNTRK_FWD16_CASE(2, 4)
    if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_forward_bf16: no instantiation");
    return check_launch("mlp_forward_bf16");
}

}  // namespace gnntrk
