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
#pragma once
#include <cstdio>
#include <cstring>
#include <type_traits>

#include <hip/hip_runtime.h>

#include "host_util.h"
#include "tile_bf16.h"

namespace gnntrk {
namespace {

constexpr uint32_t kBf16One = 0x3f80u;
constexpr int kFwdDepth = 4;  // tiles per wave and pipeline stage (forward)

// floats of one partial block = all parameters in order W1, b1, [W2, b2,] W3, b3 (the layout
// of part_layout() in mlp.hip: bias slots exist whether or not the layer has a bias)
__host__ __device__ inline int part_total(const gnntrk_mlp &m) {
    int n = m.hidden * m.in_dim + m.hidden + m.out_dim * m.hidden + m.out_dim;
    if (m.n_layers == 3) n += m.hidden * m.hidden + m.hidden;
    return n;
}

__device__ __forceinline__ void stage_seg_args(gnntrk_seg *dst, const gnntrk_seg (&seg)[GNNTRK_MAX_SEGS],
                                               int tid) {
#pragma unroll
    for (int j = 0; j < GNNTRK_MAX_SEGS; ++j)
        if (tid == j) dst[j] = seg[j];
}

// per-lane description of the chunks this lane loads: lane (g, c) owns chunks 8kk + 2g + h.
//
// Every VMEM instruction of the tile loops is executed unconditionally by all lanes: a
// load or store inside a divergent branch makes the outstanding-operation count unknown
// to the compiler's s_waitcnt placement, which then drains the whole queue (vmcnt(0)) in
// front of the next use and serialises the software pipeline on the memory latency.
// Lanes without a chunk read chunk 0 (masked to zero by `keep`), row ids of identity
// segments are read from a harmless in-bounds location and discarded by a select.
template <int KI>
struct LaneChunks {
    gch_ptr base[KI][2];
    gci_ptr idx[KI][2];  // never NULL (see above)
    int32_t stride[KI][2];
    uint32_t keep[KI][2][2], ones[KI][2][2], rmin[KI][2];
    bool has_idx[KI][2];
    bool pair8[KI];  // wide mode: the 16-byte load covers rows (r & ~1, r | 1) of an 8-byte-row tensor

    __device__ __forceinline__ void init(const SlotPlan &P, const gnntrk_seg *seg, int g) {
#pragma unroll
        for (int kk = 0; kk < KI; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = 8 * kk + 2 * g + h;
                const int j = (p < P.n_chunks) ? P.seg[p] : -1;
                const int js = j >= 0 ? j : P.seg[0];   // chunk 0 always belongs to a segment
                const int first = j >= 0 ? P.first[p] : 0;
                int d = 0;
                rmin[kk][h] = 0x80008000u;
                if (j >= 0) {
                    d = seg[j].dim - 4 * first;
                    d = d > 4 ? 4 : d;
                    if (seg[j].relu) rmin[kk][h] = 0u;
                }
                base[kk][h] = (gch_ptr)(reinterpret_cast<const uint16_t *>(seg[js].ptr) + 4 * first);
                stride[kk][h] = seg[js].stride;
                has_idx[kk][h] = seg[js].idx != nullptr;
                // identity segment: n_rows rows of >= 8 bytes -> its own data is a valid int32[n_rows]
                idx[kk][h] = has_idx[kk][h] ? (gci_ptr)seg[js].idx : (gci_ptr) reinterpret_cast<const int32_t *>(seg[js].ptr);
                if (h == 0) {
                    // wide mode (see wide_ok()): lanes without chunks mirror pair 0 of the layout
                    const int p0 = j >= 0 ? p : 0;
                    const bool second = (p0 + 1 < P.n_chunks) && P.seg[p0 + 1] >= 0;
                    pair8[kk] = !second;
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
        for (int h = 0; h < 2; ++h) {
            const int32_t v = L.idx[kk][h][row];
            r.v[kk][h] = L.has_idx[kk][h] ? v : row;
        }
}
template <int KI>
__device__ __forceinline__ void load_raw(const LaneChunks<KI> &L, const RowIds<KI> &r, RawTile<KI> &t) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            t.v[kk][h] = *reinterpret_cast<const u32x2 GNNTRK_GLOBAL *>(
                L.base[kk][h] + (int64_t)r.v[kk][h] * L.stride[kk][h]);
}
// Wide mode: ONE 16-byte load per lane and k-step (8-byte-per-lane streams top out near
// 3.2 TB/s on MI355X, 16-byte ones reach 6 TB/s).  Every chunk pair (2g, 2g+1) is either
// one 16-byte row of a segment ("row16") or a single 4-feature chunk of an identity segment
// with 8-byte rows ("pair8": the load covers two adjacent rows, the lane keeps its half).
// Host-side eligibility: wide_ok().
template <int KI>
__device__ __forceinline__ void load_row_ids_wide(const LaneChunks<KI> &L, int32_t row, RowIds<KI> &r) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) {
        const int32_t v = L.idx[kk][0][row];
        r.v[kk][0] = L.has_idx[kk][0] ? v : row;
        r.v[kk][1] = row;
    }
}
template <int KI>
__device__ __forceinline__ void load_raw_wide(const LaneChunks<KI> &L, const RowIds<KI> &r, int32_t pair_cap,
                                              RawTile<KI> &t) {
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) {
        const int32_t row = r.v[kk][0];
        int32_t base_row = row & ~1;
        base_row = base_row < pair_cap ? base_row : pair_cap;  // keep both rows of the pair in bounds
        const int32_t rr = L.pair8[kk] ? base_row : row;
        // (a pair capped at the end of an odd-length tensor starts on an odd row: 8-byte aligned)
        typedef u32x4 __attribute__((aligned(8))) u32x4_a8;
        const u32x4 w = *reinterpret_cast<const u32x4_a8 GNNTRK_GLOBAL *>(L.base[kk][0] + (int64_t)rr * L.stride[kk][0]);
        const bool hi = L.pair8[kk] && (row != base_row);
        t.v[kk][0][0] = hi ? w[2] : w[0];
        t.v[kk][0][1] = hi ? w[3] : w[1];
        t.v[kk][1][0] = w[2];
        t.v[kk][1][1] = w[3];
    }
}
// the host-side condition for wide mode
inline bool wide_ok(const SlotPlan &P, const gnntrk_seg *seg, int64_t n_rows) {
    for (int p = 0; p < 8 * P.KI; p += 2) {
        const int j0 = p < P.n_chunks ? P.seg[p] : -1, j1 = p + 1 < P.n_chunks ? P.seg[p + 1] : -1;
        if (j0 < 0) {
            if (j1 >= 0) return false;
            continue;  // empty pair (or the ones-only chunk): mirrors pair 0
        }
        if (((uintptr_t)seg[j0].ptr & 15) != 0) return false;
        if (j1 == j0 && P.first[p + 1] == P.first[p] + 1 && (P.first[p] & 1) == 0 && seg[j0].stride % 8 == 0)
            continue;  // row16
        if (j1 < 0 && seg[j0].idx == nullptr && seg[j0].stride == 4 && P.first[p] == 0 && n_rows >= 2)
            continue;  // pair8
        return false;
    }
    return true;
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

// Bias table of the accumulator-initialisation form (SlotPlan::bias_init): btab[16 t + f] = bias of hidden
// feature 16 t + f of the MIDDLE layer, btab[16 HT + o] = bias of output o of the last layer, both rounded
// to bf16 (what the fragment form stores), 0 where there is none.
template <int HT>
__device__ __forceinline__ void fill_bias_table(float *btab, const AugWeights &w, int tid, int nthreads) {
    for (int i = tid; i < (HT + 1) * 16; i += nthreads) {
        float v = 0.f;
        if (i < 16 * HT) {
            if (i < w.hidden && w.b2) v = w.b2[i];
        } else if (i - 16 * HT < w.out_dim && w.b3) {
            v = w.b3[i - 16 * HT];
        }
        btab[i] = __uint_as_float((uint32_t)bf16_bits(v) << 16);
    }
}
// accumulator tile of lane (g, c) initialised with four consecutive table entries (features 4g .. 4g + 3)
__device__ __forceinline__ f32x4 bias_frag(const float *btab16, int lane) {
    return *reinterpret_cast<const f32x4 *>(btab16 + 4 * (lane >> 4));
}

// layers 1..(last-1): inputs B -> packed hidden activations feeding the last layer
template <int KI, int HT, bool THREE, bool BI = false>
__device__ __forceinline__ void hidden_chain(const uint32_t *img, const u32x4 (&B)[KI], int lane,
                                             u32x2 (&P1)[HT], u32x2 (&P2)[HT], const float *btab = nullptr) {
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
            P2[t] = pack_tile_relu(contract_hidden<HT>(img + I::kA2 + t * hid_k_dwords(HT), P1, lane,
                                                       BI ? bias_frag(btab + 16 * t, lane) : zero));
    }
}

// The same for D independent tiles with ONE LDS read of every weight fragment (backward kernel:
// the fragments cannot stay in registers next to the weight-gradient accumulators).
template <int HT, int D>
__device__ __forceinline__ void contract_hidden_d(const uint32_t *img, const u32x2 (&P)[D][HT], int lane,
                                                  f32x4 (&acc)[D]) {
#pragma unroll
    for (int u = 0; u < HT / 2; ++u) {
        const u32x4 fr = frag_k32(img + 256 * u, lane);
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = mfma_bf16_k32(fr, join(P[d][2 * u], P[d][2 * u + 1]), acc[d]);
    }
    if (HT % 2) {
        const u32x2 zero = {0u, 0u};
        const u32x4 fr = frag_k32(img + 256 * (HT / 2), lane);
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = mfma_bf16_k32(fr, join(P[d][HT - 1], zero), acc[d]);
    }
}
template <int KI, int HT, bool THREE, int D, bool BI = false>
__device__ __forceinline__ void hidden_chain_d(const uint32_t *img, const u32x4 (&B)[D][KI], int lane,
                                               u32x2 (&P1)[D][HT], u32x2 (&P2)[D][HT], const float *btab = nullptr) {
    using I = FwdImg<KI, HT>;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        f32x4 acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = zero;
#pragma unroll
        for (int kk = 0; kk < KI; ++kk) {
            const u32x4 fr = frag_k32(img + I::kA1 + (t * KI + kk) * 256, lane);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = mfma_bf16_k32(fr, B[d][kk], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) P1[d][t] = pack_tile_relu(acc[d]);
    }
    if (THREE) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            f32x4 acc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = BI ? bias_frag(btab + 16 * t, lane) : zero;
            contract_hidden_d<HT, D>(img + I::kA2 + t * hid_k_dwords(HT), P1, lane, acc);
#pragma unroll
            for (int d = 0; d < D; ++d) P2[d][t] = pack_tile_relu(acc[d]);
        }
    }
}

// Store-redirect slots of the forward kernel (16 bytes per lane of the largest grid): the
// forward C ABI has no workspace argument.  Contents are never read.
constexpr int kFwdMaxBlocks = 2048;
__device__ uint8_t g_fwd_trash[(size_t)kFwdMaxBlocks * kBlock * 16];

// R = tiles sharing one output tile.  With at most four output features (the edge
// embeddings, the edge weight) only lane group 0 of the last layer's D tile carries data;
// R = 4 gives every tile of a group its own copy of the last-layer fragments with the
// rows rotated by 4v, so tile v lands in lane group v, the four last-layer MFMA chains
// accumulate into ONE tile (the other row blocks of each fragment are zero: exact), and a
// single full-wave store writes 4 x 16 rows.  R = 1 is the plain layout (wider outputs).
// Waves per SIMD the forward kernels of up to four hidden tiles are compiled for, and workgroups per CU their
// persistent grid is sized for (the two have to agree: a grid larger than what is resident runs in rounds, the last
// one part empty).  Round 5 found the hot instantiations at 216 registers = TWO waves per SIMD under a grid of five
// workgroups per CU (the max-ilp scheduling strategy of mlp_bf16_fwd.hip had doubled the 108 of round 1).
#ifndef GNNTRK_FWD16_WAVES_PER_SIMD
#define GNNTRK_FWD16_WAVES_PER_SIMD 0   // (0: no bound stated)
#endif
#ifndef GNNTRK_FWD16_BLOCKS_PER_CU
#define GNNTRK_FWD16_BLOCKS_PER_CU 5
#endif
// (measured per instantiation, same box, alternating - tools/ab_step.sh: the two-layer encoders are fastest at four
//  waves per SIMD (126 registers, no spill: edge encoder 0.58 -> 0.48 ms per 64 M rows), the three-layer bf16-output
//  shapes at three (154-165 registers: relational 0.77 -> 0.76, object 0.131 -> 0.118), the edge-weight head without a
//  bound (230 registers, two waves: 1.16 against 1.25 at three); at four the three-layer shapes spill 17-42 registers
//  and run at half speed)
template <int KI, int HT, bool THREE, bool SIG, int R, bool WIDE>
__host__ __device__ constexpr int fwd16_waves_per_simd() {
    if (GNNTRK_FWD16_WAVES_PER_SIMD > 0) return HT <= 4 ? GNNTRK_FWD16_WAVES_PER_SIMD : 1;
    if (KI != 1 || HT > 3) return 1;
    if (!THREE) return (R == 4 && WIDE && !SIG) ? 4 : 3;   // (the other two-layer forms spill 4-12 registers at four)
    return SIG ? 1 : 3;
}
template <int KI, int HT, bool THREE, bool SIG, int R_, bool WIDE_>
__global__ __launch_bounds__(kBlock, (fwd16_waves_per_simd<KI, HT, THREE, SIG, R_, WIDE_>())) void mlp16_fwd_kernel(const gnntrk_mlp_fwd_args a) {
    constexpr int R = R_, OT = 1;
    constexpr bool WIDE = WIDE_, BI = false, kFwdSkel = false;
#include "mlp_bf16_fwd_body.inc"
}
// The I/O SKELETON of a forward launch (debug_flags & 4096; results are NOT outputs): every load and every store of
// the kernel above - id streams, gathered and tile rows, the output rows - with the same prefetch distance, workgroup
// shape and launch bounds, and an XOR where the MLP is.  Its time is the floor this access pattern sets for the real
// kernel (bench.py: roofline.access_floor.forward), as mlp16_bwd_skel_kernel is for the backward.
template <int KI, int HT, bool THREE, bool SIG, int R_, bool WIDE_>
__global__ __launch_bounds__(kBlock, (fwd16_waves_per_simd<KI, HT, THREE, SIG, R_, WIDE_>())) void mlp16_fwd_skel_kernel(const gnntrk_mlp_fwd_args a) {
    constexpr int R = R_, OT = 1;
    constexpr bool WIDE = WIDE_, BI = false, kFwdSkel = true;
#include "mlp_bf16_fwd_body.inc"
}
// outputs of 17 .. 48 features (OT output tiles), up to four k-steps of inputs: plain form
template <int KI, int HT, int OT_, bool THREE>
__global__ __launch_bounds__(kBlock) void mlp16_fwd_ot_kernel(const gnntrk_mlp_fwd_args a) {
    constexpr int R = 1, OT = OT_;
    constexpr bool WIDE = false, BI = false, SIG = false, kFwdSkel = false;
#include "mlp_bf16_fwd_body.inc"
}
// hidden width 64 with biases: no constant-one row, biases as accumulator initial values
template <int KI, int HT, bool THREE, bool SIG>
__global__ __launch_bounds__(kBlock) void mlp16_fwd_bi_kernel(const gnntrk_mlp_fwd_args a) {
    constexpr int R = 1, OT = 1;
    constexpr bool WIDE = false, BI = true, kFwdSkel = false;
#include "mlp_bf16_fwd_body.inc"
}

// =========================================================================== backward
// Full recompute (only the op inputs are saved), one wave per 16-row tile:
//   S0  inputs -> B operand (as forward); recompute h1 (, h2) and, for the RELU / SIGMOID
//       epilogues, the pre-activation output
//   S1  upstream gradient: sum of the gout terms, epilogue derivative, rounded to bf16
//   S2  dX chain on W^T fragments:  g_hidden = relu'(h) * (W^T g)  ->  input gradients,
//       written per chunk (8 bytes per lane) to the row-aligned gradient slices
//   S3  weight gradients: dW[o][i] += sum_rows g[row][o] * act[row][i] - a K = rows
//       contraction; both operands go through a per-wave LDS image [row][feature] and come
//       back transposed with ds_read_b64_tr_b16.  Accumulators stay in registers for the
//       whole launch (fp32), are written once per wave as a partial block in parameter
//       layout and reduced in a fixed order (reduce_partials in mlp.hip): bit-reproducible.
// The bias gradients are the `ones` columns of the augmented dW tiles.
// dwords of the backward kernel's weight-fragment image (also the capacity of the in-LDS
// reduction of the partial blocks, see the end of mlp16_bwd_kernel)
__host__ __device__ constexpr int bwd16_img_dwords(int KI, int HT, int GT, bool three) {
    return HT * KI * 256 + (1 + HT) * hid_k_dwords(HT) /* forward A1 | A2 | A3 */ + HT * 128 +
           (three ? HT * hid_k_dwords(HT) : 0) + GT * hid_k_dwords(HT);
}

template <int KI, int HT, int GT, bool THREE>
struct BwdImg {
    using F = FwdImg<KI, HT>;
    static constexpr int kD3 = F::kTotal;                       // W_last'^T : [HT] x K16
    static constexpr int kD2 = kD3 + HT * 128;                  // W_mid'^T  : [HT] x hid_k
    static constexpr int kD1 = kD2 + (THREE ? HT * hid_k_dwords(HT) : 0);  // W1'^T : [GT] x hid_k
    static constexpr int kTotal = kD1 + GT * hid_k_dwords(HT);
    static_assert(kTotal == bwd16_img_dwords(KI, HT, GT, THREE), "image size formula out of sync");
};

// per-wave staging images (bytes): X = activations (in / h1 / h2), Gs = gradients
template <int KI, int HT>
struct BwdStage {
    // Bank-conflict-free images (64 banks x 4 bytes; b64 accesses are served 32 lanes per
    // cycle, b128 16 lanes): a 16-feature tile is [row][4 chunks of 8 bytes] with the chunk
    // position XORed with (row >> 2) & 3; the input image is [row][4 KI units of 16 bytes]
    // with a 32-byte row pad and the unit position XORed with (row >> 3) & 1.
    static constexpr int kInRow = 64 * KI + 32;  // bytes per row of the input image
    static constexpr int kIn = 16 * kInRow;  // inputs: written in S0, read by the first-layer stage
    static constexpr int kX = HT * 512;      // h2 / h1
    static constexpr int kG = HT * 512;      // g_out / g_h2 / g_h1
    static constexpr int kWave = kIn + kX + kG;
};

template <int KI, int HT, int GT, bool THREE>
__device__ __forceinline__ void pack_backward_weights(uint32_t *img, const AugWeights &w, const SlotPlan &P,
                                                      const gnntrk_seg *seg, int tid, int nthreads) {
    using I = BwdImg<KI, HT, GT, THREE>;
    for (int t = 0; t < HT; ++t)  // g_hidden_pre[f = 16t + i] = sum_o W_last'[o][f] g_out[o],  k = o = 4g + e
        pack_frag_k16(img + I::kD3 + t * 128, [&](int i, int g, int e) { return w.wlast(4 * g + e, 16 * t + i); },
                      tid, nthreads);
    if (THREE)
        for (int t = 0; t < HT; ++t)
            pack_hidden_k<HT>(img + I::kD2 + t * hid_k_dwords(HT), 16 * t,
                              [&](int f1, int f2) { return w.wmid(f2, f1); }, tid, nthreads);
    for (int T = 0; T < GT; ++T)
        pack_hidden_k<HT>(img + I::kD1 + T * hid_k_dwords(HT), 16 * T,
                          [&](int m, int f) {
                              const int q = m >> 2;
                              if (q >= P.n_gchunks) return 0.f;
                              // (pad slots and the ones slot have no input gradient: zero columns, so the
                              //  pad elements of a gradient row are +0 without a mask)
                              const int col = slot_col(P, seg, 4 * P.gchunk[q] + (m & 3));
                              return col >= 0 ? w.w1(f, col) : 0.f;
                          },
                          tid, nthreads);
}

// upstream gradient terms of one tile, raw (bf16: two terms x 4 features; fp32: 4 floats)
struct RawGout {
    u32x4 v;
    u32x2 w;   // third term (buffer form only)
};

// ---- buffer-addressed I/O of the backward tile loop ----------------------------------------
// The generic I/O above gives every lane its own pointers (lanes of one load read different
// tensors): per tile and half about 45 VALU instructions of 64-bit address arithmetic, "no index"
// selects, pad masks and store redirections next to about 95 of arithmetic (round 3 read the kernel as
// issue bound; round 5 measured its access floor and the issue costs - DESIGN.md 4.3 - and it is neither).  For the shapes of the default models the same accesses are made through
// WAVE-UNIFORM buffer descriptors (tile_bf16.h): one access per TENSOR, executed by all lanes; a
// lane that takes no part carries the offset 2^31 and falls out in the hardware range check, as do
// rows past the end; rows of the tile itself go through the scalar offset, gathered rows through
// `(id << log2(row bytes)) + lane constant`.  What an access looks like is split into a
// compile-time shape (which lanes, which id streams, width: BufOpShape, one IoXxx class per
// supported shape) and run-time arguments (BufOpArgs); the host builds both from the segment list
// (make_buf_plan) and takes the buffer form when the shape matches an IoXxx class, every tensor
// states its size (gnntrk_seg.rows) and stays below 2^31 bytes with power-of-two row sizes.
// Padding elements of input rows are read as stored: every producer of padded bf16 rows in this
// library writes them as zero (include/gnntrk.h).
constexpr int kBufLoads = 6, kBufIds = 3, kBufStores = 6, kBufGouts = 3;
constexpr uint32_t kBufOut = 0x80000000u;   // lane offset of a lane that takes no part

struct BufOpShape {
    uint8_t w16;       // 16-byte access (else 8 bytes; the fp32 upstream gradient: 4)
    uint8_t gathered;  // rows through an id stream (else the tile's own rows)
    int8_t sa, sb;     // id streams of the lanes that take part (sb = -1: one stream)
    uint8_t part;      // bit g: lane group g takes part
    uint8_t use_b;     // bit g: lane group g reads stream sb
    uint8_t slot;      // loads: dword of the B operand the data lands in (0 / 2); stores: gradient tile T
    uint8_t _pad;
};
// (what the tile loop needs at compile time; `part` / `use_b` only feed lane constants in the prologue)
__host__ __device__ constexpr bool same_shape(const BufOpShape &x, const BufOpShape &y) {
    return x.w16 == y.w16 && x.gathered == y.gathered && x.sa == y.sa && x.sb == y.sb && x.slot == y.slot;
}
struct BufOpArgs {
    const void *ptr;
    uint32_t bytes;
    uint8_t shift;    // log2(bytes per row)
    uint8_t off8[4];  // byte offset inside the row per lane group
    uint8_t _pad[3];
};
struct BufPlan {
    int32_t ok, n_load, n_ids, n_sids, n_store, n_gout;
    int32_t ones_dword;   // dword of the B operand holding the ones slot (-1: none), its lane group, its value
    int32_t ones_group;
    uint32_t ones_bits;
    int32_t any_relu;
    int32_t gate_mode;    // relu' on the input gradients: 0 no segment, 1 every segment, 2 mixed
    uint32_t rmin[4][2];  // per lane group and chunk of the pair: 0 (ReLU on load) or 0x80008000
    BufOpShape load_s[kBufLoads], gout_s[kBufGouts], store_s[kBufStores];
    BufOpArgs load[kBufLoads], gout[kBufGouts], store[kBufStores];
    BufOpArgs ids[kBufIds], sids[kBufIds];   // int32 id streams of the loads (one unit ahead) / of the stores
    // target-side fold of one gathered segment's gradient inside the kernel (gnntrk_gfold; IO::kFold): the lane
    // groups of gradient tile 0 that hold the folded slice take no per-row store - one row per NODE leaves the kernel
    int32_t fold_on;
    uint32_t fold_part;     // bit g: lane group g holds a chunk of the folded slice
    BufOpArgs fold_out;     // folded rows [n_nodes] (off8[g]: byte offset of group g's chunk inside a row)
    int32_t fold_stream;    // which id stream of the loads (ids[]) the folded segment is gathered through
    uint32_t fold_nodes;    // node rows in fold_out; the units' carry rows follow them in the same allocation
};

#ifndef GNNTRK_ABLATE
#define GNNTRK_ABLATE 0   // (timing-only ablation builds of the backward tile loop, see mlp_bf16_bwd_body.inc)
#endif
#ifndef GNNTRK_BWD_REG_FRAGS
#define GNNTRK_BWD_REG_FRAGS 0   // (1: first- / last-layer weight fragments of the two-tile buffer shapes in registers)
#endif
#ifndef GNNTRK_PROBE_SKIP_PACK
#define GNNTRK_PROBE_SKIP_PACK 0   // (1: timing probe of the kernels' prologue - no fragment packing, results are garbage)
#endif
#ifndef GNNTRK_BWD_HOT
#define GNNTRK_BWD_HOT 1      // (0: the relational shape takes the generic tile body too - A/B builds)
#endif
#ifndef GNNTRK_BWD_SGB
#define GNNTRK_BWD_SGB 0      // (1: sched_group_barrier pipeline over the tile body of the hot buffer-addressed shapes - A/B builds)
#endif
#ifndef GNNTRK_BWD_SGB_N
#define GNNTRK_BWD_SGB_N 62   // MFMAs of the tile body
#define GNNTRK_BWD_SGB_V 4    // VALU instructions per MFMA
#define GNNTRK_BWD_SGB_R 1    // DS reads per MFMA
#define GNNTRK_BWD_SGB_WP 4   // one DS write every so many MFMAs
#endif
#ifndef GNNTRK_BWD_STATIC_GATE
#define GNNTRK_BWD_STATIC_GATE 1   // (0: the input ReLU / relu' pattern of the buffer-addressed shapes stays a run-time value - A/B builds)
#endif
#ifndef GNNTRK_BWD_STATIC_EPI
#define GNNTRK_BWD_STATIC_EPI 1   // (0: the epilogue of the buffer-addressed shapes stays a run-time value - A/B builds)
#endif
constexpr int kEpiOf(int e) { return GNNTRK_BWD_STATIC_EPI ? e : -1; }

struct IoNone {   // the generic per-lane I/O
    static constexpr bool kFold = false;
    static constexpr uint32_t kFoldPart = 0;
    static constexpr int kFoldStream = 0;
    static constexpr int kEpi = -1;
    static constexpr int NL = 0, NI = 0, NSI = 0, NS = 0, NG = 0, kOnesDword = -1;
    static constexpr BufOpShape load[1] = {}, gout[1] = {}, store[1] = {};
};
// relational model of an interaction network (interaction_network.py:75-89): x[tgt] | x[src] (16-byte
// node rows, one descriptor, two id streams), e (8-byte rows of the tile); upstream gradient
// g_e~ + g_aggr[tgt]; gradient slices g_x_i (tile rows), g_x_j (rows through the source-sort
// permutation), g_e
// FOLD_ (round 6, gnntrk_gfold): g_x_i leaves the kernel summed per target node - its per-row store is gone
template <int NG_, bool FOLD_ = false>   // NG_ = 3: a third upstream term on the tile's rows (the edge-weight head's share, see ops_bf16.grad_tap)
struct IoRelational {
    static constexpr int kEpi = kEpiOf(GNNTRK_EPI_NONE);
    static constexpr bool kFold = FOLD_;
    static constexpr uint32_t kFoldPart = 0b0011;
    static constexpr int kFoldStream = 0;   // x_i is gathered through the first id stream (the CSR targets)
    static constexpr int NL = 2, NI = 2, NSI = 1, NS = FOLD_ ? 2 : 3, NG = NG_, kOnesDword = 2;
    static constexpr BufOpShape load[2] = {{1, 1, 0, 1, 0b0011, 0b0010, 0, 0}, {0, 0, -1, -1, 0b0100, 0, 0, 0}};
    static constexpr BufOpShape gout[3] = {{0, 0, -1, -1, 0b0001, 0, 0, 0}, {0, 1, 0, -1, 0b0001, 0, 0, 0},
                                           {0, 0, -1, -1, 0b0001, 0, 0, 0}};
    static constexpr BufOpShape kXi = {0, 0, -1, -1, 0b0011, 0, 0, 0}, kXj = {0, 1, 0, -1, 0b1100, 0, 0, 0},
                                kE = {0, 0, -1, -1, 0b0001, 0, 1, 0};
    static constexpr BufOpShape store[3] = {FOLD_ ? kXj : kXi, FOLD_ ? kE : kXj, FOLD_ ? BufOpShape{} : kE};
};
using IoRelationalF2 = IoRelational<2, true>;   // (names without a comma: the launch macros take them as one argument)
using IoRelationalF3 = IoRelational<3, true>;
// object model (interaction_network.py:92-103): x (16-byte rows) | aggr (8-byte rows), all rows of
// the tile; one upstream term; gradient slices g_x, g_aggr
struct IoObject {
    static constexpr bool kFold = false;
    static constexpr uint32_t kFoldPart = 0;
    static constexpr int kFoldStream = 0;
    static constexpr int kEpi = kEpiOf(GNNTRK_EPI_RESIDUAL);
    static constexpr int NL = 2, NI = 0, NSI = 0, NS = 2, NG = 1, kOnesDword = 2;
    static constexpr BufOpShape load[2] = {{1, 0, -1, -1, 0b0001, 0, 0, 0}, {0, 0, -1, -1, 0b0010, 0, 0, 0}};
    static constexpr BufOpShape gout[1] = {{0, 0, -1, -1, 0b0011, 0, 0, 0}};
    static constexpr BufOpShape store[2] = {{0, 0, -1, -1, 0b0011, 0, 0, 0}, {0, 0, -1, -1, 0b0100, 0, 0, 0}};
};
// the edge-weight head (edge_classifier.py:108-116): h[src] | h[tgt] (16-byte node rows, one descriptor,
// two id streams), four 8-byte edge tensors of the tile's rows; fp32 upstream gradient of the one
// weight per edge; gradient slices g_h[src] (rows through the source-sort permutation), g_h[tgt],
// g_e0 .. g_e3
template <bool FOLD_ = false>   // FOLD_: g_h[tgt] leaves the kernel summed per target node (gnntrk_gfold)
struct IoHeadT {
    static constexpr int kEpi = kEpiOf(GNNTRK_EPI_SIGMOID);
    static constexpr bool kFold = FOLD_;
    static constexpr uint32_t kFoldPart = 0b1100;
    static constexpr int kFoldStream = 1;   // h[src] | h[tgt]: the targets are the second id stream
    static constexpr int NL = 5, NI = 2, NSI = 1, NS = FOLD_ ? 5 : 6, NG = 1, kOnesDword = 2;
    static constexpr BufOpShape load[5] = {{1, 1, 0, 1, 0b0011, 0b0010, 0, 0}, {0, 0, -1, -1, 0b0100, 0, 0, 0},
                                           {0, 0, -1, -1, 0b0100, 0, 2, 0}, {0, 0, -1, -1, 0b1000, 0, 0, 0},
                                           {0, 0, -1, -1, 0b1000, 0, 2, 0}};
    static constexpr BufOpShape gout[1] = {{0, 0, -1, -1, 0b0001, 0, 0, 0}};
    static constexpr BufOpShape kHs = {0, 1, 0, -1, 0b0011, 0, 0, 0}, kHt = {0, 0, -1, -1, 0b1100, 0, 0, 0},
                                kE0 = {0, 0, -1, -1, 0b0001, 0, 1, 0}, kE1 = {0, 0, -1, -1, 0b0010, 0, 1, 0},
                                kE2 = {0, 0, -1, -1, 0b0100, 0, 1, 0}, kE3 = {0, 0, -1, -1, 0b1000, 0, 1, 0};
    static constexpr BufOpShape store[6] = {kHs, FOLD_ ? kE0 : kHt, FOLD_ ? kE1 : kE0, FOLD_ ? kE2 : kE1,
                                            FOLD_ ? kE3 : kE2, FOLD_ ? BufOpShape{} : kE3};
};
using IoHead = IoHeadT<false>;
using IoHeadF = IoHeadT<true>;
// bias-free encoder of 8-byte rows (edge_classifier.py:66-69 on the four edge features): one tensor,
// rows of the tile, weight gradients only
template <int NG_>
struct IoEncoder8 {
    static constexpr bool kFold = false;
    static constexpr uint32_t kFoldPart = 0;
    static constexpr int kFoldStream = 0;
    static constexpr int kEpi = kEpiOf(GNNTRK_EPI_RELU);
    static constexpr int NL = 1, NI = 0, NSI = 0, NS = 0, NG = NG_, kOnesDword = -1;
    static constexpr BufOpShape load[1] = {{0, 0, -1, -1, 0b0001, 0, 0, 0}};
    static constexpr BufOpShape gout[2] = {{0, 0, -1, -1, 0b0001, 0, 0, 0}, {0, 0, -1, -1, 0b0001, 0, 0, 0}};
    static constexpr BufOpShape store[1] = {};
};

// Waves per workgroup of the buffer-addressed backward kernels (the hot shapes).  A workgroup is one fragment image
// + one staging image per wave (9.2 KB with D = 2): four waves (60 KB) give two workgroups = 2 waves per SIMD on
// a CU; six waves (79 KB) would still be two workgroups = 3 waves per SIMD - but the kernels hold 202 - 206
// registers, and at the 168 of three waves per SIMD they spill 33 (relational) / 83 (head)
// registers into the tile loop: measured 6.76 against 2.88 ms and 9.88 against 3.67 ms per 64 M rows (round 4,
// tools/bench_bwd_io.py).  The occupancy of these kernels is set by registers AND LDS; the switch stays for A/B.
#ifndef GNNTRK_BWD16_BUF_WAVES
#define GNNTRK_BWD16_BUF_WAVES 4
#endif
constexpr int kBwd16BufWaves = GNNTRK_BWD16_BUF_WAVES;
// Tiles per iteration of the buffer-addressed kernels (2: K = 32 weight-gradient contractions over two 16-row
// halves; 1: half the per-wave state and staging - 162 registers, 42 KB: three workgroups per CU = 3 waves per
// SIMD without a spill.  Measured SLOWER, round 4: relational 2.83-2.89 against 2.71 ms, head 3.57 against 3.30 per
// 64 M rows on one box, alternating - a third wave does not pay for half-used weight-gradient MFMAs and the
// per-tile overheads no longer shared by two tiles.  A/B switch.)
#ifndef GNNTRK_BWD16_BUF_D
#define GNNTRK_BWD16_BUF_D 2
#endif
constexpr int kBwd16BufD = GNNTRK_BWD16_BUF_D;
template <class IO>
__host__ __device__ constexpr int bwd16_block_waves() {
    return IO::NL > 0 ? kBwd16BufWaves : kWaves;
}

template <int KI, int HT, int GT, bool THREE, bool G32, int D_, class IO_ = IoNone>
__global__ __launch_bounds__(64 * bwd16_block_waves<IO_>(), HT >= 5 ? 1 : ((IO_::NL > 0 && D_ == 1) ? 3 : 2) * bwd16_block_waves<IO_>() / 4) void mlp16_bwd_kernel(const gnntrk_mlp_bwd_args a, float *part,
                                                                           uint8_t *trash, const BufPlan bp) {
    constexpr int D = D_, OT = 1, kSkel = 0;
    using IO = IO_;
    constexpr bool BI = false;
#include "mlp_bf16_bwd_body.inc"
}
// The I/O SKELETON of a buffer-addressed launch (debug_flags & 4096; results are NOT gradients): every load and
// every store of the kernel above with the same descriptors, offsets, prefetch distance, workgroup shape and
// launch bounds, and no arithmetic between them.  Its time is the floor this ACCESS PATTERN sets for the real
// kernel at its occupancy, whatever the instruction stream does (bench.py: roofline.access_floor; round 5
// measured 2.52 of 2.75 ms for the relational shape on shuffled node ids, 1.35 ms on sequential ones).
template <int KI, int HT, int GT, bool THREE, bool G32, int D_, class IO_>
__global__ __launch_bounds__(64 * bwd16_block_waves<IO_>(), 2 * bwd16_block_waves<IO_>() / 4) void mlp16_bwd_skel_kernel(
    const gnntrk_mlp_bwd_args a, float *part, uint8_t *trash, const BufPlan bp) {
    constexpr int D = D_, OT = 1, kSkel = 128;
    using IO = IO_;
    constexpr bool BI = false;
#include "mlp_bf16_bwd_body.inc"
}
// outputs of 17 .. 48 features (OT output tiles) / up to four k-steps of inputs: generic I/O, one tile per
// iteration, one workgroup per CU, all 2 KI input-gradient tiles (unwanted ones go to the trash slots)
template <int KI, int HT, int OT_, bool THREE>
__global__ __launch_bounds__(kBlock, 1) void mlp16_bwd_ot_kernel(const gnntrk_mlp_bwd_args a, float *part, uint8_t *trash,
                                                                const BufPlan bp) {
    constexpr int D = 1, OT = OT_, GT = 2 * KI, kSkel = 0;
    using IO = IoNone;
    constexpr bool BI = false, G32 = false;
#include "mlp_bf16_bwd_body.inc"
}
// hidden width 64 with biases (SlotPlan::bias_init): generic I/O, one tile per iteration
template <int KI, int HT, int GT, bool THREE, bool G32>
__global__ __launch_bounds__(kBlock, (HT >= 5 || KI >= 2) ? 1 : 2) void mlp16_bwd_bi_kernel(const gnntrk_mlp_bwd_args a, float *part,
                                                                                          uint8_t *trash, const BufPlan bp) {
    constexpr int D = 1, OT = 1, kSkel = 0;
    using IO = IoNone;
    constexpr bool BI = true;
#include "mlp_bf16_bwd_body.inc"
}

// ------------------------------------------------------------------ launchers
// n_rows == 0 is a valid no-op: row pointers may then be NULL (what an empty tensor hands over)
int check_bf16_mlp(const gnntrk_mlp &m, int n_seg, const gnntrk_seg *seg, const char *who, int64_t n_rows) {
    if (m.n_layers != 2 && m.n_layers != 3) return fail(GNNTRK_EUNSUPPORTED, "mlp(bf16): n_layers must be 2 or 3");
    if (m.in_dim < 1 || m.hidden < 1 || m.hidden > 16 * kMaxHiddenTiles16 || m.out_dim < 1 || m.out_dim > kMaxOut16)
        return fail(GNNTRK_EUNSUPPORTED, "mlp(bf16): hidden must be in [1,128], out in [1,48]");
    if (n_seg < 1 || n_seg > GNNTRK_MAX_SEGS) return fail(GNNTRK_EINVAL, "mlp(bf16): bad segment count");
    int tot = 0;
    for (int j = 0; j < n_seg; ++j) {
        const int padded = (seg[j].dim + 3) / 4 * 4;
        if ((!seg[j].ptr && n_rows != 0) || seg[j].dim < 1 || seg[j].stride < padded || seg[j].stride % 4 != 0 ||
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

// ---- host: the buffer-addressed form of a backward launch (see BufPlan) ------------------------
inline bool buf_pow2(int64_t bytes, uint8_t &shift) {
    for (int sft = 2; sft <= 8; ++sft)
        if (((int64_t)1 << sft) == bytes) {
            shift = (uint8_t)sft;
            return true;
        }
    return false;
}
inline int buf_stream(BufOpArgs *arr, int &n, const void *idx, int64_t n_rows) {
    for (int i = 0; i < n; ++i)
        if (arr[i].ptr == idx) return i;
    if (n >= kBufIds) return -1;
    arr[n].ptr = idx;
    arr[n].bytes = (uint32_t)(n_rows * 4);
    arr[n].shift = 2;
    return n++;
}
// adds lane group g to an access of `ptr` (merging into an existing access of the same tensor / form)
inline bool buf_add(BufOpShape *sh, BufOpArgs *ar, int &n, int cap, const void *ptr, int64_t bytes, uint8_t shift,
                    bool w16, bool gathered, int stream, int slot, int g, int off8) {
    if (bytes <= 0 || bytes >= (int64_t)kBufOut || off8 < 0 || off8 > 255) return false;
    for (int i = 0; i < n; ++i) {
        if (ar[i].ptr != ptr || sh[i].w16 != (uint8_t)w16 || sh[i].gathered != (uint8_t)gathered || sh[i].slot != slot ||
            ((sh[i].part >> g) & 1))
            continue;
        bool use_b = false;
        if (gathered && stream != sh[i].sa) {
            if (sh[i].sb < 0)
                sh[i].sb = (int8_t)stream;
            else if (sh[i].sb != stream)
                continue;
            use_b = true;
        }
        sh[i].part |= (uint8_t)(1u << g);
        if (use_b) sh[i].use_b |= (uint8_t)(1u << g);
        ar[i].off8[g] = (uint8_t)off8;
        return true;
    }
    if (n >= cap) return false;
    sh[n] = BufOpShape{(uint8_t)w16, (uint8_t)gathered, (int8_t)(gathered ? stream : -1), -1, (uint8_t)(1u << g), 0,
                       (uint8_t)slot, 0};
    ar[n] = BufOpArgs{ptr, (uint32_t)bytes, shift, {0, 0, 0, 0}, {0, 0, 0}};
    ar[n].off8[g] = (uint8_t)off8;
    ++n;
    return true;
}

inline void make_buf_plan(BufPlan &B, const SlotPlan &P, const gnntrk_mlp_bwd_args *a, int GT) {
    memset(&B, 0, sizeof(B));
    B.ones_dword = -1;
    B.ones_group = -1;
    if (P.KI != 1 || a->n_rows <= 0) return;
    const int64_t M = a->n_rows;
    // inputs: lane group g owns the chunk pair (2g, 2g + 1)
    for (int g = 0; g < 4; ++g) {
        B.rmin[g][0] = B.rmin[g][1] = 0x80008000u;
        const int p0 = 2 * g, p1 = 2 * g + 1;
        const int j0 = (p0 < P.n_chunks) ? P.seg[p0] : -1, j1 = (p1 < P.n_chunks) ? P.seg[p1] : -1;
        for (int h = 0; h < 2; ++h) {
            const int j = h ? j1 : j0;
            if (j >= 0 && a->seg[j].relu) {
                B.rmin[g][h] = 0u;
                B.any_relu = 1;
            }
        }
        auto tensor = [&](int j, int64_t &bytes, uint8_t &shift, bool &gathered, int &stream) -> bool {
            const gnntrk_seg &sg = a->seg[j];
            if (sg.rows <= 0 || !buf_pow2((int64_t)sg.stride * 2, shift)) return false;
            bytes = (int64_t)sg.rows * sg.stride * 2;
            gathered = sg.idx != nullptr;
            stream = gathered ? buf_stream(B.ids, B.n_ids, sg.idx, M) : -1;
            if (!gathered && sg.rows < M) return false;
            return !gathered || stream >= 0;
        };
        int64_t bytes;
        uint8_t shift;
        bool gathered;
        int stream;
        const bool row16 = j0 >= 0 && j1 == j0 && P.first[p1] == P.first[p0] + 1 && (P.first[p0] & 1) == 0 &&
                           (a->seg[j0].stride * 2) % 16 == 0 && ((uintptr_t)a->seg[j0].ptr & 15) == 0;
        if (row16) {
            if (!tensor(j0, bytes, shift, gathered, stream)) return;
            if (!buf_add(B.load_s, B.load, B.n_load, kBufLoads, a->seg[j0].ptr, bytes, shift, true, gathered, stream, 0, g,
                         8 * P.first[p0]))
                return;
        } else {
            for (int h = 0; h < 2; ++h) {
                const int j = h ? j1 : j0, p = h ? p1 : p0;
                if (j < 0) continue;
                if (!tensor(j, bytes, shift, gathered, stream)) return;
                if (!buf_add(B.load_s, B.load, B.n_load, kBufLoads, a->seg[j].ptr, bytes, shift, false, gathered, stream,
                             2 * h, g, 8 * P.first[p]))
                    return;
            }
        }
    }
    if (B.n_load < 1) return;
    if (P.ones_slot >= 0) {
        const int p = P.ones_slot >> 2, r = P.ones_slot & 3;
        B.ones_group = p >> 1;
        B.ones_dword = 2 * (p & 1) + (r >> 1);
        B.ones_bits = kBf16One << (16 * (r & 1));
    }
    // upstream gradient terms
    const bool g32 = a->epilogue == GNNTRK_EPI_SIGMOID;
    for (int t = 0; t < a->n_gout; ++t) {
        const gnntrk_gterm &gt = a->gout[t];
        uint8_t shift;
        if (gt.rows <= 0 || !buf_pow2((int64_t)gt.stride * (g32 ? 4 : 2), shift)) return;
        if (g32 && a->mlp.out_dim != 1) return;
        const int64_t bytes = (int64_t)gt.rows * gt.stride * (g32 ? 4 : 2);
        const bool gathered = gt.idx != nullptr;
        const int stream = gathered ? buf_stream(B.ids, B.n_ids, gt.idx, M) : -1;
        if ((gathered && stream < 0) || (!gathered && gt.rows < M)) return;
        const int before = B.n_gout;
        for (int g = 0; g < 4; ++g) {
            if (4 * g >= a->mlp.out_dim) break;
            // (one access per term: a term never merges with another one)
            int n = (g == 0) ? B.n_gout : before;
            if (g == 0) {
                if (!buf_add(B.gout_s, B.gout, n, kBufGouts, gt.ptr, bytes, shift, false, gathered, stream, 0, g, 8 * g)) return;
                B.n_gout = n;
            } else {
                B.gout_s[before].part |= (uint8_t)(1u << g);
                B.gout[before].off8[g] = (uint8_t)(8 * g);
            }
        }
        if (B.n_gout != before + 1) return;
    }
    {
        int n_relu = 0, n_plain = 0;
        for (int q = 0; q < P.n_gchunks; ++q) (a->seg[P.seg[P.gchunk[q]]].relu ? n_relu : n_plain)++;
        B.gate_mode = n_relu == 0 ? 0 : n_plain == 0 ? 1 : 2;
    }
    // gradient slices: chunk q = 4T + g
    for (int T = 0; T < GT; ++T)
        for (int g = 0; g < 4; ++g) {
            const int q = 4 * T + g;
            if (q >= P.n_gchunks) continue;
            const int p = P.gchunk[q], j = P.seg[p];
            const gnntrk_gseg &gs = a->gseg[j];
            uint8_t shift;
            if (!buf_pow2((int64_t)gs.stride * 2, shift)) return;
            if (a->fold.ids && j == a->fold.seg) {
                // folded inside the kernel: no per-row store; one row per node (+ a carry row per unit) instead
                const gnntrk_seg &sg = a->seg[j];
                const int64_t units = (M + 31) / 32;
                if (T != 0 || gs.idx || !sg.idx || sg.idx != a->fold.ids || sg.rows <= 0 || shift != 4 ||
                    a->fold.n_nodes <= 0 || (a->fold.n_nodes + units) * 16 >= (int64_t)kBufOut)
                    return;
                if (!B.fold_on) {
                    B.fold_on = 1;
                    B.fold_out = BufOpArgs{gs.ptr, (uint32_t)((a->fold.n_nodes + units) * 16), 4, {0, 0, 0, 0}, {0, 0, 0}};
                    B.fold_nodes = (uint32_t)a->fold.n_nodes;
                    B.fold_stream = -1;
                    for (int i = 0; i < B.n_ids; ++i)
                        if (B.ids[i].ptr == (const void *)a->fold.ids) B.fold_stream = i;
                    if (B.fold_stream < 0) return;
                }
                B.fold_part |= 1u << g;
                B.fold_out.off8[g] = (uint8_t)(8 * P.first[p]);
                continue;
            }
            const bool gathered = gs.idx != nullptr;
            const int stream = gathered ? buf_stream(B.sids, B.n_sids, gs.idx, M) : -1;
            if (gathered && stream < 0) return;
            if (!buf_add(B.store_s, B.store, B.n_store, kBufStores, gs.ptr, M * gs.stride * 2, shift, false, gathered, stream,
                         T, g, 8 * P.first[p]))
                return;
        }
    if (a->fold.ids && !B.fold_on) return;   // (a fold that found no gradient chunk of its segment)
    if (B.fold_on && B.gate_mode == 2) return;   // (the folded gate is the uniform one)
    B.ok = 1;
}

template <class IO>
inline bool buf_plan_is(const BufPlan &B) {
    if ((B.fold_on != 0) != IO::kFold || (IO::kFold && (B.fold_part != IO::kFoldPart || B.fold_stream != IO::kFoldStream))) return false;
    if (!B.ok || B.n_load != IO::NL || B.n_ids != IO::NI || B.n_sids != IO::NSI || B.n_store != IO::NS ||
        B.n_gout != IO::NG || B.ones_dword != IO::kOnesDword)
        return false;
    for (int i = 0; i < IO::NL; ++i)
        if (!same_shape(B.load_s[i], IO::load[i])) return false;
    for (int i = 0; i < IO::NG; ++i)
        if (!same_shape(B.gout_s[i], IO::gout[i])) return false;
    for (int i = 0; i < IO::NS; ++i)
        if (!same_shape(B.store_s[i], IO::store[i])) return false;
    return true;
}

constexpr int kFwd16BlocksPerCu = GNNTRK_FWD16_BLOCKS_PER_CU;
constexpr int kBwd16BlocksPerCu = 2;
// weight-gradient-only launches (GT = 0: the encoders of raw dataset features) need 50 KB of LDS and
// under 100 registers: three workgroups per CU are resident, and the latency-bound tile loop takes them
constexpr int kBwd16BlocksPerCuLight = 3;
constexpr int kBwd16BlocksPerCuMax = 3;   // (sizes the workspace)


// the buffer-addressed instantiation of a launch, by name ("" = the generic per-lane I/O)
inline const char *buf_io_name(const BufPlan &B, int KI, int HT, int GT, bool three, bool g32, int debug_flags,
                               int epilogue) {
    if (!B.ok || (debug_flags & (64 | 128)) || KI != 1 || (HT != 1 && HT != 3)) return "";
    auto epi_ok = [&](int k) { return k < 0 || k == epilogue; };   // (a class with a static epilogue only takes that one)
    if (g32)
        return (GT == 2 && three && epi_ok(IoHead::kEpi) && buf_plan_is<IoHead>(B)) ? "IoHead"
               : (GT == 2 && three && HT == 3 && kBwd16BufD == 2 && epi_ok(IoHeadF::kEpi) && buf_plan_is<IoHeadF>(B)) ? "IoHeadF" : "";
    if (GT == 2 && three && epi_ok(IoRelational<2>::kEpi) && buf_plan_is<IoRelational<2>>(B)) return "IoRelational<2>";
    if (GT == 2 && three && epi_ok(IoRelational<3>::kEpi) && buf_plan_is<IoRelational<3>>(B)) return "IoRelational<3>";
    if (GT == 2 && three && HT == 3 && kBwd16BufD == 2 && epi_ok(IoRelationalF2::kEpi) && buf_plan_is<IoRelationalF2>(B)) return "IoRelationalF2";
    if (GT == 2 && three && HT == 3 && kBwd16BufD == 2 && epi_ok(IoRelationalF3::kEpi) && buf_plan_is<IoRelationalF3>(B)) return "IoRelationalF3";
    if (GT == 1 && three && epi_ok(IoObject::kEpi) && buf_plan_is<IoObject>(B)) return "IoObject";
    if (GT == 0 && !three && epi_ok(IoEncoder8<1>::kEpi) && buf_plan_is<IoEncoder8<1>>(B)) return "IoEncoder8<1>";
    if (GT == 0 && !three && epi_ok(IoEncoder8<2>::kEpi) && buf_plan_is<IoEncoder8<2>>(B)) return "IoEncoder8<2>";
    return "";
}

// launches the backward instantiation for (plan, GT, three); G32 = fp32 upstream gradient
// (grid: workgroups of kWaves waves; grid_buf: of kBwd16BufWaves waves - what the buffer-addressed kernels take;
// *used = {workgroups, waves per workgroup} of the launch: grid * waves partial blocks unless reduced in LDS)
template <bool G32>
int launch_bwd16(const gnntrk_mlp_bwd_args *a, const SlotPlan &P, int GT, int grid, int grid_buf, int *used,
                 float *part, uint8_t *trash, hipStream_t stream) {
    used[0] = grid;
    used[1] = kWaves;
    const bool three = a->mlp.n_layers == 3;
    bool launched = false;
    BufPlan B;
    make_buf_plan(B, P, a, GT);
    if (a->mlp.out_dim > 16 || P.KI > 2) {   // output tiles / wide inputs (three hidden tiles; GT = 2 KI)
        if (G32 || a->epilogue == GNNTRK_EPI_RELU || a->epilogue == GNNTRK_EPI_SIGMOID)
            return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: outputs over 16 / inputs over 64 slots take the NONE and RESIDUAL epilogues");
        const int ot = (a->mlp.out_dim + 15) / 16;
#define GNNTRK_BWD16_OT(KI_, OT_)                                                              \
    if (!launched && P.KI == KI_ && ot == OT_ && P.HT == 3 && GT == 2 * KI_) {                 \
        if (three) { auto kfn = mlp16_bwd_ot_kernel<KI_, 3, OT_, true>;                        \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B); } \
        else { auto kfn = mlp16_bwd_ot_kernel<KI_, 3, OT_, false>;                             \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B); } \
        launched = true;                                                                       \
    }
        if constexpr (!G32) {
            GNNTRK_BWD16_OT(1, 2) GNNTRK_BWD16_OT(1, 3) GNNTRK_BWD16_OT(2, 2) GNNTRK_BWD16_OT(2, 3)
            GNNTRK_BWD16_OT(3, 1) GNNTRK_BWD16_OT(3, 2) GNNTRK_BWD16_OT(3, 3)
            GNNTRK_BWD16_OT(4, 1) GNNTRK_BWD16_OT(4, 2) GNNTRK_BWD16_OT(4, 3)
        }
#undef GNNTRK_BWD16_OT
        if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: no instantiation (output tiles)");
        return check_launch("mlp_backward_bf16");
    }
    if (P.bias_init && P.HT == 8)   // hidden width 128: own translation unit (mlp_bf16_bi8.hip says why)
        return launch_bwd16_bi8(a, P, GT, G32 ? 1 : 0, grid, part, trash, stream);
    if (P.bias_init) {   // hidden width 64 with biases: the accumulator-initialised kernels
#define GNNTRK_BWD16_BI(KI_, HT_, GT_)                                                         \
    if (!launched && P.KI == KI_ && P.HT == HT_ && GT == GT_) {                                \
        if (three) { auto kfn = mlp16_bwd_bi_kernel<KI_, HT_, GT_, true, G32>;                 \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B); } \
        else { auto kfn = mlp16_bwd_bi_kernel<KI_, HT_, GT_, false, G32>;                      \
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B); } \
        launched = true;                                                                       \
    }
        GNNTRK_BWD16_BI(1, 4, 0) GNNTRK_BWD16_BI(1, 4, 1) GNNTRK_BWD16_BI(1, 4, 2)
        GNNTRK_BWD16_BI(2, 4, 1) GNNTRK_BWD16_BI(2, 4, 4)
#undef GNNTRK_BWD16_BI
        if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: no instantiation (bias_init)");
        return check_launch("mlp_backward_bf16");
    }
    // the shapes of the default models go through buffer descriptors (debug_flags & 128: generic I/O)
    {
        const char *io = buf_io_name(B, P.KI, P.HT, GT, three, G32, a->debug_flags, a->epilogue);
        if (a->debug_flags & 256)   // (diagnostics: which I/O form a launch takes)
            fprintf(stderr, "mlp_backward_bf16: KI %d HT %d GT %d three %d rows %lld plan ok %d (loads %d ids %d+%d gout %d stores %d ones %d) -> %s\n",
                    P.KI, P.HT, GT, (int)three, (long long)a->n_rows, B.ok, B.n_load, B.n_ids, B.n_sids, B.n_gout,
                    B.n_store, B.ones_dword, io[0] ? io : "generic");
#define GNNTRK_BWD16_SKEL(IO_)                                                                        \
    if (!launched && (a->debug_flags & 4096) && P.HT == 3 && kBwd16BufD == 2 && strcmp(io, #IO_) == 0) { \
        auto kfn = mlp16_bwd_skel_kernel<1, 3, 2, true, G32, 2, IO_>;                                 \
        hipLaunchKernelGGL(kfn, dim3(grid_buf), dim3(64 * kBwd16BufWaves), 0, stream, *a, part, trash, B); \
        used[0] = grid_buf;                                                                           \
        used[1] = kBwd16BufWaves;                                                                     \
        launched = true;                                                                              \
    }
        if constexpr (G32) {
            GNNTRK_BWD16_SKEL(IoHead)
            GNNTRK_BWD16_SKEL(IoHeadF)
        } else {
            GNNTRK_BWD16_SKEL(IoRelational<2>)
            GNNTRK_BWD16_SKEL(IoRelational<3>)
            GNNTRK_BWD16_SKEL(IoRelationalF2)
            GNNTRK_BWD16_SKEL(IoRelationalF3)
        }
#undef GNNTRK_BWD16_SKEL
#define GNNTRK_BWD16_BUF(HT_, GT_, T_, IO_)                                                           \
    if (!launched && P.HT == HT_ && strcmp(io, #IO_) == 0) {                                          \
        auto kfn = mlp16_bwd_kernel<1, HT_, GT_, T_, G32, kBwd16BufD, IO_>;                           \
        hipLaunchKernelGGL(kfn, dim3(grid_buf), dim3(64 * kBwd16BufWaves), 0, stream, *a, part, trash, B); \
        used[0] = grid_buf;                                                                           \
        used[1] = kBwd16BufWaves;                                                                     \
        launched = true;                                                                              \
    }
        if constexpr (G32) {
            GNNTRK_BWD16_BUF(3, 2, true, IoHead)
            GNNTRK_BWD16_BUF(1, 2, true, IoHead)
            GNNTRK_BWD16_BUF(3, 2, true, IoHeadF)
        } else {
            GNNTRK_BWD16_BUF(3, 2, true, IoRelational<2>)
            GNNTRK_BWD16_BUF(3, 2, true, IoRelational<3>)
            GNNTRK_BWD16_BUF(3, 2, true, IoRelationalF2)
            GNNTRK_BWD16_BUF(3, 2, true, IoRelationalF3)
            GNNTRK_BWD16_BUF(1, 2, true, IoRelational<2>)
            GNNTRK_BWD16_BUF(1, 2, true, IoRelational<3>)
            GNNTRK_BWD16_BUF(3, 1, true, IoObject)
            GNNTRK_BWD16_BUF(1, 1, true, IoObject)
            GNNTRK_BWD16_BUF(3, 0, false, IoEncoder8<1>)
            GNNTRK_BWD16_BUF(3, 0, false, IoEncoder8<2>)
            GNNTRK_BWD16_BUF(1, 0, false, IoEncoder8<1>)
            GNNTRK_BWD16_BUF(1, 0, false, IoEncoder8<2>)
        }
#undef GNNTRK_BWD16_BUF
        if (launched) return check_launch("mlp_backward_bf16");
    }
    if (a->fold.ids) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: this launch does not take a fold (gnntrk_mlp_backward_bf16_can_fold)");
// D = 2 (two 16-row halves per iteration, K = 32 weight-gradient contractions) wherever the
// doubled staging images fit the workgroup's LDS budget: one k-step, up to three hidden tiles -
// every shape of the reference's default models.  debug_flags & 64 forces D = 1 (A/B timing).
#define GNNTRK_BWD16_LAUNCH(KI_, HT_, GT_, T_, D_)                                             \
    {                                                                                          \
        auto kfn = mlp16_bwd_kernel<KI_, HT_, GT_, T_, G32, D_>;                               \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kBlock), 0, stream, *a, part, trash, B);      \
    }
#define GNNTRK_BWD16_CASE(KI_, HT_, GT_)                                                       \
    if (P.KI == KI_ && P.HT == HT_ && GT == GT_) {                                             \
        constexpr bool kTwo = (KI_ == 1 && HT_ <= 3);                                          \
        if (kTwo && !(a->debug_flags & 64)) {                                                  \
            if (three) GNNTRK_BWD16_LAUNCH(KI_, HT_, GT_, true, (kTwo ? 2 : 1))                \
            else GNNTRK_BWD16_LAUNCH(KI_, HT_, GT_, false, (kTwo ? 2 : 1))                     \
        } else {                                                                               \
            if (three) GNNTRK_BWD16_LAUNCH(KI_, HT_, GT_, true, 1)                             \
            else GNNTRK_BWD16_LAUNCH(KI_, HT_, GT_, false, 1)                                  \
        }                                                                                      \
        launched = true;                                                                       \
    }
#define GNNTRK_BWD16_HT(KI_, GT_) \
    GNNTRK_BWD16_CASE(KI_, 1, GT_) GNNTRK_BWD16_CASE(KI_, 2, GT_) GNNTRK_BWD16_CASE(KI_, 3, GT_) \
        GNNTRK_BWD16_CASE(KI_, 4, GT_) GNNTRK_BWD16_CASE(KI_, 5, GT_) GNNTRK_BWD16_CASE(KI_, 6, GT_)
    GNNTRK_BWD16_HT(1, 0)   // weight gradients only (the encoders of raw dataset features)
    GNNTRK_BWD16_HT(1, 1)
    GNNTRK_BWD16_HT(1, 2)
    GNNTRK_BWD16_CASE(1, 7, 0) GNNTRK_BWD16_CASE(1, 8, 0)   // hidden widths 96 .. 127: one k-step of inputs
    GNNTRK_BWD16_CASE(1, 7, 1) GNNTRK_BWD16_CASE(1, 8, 1)
    GNNTRK_BWD16_CASE(1, 7, 2) GNNTRK_BWD16_CASE(1, 8, 2)
    GNNTRK_BWD16_HT(2, 1)
    GNNTRK_BWD16_HT(2, 4)
#undef GNNTRK_BWD16_HT
#undef GNNTRK_BWD16_CASE
#undef GNNTRK_BWD16_LAUNCH
    if (!launched) return fail(GNNTRK_EUNSUPPORTED, "mlp_backward_bf16: no instantiation");
    return check_launch("mlp_backward_bf16");
}

}  // namespace
}  // namespace gnntrk
