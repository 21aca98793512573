// bf16-storage / fp32-accumulate building blocks of the fused gather-MLP kernels
// (gfx950 / CDNA4): BASELINE config 3/4 ("bf16 storage for x, e, e~, aggr and the MFMA
// inputs, fp32 accumulate").
//
// Orientation is the one of tile_mlp.h - activations TRANSPOSED, features x rows, in the
// accumulator layout  D[4g + r][c]  (lane l = 16 g + c)  - but the contraction runs on
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16:
//     A[i = l&15][k = K/4 * (l>>4) + e],   B[k = K/4 * (l>>4) + e][j = l&15],
// e = 0..7 (K = 32, four VGPRs) or 0..3 (K = 16, two VGPRs).  Two D tiles of layer n,
// rounded to bf16 and packed pairwise, ARE one K = 32 B operand of layer n+1: k-position
// (g, e) holds feature 16(2u) + 4g + e for e < 4 and 16(2u+1) + 4g + e-4 for e >= 4; an odd
// last tile feeds a K = 16 step.  The weight (A) fragments are packed with the same
// permutation of k, so nothing moves between lanes in the forward or the dX chain.
// The K = rows contractions of the weight gradients take both operands through an LDS
// image [row][feature] and ds_read_b64_tr_b16 (hardware 4x4 transpose read).
//
// Layouts and lane maps were checked on an MI355X against a host model
// (tools/probe_bf16.hip): MFMA k-maps as above, v_cvt_pk_bf16_f32 = round-to-nearest-even,
// v_pk_max_i16(x, 0) = ReLU on packed bf16, transpose read as lds_read_tr16 documents.
#pragma once

#include "tile_mlp.h"

namespace gnntrk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const uint16_t GNNTRK_GLOBAL *gch_ptr;  // bf16 storage
typedef uint16_t GNNTRK_GLOBAL *gh_ptr;

// ---- hardware primitives -----------------------------------------------------------
// (tests/emul/shim/hip/hip_runtime.h defines GNNTRK_BF16_PRIMITIVES and host models of
// the same functions for the CPU wave emulator.)
#ifndef GNNTRK_BF16_PRIMITIVES
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef short i16x4_hw __attribute__((ext_vector_type(4)));
typedef short i16x2_hw __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 (lo in bits 0..15), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t bf16x2_pack(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
// per 16-bit half, as signed integers: v_pk_max_i16 / v_pk_min_i16 / v_pk_mul_lo_u16
__device__ __forceinline__ uint32_t i16x2_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_hw, a),
                                                                  __builtin_bit_cast(i16x2_hw, b)));
}
__device__ __forceinline__ uint32_t i16x2_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(i16x2_hw, a),
                                                                  __builtin_bit_cast(i16x2_hw, b)));
}
__device__ __forceinline__ uint32_t u16x2_mul(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t,
                              __builtin_bit_cast(u16x2_hw, a) * __builtin_bit_cast(u16x2_hw, b));
}
__device__ __forceinline__ f32x4 mfma_bf16_k32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                                   __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16_k16(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(i16x4_hw, a),
                                                     __builtin_bit_cast(i16x4_hw, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16.  Every lane passes the LDS address of four consecutive bf16
// (8-byte aligned).  Inside each 16-lane group the 16 x 4 elements form a 4 x 16 block
// (lane p supplies row p>>2, columns 4(p&3)..+3); lane p receives column p of the four
// rows.  With an image X[row][16 features] and lane (g, p) pointing at
// X[4g + (p>>2)][4(p&3)], lane (g, p) gets X[4g + 0..3][p]: the K = rows operand.
__device__ __forceinline__ u32x2 lds_read_tr16(const uint16_t *p) {
    typedef __attribute__((address_space(3))) i16x4_hw *lds_ptr;
    const i16x4_hw v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_ptr)(__attribute__((address_space(3))) void *)p);
    return __builtin_bit_cast(u32x2, v);
}
// An offset of zero the optimiser cannot see through.  The backward kernel re-reads its
// weight fragments from LDS in every tile (about 25 ds_read_b128): hoisted out of the tile
// loop they would take 70+ registers next to the 72 weight-gradient accumulators.
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
// A value the optimiser cannot see through (no instruction is emitted).  gate_bf16x2 takes
// its constant 0x00010001 through it: knowing the constant, the optimiser expands
// "x * min(p, 1)" into five compares / selects / permutes per register instead of
// v_pk_min_i16 + v_pk_mul_lo_u16.
__device__ __forceinline__ uint32_t opaque_u32(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
// acc += A * B (K = 16) for the weight-gradient accumulators that live across the whole
// tile loop.  They are updated in place as long as the tile loop is a uniform counted loop
// (DESIGN.md 4.3): in a divergent loop every accumulator is copied once per iteration.
__device__ __forceinline__ void mfma_bf16_k16_acc(u32x2 a, u32x2 b, f32x4 &acc) {
    acc = mfma_bf16_k16(a, b, acc);
}
__device__ __forceinline__ void drain_mfma() {}

// ---- buffer addressing (wave-uniform 128-bit descriptor in SGPRs + 32-bit lane offset) ----------
// raw buffer (stride 0): byte offset = voffset + soffset; an access whose offset is not below
// num_records returns 0 (loads) / is dropped (stores) - per lane, soffset included (checked on an
// MI355X: tools/probe_buf.hip).  The tile loops lean on exactly that: lanes that take no part in an
// access carry the offset 2^31 (tensors are below 2^31 bytes), rows past the end of a tensor fall
// out by themselves - no per-lane pointers, no masks, no store redirection.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t buf_make(const void *base, uint32_t num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)num_bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_load_u32(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ u32x2 buf_load_u32x2(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x4 buf_load_u32x4(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store_u32(uint32_t v, buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store_u32x2(u32x2 v, buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    typedef unsigned int v2u_hw __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_hw, v), r, (int)voff, (int)soff, 0);
}
#endif  // GNNTRK_BF16_PRIMITIVES

__device__ __forceinline__ float bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) { return i16x2_max(v, 0u); }
// x where the matching half of p is a positive bf16 (p is a ReLU output: >= +0), else 0:
// x * min(p, 1) per 16-bit half as integers; `one` = opaque_u32(0x00010001)
__device__ __forceinline__ uint32_t gate_bf16x2(uint32_t x, uint32_t p, uint32_t one) {
    return u16x2_mul(x, i16x2_min(p, one));
}
__device__ __forceinline__ u32x2 pack_tile(const f32x4 &v) {
    u32x2 r;
    r[0] = bf16x2_pack(v[0], v[1]);
    r[1] = bf16x2_pack(v[2], v[3]);
    return r;
}
__device__ __forceinline__ u32x2 pack_tile_relu(const f32x4 &v) {
    u32x2 r = pack_tile(v);
    r[0] = relu_bf16x2(r[0]);
    r[1] = relu_bf16x2(r[1]);
    return r;
}
__device__ __forceinline__ u32x4 join(const u32x2 &a, const u32x2 &b) {
    u32x4 r;
    r[0] = a[0];
    r[1] = a[1];
    r[2] = b[0];
    r[3] = b[1];
    return r;
}
__device__ __forceinline__ uint16_t bf16_bits(float x) {  // RNE (weight packing, prologue only)
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---- slot plan ------------------------------------------------------------------------
// The layer-1 contraction dimension is KI k-steps of 32 "slots".  Every input segment
// contributes ceil(dim/4) chunks of four slots (one 8-byte bf16 load per lane and chunk);
// chunk p sits in k-step p/8, lane group (p%8)/2, half p%2, so lane (g, c) loads chunks
// 8kk + 2g + {0,1} of row c.  Pad slots are forced to zero after the load, one of them is
// forced to 1.0 when any layer has a bias: the biases ride in the weight fragments
// (column `ones_slot` of W1', column `hid_ones` of W2'/W3'; row `hid_ones` of W1'/W2'
// regenerates the constant), rounded to bf16 like autocast rounds them.
constexpr int kMaxChunks16 = 32;   // four k-steps of 32 slots (more than two: the output-tile kernels, three hidden tiles)
// hidden tiles of 16 (hidden features + the constant-one row): up to 4 in the hot instantiations
// (two workgroups per CU), 5 and 6 - hidden widths 64 .. 95 - in plain ones (one tile per iteration,
// one workgroup per CU: their weight-gradient accumulators alone take up to 264 registers)
constexpr int kMaxOut16 = 48;          // output features: up to three output tiles (with three hidden tiles)
constexpr int kMaxHiddenTiles16 = 8;   // (seven and eight: one 32-slot k-step of inputs only)

struct SlotPlan {
    int32_t n_chunks;   // chunks in use (incl. a ones-only chunk)
    int32_t KI;         // 32-slot k-steps: 1 or 2
    int32_t ones_slot;  // -1: no bias anywhere
    int32_t HT;         // hidden tiles (hidden + ones row)
    int32_t hid_ones;   // hidden feature index of the constant-one row, -1: none
    int32_t n_gchunks;  // chunks whose gradient is wanted (backward)
    int32_t GT;         // gradient M tiles = ceil(n_gchunks / 4)
    int32_t ok;         // 0: shape not supported
    int32_t bias_init;  // 1: no constant-one hidden row - the biases of the layers after the first are the
                        //    initial values of their accumulators (hidden widths 64 / 128, see make_slot_plan)
    int8_t seg[kMaxChunks16];    // segment of the chunk, -1: none
    int8_t first[kMaxChunks16];  // first feature of the chunk inside the segment / 4
    int8_t gchunk[kMaxChunks16]; // gradient chunk q -> input chunk
    int16_t colbase[GNNTRK_MAX_SEGS];  // first W1 column of the segment
};

__host__ __device__ inline void make_slot_plan(SlotPlan &P, const gnntrk_mlp &m, int n_seg,
                                               const gnntrk_seg *seg, const gnntrk_gseg *gseg) {
    P.ok = 1;
    int n = 0, col = 0;
    for (int p = 0; p < kMaxChunks16; ++p) {
        P.seg[p] = -1;
        P.first[p] = 0;
        P.gchunk[p] = -1;
    }
    for (int j = 0; j < GNNTRK_MAX_SEGS; ++j) P.colbase[j] = 0;
    for (int j = 0; j < n_seg; ++j) {
        P.colbase[j] = (int16_t)col;
        col += seg[j].dim;
        for (int ch = 0; 4 * ch < seg[j].dim; ++ch, ++n) {
            if (n >= kMaxChunks16) {
                P.ok = 0;
                return;
            }
            P.seg[n] = (int8_t)j;
            P.first[n] = (int8_t)ch;
        }
    }
    const bool bias = m.b[0] != nullptr || m.b[1] != nullptr || (m.n_layers == 3 && m.b[2] != nullptr);
    P.ones_slot = -1;
    if (bias) {
        for (int p = 0; p < n && P.ones_slot < 0; ++p) {
            const int d = seg[P.seg[p]].dim - 4 * P.first[p];
            if (d < 4) P.ones_slot = 4 * p + d;
        }
        if (P.ones_slot < 0) {
            if (n >= kMaxChunks16) {
                P.ok = 0;
                return;
            }
            P.ones_slot = 4 * n;
            ++n;
        }
    }
    P.n_chunks = n;
    P.KI = (n + 7) / 8;
    const bool hid_bias = bias && (m.b[1] != nullptr || (m.n_layers == 3 && m.b[2] != nullptr));
    // Hidden width 64 - the power of two people pick - has no spare row in its last hidden tile: a
    // constant-one row costs a whole tile (five instead of four: one workgroup per CU instead of two).
    // There the biases of the layers after the first enter as the INITIAL VALUES of the fp32 accumulators
    // (rounded to bf16 first, as the fragment form rounds them), and their gradients are one extra MFMA per
    // gradient tile against a tile of ones.  128 takes eight tiles this way instead of nine (which no wave
    // has the registers for) where the inputs fit one k-step; its kernels live in mlp_bf16_bi8.hip.
    P.bias_init = (hid_bias && (m.hidden == 64 || (m.hidden == 128 && P.KI == 1))) ? 1 : 0;
    P.hid_ones = (hid_bias && !P.bias_init) ? m.hidden : -1;
    P.HT = (m.hidden + (P.hid_ones >= 0 ? 1 : 0) + 15) / 16;
    int q = 0;
    if (gseg)
        for (int p = 0; p < n; ++p)
            if (P.seg[p] >= 0 && gseg[P.seg[p]].ptr) P.gchunk[q++] = (int8_t)p;
    P.n_gchunks = q;
    P.GT = (q + 3) / 4;
    if (P.HT > kMaxHiddenTiles16 || (P.HT > 6 && P.KI > 1) || m.out_dim > kMaxOut16) P.ok = 0;
    // outputs over 16 features (two / three output tiles) and inputs over 64 slots: three hidden tiles, ones-row
    // biases, bf16 output - the shapes of GraphConstructionResIN(hidden_dim=40)
    if ((m.out_dim > 16 || P.KI > 2) && (P.HT != 3 || P.bias_init)) P.ok = 0;
}

// W1 column of input slot s: >= 0 column, -1 pad, -2 the ones slot
__host__ __device__ inline int slot_col(const SlotPlan &P, const gnntrk_seg *seg, int s) {
    if (s == P.ones_slot) return -2;
    const int p = s >> 2;
    if (p >= P.n_chunks || P.seg[p] < 0) return -1;
    const int f = 4 * P.first[p] + (s & 3);
    return f < seg[P.seg[p]].dim ? P.colbase[P.seg[p]] + f : -1;
}

// hidden feature at k-position (g, e) of K = 32 pair u / of the odd last tile (K = 16)
__host__ __device__ inline int hid_feat_k32(int u, int g, int e) {
    return e < 4 ? 32 * u + 4 * g + e : 32 * u + 16 + 4 * g + (e - 4);
}
__host__ __device__ inline int hid_feat_k16(int t, int g, int e) { return 16 * t + 4 * g + e; }

// Augmented weights (see SlotPlan): value of W1'[o][slot], Wmid'[o][f], Wlast'[o][f].
struct AugWeights {
    const float *W1, *b1, *W2, *b2, *W3, *b3;  // W2/b2 = middle layer (NULL for 2 layers)
    int in_dim, hidden, out_dim, hid_ones;
    __device__ __forceinline__ float w1(int o, int col) const {  // col from slot_col
        if (o < hidden) {
            if (col >= 0) return W1[o * in_dim + col];
            if (col == -2 && b1) return b1[o];
            return 0.f;
        }
        return (o == hid_ones && col == -2) ? 1.f : 0.f;
    }
    __device__ __forceinline__ float wmid(int o, int f) const {
        if (o < hidden) {
            if (f < hidden) return W2[o * hidden + f];
            return (f == hid_ones && b2) ? b2[o] : 0.f;
        }
        return (o == hid_ones && f == hid_ones) ? 1.f : 0.f;
    }
    __device__ __forceinline__ float wlast(int o, int f) const {
        if (o >= out_dim) return 0.f;
        if (f < hidden) return W3[o * hidden + f];
        return (f == hid_ones && b3) ? b3[o] : 0.f;
    }
};
__device__ inline AugWeights make_aug(const gnntrk_mlp &m, const SlotPlan &P) {
    AugWeights w;
    const bool three = m.n_layers == 3;
    w.W1 = m.W[0];
    w.b1 = m.b[0];
    w.W2 = three ? m.W[1] : nullptr;
    w.b2 = three ? m.b[1] : nullptr;
    w.W3 = three ? m.W[2] : m.W[1];
    w.b3 = three ? m.b[2] : m.b[1];
    w.in_dim = m.in_dim;
    w.hidden = m.hidden;
    w.out_dim = m.out_dim;
    w.hid_ones = P.hid_ones;
    return w;
}

// Fragment images in LDS: K = 32 fragment = 64 lanes x 4 dwords, K = 16 = 64 x 2 dwords.
// `val(i, g, e)` is the matrix element for fragment row i and k-position (g, e).
template <class F>
__device__ __forceinline__ void pack_frag_k32(uint32_t *dst, F val, int tid, int nthreads) {
    for (int t = tid; t < 256; t += nthreads) {
        const int l = t >> 2, d = t & 3, i = l & 15, g = l >> 4;
        dst[t] = (uint32_t)bf16_bits(val(i, g, 2 * d)) | ((uint32_t)bf16_bits(val(i, g, 2 * d + 1)) << 16);
    }
}
template <class F>
__device__ __forceinline__ void pack_frag_k16(uint32_t *dst, F val, int tid, int nthreads) {
    for (int t = tid; t < 128; t += nthreads) {
        const int l = t >> 1, d = t & 1, i = l & 15, g = l >> 4;
        dst[t] = (uint32_t)bf16_bits(val(i, g, 2 * d)) | ((uint32_t)bf16_bits(val(i, g, 2 * d + 1)) << 16);
    }
}
__device__ __forceinline__ u32x4 frag_k32(const uint32_t *img, int lane) {
    return *reinterpret_cast<const u32x4 *>(img + 4 * lane);
}
__device__ __forceinline__ u32x2 frag_k16(const uint32_t *img, int lane) {
    return *reinterpret_cast<const u32x2 *>(img + 2 * lane);
}

// dwords of the K-side fragments of ONE M tile contracting over HT hidden tiles.  An odd
// last tile is contracted with a K = 32 step whose upper half is zero (fragment and
// operand): accumulating a K = 16 MFMA onto the result of a K = 32 one (different pass
// counts, SrcC = previous vDst) returned wrong sums on gfx950 with ROCm 7.2's hazard
// handling, so one accumulation chain never mixes MFMA opcodes.
__host__ __device__ constexpr int hid_k_dwords(int HT) { return ((HT + 1) / 2) * 256; }

// acc += A(row tile) * B(hidden activations packed as HT tiles of u32x2)
template <int HT>
__device__ __forceinline__ f32x4 contract_hidden(const uint32_t *img, const u32x2 (&P)[HT], int lane,
                                                 f32x4 acc) {
#pragma unroll
    for (int u = 0; u < HT / 2; ++u)
        acc = mfma_bf16_k32(frag_k32(img + 256 * u, lane), join(P[2 * u], P[2 * u + 1]), acc);
    if (HT % 2) {
        const u32x2 zero = {0u, 0u};
        acc = mfma_bf16_k32(frag_k32(img + 256 * (HT / 2), lane), join(P[HT - 1], zero), acc);
    }
    return acc;
}

}  // namespace gnntrk
