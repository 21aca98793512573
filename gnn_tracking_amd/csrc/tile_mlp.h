// Device-side building blocks of the fused gather-MLP kernels (gfx950 / CDNA4).
//
// Orientation: every activation tile is held TRANSPOSED, features x rows, in the
// accumulator layout of v_mfma_f32_16x16x4_f32:
//     lane l = 16*g + c  (g = l>>4, c = l&15), register r in 0..3
//     D[4g + r][c]   <->   feature feat_of(map, tile, g, r)  of  row (tile_base + c)
// With that convention the D registers of layer n ARE the B operand of layer n+1
// (B[k = g][j = c]): k-step (tile t, reg r) contracts the four features
// {feat_of(map,t,g,r) : g=0..3}.  The k order is a permutation of the features, the
// weight (A) fragments are packed with the same permutation, so nothing ever moves
// between lanes in the forward chain.  fp32 MFMA is bit-exact an fmaf chain
// (cdna_hip_programming.md section 3), so results differ from the torch reference
// only by summation order.
//
// A feature dimension D is split into q = D/16 full tiles (feature 16t+4g+r) and one
// remainder tile that is packed "R registers deep": feature 16q + R*g + r, r < R,
// R = ceil((D%16)/4).  That keeps the number of k-steps at ceil(D/4) (10 for the
// default hidden width 40) instead of 4*ceil(D/16) (12).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gnntrk.h"

namespace gnntrk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;      // threads per workgroup (4 waves, one per SIMD)
constexpr int kWaves = 4;        // waves per workgroup
constexpr int kTileRows = 16;    // rows (edges / nodes) per MFMA tile
constexpr int kTbLd = 20;        // leading dim (floats) of the transpose buffer
constexpr int kTbRows = 64;      // max features staged at once

struct DimMap {
    int D;   // features
    int q;   // full 16-feature tiles
    int R;   // registers used in the remainder tile (0 if none)
    int nt;  // tiles
    int ks;  // k-steps = 4q + R
};

__host__ __device__ inline DimMap make_dimmap(int D) {
    DimMap m;
    m.D = D;
    m.q = D / 16;
    const int rem = D % 16;
    m.R = (rem + 3) / 4;
    m.nt = m.q + (rem ? 1 : 0);
    m.ks = 4 * m.q + m.R;
    return m;
}

// feature held by lane-group g, register r of tile t; -1 = padding
__host__ __device__ inline int feat_of(const DimMap &m, int t, int g, int r) {
    if (t < m.q) return 16 * t + 4 * g + r;
    if (t == m.q && r < m.R) {
        const int f = m.R * g + r;
        return (f < m.D - 16 * m.q) ? 16 * m.q + f : -1;
    }
    return -1;
}
// is (tile t, register r) a k-step of the map?  (wave-uniform)
__host__ __device__ inline bool kvalid(const DimMap &m, int t, int r) {
    return t < m.q || (t == m.q && r < m.R);
}
__host__ __device__ inline int kindex(const DimMap &m, int t, int r) {
    return t < m.q ? 4 * t + r : 4 * m.q + r;
}
__host__ __device__ inline void kstep_tr(const DimMap &m, int ks, int &t, int &r) {
    if (ks < 4 * m.q) {
        t = ks >> 2;
        r = ks & 3;
    } else {
        t = m.q;
        r = ks - 4 * m.q;
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Order the LDS traffic of ONE wave (write by some lanes, read by others).  LDS ops
// of a wave execute in order; the fences stop the compiler from reordering them and
// make it wait for the writes (lgkmcnt) before the dependent reads are issued.
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A-operand fragments of a matrix Mat[rowfeat][kfeat] held in nn.Linear storage
// W[out][in] (ld = in_dim):  forward layer: Mat = W  (rows = out, k = in);
// transposed (backward dX): Mat = W^T (rows = in, k = out).
// dst[(to * kmap.ks + ks) * 64 + lane];  lane (g,c) holds
//     Mat[feat_of(rowmap,to,c>>2,c&3)][feat_of(kmap,tk,g,rk)],  (tk,rk) = k-step ks.
__device__ inline void fill_frags(float *dst, const float *W, int ld, const DimMap rowmap,
                                  const DimMap kmap, bool transposed, int tid, int nthreads) {
    const int n = rowmap.nt * kmap.ks * 64;
    for (int i = tid; i < n; i += nthreads) {
        const int fr = i >> 6, l = i & 63;
        const int to = fr / kmap.ks, ks = fr - to * kmap.ks;
        int tk, rk;
        kstep_tr(kmap, ks, tk, rk);
        const int g = l >> 4, c = l & 15;
        const int rf = feat_of(rowmap, to, c >> 2, c & 3);
        const int kf = feat_of(kmap, tk, g, rk);
        float v = 0.f;
        if (rf >= 0 && kf >= 0) v = transposed ? W[(int64_t)kf * ld + rf] : W[(int64_t)rf * ld + kf];
        dst[i] = v;
    }
}

// bias in accumulator layout: dst[t*64 + lane][r] = b[feat_of(map,t,g,r)] (0 if pad / no bias)
__device__ inline void fill_bias(f32x4 *dst, const float *b, const DimMap map, int tid,
                                 int nthreads) {
    for (int i = tid; i < map.nt * 64; i += nthreads) {
        const int t = i >> 6, l = i & 63, g = l >> 4;
        f32x4 v;
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(map, t, g, r);
            v[r] = (b != nullptr && f >= 0) ? b[f] : 0.f;
        }
        dst[i] = v;
    }
}

struct Maps {
    DimMap in, hid, out;
    bool three;  // 3 linear layers (else 2)
};

// XCD-aware persistent tile schedule: block b runs on XCD b % 8 (observed dispatch
// order, MI355X_MICROARCH.md "Workgroup dispatch"); give every XCD one contiguous
// range of tiles so the node rows gathered by neighbouring tiles stay in ITS L2.
// Correctness does not depend on the placement.
struct TileSched {
    int64_t cur, end, step;
};
__device__ inline TileSched make_sched(int64_t n_tiles) {
    const int G = gridDim.x;
    const int nx = (G % 8 == 0) ? 8 : 1;
    const int x = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int64_t t0 = n_tiles * x / nx, t1 = n_tiles * (x + 1) / nx;
    TileSched s;
    s.cur = t0 + lb * kWaves + (threadIdx.x >> 6);
    s.end = t1;
    s.step = (int64_t)bpx * kWaves;
    return s;
}

// one concatenated-input feature slot of a lane: where to read it from
struct InSlot {
    const float *base;   // segment ptr + feature offset; nullptr = padding
    const int32_t *idx;  // row gather index or nullptr
    int32_t stride;
};

// Segment descriptors staged in LDS (kernel arguments cannot be indexed dynamically
// without a scratch copy).  Filled by stage_segs() before the first __syncthreads().
struct SegTable {
    const float *ptr[GNNTRK_MAX_SEGS];
    const int32_t *idx[GNNTRK_MAX_SEGS];
    float *gptr[GNNTRK_MAX_SEGS];
    const int32_t *gidx[GNNTRK_MAX_SEGS];
    int32_t dim[GNNTRK_MAX_SEGS];
    int32_t stride[GNNTRK_MAX_SEGS];
    int32_t relu[GNNTRK_MAX_SEGS];
    int32_t gstride[GNNTRK_MAX_SEGS];
    int32_t gacc[GNNTRK_MAX_SEGS];
};

__device__ __forceinline__ void stage_segs(SegTable &tab, const gnntrk_seg (&seg)[GNNTRK_MAX_SEGS],
                                           const gnntrk_gseg *gseg, int n_seg, int tid) {
#pragma unroll
    for (int j = 0; j < GNNTRK_MAX_SEGS; ++j)
        if (tid == j) {
            const bool on = j < n_seg;
            tab.ptr[j] = on ? seg[j].ptr : nullptr;
            tab.idx[j] = on ? seg[j].idx : nullptr;
            tab.dim[j] = on ? seg[j].dim : 0;
            tab.stride[j] = on ? seg[j].stride : 0;
            tab.relu[j] = on ? seg[j].relu : 0;
            tab.gptr[j] = (on && gseg) ? gseg[j].ptr : nullptr;
            tab.gidx[j] = (on && gseg) ? gseg[j].idx : nullptr;
            tab.gstride[j] = (on && gseg) ? gseg[j].stride : 0;
            tab.gacc[j] = (on && gseg) ? gseg[j].accumulate : 0;
        }
}

template <int KT>
__device__ __forceinline__ void setup_in_slots(const SegTable &tab, int n_seg, const DimMap &in,
                                               int g, InSlot (&slot)[KT * 4], unsigned &relu_bits) {
    relu_bits = 0;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            InSlot s;
            s.base = nullptr;
            s.idx = nullptr;
            s.stride = 0;
            const int f = feat_of(in, t, g, r);
            int off = 0;
            for (int j = 0; j < n_seg; ++j) {
                const int d = tab.dim[j];
                if (f >= off && f < off + d) {
                    s.base = tab.ptr[j] + (f - off);
                    s.idx = tab.idx[j];
                    s.stride = tab.stride[j];
                    if (tab.relu[j]) relu_bits |= 1u << (t * 4 + r);
                }
                off += d;
            }
            slot[t * 4 + r] = s;
        }
}

template <int KT>
__device__ __forceinline__ void load_inputs(const InSlot (&slot)[KT * 4], unsigned relu_bits,
                                            const DimMap &in, int64_t row, bool valid,
                                            f32x4 (&bin)[KT]) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < in.nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const InSlot &s = slot[t * 4 + r];
                float x = 0.f;
                if (valid && s.base != nullptr) {
                    const int64_t rr = s.idx ? (int64_t)s.idx[row] : row;
                    x = s.base[rr * s.stride];
                    if ((relu_bits >> (t * 4 + r)) & 1u) x = fmaxf(x, 0.f);
                }
                v[r] = x;
            }
        }
        bin[t] = v;
    }
}

// Forward of the (2- or 3-layer) MLP on one 16-row tile.  a1/a2 are post-ReLU hidden
// activations, y the linear output (all in accumulator layout).  For two layers a2 is
// not computed and the output layer reads a1.
template <int KT, int HT>
__device__ __forceinline__ void mlp_tile_forward(const Maps &mp, const float *w1, const float *w2,
                                                 const float *w3, const f32x4 *b1, const f32x4 *b2,
                                                 const f32x4 *b3, int lane, const f32x4 (&bin)[KT],
                                                 f32x4 (&a1)[HT], f32x4 (&a2)[HT], f32x4 &y,
                                                 bool want_y) {
    // layer 1: in -> hid
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        if (to < mp.hid.nt)
            a1[to] = b1[to * 64 + lane];
        else
            a1[to] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (kvalid(mp.in, t, r)) {
                const int ks = kindex(mp.in, t, r);
#pragma unroll
                for (int to = 0; to < HT; ++to)
                    if (to < mp.hid.nt)
                        a1[to] = mfma4(w1[(to * mp.in.ks + ks) * 64 + lane], bin[t][r], a1[to]);
            }
#pragma unroll
    for (int to = 0; to < HT; ++to)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[to][r] = fmaxf(a1[to][r], 0.f);

    // layer 2: hid -> hid
    if (mp.three) {
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            if (to < mp.hid.nt)
                a2[to] = b2[to * 64 + lane];
            else
                a2[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kvalid(mp.hid, t, r)) {
                    const int ks = kindex(mp.hid, t, r);
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        if (to < mp.hid.nt)
                            a2[to] = mfma4(w2[(to * mp.hid.ks + ks) * 64 + lane], a1[t][r], a2[to]);
                }
#pragma unroll
        for (int to = 0; to < HT; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) a2[to][r] = fmaxf(a2[to][r], 0.f);
    }

    // output layer: hid -> out (one tile)
    y = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!mp.three) {
#pragma unroll
        for (int to = 0; to < HT; ++to) a2[to] = a1[to];
    }
    if (want_y) {
        y = b3[lane];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kvalid(mp.hid, t, r)) {
                    const int ks = kindex(mp.hid, t, r);
                    y = mfma4(w3[ks * 64 + lane], a2[t][r], y);
                }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

}  // namespace gnntrk
