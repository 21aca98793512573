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
// A feature dimension D is split into q tiles in the plain mapping (feature
// 16t+4g+r) and, when 1 <= D%16 <= 12, one remainder tile packed "R registers deep":
// feature 16q + R*g + r, r < R, R = ceil((D%16)/4).  That keeps the number of k-steps
// at ceil(D/4) (10 for the default hidden width 40) instead of 4*ceil(D/16) (12).
// The k-step count alone fixes (q, R) = (ks/4, ks%4), which is what lets the hot
// shapes be compiled with every loop bound static (StaticDims) - straight-line MFMA
// code the scheduler can pipeline - next to a generic run-time version (DynDims).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gnntrk.h"

namespace gnntrk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers the kernels dereference in the tile loop are tagged as GLOBAL address space:
// generic ("flat") loads count on vmcnt AND lgkmcnt, so every LDS wait would also drain
// the software-prefetched gathers of the next tile.
#define GNNTRK_GLOBAL __attribute__((address_space(1)))
#define GNNTRK_LDS __attribute__((address_space(3)))
typedef const float GNNTRK_GLOBAL *gcf_ptr;
typedef float GNNTRK_GLOBAL *gf_ptr;
typedef const int32_t GNNTRK_GLOBAL *gci_ptr;

constexpr int kBlock = 256;    // threads per workgroup (4 waves, one per SIMD)
constexpr int kWaves = 4;      // waves per workgroup
constexpr int kTileRows = 16;  // rows (edges / nodes) per MFMA tile
constexpr int kTbLd = 20;      // leading dim (floats) of a transpose buffer
constexpr int kTbRows = 64;    // max features staged at once (+1 garbage row for padding)
constexpr int kTbBufs = 3;     // rotating transpose buffers per wave
constexpr int kTbSize = (kTbRows + 1) * kTbLd;

struct DimMap {
    int D;   // features
    int q;   // tiles in the plain mapping
    int R;   // registers of the packed remainder tile (0: none)
    int nt;  // tiles = q + (R ? 1 : 0)
    int ks;  // k-steps = 4q + R
};

__host__ __device__ inline DimMap make_dimmap(int D) {
    DimMap m;
    m.D = D;
    const int rem = D % 16;
    if (rem == 0 || rem > 12) {
        m.q = (D + 15) / 16;
        m.R = 0;
    } else {
        m.q = D / 16;
        m.R = (rem + 3) / 4;
    }
    m.nt = m.q + (m.R ? 1 : 0);
    m.ks = 4 * m.q + m.R;
    return m;
}

// feature held by lane-group g, register r of tile t; -1 = padding
__host__ __device__ inline int feat_of(const DimMap &m, int t, int g, int r) {
    if (t < m.q) {
        const int f = 16 * t + 4 * g + r;
        return f < m.D ? f : -1;
    }
    if (t == m.q && r < m.R) {
        const int f = 16 * m.q + m.R * g + r;
        return f < m.D ? f : -1;
    }
    return -1;
}

// ---- loop-bound policies -------------------------------------------------------
// k-step (t, r) of a dimension with `ks` k-steps has index 4t + r and exists iff
// 4t + r < ks (both mappings).  Tiles: ceil(ks / 4).
struct DynDims {
    static constexpr bool kStatic = false;
    static constexpr int kItems = 0;
    DimMap in, hid, out;
    bool three_;
    __device__ __forceinline__ int in_ks() const { return in.ks; }
    __device__ __forceinline__ int hid_ks() const { return hid.ks; }
    __device__ __forceinline__ int out_ks() const { return out.ks; }
    __device__ __forceinline__ bool three() const { return three_; }
    __device__ __forceinline__ void set_three(bool v) { three_ = v; }
};
template <int KSI, int KSH, int KSO, bool THREE, int NITEMS = 0>
struct StaticDims {
    static constexpr bool kStatic = true;  // host dispatch guarantees the ones-row conditions
    static constexpr int kItems = NITEMS;  // load-list capacity (0: 4*KT+4)
    DimMap in, hid, out;
    __device__ __forceinline__ constexpr int in_ks() const { return KSI; }
    __device__ __forceinline__ constexpr int hid_ks() const { return KSH; }
    __device__ __forceinline__ constexpr int out_ks() const { return KSO; }
    __device__ __forceinline__ constexpr bool three() const { return THREE; }
    __device__ __forceinline__ void set_three(bool) {}
};
template <class Dims>
__device__ __forceinline__ Dims make_dims(const gnntrk_mlp &m) {
    Dims d;
    d.in = make_dimmap(m.in_dim);
    d.hid = make_dimmap(m.hidden);
    d.out = make_dimmap(m.out_dim);
    d.set_three(m.n_layers == 3);
    return d;
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
// The same ordering WITHOUT the drain.  The DS instructions of one wave are executed by the LDS in the
// order they were issued, so a read (transposed reads included) issued after a write of the same wave sees
// the written data - whichever lane wrote it - and a write issued after a read cannot overtake it.  What is
// left to enforce is that the COMPILER keeps the DS instructions in program order; the fences above do
// that too, but they also emit s_waitcnt lgkmcnt(0) (an empty LDS queue before the first dependent read is
// even issued: five drains of 100-200 cycles per pair of tiles in the bf16 backward) and stop every other
// instruction from being scheduled across.  A fence at WAVEFRONT scope is exactly that: the memory model
// orders the wave's own accesses on both sides of it - other lanes' included, the compiler may not move a
// read above the write whatever it can prove about one lane's addresses - and, a wave's memory operations
// being in order already, it costs no instruction.  (The wave64 emulator runs lanes as fibers: there the
// barrier is real.)
#ifndef GNNTRK_LDS_INORDER
#define GNNTRK_LDS_INORDER 1
#endif
__device__ __forceinline__ void lds_wave_order() {
#if defined(GNNTRK_BF16_PRIMITIVES) || !GNNTRK_LDS_INORDER
    lds_wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
}

// A-operand fragments of a matrix Mat[rowfeat][kfeat] held in nn.Linear storage
// W[out][in] (ld = in_dim):  forward layer: Mat = W  (rows = out, k = in);
// transposed (backward dX): Mat = W^T (rows = in, k = out).
// dst[(to * kmap.ks + ks) * 64 + lane];  lane (g,c) holds
//     Mat[feat_of(rowmap,to,c>>2,c&3)][feat_of(kmap,ks>>2,g,ks&3)].
__device__ inline void fill_frags(float *dst, const float *W, int ld, const DimMap rowmap,
                                  const DimMap kmap, bool transposed, int tid, int nthreads) {
    const int n = rowmap.nt * kmap.ks * 64;
    for (int i = tid; i < n; i += nthreads) {
        const int fr = i >> 6, l = i & 63;
        const int to = fr / kmap.ks, ks = fr - to * kmap.ks;
        const int g = l >> 4, c = l & 15;
        const int rf = feat_of(rowmap, to, c >> 2, c & 3);
        const int kf = feat_of(kmap, ks >> 2, g, ks & 3);
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

// XCD-aware persistent tile schedule: block b runs on XCD b % 8 (observed dispatch
// order, MI355X_MICROARCH.md "Workgroup dispatch"); give every XCD one contiguous
// range of tiles so the node rows gathered by neighbouring tiles stay in ITS L2.
// Correctness does not depend on the placement.
struct TileSched {
    int64_t cur, end, step;
};
__device__ inline TileSched make_sched(int64_t n_tiles, int waves = kWaves) {
    const int G = gridDim.x;
    const int nx = (G % 8 == 0) ? 8 : 1;
    const int x = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = G / nx;
    const int64_t t0 = n_tiles * x / nx, t1 = n_tiles * (x + 1) / nx;
    TileSched s;
    // the wave index is wave-uniform: taken through readfirstlane the whole schedule lives in
    // SGPRs and the tile loop is a uniform (scalar-branch) loop.  As a divergent loop its
    // loop-carried MFMA accumulators needed an exec-masked copy per register and iteration
    // (an MFMA writes all lanes regardless of exec).
    s.cur = t0 + lb * waves + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s.end = t1;
    s.step = (int64_t)bpx * waves;
    return s;
}

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

// Weight (A-operand) fragments of one layer: either read from LDS at every use
// (generic kernels) or copied once into registers (static kernels: with one wave per
// SIMD and a 512-entry register file they are loop-invariant operands, and the MFMA
// chain runs without a single LDS wait).
template <int N>
struct RegFrags {
    float v[N];
    __device__ __forceinline__ void load(const float *lds, int n, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (i < n) ? lds[i * 64 + lane] : 0.f;
    }
    __device__ __forceinline__ float get(int i, int) const { return v[i]; }
};
template <int N>
struct LdsFrags {
    const float *p;
    __device__ __forceinline__ void load(const float *lds, int, int) { p = lds; }
    __device__ __forceinline__ float get(int i, int lane) const { return p[i * 64 + lane]; }
};
template <int N>
struct RegBias {
    f32x4 v[N];
    __device__ __forceinline__ void load(const f32x4 *lds, int n, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (i < n) ? lds[i * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ f32x4 get(int i, int) const { return v[i]; }
};
template <int N>
struct LdsBias {
    const f32x4 *p;
    __device__ __forceinline__ void load(const f32x4 *lds, int, int) { p = lds; }
    __device__ __forceinline__ f32x4 get(int i, int lane) const { return p[i * 64 + lane]; }
};

// ---- row-staged input / output through LDS ------------------------------------------
// The concatenated input of a tile is assembled in an LDS staging buffer laid out
// [feature][row] (leading dim kTbLd): the wave walks a uniform "load list" - one item per
// (segment, 4-feature chunk) - with lane = (row c = lane & 15, element part = lane >> 4), so
// every global access is wave-uniform in control flow, addressed as uniform base + 32-bit
// offset, and row-contiguous.  The MFMA B operand is then one ds_read_b32 per k-step from the
// buffer, and the transposed copy the weight-gradient MFMAs need is a ds_read_b128 of the
// SAME buffer.  Gradient slices leave the same way in reverse.
constexpr int kMaxItems = 16;

struct LoadList {  // lives in LDS, built once per workgroup
    const float *ptr[kMaxItems];    // segment base + 4 * chunk
    const int32_t *idx[kMaxItems];  // row gather index or nullptr
    float *gptr[kMaxItems];         // gradient slice base + 4 * chunk, or nullptr
    const int32_t *gidx[kMaxItems];
    int32_t stride[kMaxItems], gstride[kMaxItems];
    int32_t rem[kMaxItems];   // features in this chunk: min(4, dim - 4 * chunk)
    int32_t frow[kMaxItems];  // first feature row of the chunk in the staging buffer
    int32_t relu[kMaxItems], gacc[kMaxItems];
    int32_t n;
};

__device__ inline void build_load_list(LoadList &L, const SegTable &tab, int n_seg) {
    int n = 0, foff = 0;
    for (int j = 0; j < n_seg; ++j) {
        const int d = tab.dim[j];
        for (int ch = 0; 4 * ch < d && n < kMaxItems; ++ch, ++n) {
            L.ptr[n] = tab.ptr[j] + 4 * ch;
            L.idx[n] = tab.idx[j];
            L.gptr[n] = tab.gptr[j] ? tab.gptr[j] + 4 * ch : nullptr;
            L.gidx[n] = tab.gidx[j];
            L.stride[n] = tab.stride[j];
            L.gstride[n] = tab.gstride[j];
            L.rem[n] = (d - 4 * ch < 4) ? d - 4 * ch : 4;
            L.frow[n] = foff + 4 * ch;
            L.relu[n] = tab.relu[j];
            L.gacc[n] = tab.gacc[j];
        }
        foff += d;
    }
    L.n = n;
}

__device__ __forceinline__ uint32_t uni32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T uni_ptr(const void *p) {  // wave-uniform pointer -> SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)uni32((uint32_t)(v >> 32)) << 32) |
                                 (unsigned long long)uni32((uint32_t)v);
    return (T)u;
}

// uniform per-item descriptors held by every wave (SGPRs)
// Uniform per-item descriptors.  Static shapes hoist them into SGPRs once (HOIST = true);
// the generic kernels re-read them from the LDS load list at every use - they already run
// out of scalar registers, and hundreds of SGPR spills are not worth a few LDS reads.
template <int NI, bool HOIST>
struct Items;

template <int NI>
struct Items<NI, true> {
    gcf_ptr ptr_[NI];
    gci_ptr idx_[NI];
    gf_ptr gptr_[NI];
    int32_t stride_[NI], meta_[NI], gstride_[NI];  // meta = rem | frow << 8 | relu << 24
    int32_t n;
    __device__ __forceinline__ void load(const LoadList &L) {
        n = (int32_t)uni32((uint32_t)L.n);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool on = i < n;
            ptr_[i] = on ? uni_ptr<gcf_ptr>(L.ptr[i]) : nullptr;
            idx_[i] = on ? uni_ptr<gci_ptr>(L.idx[i]) : nullptr;
            gptr_[i] = on ? uni_ptr<gf_ptr>(L.gptr[i]) : nullptr;
            stride_[i] = on ? (int32_t)uni32((uint32_t)L.stride[i]) : 0;
            gstride_[i] = on ? (int32_t)uni32((uint32_t)L.gstride[i]) : 0;
            meta_[i] = on ? (int32_t)uni32((uint32_t)(L.rem[i] | (L.frow[i] << 8) | (L.relu[i] << 24))) : 0;
        }
    }
    __device__ __forceinline__ gcf_ptr ptr(int i) const { return ptr_[i]; }
    __device__ __forceinline__ gci_ptr idx(int i) const { return idx_[i]; }
    __device__ __forceinline__ gf_ptr gptr(int i) const { return gptr_[i]; }
    __device__ __forceinline__ int stride(int i) const { return stride_[i]; }
    __device__ __forceinline__ int gstride(int i) const { return gstride_[i]; }
    __device__ __forceinline__ int rem(int i) const { return meta_[i] & 0xff; }
    __device__ __forceinline__ int frow(int i) const { return (meta_[i] >> 8) & 0xffff; }
    __device__ __forceinline__ bool relu(int i) const { return (meta_[i] >> 24) != 0; }
};

template <int NI>
struct Items<NI, false> {
    const LoadList *L;
    int32_t n;
    __device__ __forceinline__ void load(const LoadList &l) {
        L = &l;
        n = (int32_t)uni32((uint32_t)l.n);
    }
    __device__ __forceinline__ gcf_ptr ptr(int i) const { return uni_ptr<gcf_ptr>(L->ptr[i]); }
    __device__ __forceinline__ gci_ptr idx(int i) const { return uni_ptr<gci_ptr>(L->idx[i]); }
    __device__ __forceinline__ gf_ptr gptr(int i) const { return uni_ptr<gf_ptr>(L->gptr[i]); }
    __device__ __forceinline__ int stride(int i) const { return (int)uni32((uint32_t)L->stride[i]); }
    __device__ __forceinline__ int gstride(int i) const { return (int)uni32((uint32_t)L->gstride[i]); }
    __device__ __forceinline__ int rem(int i) const { return (int)uni32((uint32_t)L->rem[i]); }
    __device__ __forceinline__ int frow(int i) const { return (int)uni32((uint32_t)L->frow[i]); }
    __device__ __forceinline__ bool relu(int i) const { return uni32((uint32_t)L->relu[i]) != 0; }
};

// row ids of one tile row for every item (row is pre-clamped to a valid row)
template <int NI, bool H>
__device__ __forceinline__ void item_row_ids(const Items<NI, H> &it, int64_t row, int32_t (&rid)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int32_t v = (int32_t)row;
        if (i < it.n) {
            const gci_ptr ix = it.idx(i);
            if (ix != nullptr) v = ix[row];
        }
        rid[i] = v;
    }
}
template <int NI, bool H>
__device__ __forceinline__ void item_values(const Items<NI, H> &it, const int32_t (&rid)[NI],
                                            int part, bool valid, float (&pv)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float v = 0.f;
        if (i < it.n) {
            const gcf_ptr p = it.ptr(i);
            const int st = it.stride(i), rm = it.rem(i);
            if (valid && part < rm) v = p[(int64_t)rid[i] * st + part];
        }
        pv[i] = v;
    }
}
// staging buffer [feature][row]: value of (item i, element part) of row c
template <int NI, bool H>
__device__ __forceinline__ void stage_items(const Items<NI, H> &it, float *sbuf, int part, int c,
                                            const float (&pv)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (i < it.n) {
            const int rm = it.rem(i), fr = it.frow(i);
            const bool rl = it.relu(i);
            if (part < rm) sbuf[(fr + part) * kTbLd + c] = rl ? fmaxf(pv[i], 0.f) : pv[i];
        }
}
// LDS offsets of a lane's B-operand features (padding -> zero row `zrow`)
template <int NT>
__device__ __forceinline__ void operand_offsets(const DimMap &map, int g, int c, int zrow,
                                                int (&off)[NT * 4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = feat_of(map, t, g, r);
            off[t * 4 + r] = (f >= 0 ? f : zrow) * kTbLd + c;
        }
}
template <int NT>
__device__ __forceinline__ void read_operand(const float *sbuf, const int (&off)[NT * 4], int nt,
                                             f32x4 (&x)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sbuf[off[t * 4 + r]];
        }
        x[t] = v;
    }
}

// Forward of the (2- or 3-layer) MLP on one 16-row tile.  a1/a2 are post-ReLU hidden
// activations (a2 = a1 for two layers), y the linear output, all in accumulator layout.
template <int KT, int HT, class Dims, class W1, class B1>
__device__ __forceinline__ void mlp_layer1(const Dims &dm, const W1 &w1, const B1 &b1, int lane,
                                           const f32x4 (&bin)[KT], f32x4 (&a1)[HT]) {
    const int nth = (dm.hid_ks() + 3) >> 2;
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        if (to < nth)
            a1[to] = b1.get(to, lane);
        else
            a1[to] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * t + r < dm.in_ks()) {
#pragma unroll
                for (int to = 0; to < HT; ++to)
                    if (to < nth)
                        a1[to] = mfma4(w1.get(to * dm.in_ks() + 4 * t + r, lane), bin[t][r], a1[to]);
            }
#pragma unroll
    for (int to = 0; to < HT; ++to)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[to][r] = fmaxf(a1[to][r], 0.f);
}

template <int HT, class Dims, class W2, class B2>
__device__ __forceinline__ void mlp_layer2(const Dims &dm, const W2 &w2, const B2 &b2, int lane,
                                           const f32x4 (&a1)[HT], f32x4 (&a2)[HT]) {
    const int nth = (dm.hid_ks() + 3) >> 2;
    if (dm.three()) {
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            if (to < nth)
                a2[to] = b2.get(to, lane);
            else
                a2[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * t + r < dm.hid_ks()) {
#pragma unroll
                    for (int to = 0; to < HT; ++to)
                        if (to < nth)
                            a2[to] = mfma4(w2.get(to * dm.hid_ks() + 4 * t + r, lane), a1[t][r],
                                           a2[to]);
                }
#pragma unroll
        for (int to = 0; to < HT; ++to)
#pragma unroll
            for (int r = 0; r < 4; ++r) a2[to][r] = fmaxf(a2[to][r], 0.f);
    } else {
#pragma unroll
        for (int to = 0; to < HT; ++to) a2[to] = a1[to];
    }
}

template <int HT, class Dims, class W3>
__device__ __forceinline__ f32x4 mlp_layer3(const Dims &dm, const W3 &w3, f32x4 b3, int lane,
                                            const f32x4 (&a2)[HT]) {
    f32x4 y = b3;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * t + r < dm.hid_ks()) y = mfma4(w3.get(4 * t + r, lane), a2[t][r], y);
    return y;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

}  // namespace gnntrk
