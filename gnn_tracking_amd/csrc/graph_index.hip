// Graph index: stable target-sorted (CSR) and source-sorted views of a COO edge list.
// See gnntrk_graph_index in include/gnntrk.h (replaces the gather / scatter bookkeeping of PyG's
// MessagePassing.propagate, models/interaction_network.py:36,67).  HBM-bound integer work.
//
// Own two-level counting sort (round 3; the library radix sort stays for shapes outside it):
//
//   a node id is split into (bucket = id >> 8, low = id & 255).  The edge list is cut into chunks of
//   consecutive edges (one workgroup each).
//     1. count   per (bucket, chunk) edge counts: LDS histogram per chunk, written into a dense
//                bucket-major table
//     2. scan    one exclusive scan over the flattened table = the output offset of every
//                (bucket, chunk) run
//     3. split   every edge goes to its bucket's region (position from an LDS cursor per bucket) as
//                one 8-byte record {low | payload, value}
//     4. sort    one workgroup per bucket (256 nodes, a few thousand edges, held in LDS): histogram
//                over `low` -> row pointers of the bucket's nodes; the rank of a record inside its
//                node's list is the number of records of that node with a smaller value (edge id /
//                CSR position), so the result is the STABLE order whatever order step 3's cursors
//                handed out - deterministic, bit-identical to a stable sort.
//   Run twice: by target over the COO list (value = edge id, payload = source) and by source over
//   the CSR list (value = CSR position).  Row pointers, `src[perm]`, `tgt` and the inverse
//   permutation fall out of step 4 - no key extraction, gather or boundary passes.
//
//   What makes steps 1 / 3 stream: a collated batch is a list of events with disjoint id ranges and
//   contiguous edge ranges, so a chunk only touches the few hundred buckets of its own event: the
//   counters of a chunk live in an LDS window of 4096 buckets (kWin) placed a third below the chunk's first id and the
//   runs it writes are hundreds of bytes long.  Ids outside the window (one giant unsorted graph)
//   take global atomics on the table itself - slower, same result.  A bucket beyond the LDS
//   capacity (hub nodes) is ranked from global memory in tiles - slower, same result.
//
//   Algorithmic bytes per edge: count 8 + split 16 + 8 + sort 8 + 12, then 4 + 4 + 8 + 8 + 8 = 84
//   (+ 0.3 table entries per edge x 12 B); the library path moved about 260.
#include "host_util.h"
#include "tile_bf16.h"

namespace gnntrk {

constexpr int kTpb = 256;

#ifndef GNNTRK_REMAP_NT
#define GNNTRK_REMAP_NT 3   // (bit 0: non-temporal id loads, bit 1: non-temporal stores of gi_remap_kernel - A/B builds)
#endif
#ifndef GNNTRK_NO_LIBRARY_SORT_OFF
#define GNNTRK_NO_LIBRARY_SORT_OFF 0   // (1: gnntrk_node_order always takes the exact radix form - A/B builds)
#endif
// ------------------------------------------------------------------ own counting sort
#ifndef GNNTRK_GI_SH
#define GNNTRK_GI_SH 8
#endif
#ifndef GNNTRK_GI_SORT_TPB
#define GNNTRK_GI_SORT_TPB 512
#endif
constexpr int kSH = GNNTRK_GI_SH;       // nodes per bucket = 256
constexpr int kBins = 1 << kSH;
constexpr int kSortTpb = GNNTRK_GI_SORT_TPB;
constexpr int kSortR = 10;              // records a thread of the bucket sort holds in registers
constexpr int kCap = kSortTpb * kSortR; // records of a bucket the bucket sort holds in LDS (5120)
constexpr int kWin = 4096;              // bucket counters a chunk holds in LDS
// One split workgroup of 1024 lanes per CU.  Two resident workgroups (768 lanes at <= 85 registers) measured
// SLOWER (split by target 1.18 against 1.12 ms, by source 0.38 against 0.33): the open (bucket, chunk) runs of
// the resident chunks - about 600 partially written lines per chunk and output array - then no longer fit an
// XCD's 4 MB L2, and the runs leave it in pieces.
#ifndef GNNTRK_GI_SPLIT_TPB
#define GNNTRK_GI_SPLIT_TPB 1024
#endif
#ifndef GNNTRK_GI_SPLIT_WG
#define GNNTRK_GI_SPLIT_WG 1
#endif
constexpr int kSplitTpb = GNNTRK_GI_SPLIT_TPB;
constexpr int kCountTpb = 1024;
constexpr int kSplitR = 4;              // edges per thread and tile of the split
constexpr int kSplitTile = kSplitTpb * kSplitR;
constexpr int kMaxChunks = 512;
#ifndef GNNTRK_GI_HUB_BITMAP_MIN
#define GNNTRK_GI_HUB_BITMAP_MIN 65536   // hub nodes above this degree are ranked through per-run bitmaps (the emulator build: 8192)
#endif
constexpr int64_t kMaxTable = (int64_t)96 << 20;   // table entries (4 B each)
constexpr int kScanTpb = 256, kScanItems = 16, kScanTile = kScanTpb * kScanItems;

// first sort: key = target (COO row 1), payload = source (row 0), value = edge id.  Two per-edge
// inputs of the caller can ride along into CSR order (gnntrk_graph_index_carry): a 1-byte label in
// bit 31 of the value (edge ids are below 2^31), a row of four floats as four bf16 in a second
// 8-byte record array
struct KeysCoo {
    const int64_t *src, *tgt;
    int64_t N;
    const uint8_t *label;
    const float *rows;
    int rows_stride;
    __device__ __forceinline__ uint32_t key(int64_t e, int *bad) const { return checked(tgt[e], bad); }
    __device__ __forceinline__ uint32_t payload(int64_t e, int *bad) const { return checked(src[e], bad); }
    __device__ __forceinline__ uint32_t value(int64_t e) const {
        return (uint32_t)e | ((label && label[e]) ? 0x80000000u : 0u);
    }
    __device__ __forceinline__ uint2 row(int64_t e) const {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(rows + e * rows_stride);
        return uint2{bf16x2_pack(v[0], v[1]), bf16x2_pack(v[2], v[3])};
    }
    // four consecutive edges e .. e + 3 of one lane (e a multiple of four): with `vec` (16-byte aligned id rows,
    // 4-byte aligned labels) the ids come as 16-byte loads - 8-byte-per-lane streams top out near 3 TB/s here
    int vec;
    // optional renumbering of the nodes (gnntrk_graph_index_carry.node_rank: old id -> new id, a permutation of
    // [0, N) that keeps every event's id range): applied where an id is read, so the whole index comes out in
    // the new numbering.  The table of one event (600 KB at 150 k hits) stays in the XCD's L2.
    const int32_t *rank;
    __device__ __forceinline__ uint32_t checked(int64_t v, int *bad) const {
        if (v < 0 || v >= N) {
            if (bad) atomicAdd(bad, 1);
            v = v < 0 ? 0 : N - 1;
        }
        return rank ? (uint32_t)rank[v] : (uint32_t)v;
    }
    __device__ __forceinline__ void keys4(int64_t e, int64_t e1, uint32_t k[4], int *bad) const {
        if (vec && e + 4 <= e1) {
            const longlong2 a = *reinterpret_cast<const longlong2 *>(tgt + e);
            const longlong2 b = *reinterpret_cast<const longlong2 *>(tgt + e + 2);
            k[0] = checked(a.x, bad);
            k[1] = checked(a.y, bad);
            k[2] = checked(b.x, bad);
            k[3] = checked(b.y, bad);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) k[q] = e + q < e1 ? key(e + q, bad) : 0u;
        }
    }
    // the split holds the NEXT tile's ids as loaded (converted only when their tile starts: a conversion at the
    // load would wait for it on the spot)
    struct Raw {
        longlong2 ta, tb, sa, sb;
        uint32_t l4;
    };
    __device__ __forceinline__ void load_raw(int64_t e, int64_t e1, Raw &r) const {
        if (vec && e + 4 <= e1) {
            r.ta = *reinterpret_cast<const longlong2 *>(tgt + e);
            r.tb = *reinterpret_cast<const longlong2 *>(tgt + e + 2);
            r.sa = *reinterpret_cast<const longlong2 *>(src + e);
            r.sb = *reinterpret_cast<const longlong2 *>(src + e + 2);
            r.l4 = label ? *reinterpret_cast<const uint32_t *>(label + e) : 0u;
        } else {
            long long t[4], u[4];
            r.l4 = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = e + q < e1;
                t[q] = in ? tgt[e + q] : 0;
                u[q] = in ? src[e + q] : 0;
                if (in && label && label[e + q]) r.l4 |= 1u << (8 * q);
            }
            r.ta = longlong2{t[0], t[1]};
            r.tb = longlong2{t[2], t[3]};
            r.sa = longlong2{u[0], u[1]};
            r.sb = longlong2{u[2], u[3]};
        }
    }
    // (ids out of range: the targets were counted by step 1, the sources are counted here)
    __device__ __forceinline__ void decode(const Raw &r, int64_t e, uint32_t k[4], uint32_t p[4], uint32_t v[4],
                                           int *bad) const {
        k[0] = checked(r.ta.x, nullptr);
        k[1] = checked(r.ta.y, nullptr);
        k[2] = checked(r.tb.x, nullptr);
        k[3] = checked(r.tb.y, nullptr);
        p[0] = checked(r.sa.x, bad);
        p[1] = checked(r.sa.y, bad);
        p[2] = checked(r.sb.x, bad);
        p[3] = checked(r.sb.y, bad);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (uint32_t)(e + q) | (((r.l4 >> (8 * q)) & 0xffu) ? 0x80000000u : 0u);
    }
    __device__ __forceinline__ f32x4 row_raw(int64_t e) const {
        return *reinterpret_cast<const f32x4 *>(rows + e * rows_stride);
    }
};
// first sort over a RENUMBERED edge list (gnntrk_graph_index_carry.node_rank): gi_remap_kernel has written both
// endpoint rows as validated int32 new ids (one streaming pass with every lane's eight table lookups in flight:
// inside the count / split kernels, whose occupancy is one workgroup per CU, the same lookups cost 1.7 ms per
// 64 M edges, here about a third of that); the count and split then read 4 + 8 bytes per edge instead of 8 + 16
struct KeysCoo32 {
    const int32_t *src, *tgt;
    const uint8_t *label;
    const float *rows;
    int rows_stride;
    int vec;   // 4-byte aligned labels
    __device__ __forceinline__ uint32_t key(int64_t e, int *) const { return (uint32_t)tgt[e]; }
    __device__ __forceinline__ uint32_t payload(int64_t e, int *) const { return (uint32_t)src[e]; }
    __device__ __forceinline__ uint32_t value(int64_t e) const {
        return (uint32_t)e | ((label && label[e]) ? 0x80000000u : 0u);
    }
    __device__ __forceinline__ uint2 row(int64_t e) const {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(rows + e * rows_stride);
        return uint2{bf16x2_pack(v[0], v[1]), bf16x2_pack(v[2], v[3])};
    }
    __device__ __forceinline__ void keys4(int64_t e, int64_t e1, uint32_t k[4], int *) const {
        if (e + 4 <= e1) {   // (own 16-byte aligned arrays; e is a multiple of four)
            const int4 a = *reinterpret_cast<const int4 *>(tgt + e);
            k[0] = (uint32_t)a.x;
            k[1] = (uint32_t)a.y;
            k[2] = (uint32_t)a.z;
            k[3] = (uint32_t)a.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) k[q] = e + q < e1 ? (uint32_t)tgt[e + q] : 0u;
        }
    }
    struct Raw {
        int4 t, s;
        uint32_t l4;
    };
    __device__ __forceinline__ void load_raw(int64_t e, int64_t e1, Raw &r) const {
        if (e + 4 <= e1) {
            r.t = *reinterpret_cast<const int4 *>(tgt + e);
            r.s = *reinterpret_cast<const int4 *>(src + e);
            if (!label) {
                r.l4 = 0u;
            } else if (vec) {
                r.l4 = *reinterpret_cast<const uint32_t *>(label + e);
            } else {
                r.l4 = (label[e] ? 1u : 0u) | (label[e + 1] ? 0x100u : 0u) | (label[e + 2] ? 0x10000u : 0u) |
                       (label[e + 3] ? 0x1000000u : 0u);
            }
        } else {
            int t[4], u[4];
            r.l4 = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = e + q < e1;
                t[q] = in ? tgt[e + q] : 0;
                u[q] = in ? src[e + q] : 0;
                if (in && label && label[e + q]) r.l4 |= 1u << (8 * q);
            }
            r.t = int4{t[0], t[1], t[2], t[3]};
            r.s = int4{u[0], u[1], u[2], u[3]};
        }
    }
    __device__ __forceinline__ void decode(const Raw &r, int64_t e, uint32_t k[4], uint32_t p[4], uint32_t v[4],
                                           int *) const {
        k[0] = (uint32_t)r.t.x;
        k[1] = (uint32_t)r.t.y;
        k[2] = (uint32_t)r.t.z;
        k[3] = (uint32_t)r.t.w;
        p[0] = (uint32_t)r.s.x;
        p[1] = (uint32_t)r.s.y;
        p[2] = (uint32_t)r.s.z;
        p[3] = (uint32_t)r.s.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (uint32_t)(e + q) | (((r.l4 >> (8 * q)) & 0xffu) ? 0x80000000u : 0u);
    }
    __device__ __forceinline__ f32x4 row_raw(int64_t e) const {
        return *reinterpret_cast<const f32x4 *>(rows + e * rows_stride);
    }
};
// both endpoint rows of the COO list through the renumbering: ids checked (counted, clamped), translated, written
// as int32.  A lane takes four consecutive edges: 16-byte id loads, eight lookups in flight, 16-byte stores.
__global__ __launch_bounds__(kTpb) void gi_remap_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ tgt,
                                                        int64_t E, int64_t N, const int32_t *__restrict__ rank, int vec,
                                                        int32_t *__restrict__ src32, int32_t *__restrict__ tgt32,
                                                        int *__restrict__ bad) {
    auto chk = [&](long long v) -> long long {
        if (v < 0 || v >= N) {
            atomicAdd(bad, 1);
            v = v < 0 ? 0 : N - 1;
        }
        return v;
    };
    const int64_t n4 = (E + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kTpb) {
        const int64_t e = 4 * i;
        if (vec && e + 4 <= E) {
            // (the id streams and the outputs pass through once: non-temporal, the lookup table keeps the L2)
            typedef int i4_hw __attribute__((ext_vector_type(4)));
            auto ld2 = [](const int64_t *p, long long &a, long long &b) {   // two ids as one 16-byte non-temporal load
#if GNNTRK_REMAP_NT & 1
                const i4_hw v = __builtin_nontemporal_load(reinterpret_cast<const i4_hw *>(p));
#else
                const i4_hw v = *reinterpret_cast<const i4_hw *>(p);
#endif
                a = (long long)(((unsigned long long)(uint32_t)v[1] << 32) | (uint32_t)v[0]);
                b = (long long)(((unsigned long long)(uint32_t)v[3] << 32) | (uint32_t)v[2]);
            };
            long long it[4], is[4];
            ld2(tgt + e, it[0], it[1]);
            ld2(tgt + e + 2, it[2], it[3]);
            ld2(src + e, is[0], is[1]);
            ld2(src + e + 2, is[2], is[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                it[q] = chk(it[q]);
                is[q] = chk(is[q]);
            }
#if GNNTRK_REMAP_NT & 4   // (timing probe: no table lookups - results are NOT a renumbering)
            const i4_hw ot = {(int)it[0], (int)it[1], (int)it[2], (int)it[3]};
            const i4_hw os = {(int)is[0], (int)is[1], (int)is[2], (int)is[3]};
#else
            const i4_hw ot = {rank[it[0]], rank[it[1]], rank[it[2]], rank[it[3]]};
            const i4_hw os = {rank[is[0]], rank[is[1]], rank[is[2]], rank[is[3]]};
#endif
#if GNNTRK_REMAP_NT & 2
            __builtin_nontemporal_store(ot, reinterpret_cast<i4_hw *>(tgt32 + e));
            __builtin_nontemporal_store(os, reinterpret_cast<i4_hw *>(src32 + e));
#else
            *reinterpret_cast<i4_hw *>(tgt32 + e) = ot;
            *reinterpret_cast<i4_hw *>(src32 + e) = os;
#endif
        } else {
            for (int q = 0; q < 4 && e + q < E; ++q) {
                tgt32[e + q] = rank[chk(tgt[e + q])];
                src32[e + q] = rank[chk(src[e + q])];
            }
        }
    }
}
// second sort: key = source of the CSR-ordered list (already validated), value = CSR position
struct KeysCsr {
    const int32_t *src;
    __device__ __forceinline__ uint32_t key(int64_t e, int *) const { return (uint32_t)src[e]; }
    __device__ __forceinline__ uint32_t payload(int64_t, int *) const { return 0u; }
    __device__ __forceinline__ uint32_t value(int64_t e) const { return (uint32_t)e; }
    __device__ __forceinline__ uint2 row(int64_t) const { return uint2{0u, 0u}; }
    __device__ __forceinline__ void keys4(int64_t e, int64_t e1, uint32_t k[4], int *) const {
        if (e + 4 <= e1) {   // (the CSR list is an own 16-byte aligned array; e is a multiple of four)
            const int4 a = *reinterpret_cast<const int4 *>(src + e);
            k[0] = (uint32_t)a.x;
            k[1] = (uint32_t)a.y;
            k[2] = (uint32_t)a.z;
            k[3] = (uint32_t)a.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) k[q] = e + q < e1 ? (uint32_t)src[e + q] : 0u;
        }
    }
    struct Raw {
        int4 a;
    };
    __device__ __forceinline__ void load_raw(int64_t e, int64_t e1, Raw &r) const {
        if (e + 4 <= e1) {
            r.a = *reinterpret_cast<const int4 *>(src + e);
        } else {
            r.a = int4{e < e1 ? src[e] : 0, e + 1 < e1 ? src[e + 1] : 0, e + 2 < e1 ? src[e + 2] : 0, 0};
        }
    }
    __device__ __forceinline__ void decode(const Raw &r, int64_t e, uint32_t k[4], uint32_t p[4], uint32_t v[4],
                                           int *) const {
        k[0] = (uint32_t)r.a.x;
        k[1] = (uint32_t)r.a.y;
        k[2] = (uint32_t)r.a.z;
        k[3] = (uint32_t)r.a.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] = 0u;
            v[q] = (uint32_t)(e + q);
        }
    }
    __device__ __forceinline__ f32x4 row_raw(int64_t) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
};

// The LDS window of a chunk starts a third of its width below the bucket of the chunk's first id: a
// chunk that begins inside event A and runs into event B sees ids from A's first to B's last
__device__ __forceinline__ int gi_window_start(uint32_t first_key) {
    const int b = (int)(first_key >> kSH) - kWin / 3;
    return b < 0 ? 0 : b;
}

// step 1.  tbl[bucket * n_chunks + chunk] = edges of the chunk in the bucket (tbl zeroed before);
// range[chunk] = (first, last) touched window slot
template <class K>
__global__ __launch_bounds__(kCountTpb) void gi_count_kernel(K keys, int64_t E, int chunk, int n_chunks, int NB,
                                                             uint32_t *__restrict__ tbl, int2 *__restrict__ range,
                                                             int *__restrict__ bad) {
    __shared__ uint32_t s_cnt[kWin];
    __shared__ int s_lo, s_hi;
    const int tid = threadIdx.x, c = blockIdx.x;
    const int64_t e0 = (int64_t)c * chunk, e1 = e0 + chunk < E ? e0 + chunk : E;
    for (int d = tid; d < kWin; d += kCountTpb) s_cnt[d] = 0;
    if (tid == 0) {
        s_lo = kWin;
        s_hi = -1;
    }
    __syncthreads();
    const int w0 = gi_window_start(keys.key(e0, nullptr));   // (e0 is counted below)
    for (int64_t e = e0 + 4 * tid; e < e1; e += 4 * kCountTpb) {
        uint32_t k4[4];
        keys.keys4(e, e1, k4, bad);
        // (consecutive edges of an ordered list share their bucket: one counter update per run of a lane's four)
        auto bump = [&](uint32_t b, uint32_t by) {
            const uint32_t d = b - (uint32_t)w0;
            if (d < (uint32_t)kWin)
                atomicAdd(&s_cnt[d], by);
            else
                atomicAdd(&tbl[(size_t)b * n_chunks + c], by);
        };
        uint32_t run_b = k4[0] >> kSH, run_n = 1;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            if (e + q >= e1) break;
            const uint32_t b = k4[q] >> kSH;
            if (b == run_b) {
                run_n += 1;
            } else {
                bump(run_b, run_n);
                run_b = b;
                run_n = 1;
            }
        }
        bump(run_b, run_n);
    }
    __syncthreads();
    int lo = kWin, hi = -1;
    for (int d = tid; d < kWin; d += kCountTpb) {
        const uint32_t v = s_cnt[d];
        if (v) {
            tbl[(size_t)(w0 + d) * n_chunks + c] = v;
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
    }
    if (hi >= 0) {
        atomicMin(&s_lo, lo);
        atomicMax(&s_hi, hi);
    }
    __syncthreads();
    if (tid == 0) range[c] = int2{s_lo, s_hi};
    (void)NB;
}

// step 2: exclusive scan of tbl[0 .. T] in place (tbl[T] = total), three kernels
__global__ __launch_bounds__(kScanTpb) void gi_scan_sums_kernel(const uint32_t *__restrict__ v, int64_t n,
                                                                uint32_t *__restrict__ sums) {
    __shared__ uint32_t s[kScanTpb];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
    uint32_t t = 0;
    if (i0 + kScanItems <= n) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4 *>(v + i0 + 4 * q);
            t += a.x + a.y + a.z + a.w;
        }
    } else {
        for (int q = 0; q < kScanItems; ++q) t += i0 + q < n ? v[i0 + q] : 0u;
    }
    s[tid] = t;
    __syncthreads();
    for (int d = kScanTpb / 2; d > 0; d >>= 1) {
        if (tid < d) s[tid] += s[tid + d];
        __syncthreads();
    }
    if (tid == 0) sums[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void gi_scan_tiles_kernel(uint32_t *__restrict__ sums, int n) {
    __shared__ uint32_t s[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t mine = 0;
    for (int i = lo; i < hi; ++i) mine += sums[i];
    s[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? s[tid - d] : 0u;
        __syncthreads();
        s[tid] += v;
        __syncthreads();
    }
    uint32_t run = s[tid] - mine;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = sums[i];
        sums[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(kScanTpb) void gi_scan_apply_kernel(uint32_t *__restrict__ v, int64_t n,
                                                                 const uint32_t *__restrict__ sums) {
    __shared__ uint32_t s[kScanTpb];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
    uint32_t x[kScanItems];
    const bool full = i0 + kScanItems <= n;
    if (full) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4 *>(v + i0 + 4 * q);
            x[4 * q] = a.x;
            x[4 * q + 1] = a.y;
            x[4 * q + 2] = a.z;
            x[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kScanItems; ++q) x[q] = i0 + q < n ? v[i0 + q] : 0u;
    }
    uint32_t t = 0;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) t += x[q];
    s[tid] = t;
    __syncthreads();
    for (int d = 1; d < kScanTpb; d <<= 1) {
        const uint32_t u = tid >= d ? s[tid - d] : 0u;
        __syncthreads();
        s[tid] += u;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + s[tid] - t;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        const uint32_t c = x[q];
        x[q] = run;
        run += c;
    }
    if (full) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q)
            *reinterpret_cast<uint4 *>(v + i0 + 4 * q) = uint4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    } else {
#pragma unroll
        for (int q = 0; q < kScanItems; ++q)
            if (i0 + q < n) v[i0 + q] = x[q];
    }
}

// base[b] = first record of bucket b (b <= NB), taken before step 3's cursors touch the table
__global__ __launch_bounds__(kTpb) void gi_bases_kernel(const uint32_t *__restrict__ tbl, int n_chunks, int NB,
                                                        uint32_t *__restrict__ base) {
    const int b = blockIdx.x * kTpb + threadIdx.x;
    if (b <= NB) base[b] = tbl[(size_t)b * n_chunks];
}

// exclusive scan of one value per thread over a workgroup of kSplitTpb threads (wave shuffles + one
// LDS round for the 16 wave totals); also returns the total
__device__ __forceinline__ uint32_t gi_block_exscan(uint32_t v, uint32_t *s_wsum, uint32_t *total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl(inc, lane - d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSplitTpb / 64; ++w) {
        const uint32_t t = s_wsum[w];
        pre += w < wv ? t : 0u;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// step 3.  part[pos] = {lo: value (edge id [| label << 31] / CSR position), hi: low << pbits | payload}.
// A tile of 4096 edges is first grouped by bucket in LDS (rank inside the tile's bucket from an LDS
// counter, tile-local starts from a scan over the chunk's touched window range), then written out
// with consecutive lanes on consecutive records of a run: the stores coalesce.  ROWS: the
// caller's per-edge rows (four floats -> four bf16) take the same way into part_rows[pos].
template <class K, bool ROWS>
__global__ __launch_bounds__(kSplitTpb, GNNTRK_GI_SPLIT_WG * kSplitTpb / 256) void gi_split_kernel(K keys, int64_t E, int chunk, int n_chunks, int pbits,
                                                             uint32_t *__restrict__ tbl,
                                                             const int2 *__restrict__ range,
                                                             uint2 *__restrict__ part, uint2 *__restrict__ part_rows,
                                                             int *__restrict__ bad) {
    __shared__ uint32_t s_off[kWin];     // next free record of the (bucket, chunk) run
    __shared__ uint32_t s_cnt[kWin];     // tile: records per bucket, then the tile-local start
    __shared__ uint2 s_stage[kSplitTile];
    __shared__ uint16_t s_slot[kSplitTile];
    __shared__ uint32_t s_wsum[kSplitTpb / 64];
    const int tid = threadIdx.x, c = blockIdx.x;
    const int64_t e0 = (int64_t)c * chunk, e1 = e0 + chunk < E ? e0 + chunk : E;
    const int w0 = gi_window_start(keys.key(e0, nullptr));
    const int2 r = range[c];
    const int n_r = r.y >= r.x ? r.y - r.x + 1 : 0;         // touched window slots
    const int per = (n_r + kSplitTpb - 1) / kSplitTpb;      // ... per thread in the scans (1 for a collated batch)
    for (int d = r.x + tid; d <= r.y; d += kSplitTpb) {
        s_off[d] = tbl[(size_t)(w0 + d) * n_chunks + c];
        s_cnt[d] = 0;
    }
    __syncthreads();
    // the next tile's ids are loaded while the current tile goes through its LDS phases; the carried rows of
    // the current tile are loaded at its start and used in its last phase
    static_assert(kSplitR == 4, "a lane takes four consecutive edges of a tile");
    typename K::Raw nraw;
    keys.load_raw(e0 + 4 * tid, e1, nraw);
    for (int64_t t0 = e0; t0 < e1; t0 += kSplitTile) {
        uint2 rec[kSplitR];
        f32x4 rraw[ROWS ? kSplitR : 1];
        uint32_t slot[kSplitR], rk[kSplitR];
        const int64_t et = t0 + 4 * tid;
        {
            uint32_t nk[kSplitR], np[kSplitR], nv[kSplitR];
            keys.decode(nraw, et, nk, np, nv, bad);
            keys.load_raw(et + kSplitTile, e1, nraw);
#pragma unroll
            for (int q = 0; q < kSplitR; ++q) {
                const int64_t e = et + q;
                slot[q] = 0xffffffffu;
                if (ROWS) rraw[q] = e < e1 ? keys.row_raw(e) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < e1) {
                    const uint32_t v = nk[q], pl = np[q];
                    const uint32_t b = v >> kSH, low = v & (kBins - 1);
                    const uint32_t d = b - (uint32_t)w0;
                    rec[q] = uint2{nv[q], (low << pbits) | pl};
                    if (d < (uint32_t)kWin) {
                        slot[q] = d;
                        rk[q] = atomicAdd(&s_cnt[d], 1u);
                    } else {   // outside the chunk's LDS window: cursor in the table itself, direct stores
                        const uint32_t pos = atomicAdd(&tbl[(size_t)b * n_chunks + c], 1u);
                        part[pos] = rec[q];
                        if (ROWS) {
                            const f32x4 rv = rraw[q];
                            part_rows[pos] = uint2{bf16x2_pack(rv[0], rv[1]), bf16x2_pack(rv[2], rv[3])};
                        }
                    }
                }
            }
        }
        __syncthreads();
        // tile-local starts: exclusive scan of the counts over the touched range
        const int d0 = r.x + tid * per;
        uint32_t mine = 0;
#pragma unroll 1
        for (int i = 0; i < per; ++i)
            if (d0 + i <= r.y) mine += s_cnt[d0 + i];
        uint32_t tile_n;
        uint32_t run = gi_block_exscan(mine, s_wsum, &tile_n);
#pragma unroll 1
        for (int i = 0; i < per; ++i)
            if (d0 + i <= r.y) {
                const uint32_t cn = s_cnt[d0 + i];
                s_cnt[d0 + i] = run;
                run += cn;
            }
        __syncthreads();
        uint32_t sp[kSplitR];
#pragma unroll
        for (int q = 0; q < kSplitR; ++q)
            if (slot[q] != 0xffffffffu) {
                sp[q] = s_cnt[slot[q]] + rk[q];
                s_stage[sp[q]] = rec[q];
                s_slot[sp[q]] = (uint16_t)slot[q];
            }
        __syncthreads();
        for (uint32_t j = tid; j < tile_n; j += kSplitTpb) {
            const uint32_t d = s_slot[j];
            part[s_off[d] - s_cnt[d] + j] = s_stage[j];
        }
        if (ROWS) {   // the rows through the same staging image
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kSplitR; ++q)
                if (slot[q] != 0xffffffffu) {
                    const f32x4 rv = rraw[q];
                    s_stage[sp[q]] = uint2{bf16x2_pack(rv[0], rv[1]), bf16x2_pack(rv[2], rv[3])};
                }
            __syncthreads();
            for (uint32_t j = tid; j < tile_n; j += kSplitTpb) {
                const uint32_t d = s_slot[j];
                part_rows[s_off[d] - s_cnt[d] + j] = s_stage[j];
            }
        }
        __syncthreads();
        // advance the run cursors by the tile's counts (count = next start - own start; the last
        // touched entry ends at tile_n), then clear the counters for the next tile
#pragma unroll 1
        for (int i = 0; i < per; ++i)
            if (d0 + i <= r.y) {
                const int d = d0 + i;
                s_off[d] += (d == r.y ? tile_n : s_cnt[d + 1]) - s_cnt[d];
            }
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < per; ++i)
            if (d0 + i <= r.y) s_cnt[d0 + i] = 0;
        __syncthreads();
    }
}

// step 4 outputs.  kValMask: bits of a record's value that order it; kPos: the kernel keeps the
// slot -> partition position map (the carried rows are fetched from part_rows at write-out)
template <bool ROWS>
struct OutCsr {   // first sort
    static constexpr uint32_t kValMask = 0x7fffffffu;
    static constexpr bool kPos = ROWS;
    int32_t *perm, *tgt, *src, *rowptr;
    uint32_t pmask;
    uint8_t *label_csr;
    const uint2 *part_rows;
    uint16_t *rows_csr;
    int rows_out_stride;
    __device__ __forceinline__ void ranked(uint32_t k, uint32_t val, uint32_t hi, uint32_t pos) const {
        perm[k] = (int32_t)(val & kValMask);
        src[k] = (int32_t)(hi & pmask);
        if (label_csr) label_csr[k] = (uint8_t)(val >> 31);
        if (ROWS) *reinterpret_cast<uint2 *>(rows_csr + (size_t)k * rows_out_stride) = part_rows[pos];
    }
    __device__ __forceinline__ void slot(uint32_t k, uint32_t node) const { tgt[k] = (int32_t)node; }
};
struct OutSrc {   // second sort
    static constexpr uint32_t kValMask = 0xffffffffu;
    static constexpr bool kPos = false;
    int32_t *spos, *spos_inv, *rowptr;
    __device__ __forceinline__ void ranked(uint32_t k, uint32_t val, uint32_t, uint32_t) const {
        spos[k] = (int32_t)val;
        if (spos_inv) spos_inv[val] = (int32_t)k;
    }
    __device__ __forceinline__ void slot(uint32_t, uint32_t) const {}
};

template <class O>
__global__ __launch_bounds__(kSortTpb) void gi_bucket_sort_kernel(const uint2 *__restrict__ part,
                                                                  const uint32_t *__restrict__ base_arr, int NB,
                                                                  int64_t N, int64_t E, int pbits, int chunk, O out) {
    constexpr uint32_t VM = O::kValMask;
    __shared__ uint32_t s_hist[kBins], s_start[kBins], s_wsum[kBins / 64];
    __shared__ uint2 s_rec[kCap];
    __shared__ uint16_t s_pos[O::kPos ? kCap : 1];
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t base = base_arr[b], M = base_arr[b + 1] - base;
    const uint32_t node0 = (uint32_t)b << kSH;
    const bool fits = M <= (uint32_t)kCap;
    if (tid < kBins) s_hist[tid] = 0;
    __syncthreads();
    // the bucket's records: all loads of a thread in flight at once, kept in registers until placed
    uint2 rec[kSortR];
    uint32_t ord[kSortR];
    if (fits) {
#pragma unroll
        for (int q = 0; q < kSortR; ++q) {
            const uint32_t i = tid + q * kSortTpb;
            rec[q] = i < M ? part[base + i] : uint2{0u, 0u};
        }
#pragma unroll
        for (int q = 0; q < kSortR; ++q)
            if (tid + q * kSortTpb < M) ord[q] = atomicAdd(&s_hist[rec[q].y >> pbits], 1u);
    } else {
        for (uint32_t i = tid; i < M; i += kSortTpb) atomicAdd(&s_hist[part[base + i].y >> pbits], 1u);
    }
    __syncthreads();
    // exclusive scan of the 256 node counts (waves 0..3: shuffle scan + wave totals)
    uint32_t inc = 0, cnt_mine = 0;
    if (tid < kBins) {
        cnt_mine = s_hist[tid];
        inc = cnt_mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl(inc, lane - d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) s_wsum[wv] = inc;
    }
    __syncthreads();
    if (tid < kBins) {
        uint32_t pre = 0;
#pragma unroll
        for (int w = 0; w < kBins / 64; ++w) pre += w < wv ? s_wsum[w] : 0u;
        const uint32_t st = pre + inc - cnt_mine;
        s_start[tid] = st;
        if ((int64_t)node0 + tid < N) out.rowptr[node0 + tid] = (int32_t)(base + st);
    }
    if (b == NB - 1 && tid == 0) out.rowptr[N] = (int32_t)E;
    __syncthreads();
    if (fits) {
#pragma unroll
        for (int q = 0; q < kSortR; ++q)
            if (tid + q * kSortTpb < M) {
                const uint32_t sl = s_start[rec[q].y >> pbits] + ord[q];
                s_rec[sl] = rec[q];
                if (O::kPos) s_pos[sl] = (uint16_t)(tid + q * kSortTpb);
            }
        __syncthreads();
        // rank every record inside its node's list, then bring the image into the final order in place, so
        // that the arrays leave with consecutive lanes on consecutive positions (a wave's store touches a few
        // lines instead of up to 64)
        uint32_t dst[kSortR];
        uint16_t ps[O::kPos ? kSortR : 1];
#pragma unroll
        for (int q = 0; q < kSortR; ++q) {
            const uint32_t j = tid + q * kSortTpb;
            if (j < M) {
                const uint2 r = s_rec[j];
                const uint32_t low = r.y >> pbits, st = s_start[low], n = s_hist[low];
                uint32_t cnt = 0;
                for (uint32_t t = 0; t < n; ++t) cnt += (s_rec[st + t].x & VM) < (r.x & VM) ? 1u : 0u;
                rec[q] = r;
                dst[q] = st + cnt;
                if (O::kPos) ps[q] = s_pos[j];
                out.slot(base + j, node0 + low);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kSortR; ++q)
            if (tid + q * kSortTpb < M) {
                s_rec[dst[q]] = rec[q];
                if (O::kPos) s_pos[dst[q]] = ps[q];
            }
        __syncthreads();
        for (uint32_t j = tid; j < M; j += kSortTpb) {
            const uint2 r = s_rec[j];
            out.ranked(base + j, r.x, r.y, base + (O::kPos ? (uint32_t)s_pos[j] : 0u));
        }
    } else {
        // a bucket beyond the LDS capacity: groups of consecutive nodes whose lists fit are taken one
        // after the other (one pass over the bucket's records per group, served by the L2); a single
        // node beyond the capacity (a hub) is ranked against tiles of its own list streamed through
        // LDS - quadratic in the hub's degree, same result
        __shared__ uint32_t s_cur[kBins];
        // (with carried rows the last third of the record image holds 32-bit partition positions)
        constexpr uint32_t kCapG = O::kPos ? (uint32_t)kCap * 2 / 3 : (uint32_t)kCap;
        uint32_t *s_pos32 = reinterpret_cast<uint32_t *>(s_rec + kCapG);
        int g0 = 0;
        while (g0 < kBins) {
            const uint32_t st0 = s_start[g0];
            int g1 = g0;
            while (g1 < kBins && (g1 + 1 < kBins ? s_start[g1 + 1] : M) - st0 <= kCapG) ++g1;
            if (g1 == g0) {   // hub node g0: its list alone exceeds the LDS capacity
                // The bucket's region is a sequence of runs, one per CHUNK of the edge list, in chunk order (step 2's
                // scan lays (bucket, chunk) out chunk after chunk), and a record's value - an edge id / a CSR position -
                // says which chunk it came from: value / chunk.  So a hub record's rank is the number of hub records
                // in earlier runs (a histogram over the chunks) + the number of hub records of its OWN run with a
                // smaller value (a walk over that run only).  Until round 5 every hub record was compared with every
                // record of the bucket: O(M^2 / 512) per workgroup - seconds for a million-edge hub.
                const uint32_t n = s_hist[g0];
                uint32_t *h_all = reinterpret_cast<uint32_t *>(s_rec), *h_hub = h_all + kMaxChunks;   // (s_rec is free here)
                __syncthreads();
                for (int t = tid; t < 2 * kMaxChunks; t += kSortTpb) h_all[t] = 0u;
                __syncthreads();
                for (uint32_t i = tid; i < M; i += kSortTpb) {
                    const uint2 r = part[base + i];
                    const uint32_t ck = (r.x & VM) / (uint32_t)chunk;
                    atomicAdd(&h_all[ck], 1u);
                    if ((int)(r.y >> pbits) == g0) atomicAdd(&h_hub[ck], 1u);
                }
                __syncthreads();
                if (tid == 0) {   // exclusive scans over at most 512 chunks
                    uint32_t a_ = 0, h_ = 0;
                    for (int ck = 0; ck < kMaxChunks; ++ck) {
                        const uint32_t va = h_all[ck], vh = h_hub[ck];
                        h_all[ck] = a_;
                        h_hub[ck] = h_;
                        a_ += va;
                        h_ += vh;
                    }
                }
                __syncthreads();
                if (n <= (uint32_t)GNNTRK_GI_HUB_BITMAP_MIN) {   // a walk over the record's own run
                    for (uint32_t i = tid; i < M; i += kSortTpb) {
                        const uint2 r = part[base + i];
                        if ((int)(r.y >> pbits) != g0) continue;
                        const uint32_t ck = (r.x & VM) / (uint32_t)chunk;
                        const uint32_t r0 = h_all[ck], r1 = ck + 1 < (uint32_t)kMaxChunks ? h_all[ck + 1] : M;
                        uint32_t cnt = h_hub[ck];
                        for (uint32_t t = r0; t < r1; ++t) {
                            const uint2 q = part[base + t];
                            cnt += ((int)(q.y >> pbits) == g0 && (q.x & VM) < (r.x & VM)) ? 1u : 0u;
                        }
                        out.ranked(base + st0 + cnt, r.x, r.y, base + i);
                    }
                } else {
                    // a large hub: the values of a run are distinct ids of ONE chunk, so the hub records of a run
                    // are the set bits of a bitmap over the chunk's id range (spans of 128 K ids: 16 KB of LDS) and
                    // a record's rank inside its run is the number of set bits below its own - linear in the
                    // bucket's size (a 2 M-edge hub: 5 s with the walk above, milliseconds this way)
                    constexpr uint32_t kSpanW = 4096, kSpan = kSpanW * 32;
                    uint32_t *bm = h_all + 2 * kMaxChunks, *pw = bm + kSpanW;
                    static_assert((2 * kMaxChunks + 2 * 4096) * sizeof(uint32_t) <= sizeof(s_rec), "hub scratch exceeds s_rec");
                    for (int ck = 0; ck < kMaxChunks; ++ck) {
                        const uint32_t r0 = h_all[ck], r1 = ck + 1 < kMaxChunks ? h_all[ck + 1] : M;
                        const uint32_t hub_here = (ck + 1 < kMaxChunks ? h_hub[ck + 1] : n) - h_hub[ck];
                        if (hub_here == 0u) continue;   // (uniform)
                        uint32_t before = h_hub[ck];
                        for (uint32_t lo = (uint32_t)ck * (uint32_t)chunk, sub = 0; sub < (uint32_t)chunk; sub += kSpan, lo += kSpan) {
                            __syncthreads();
                            for (uint32_t w = tid; w < kSpanW; w += kSortTpb) bm[w] = 0u;
                            __syncthreads();
                            for (uint32_t t = r0 + tid; t < r1; t += kSortTpb) {
                                const uint2 q = part[base + t];
                                const uint32_t d = (q.x & VM) - lo;
                                if ((int)(q.y >> pbits) == g0 && d < kSpan) atomicOr(&bm[d >> 5], 1u << (d & 31u));
                            }
                            __syncthreads();
                            // exclusive prefix of the words' bit counts: eight words per thread + a scan of the
                            // 512 thread sums through pw's upper half
                            uint32_t mine = 0;
                            constexpr uint32_t kPer = kSpanW / kSortTpb;
                            for (uint32_t j = 0; j < kPer; ++j) mine += (uint32_t)__popc(bm[tid * kPer + j]);
                            uint32_t *ts = pw;   // thread sums, scanned in place
                            ts[tid] = mine;
                            __syncthreads();
                            for (uint32_t off = 1; off < (uint32_t)kSortTpb; off <<= 1) {
                                const uint32_t v = tid >= (int)off ? ts[tid - off] : 0u;
                                __syncthreads();
                                ts[tid] += v;
                                __syncthreads();
                            }
                            const uint32_t excl = ts[tid] - mine, span_total = ts[kSortTpb - 1];
                            __syncthreads();
                            uint32_t run_ = excl;
                            for (uint32_t j = 0; j < kPer; ++j) {
                                const uint32_t w = tid * kPer + j, c_ = (uint32_t)__popc(bm[w]);
                                pw[w] = run_;
                                run_ += c_;
                            }
                            __syncthreads();
                            for (uint32_t t = r0 + tid; t < r1; t += kSortTpb) {
                                const uint2 q = part[base + t];
                                const uint32_t d = (q.x & VM) - lo;
                                if ((int)(q.y >> pbits) == g0 && d < kSpan) {
                                    const uint32_t w = d >> 5;
                                    const uint32_t below = (uint32_t)__popc(bm[w] & ((1u << (d & 31u)) - 1u));
                                    out.ranked(base + st0 + before + pw[w] + below, q.x, q.y, base + t);
                                }
                            }
                            before += span_total;
                        }
                    }
                }
                for (uint32_t j = tid; j < n; j += kSortTpb) out.slot(base + st0 + j, node0 + (uint32_t)g0);
                __syncthreads();   // (the histograms lived in s_rec: the next group refills it)
                g0 += 1;
                continue;
            }
            const uint32_t n_g = (g1 < kBins ? s_start[g1] : M) - st0;
            __syncthreads();
            if (tid < kBins) s_cur[tid] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < M; i += kSortTpb) {
                const uint2 r = part[base + i];
                const int low = (int)(r.y >> pbits);
                if (low >= g0 && low < g1) {
                    const uint32_t sl = s_start[low] - st0 + atomicAdd(&s_cur[low], 1u);
                    s_rec[sl] = r;
                    if (O::kPos) s_pos32[sl] = i;
                }
            }
            __syncthreads();
            for (uint32_t j = tid; j < n_g; j += kSortTpb) {
                const uint2 r = s_rec[j];
                const uint32_t low = r.y >> pbits, st = s_start[low] - st0, n = s_hist[low];
                uint32_t cnt = 0;
                for (uint32_t t = 0; t < n; ++t) cnt += (s_rec[st + t].x & VM) < (r.x & VM) ? 1u : 0u;
                out.ranked(base + st0 + st + cnt, r.x, r.y, base + (O::kPos ? s_pos32[j] : 0u));
                out.slot(base + st0 + j, node0 + low);
            }
            g0 = g1;
        }
    }
}

struct OwnPlan {
    bool ok;      // the own sort can run this shape
    bool dense;   // ... but most buckets would overflow the LDS capacity: the library form is faster
    int NB, n_chunks, chunk, bitsN;
    int64_t T;   // table entries (+ 1 total)
    int n_tiles;
    size_t off_range, off_base, off_tbl, off_sums, off_part, off_rows, bytes;
};

static int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n) ++b;
    return b;
}

static OwnPlan own_plan(int64_t N, int64_t E, bool rows) {
    OwnPlan p{};
    p.ok = false;
    if (N < 1 || E < 1) return p;
    p.bitsN = bits_for(N > 1 ? N : 2);
    if (p.bitsN + kSH > 32) return p;
    p.NB = (int)ceil_div(N, kBins);
    int nc = (int)ceil_div(E, 16384);
    if (nc > kMaxChunks) nc = kMaxChunks;
    int64_t chunk = ceil_div(E, nc);
    chunk = ceil_div(chunk, 4) * 4;   // (a lane of the count / split kernels takes four consecutive edges)
    p.chunk = (int)chunk;
    p.n_chunks = (int)ceil_div(E, chunk);
    p.T = (int64_t)p.NB * p.n_chunks;
    if (p.T > kMaxTable) return p;
    // very dense (multi)graphs: most buckets would overflow the LDS capacity of the bucket sort and take its tiled
    // path.  (The bar was 3/4 of the capacity until round 5: a kNN graph with k = 16 - 200 k hits, 3.07 M edges, at
    // most 4 096 records per bucket of 256 nodes - sat at 0.77 and took the library form: two radix sorts for a
    // graph whose every bucket fits.)
    p.dense = E / p.NB > kCap * 9 / 10;
    p.n_tiles = (int)ceil_div(p.T + 1, kScanTile);
    size_t o = 256;
    p.off_range = o;
    o += align_up((size_t)p.n_chunks * sizeof(int2), 256);
    p.off_base = o;
    o += align_up((size_t)(p.NB + 1) * 4, 256);
    p.off_tbl = o;
    o += align_up((size_t)(p.T + 1) * 4, 256);
    p.off_sums = o;
    o += align_up((size_t)p.n_tiles * 4, 256);
    p.off_part = o;
    o += align_up((size_t)E * 8, 256);
    p.off_rows = o;   // the carried rows in partition order
    if (rows) o += align_up((size_t)E * 8, 256);
    p.bytes = o;
    p.ok = true;
    return p;
}

template <class K, class O, bool ROWS>
static int own_sort(const OwnPlan &p, K keys, O out, int pbits, int64_t N, int64_t E, char *ws, int *bad,
                    hipStream_t stream) {
    int2 *range = reinterpret_cast<int2 *>(ws + p.off_range);
    uint32_t *base = reinterpret_cast<uint32_t *>(ws + p.off_base);
    uint32_t *tbl = reinterpret_cast<uint32_t *>(ws + p.off_tbl);
    uint32_t *sums = reinterpret_cast<uint32_t *>(ws + p.off_sums);
    uint2 *part = reinterpret_cast<uint2 *>(ws + p.off_part);
    int rc = check_hip(hipMemsetAsync(tbl, 0, (size_t)(p.T + 1) * 4, stream), "graph_index_build(memset)");
    if (rc) return rc;
    hipLaunchKernelGGL((gi_count_kernel<K>), dim3(p.n_chunks), dim3(kCountTpb), 0, stream, keys, E, p.chunk,
                       p.n_chunks, p.NB, tbl, range, bad);
    hipLaunchKernelGGL(gi_scan_sums_kernel, dim3(p.n_tiles), dim3(kScanTpb), 0, stream, tbl, p.T + 1, sums);
    hipLaunchKernelGGL(gi_scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, sums, p.n_tiles);
    hipLaunchKernelGGL(gi_scan_apply_kernel, dim3(p.n_tiles), dim3(kScanTpb), 0, stream, tbl, p.T + 1, sums);
    hipLaunchKernelGGL(gi_bases_kernel, dim3((int)ceil_div(p.NB + 1, kTpb)), dim3(kTpb), 0, stream, tbl, p.n_chunks,
                       p.NB, base);
    uint2 *part_rows = reinterpret_cast<uint2 *>(ws + p.off_rows);
    hipLaunchKernelGGL((gi_split_kernel<K, ROWS>), dim3(p.n_chunks), dim3(kSplitTpb), 0, stream, keys, E, p.chunk,
                       p.n_chunks, pbits, tbl, range, part, part_rows, bad);
    hipLaunchKernelGGL((gi_bucket_sort_kernel<O>), dim3(p.NB), dim3(kSortTpb), 0, stream, part, base, p.NB, N, E,
                       pbits, p.chunk, out);
    return GNNTRK_OK;
}

// -------------------------------------------- library path (shapes outside the own sort)
// (ids out of range are counted and clamped, as in the own form; `rank`: see KeysCoo)
__device__ __forceinline__ uint32_t gi_checked(int64_t v, int64_t N, const int32_t *rank, int *bad) {
    if (v < 0 || v >= N) {
        atomicAdd(bad, 1);
        v = v < 0 ? 0 : N - 1;
    }
    return rank ? (uint32_t)rank[v] : (uint32_t)v;
}
__global__ __launch_bounds__(kTpb) void gi_keys_kernel(const int64_t *__restrict__ ids, int64_t E,
                                                       int64_t N, const int32_t *__restrict__ rank,
                                                       uint32_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals,
                                                       int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * kTpb + threadIdx.x; e < E;
         e += (int64_t)gridDim.x * kTpb) {
        keys[e] = gi_checked(ids[e], N, rank, bad);
        vals[e] = (uint32_t)e;
    }
}

// src_sorted[k] = src[perm[k]]; also emits the keys/vals of the second (by-source) sort
__global__ __launch_bounds__(kTpb) void gi_gather_src_kernel(const int64_t *__restrict__ src,
                                                             const uint32_t *__restrict__ perm,
                                                             int64_t E, int64_t N, const int32_t *__restrict__ rank,
                                                             int32_t *__restrict__ src_s,
                                                             uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals,
                                                             int *__restrict__ bad) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < E;
         k += (int64_t)gridDim.x * kTpb) {
        const uint32_t s = gi_checked(src[perm[k]], N, rank, bad);
        src_s[k] = (int32_t)s;
        keys[k] = s;
        vals[k] = (uint32_t)k;
    }
}

// rowptr[n] = first position k with sorted[k] >= n  (sorted ascending, E entries)
__global__ __launch_bounds__(kTpb) void gi_rowptr_kernel(const uint32_t *__restrict__ sorted,
                                                         int64_t E, int64_t N,
                                                         int32_t *__restrict__ rowptr) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k <= E;
         k += (int64_t)gridDim.x * kTpb) {
        const int64_t prev = (k == 0) ? -1 : (int64_t)sorted[k - 1];
        const int64_t cur = (k == E) ? N : (int64_t)sorted[k];
        for (int64_t n = prev + 1; n <= cur && n <= N; ++n) rowptr[n] = (int32_t)k;
    }
}

// inv[spos[k]] = k
__global__ __launch_bounds__(kTpb) void gi_invert_kernel(const int32_t *__restrict__ spos, int64_t E,
                                                         int32_t *__restrict__ inv) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < E; k += (int64_t)gridDim.x * kTpb)
        inv[spos[k]] = (int32_t)k;
}

static int stream_grid(int64_t n) {
    int64_t g = ceil_div(n, kTpb);
    const int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

static size_t library_ws_bytes(int64_t E) {
    const size_t arr = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint32_t), 256);
    return 256 /* flags */ + 3 * arr + align_up(sort_pairs_temp_bytes(E), 256);
}

// (carry: bit 0 = carried rows, bit 1 = node_rank given: room for the renumbered int32 edge list behind the plan)
static size_t remap_arr_bytes(int64_t E) { return align_up((size_t)(E > 0 ? E : 1) * sizeof(int32_t), 256); }
size_t graph_index_ws_bytes(int64_t N, int64_t E, int carry) {
    const size_t lib = library_ws_bytes(E);
    const OwnPlan p = own_plan(N, E, (carry & 1) != 0);
    const size_t own = p.ok ? align_up(p.bytes, 256) + ((carry & 2) ? 2 * remap_arr_bytes(E) : 0) : 0;
    return own > lib ? own : lib;
}

// the carried per-edge inputs behind the library form: plain gathers through perm
__global__ __launch_bounds__(kTpb) void gi_gather_label_kernel(const uint8_t *__restrict__ label,
                                                               const int32_t *__restrict__ perm, int64_t E,
                                                               uint8_t *__restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * kTpb + threadIdx.x; k < E; k += (int64_t)gridDim.x * kTpb)
        out[k] = label[perm[k]] ? 1 : 0;
}

static int library_build(const int64_t *edge_index, const gnntrk_graph_index *o, char *p, int *bad, int64_t N,
                         int64_t E, const int32_t *rank, hipStream_t stream) {
    p += 256;
    const size_t arr = align_up((size_t)(E > 0 ? E : 1) * sizeof(uint32_t), 256);
    uint32_t *keys_a = reinterpret_cast<uint32_t *>(p);
    p += arr;
    uint32_t *vals_a = reinterpret_cast<uint32_t *>(p);
    p += arr;
    uint32_t *keys_b = reinterpret_cast<uint32_t *>(p);
    p += arr;
    void *temp = p;
    const size_t temp_bytes = sort_pairs_temp_bytes(E);
    const int bits = bits_for(N > 1 ? N : 2);
    const int grid = stream_grid(E + 1);
    const int64_t *src = edge_index, *tgt = edge_index + E;
    hipLaunchKernelGGL(gi_keys_kernel, dim3(grid), dim3(kTpb), 0, stream, tgt, E, N, rank, keys_a, vals_a, bad);
    // the sorted keys ARE the CSR targets: sort straight into the output array
    uint32_t *tgt_sorted = reinterpret_cast<uint32_t *>(o->tgt);
    int rc = sort_pairs_u32(keys_a, tgt_sorted, vals_a, reinterpret_cast<uint32_t *>(o->perm), E, bits, temp,
                            temp_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gi_rowptr_kernel, dim3(grid), dim3(kTpb), 0, stream, tgt_sorted, E, N, o->rowptr_t);
    hipLaunchKernelGGL(gi_gather_src_kernel, dim3(grid), dim3(kTpb), 0, stream, src,
                       reinterpret_cast<const uint32_t *>(o->perm), E, N, rank, o->src, keys_a, vals_a, bad);
    rc = sort_pairs_u32(keys_a, keys_b, vals_a, reinterpret_cast<uint32_t *>(o->spos), E, bits, temp, temp_bytes,
                        stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gi_rowptr_kernel, dim3(grid), dim3(kTpb), 0, stream, keys_b, E, N, o->rowptr_s);
    if (o->spos_inv)
        hipLaunchKernelGGL(gi_invert_kernel, dim3(grid), dim3(kTpb), 0, stream, o->spos, E, o->spos_inv);
    return GNNTRK_OK;
}

int graph_index_build(const int64_t *edge_index, const gnntrk_graph_index *o, const gnntrk_graph_index_carry *cy,
                      void *ws, size_t ws_bytes, int flags, hipStream_t stream) {
    if (!o) return fail(GNNTRK_EINVAL, "graph_index_build: NULL output descriptor");
    const int64_t E = o->n_edges, N = o->n_nodes;
    if (E < 0 || N < 0 || E > 0x7fffffff || N > 0x7fffffff)
        return fail(GNNTRK_EUNSUPPORTED, "graph_index_build: sizes must fit int32");
    if (!o->rowptr_t || !o->rowptr_s || (E > 0 && (!o->perm || !o->tgt || !o->src || !o->spos)))
        return fail(GNNTRK_EINVAL, "graph_index_build: NULL output array");
    if (E > 0 && !edge_index) return fail(GNNTRK_EINVAL, "graph_index_build: NULL edge_index");
    const uint8_t *label = cy ? cy->edge_label : nullptr;
    const float *rows = cy ? cy->edge_rows : nullptr;
    const int32_t *rank = cy ? cy->node_rank : nullptr;
    if (E > 0 && label && !cy->label_csr) return fail(GNNTRK_EINVAL, "graph_index_build: carried label without output");
    if (E > 0 && rows &&
        (!cy->rows_csr_bf16 || cy->rows_stride < 4 || cy->rows_stride % 4 != 0 || ((uintptr_t)rows & 15) != 0 ||
         cy->out_stride < 4 || cy->out_stride % 4 != 0 || ((uintptr_t)cy->rows_csr_bf16 & 7) != 0))
        return fail(GNNTRK_EINVAL,
                    "graph_index_build: carried rows must be 16-byte aligned fp32 [E, 4] with a row stride that is a "
                    "multiple of 4 floats; the bf16 output 8-byte aligned with a stride that is a multiple of 4");
    if (!ws || ws_bytes < graph_index_ws_bytes(N, E, (rows ? 1 : 0) | (rank ? 2 : 0)))
        return fail(GNNTRK_EINVAL, "graph_index_build: workspace too small");

    char *p = reinterpret_cast<char *>(ws);
    int *bad = reinterpret_cast<int *>(p);
    int rc = check_hip(hipMemsetAsync(bad, 0, 256, stream), "graph_index_build(memset)");
    if (rc) return rc;
    if (E > 0) {
        const OwnPlan plan = own_plan(N, E, rows != nullptr);
        if (plan.ok && !(flags & 1) && (!plan.dense || (flags & 2))) {
            const int vec = (((uintptr_t)edge_index | (uintptr_t)(edge_index + E)) & 15) == 0 &&
                            (!label || ((uintptr_t)label & 3) == 0);
            const KeysCoo k1{edge_index, edge_index + E, N, label, rows, rows ? cy->rows_stride : 0, vec, nullptr};
            const uint32_t pmask = (uint32_t)(((uint64_t)1 << plan.bitsN) - 1);
            const OutCsr<true> o1r{o->perm, o->tgt, o->src, o->rowptr_t, pmask, label ? cy->label_csr : nullptr,
                                   reinterpret_cast<const uint2 *>(p + plan.off_rows), rows ? cy->rows_csr_bf16 : nullptr,
                                   rows ? cy->out_stride : 0};
            const OutCsr<false> o1{o->perm, o->tgt, o->src, o->rowptr_t, pmask, label ? cy->label_csr : nullptr,
                                   nullptr, nullptr, 0};
            if (rank) {   // renumbered: one pass writes the translated int32 list, the sort reads that
                int32_t *src32 = reinterpret_cast<int32_t *>(p + align_up(plan.bytes, 256));
                int32_t *tgt32 = reinterpret_cast<int32_t *>(p + align_up(plan.bytes, 256) + remap_arr_bytes(E));
                const int vec_ids = (((uintptr_t)edge_index | (uintptr_t)(edge_index + E)) & 15) == 0;
                hipLaunchKernelGGL(gi_remap_kernel, dim3(stream_grid((E + 3) / 4)), dim3(kTpb), 0, stream, edge_index,
                                   edge_index + E, E, N, rank, vec_ids, src32, tgt32, bad);
                const KeysCoo32 k32{src32, tgt32, label, rows, rows ? cy->rows_stride : 0,
                                    (!label || ((uintptr_t)label & 3) == 0) ? 1 : 0};
                rc = rows ? own_sort<KeysCoo32, OutCsr<true>, true>(plan, k32, o1r, plan.bitsN, N, E, p, bad, stream)
                          : own_sort<KeysCoo32, OutCsr<false>, false>(plan, k32, o1, plan.bitsN, N, E, p, bad, stream);
            } else if (rows) {
                rc = own_sort<KeysCoo, OutCsr<true>, true>(plan, k1, o1r, plan.bitsN, N, E, p, bad, stream);
            } else {
                rc = own_sort<KeysCoo, OutCsr<false>, false>(plan, k1, o1, plan.bitsN, N, E, p, bad, stream);
            }
            if (rc) return rc;
            const KeysCsr k2{o->src};
            const OutSrc o2{o->spos, o->spos_inv, o->rowptr_s};
            rc = own_sort<KeysCsr, OutSrc, false>(plan, k2, o2, 0, N, E, p, bad, stream);
        } else {
            rc = library_build(edge_index, o, p, bad, N, E, rank, stream);
            if (rc) return rc;
            if (label)
                hipLaunchKernelGGL(gi_gather_label_kernel, dim3(stream_grid(E)), dim3(kTpb), 0, stream, label, o->perm, E,
                                   cy->label_csr);
            if (rows)
                rc = rows_to_bf16_launch(rows, 4, cy->rows_stride, o->perm, E, cy->rows_csr_bf16, cy->out_stride, stream);
        }
        if (rc) return rc;
    } else {
        rc = check_hip(hipMemsetAsync(o->rowptr_t, 0, (size_t)(N + 1) * 4, stream), "memset");
        if (rc) return rc;
        rc = check_hip(hipMemsetAsync(o->rowptr_s, 0, (size_t)(N + 1) * 4, stream), "memset");
        if (rc) return rc;
    }
    return check_launch("graph_index_build");
}

// ------------------------------------------------------------------ collation of per-event indices
// A collated batch is a list of disjoint graphs with contiguous id / edge ranges, so its stable target sort, its
// source sort and their inverse are the events' own, shifted: one streaming pass per event places a cached
// per-event index into the batch arrays (20 B/edge + 8 B/node read and written, + the carried label / row bytes)
// instead of sorting the batch again (gnntrk_graph_index_place).
__global__ __launch_bounds__(kTpb) void gi_place_kernel(gnntrk_graph_index part, gnntrk_graph_index batch, int32_t noff,
                                                        int32_t eoff, const uint8_t *__restrict__ lab_in,
                                                        uint8_t *__restrict__ lab_out, const uint2 *__restrict__ rows_in,
                                                        uint2 *__restrict__ rows_out, const int32_t *__restrict__ nperm_in,
                                                        int32_t *__restrict__ nperm_out, const int32_t *__restrict__ nrank_in,
                                                        int32_t *__restrict__ nrank_out) {
    const int64_t E = part.n_edges, N = part.n_nodes;
    const int64_t n4 = (E + 3) / 4;
    const bool vec = ((eoff & 3) == 0);   // (16-byte aligned destinations)
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kTpb) {
        const int64_t k = 4 * i;
        if (vec && k + 4 <= E) {
            auto mv = [&](const int32_t *src, int32_t *dst, int32_t add) {
                int4 v = *reinterpret_cast<const int4 *>(src + k);
                v.x += add;
                v.y += add;
                v.z += add;
                v.w += add;
                *reinterpret_cast<int4 *>(dst + eoff + k) = v;
            };
            mv(part.perm, batch.perm, eoff);
            mv(part.tgt, batch.tgt, noff);
            mv(part.src, batch.src, noff);
            mv(part.spos, batch.spos, eoff);
            if (part.spos_inv && batch.spos_inv) mv(part.spos_inv, batch.spos_inv, eoff);
            if (lab_in) *reinterpret_cast<uint32_t *>(lab_out + eoff + k) = *reinterpret_cast<const uint32_t *>(lab_in + k);
            if (rows_in) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rows_out[eoff + k + q] = rows_in[k + q];
            }
        } else {
            for (int q = 0; q < 4 && k + q < E; ++q) {
                const int64_t j = k + q;
                batch.perm[eoff + j] = part.perm[j] + eoff;
                batch.tgt[eoff + j] = part.tgt[j] + noff;
                batch.src[eoff + j] = part.src[j] + noff;
                batch.spos[eoff + j] = part.spos[j] + eoff;
                if (part.spos_inv && batch.spos_inv) batch.spos_inv[eoff + j] = part.spos_inv[j] + eoff;
                if (lab_in) lab_out[eoff + j] = lab_in[j];
                if (rows_in) rows_out[eoff + j] = rows_in[j];
            }
        }
    }
    for (int64_t n = (int64_t)blockIdx.x * kTpb + threadIdx.x; n <= N; n += (int64_t)gridDim.x * kTpb) {
        batch.rowptr_t[noff + n] = part.rowptr_t[n] + eoff;   // (entry N of an event = entry 0 of the next: same value)
        batch.rowptr_s[noff + n] = part.rowptr_s[n] + eoff;
        if (n < N && nperm_in) {
            nperm_out[noff + n] = nperm_in[n] + noff;
            nrank_out[noff + n] = nrank_in[n] + noff;
        }
    }
}
int graph_index_place(const gnntrk_graph_index *part, int64_t node_offset, int64_t edge_offset,
                      const gnntrk_graph_index *batch, const uint8_t *label_part, uint8_t *label_batch,
                      const uint16_t *rows_part, uint16_t *rows_batch, const int32_t *node_perm_part,
                      int32_t *node_perm_batch, const int32_t *node_rank_part, int32_t *node_rank_batch,
                      hipStream_t stream) {
    if (!part || !batch) return fail(GNNTRK_EINVAL, "graph_index_place: NULL descriptor");
    if (node_offset < 0 || edge_offset < 0 || node_offset + part->n_nodes > batch->n_nodes ||
        edge_offset + part->n_edges > batch->n_edges || batch->n_edges > 0x7fffffff || batch->n_nodes > 0x7fffffff)
        return fail(GNNTRK_EINVAL, "graph_index_place: the event does not fit the batch at these offsets");
    if ((label_part != nullptr) != (label_batch != nullptr) || (rows_part != nullptr) != (rows_batch != nullptr) ||
        (node_perm_part != nullptr) != (node_perm_batch != nullptr) || (node_perm_part != nullptr) != (node_rank_part != nullptr) ||
        (node_rank_part != nullptr) != (node_rank_batch != nullptr))
        return fail(GNNTRK_EINVAL, "graph_index_place: carried / node-order arrays must come in pairs");
    if (rows_part && ((((uintptr_t)rows_part | (uintptr_t)rows_batch) & 7) != 0))
        return fail(GNNTRK_EINVAL, "graph_index_place: carried rows are 8-byte rows");
    const int64_t work = part->n_edges / 4 > part->n_nodes ? part->n_edges / 4 : part->n_nodes;
    hipLaunchKernelGGL(gi_place_kernel, dim3(stream_grid(work + 1)), dim3(kTpb), 0, stream, *part, *batch,
                       (int32_t)node_offset, (int32_t)edge_offset, label_part, label_batch,
                       reinterpret_cast<const uint2 *>(rows_part), reinterpret_cast<uint2 *>(rows_batch), node_perm_part,
                       node_perm_batch, node_rank_part, node_rank_batch);
    return check_launch("graph_index_place");
}

// ------------------------------------------------------------------ node order (gnntrk_node_order)
// Per-event renumbering of the nodes by a caller-supplied key (one float per node: the hits' azimuth for
// tracking graphs, whose edges join hits of neighbouring azimuth): new ids = rank of (event, key, old id).
// One 64-bit radix sort of N pairs; the graph-index build then reads every node id through `rank`
// (gnntrk_graph_index_carry.node_rank), so neighbouring nodes of the graph get neighbouring rows - the
// gathers of x[src] and the source-sorted gradient stores become local.
__global__ __launch_bounds__(kTpb) void no_keys_kernel(const float *__restrict__ key, int64_t stride,
                                                       const int64_t *__restrict__ batch, int64_t n,
                                                       unsigned long long *__restrict__ keys,
                                                       uint32_t *__restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTpb) {
        uint32_t u = __float_as_uint(key[i * stride]);
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;   // order-preserving map of the floats (NaNs at the ends)
        const unsigned long long b = batch ? (unsigned long long)(uint32_t)batch[i] : 0ull;
        keys[i] = (b << 32) | u;
        vals[i] = (uint32_t)i;
    }
}
// the same as ONE 32-bit key where the event ids need at most eight bits: event in the top `ebits` bits, below it
// the top 32 - ebits bits of the key's order-preserving image (>= 24: sign, exponent and 15 mantissa bits - keys
// that agree that far keep their old order) - four radix passes over 4-byte keys instead of five over 8-byte ones
__global__ __launch_bounds__(kTpb) void no_keys32_kernel(const float *__restrict__ key, int64_t stride,
                                                         const int64_t *__restrict__ batch, int64_t n, int ebits,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTpb) {
        uint32_t u = __float_as_uint(key[i * stride]);
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
        const uint32_t b = (batch && ebits > 0) ? (uint32_t)batch[i] << (32 - ebits) : 0u;
        keys[i] = b | (ebits > 0 ? u >> ebits : u);
        vals[i] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(kTpb) void no_finish_kernel(const uint32_t *__restrict__ sorted, int64_t n,
                                                         int32_t *__restrict__ perm, int32_t *__restrict__ rank) {
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTpb) {
        const uint32_t v = sorted[i];
        perm[i] = (int32_t)v;
        rank[v] = (int32_t)i;
    }
}
// ---- own form (round 6): a counting sort on a QUANTISED key - locality is all the key has to buy (locality.py), so
// the per-event order need not resolve more than the 65 536 levels of
//     q = trunc((key - lo) * (65535 / (hi - lo))),   lo / hi = min / max of the key over the nodes of the call (fp32),
// and the pair (event, q) is one small integer: the two-level counting sort of the edge lists above takes it as a
// "node id" (records = nodes, value = old id, stable: nodes that share a level keep their old order).  One min / max
// pass, one key pass and the sort's own passes over N records instead of four radix passes of the library.
constexpr int kNoLevels = 1 << 16;
constexpr int kNoMaxEvents = 64;
// per EVENT: mm[2 b] = max of the INVERTED order-preserving image of the key (= the minimum), mm[2 b + 1] = max of the
// image; both start from a zero fill.  (Per event, not per call: an event then gets the same levels - hence the same
// order - whether it is ordered alone or inside a collated batch, which is what lets cached per-event indices be
// placed into a batch, gnntrk_graph_index_place.)
__global__ __launch_bounds__(kTpb) void no_minmax_kernel(const float *__restrict__ key, int64_t stride,
                                                         const int64_t *__restrict__ batch, int64_t n, int64_t n_events,
                                                         uint32_t *__restrict__ mm) {
    __shared__ uint32_t s_mm[2 * kNoMaxEvents];
    for (int t = threadIdx.x; t < 2 * kNoMaxEvents; t += kTpb) s_mm[t] = 0u;
    __syncthreads();
    for (int64_t i0 = (int64_t)blockIdx.x * kTpb; i0 < n; i0 += (int64_t)gridDim.x * kTpb) {
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < n;
        const float v = in ? key[i * stride] : 0.f;
        long long b = (in && batch) ? batch[i] : 0;
        b = b < 0 ? 0 : b >= n_events ? n_events - 1 : b;   // (counted as bad in the key pass)
        uint32_t u = __float_as_uint(v);
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
        const bool on = in && v == v;   // (NaN keys take the last level, they do not set the range)
        uint32_t nlo = on ? ~u : 0u, hi = on ? u : 0u;
        const int b0 = __shfl((int)b, 0);
        if (__ballot(in && (int)b != b0) == 0ull) {   // one event per wave (the rule): one LDS atomic pair per wave
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t l2 = (uint32_t)__shfl_xor((int)nlo, o), h2 = (uint32_t)__shfl_xor((int)hi, o);
                nlo = l2 > nlo ? l2 : nlo;
                hi = h2 > hi ? h2 : hi;
            }
            if ((threadIdx.x & 63) == 0) {
                atomicMax(&s_mm[2 * b0], nlo);
                atomicMax(&s_mm[2 * b0 + 1], hi);
            }
        } else if (on) {
            atomicMax(&s_mm[2 * (int)b], nlo);
            atomicMax(&s_mm[2 * (int)b + 1], hi);
        }
    }
    __syncthreads();
    // (maxima: the result does not depend on the order of the atomics)
    for (int t = threadIdx.x; t < 2 * kNoMaxEvents; t += kTpb)
        if (s_mm[t]) atomicMax(&mm[t], s_mm[t]);
}
__device__ __forceinline__ float no_unimage(uint32_t u) {
    return __uint_as_float((u >> 31) ? (u ^ 0x80000000u) : ~u);
}
__global__ __launch_bounds__(kTpb) void no_qkeys_kernel(const float *__restrict__ key, int64_t stride,
                                                        const int64_t *__restrict__ batch, int64_t n, int64_t n_events,
                                                        const uint32_t *__restrict__ mm, int32_t *__restrict__ q,
                                                        int *__restrict__ bad) {
    __shared__ float s_lo[kNoMaxEvents], s_scale[kNoMaxEvents];
    for (int t = threadIdx.x; t < kNoMaxEvents; t += kTpb) {
        const uint32_t ulo = ~mm[2 * t], uhi = mm[2 * t + 1];
        const bool any = t < n_events && ulo <= uhi;   // (no finite key in the event: one level)
        const float lo = any ? no_unimage(ulo) : 0.f, hi = any ? no_unimage(uhi) : 0.f;
        s_lo[t] = lo;
        s_scale[t] = hi > lo ? (float)(kNoLevels - 1) / (hi - lo) : 0.f;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kTpb + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTpb) {
        const float v = key[i * stride];
        long long b = batch ? batch[i] : 0;
        if (b < 0 || b >= n_events) {   // (an event id outside the stated count: counted, clamped - the caller checks)
            atomicAdd(bad, 1);
            b = b < 0 ? 0 : n_events - 1;
        }
        float t = (v - s_lo[b]) * s_scale[b];
        t = t < 0.f ? 0.f : t;
        const uint32_t lvl = (v != v || t >= (float)(kNoLevels - 1)) ? (uint32_t)(kNoLevels - 1) : (uint32_t)t;
        q[i] = (int32_t)((uint32_t)b * (uint32_t)kNoLevels + lvl);
    }
}
struct OutNode {   // the sort's outputs for the node order: perm[new] = old, rank[old] = new
    static constexpr uint32_t kValMask = 0xffffffffu;
    static constexpr bool kPos = false;
    int32_t *perm, *rank, *rowptr;
    __device__ __forceinline__ void ranked(uint32_t k, uint32_t val, uint32_t, uint32_t) const {
        perm[k] = (int32_t)val;
        rank[val] = (int32_t)k;
    }
    __device__ __forceinline__ void slot(uint32_t, uint32_t) const {}
};
// (event, level) pairs the own form takes: the table of the counting sort is bounded by it
constexpr int64_t kNoMaxKeys = (int64_t)kNoMaxEvents * kNoLevels;
static bool node_order_own(int64_t n, int64_t n_events, const int64_t *batch, OwnPlan &plan, int64_t &keys) {
    if (GNNTRK_NO_LIBRARY_SORT_OFF) return false;
    const int64_t ev = batch ? n_events : 1;
    if (ev < 1) return false;   // (event count not stated: the exact radix form)
    if (ev > kNoMaxEvents) return false;
    keys = ev * kNoLevels;
    if (keys > kNoMaxKeys) return false;
    plan = own_plan(keys, n, false);
    return plan.ok && !plan.dense;
}
size_t node_order_ws_bytes(int64_t n) {
    const size_t m = (size_t)(n > 0 ? n : 1);
    const size_t t64 = sort_pairs_u64_temp_bytes(n), t32 = sort_pairs_temp_bytes(n);
    const size_t lib = 2 * align_up(m * 8, 256) + 2 * align_up(m * 4, 256) + align_up(t64 > t32 ? t64 : t32, 256);
    // own form: min / max + flag words | quantised keys | row pointers of the (event, level) pairs | the sort's plan
    const OwnPlan p = own_plan(kNoMaxKeys, n > 0 ? n : 1, false);
    const size_t own = 1024 + align_up(m * 4, 256) + align_up((size_t)(kNoMaxKeys + 1) * 4, 256) + (p.ok ? p.bytes : 0);
    return lib > own ? lib : own;
}
int node_order(const float *key, int64_t key_stride, const int64_t *batch, int64_t n_events, int64_t n, int32_t *perm,
               int32_t *rank, void *ws, size_t ws_bytes, hipStream_t stream) {
    if (n < 0 || n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "node_order: sizes must fit int32");
    if (n == 0) return GNNTRK_OK;
    if (!key || key_stride < 1 || !perm || !rank) return fail(GNNTRK_EINVAL, "node_order: NULL argument");
    if (!ws || ws_bytes < node_order_ws_bytes(n)) return fail(GNNTRK_EINVAL, "node_order: workspace too small");
    char *p = reinterpret_cast<char *>(ws);
    const int grid = stream_grid(n);
    OwnPlan plan;
    int64_t n_keys = 0;
    if (node_order_own(n, n_events, batch, plan, n_keys)) {
        uint32_t *mm = reinterpret_cast<uint32_t *>(p);          // per event {~min image, max image}; [128]: bad event ids
        int32_t *q = reinterpret_cast<int32_t *>(p + 1024);
        int32_t *rowptr = reinterpret_cast<int32_t *>(p + 1024 + align_up((size_t)n * 4, 256));
        char *sort_ws = reinterpret_cast<char *>(rowptr) + align_up((size_t)(kNoMaxKeys + 1) * 4, 256);
        int rc = check_hip(hipMemsetAsync(mm, 0, 1024, stream), "node_order(memset)");
        if (rc) return rc;
        const int64_t ev = batch ? n_events : 1;
        hipLaunchKernelGGL(no_minmax_kernel, dim3(grid), dim3(kTpb), 0, stream, key, key_stride, batch, n, ev, mm);
        hipLaunchKernelGGL(no_qkeys_kernel, dim3(grid), dim3(kTpb), 0, stream, key, key_stride, batch, n, ev, mm, q,
                           reinterpret_cast<int *>(mm + 2 * kNoMaxEvents));
        KeysCsr keys{q};
        OutNode out{perm, rank, rowptr};
        rc = own_sort<KeysCsr, OutNode, false>(plan, keys, out, 0, n_keys, n, sort_ws, reinterpret_cast<int *>(mm + 2 * kNoMaxEvents), stream);
        if (rc) return rc;
        return check_launch("node_order");
    }
    const size_t k8 = align_up((size_t)n * 8, 256), v4 = align_up((size_t)n * 4, 256);
    unsigned long long *ka = reinterpret_cast<unsigned long long *>(p), *kb = reinterpret_cast<unsigned long long *>(p + k8);
    uint32_t *va = reinterpret_cast<uint32_t *>(p + 2 * k8), *vb = reinterpret_cast<uint32_t *>(p + 2 * k8 + v4);
    void *temp = p + 2 * k8 + 2 * v4;
    // sorted bits: the 32 of the key + what the event ids need (n_events <= 0: not stated - all 32)
    int ebits = batch ? 32 : 0;
    if (batch && n_events > 0) {
        ebits = 0;
        while (ebits < 32 && ((int64_t)1 << ebits) < n_events) ++ebits;
    }
    int rc;
    if (ebits <= 8) {   // one 32-bit key (see no_keys32_kernel)
        uint32_t *k32a = reinterpret_cast<uint32_t *>(ka), *k32b = reinterpret_cast<uint32_t *>(kb);
        hipLaunchKernelGGL(no_keys32_kernel, dim3(grid), dim3(kTpb), 0, stream, key, key_stride, batch, n, ebits, k32a, va);
        rc = sort_pairs_u32(k32a, k32b, va, vb, n, 32, temp, sort_pairs_temp_bytes(n), stream);
    } else {
        hipLaunchKernelGGL(no_keys_kernel, dim3(grid), dim3(kTpb), 0, stream, key, key_stride, batch, n, ka, va);
        rc = sort_pairs_u64_bits(ka, kb, va, vb, n, 32 + ebits, temp, sort_pairs_u64_temp_bytes(n), stream);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(no_finish_kernel, dim3(grid), dim3(kTpb), 0, stream, vb, n, perm, rank);
    return check_launch("node_order");
}

}  // namespace gnntrk
