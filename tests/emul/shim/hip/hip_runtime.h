// gfx950 wave64 execution-model emulator for CPU-side runs of the kernel sources.
//
// TEST INFRASTRUCTURE ONLY.  The product kernels in gnn_tracking_amd/csrc/*.hip are
// written for hipcc --offload-arch=gfx950 and use HIP / __builtin_amdgcn_* directly.
// There is no GPU in the build container, so tests/emul/build_emul.py compiles the
// SAME source files with g++ (optionally -fsanitize=address,undefined) against this
// header, which stands in for <hip/hip_runtime.h>: every "lane" is a fiber of the calling
// thread, a workgroup is blockDim.x of them, wave-level builtins (MFMA 16x16x4 f32,
// shuffles, wave barrier) rendezvous the 64 threads of a wave and reproduce the
// documented gfx950 lane layouts bit-for-bit (MFMA = k-ordered fmaf chain,
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]).
// A divergent or out-of-bounds kernel deadlocks (-> timeout abort) or trips ASan
// here instead of on the GPU box.  Nothing in gnn_tracking_amd/ references this.
#pragma once

#include <pthread.h>
#include <ucontext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
// clang's ext_vector_type(n) on 4-byte elements == gcc's vector_size(4n)
#define ext_vector_type(n) vector_size(4 * (n))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemul_idx {
    unsigned x, y, z;
};
extern thread_local hipemul_idx threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
typedef void *hipStream_t;
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };

inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
// (the emulated device: two workgroups of any kernel resident per CU)
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 2; return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {
    *v = 2; /* a 2-CU "device": keeps emulated grids small */
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// --------------------------------------------------------------- vector types
typedef float f32x4_emul __attribute__((vector_size(16)));
struct float4 {
    float x, y, z, w;
};
struct float2 {
    float x, y;
};
struct int2 {
    int x, y;
};
struct int4 {
    int x, y, z, w;
};
struct uint2 {
    unsigned x, y;
};
struct uint4 {
    unsigned x, y, z, w;
};
struct longlong2 {
    long long x, y;
};
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }

// ------------------------------------------------------------- execution model
// Every lane of a workgroup is a FIBER (ucontext) of the calling thread; a wave / workgroup
// collective that not all participants have reached yet switches to the next runnable lane.
// (One host thread per lane - the first version - spent nine tenths of a test run in futex and
// clone system calls.)  A full pass over the lanes without any progress is a deadlock: lanes
// diverged around a collective.
namespace hipemul {

struct Block;
extern thread_local Block *cur_block;
void fiber_yield();
void note_progress();
const char *&fiber_waiting_at();

struct Barrier {
    unsigned n = 0, count = 0, gen = 0;
    void init(unsigned n_) {
        n = n_;
        count = 0;
        gen = 0;
    }
    void destroy() {}
    void wait(const char *what) {
        const unsigned g = gen;
        if (++count == n) {
            count = 0;
            ++gen;
            note_progress();
            return;
        }
        fiber_waiting_at() = what;
        while (g == gen) fiber_yield();
    }
};

struct Wave {
    Barrier bar;
    float A[16][4];
    float B[4][16];
    float A32[16][32];
    float B32[32][16];
    uint64_t xch[64];
};

// A lane's saved context.  glibc's swapcontext() saves / restores the signal mask with two system
// calls per switch - nine tenths of an emulated test's run time; the plain build therefore switches
// with a dozen instructions of its own (callee-saved registers + stack pointer, emul_runtime.cpp).
// Sanitizer builds keep ucontext, which the sanitizers know how to follow.
#if defined(__SANITIZE_ADDRESS__) || !defined(__x86_64__)
#define HIPEMUL_UCONTEXT 1
typedef ucontext_t FiberCtx;
#else
#define HIPEMUL_UCONTEXT 0
struct FiberCtx {
    void *sp = nullptr;
};
#endif

struct Fiber {
    FiberCtx ctx;
    bool done = false;
    const char *waiting_at = "";
};

struct Block {
    Barrier bar;
    std::vector<Wave> waves;
    std::vector<Fiber> fibers;
    FiberCtx sched;
    unsigned cur = 0;
    unsigned long progress = 0;
    const std::function<void()> *fn = nullptr;
    dim3 grid, block;
    unsigned bx = 0, by = 0;
};

inline Wave &wave() { return cur_block->waves[threadIdx.x >> 6]; }
inline int lane() { return threadIdx.x & 63; }

void run_block(Block &blk);  // emul_runtime.cpp

template <class F>
void launch(dim3 grid, dim3 block, F fn) {
    if (block.x % 64 != 0 || block.y != 1 || block.z != 1 || grid.z != 1) {
        fprintf(stderr, "hipemul: unsupported launch geometry\n");
        abort();
    }
    const std::function<void()> f = fn;
    Block blk;
    blk.fn = &f;
    blk.grid = grid;
    blk.block = block;
    blk.waves.resize(block.x / 64);
    blk.fibers.resize(block.x);
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned b = 0; b < grid.x; ++b) {
            blk.bx = b;
            blk.by = by;
            run_block(blk);
        }
}
}  // namespace hipemul

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ::hipemul::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

// -------------------------------------------------------------------- builtins
inline void __syncthreads() { hipemul::cur_block->bar.wait("__syncthreads"); }
inline void __builtin_amdgcn_wave_barrier() { hipemul::wave().bar.wait("wave_barrier"); }
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline void __builtin_amdgcn_sched_barrier(int) {}   // (a scheduling fence for the device compiler: nothing to model)
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
// cache-policy hints of the device compiler: plain accesses here
template <class T> inline T __builtin_nontemporal_load(const T *p) { return *p; }
template <class T> inline void __builtin_nontemporal_store(T v, T *p) { *p = v; }
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)

// v_mfma_f32_16x16x4_f32: D = A(16x4) * B(4x16) + C, exact f32 fmaf chain in k order
inline f32x4_emul __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emul c, int,
                                                       int, int) {
    hipemul::Wave &w = hipemul::wave();
    const int l = hipemul::lane();
    w.A[l & 15][l >> 4] = a;
    w.B[l >> 4][l & 15] = b;
    w.bar.wait("mfma");
    f32x4_emul d;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.A[row][k], w.B[k][col], acc);
        d[r] = acc;
    }
    w.bar.wait("mfma");
    return d;
}

// ---- bf16 primitives of gnn_tracking_amd/csrc/tile_bf16.h (host models) ---------------
#define GNNTRK_BF16_PRIMITIVES 1
namespace gnntrk {
typedef uint32_t u32x4_emul __attribute__((vector_size(16)));
typedef uint32_t u32x2_emul __attribute__((vector_size(8)));
inline float hipemul_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t hipemul_f32_to_bf16(float f) {  // round to nearest even (v_cvt_pk_bf16_f32)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline uint32_t bf16x2_pack(float lo, float hi) {
    return (uint32_t)hipemul_f32_to_bf16(lo) | ((uint32_t)hipemul_f32_to_bf16(hi) << 16);
}
inline uint32_t i16x2_max(uint32_t a, uint32_t b) {
    const int16_t l = std::max((int16_t)(a & 0xffff), (int16_t)(b & 0xffff));
    const int16_t h = std::max((int16_t)(a >> 16), (int16_t)(b >> 16));
    return (uint32_t)(uint16_t)l | ((uint32_t)(uint16_t)h << 16);
}
inline uint32_t i16x2_min(uint32_t a, uint32_t b) {
    const int16_t l = std::min((int16_t)(a & 0xffff), (int16_t)(b & 0xffff));
    const int16_t h = std::min((int16_t)(a >> 16), (int16_t)(b >> 16));
    return (uint32_t)(uint16_t)l | ((uint32_t)(uint16_t)h << 16);
}
inline uint32_t u16x2_mul(uint32_t a, uint32_t b) {
    const uint16_t l = (uint16_t)((a & 0xffff) * (b & 0xffff));
    const uint16_t h = (uint16_t)((a >> 16) * (b >> 16));
    return (uint32_t)l | ((uint32_t)h << 16);
}
// v_mfma_f32_16x16x{32,16}_bf16: A[i=l&15][k=(K/4)(l>>4)+e], B[k][j=l&15], D[4(l>>4)+r][l&15];
// products exact in f32, accumulated in k order (the hardware's internal order is not
// documented: tests compare with a tolerance)
template <int K, class V>
inline f32x4_emul hipemul_mfma_bf16(V a, V b, f32x4_emul c) {
    hipemul::Wave &w = hipemul::wave();
    const int l = hipemul::lane();
    for (int e = 0; e < K / 4; ++e) {
        const uint32_t aw = a[e >> 1], bw = b[e >> 1];
        const uint16_t ah = (e & 1) ? (uint16_t)(aw >> 16) : (uint16_t)aw;
        const uint16_t bh = (e & 1) ? (uint16_t)(bw >> 16) : (uint16_t)bw;
        w.A32[l & 15][(K / 4) * (l >> 4) + e] = hipemul_bf16_to_f32(ah);
        w.B32[(K / 4) * (l >> 4) + e][l & 15] = hipemul_bf16_to_f32(bh);
    }
    w.bar.wait("mfma_bf16");
    f32x4_emul d;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < K; ++k) acc += w.A32[row][k] * w.B32[k][col];
        d[r] = acc;
    }
    w.bar.wait("mfma_bf16");
    return d;
}
inline f32x4_emul mfma_bf16_k32(u32x4_emul a, u32x4_emul b, f32x4_emul c) {
    return hipemul_mfma_bf16<32>(a, b, c);
}
inline f32x4_emul mfma_bf16_k16(u32x2_emul a, u32x2_emul b, f32x4_emul c) {
    return hipemul_mfma_bf16<16>(a, b, c);
}
inline int opaque_zero() { return 0; }
inline uint32_t opaque_u32(uint32_t v) { return v; }
inline void drain_mfma() {}
inline void mfma_bf16_k16_acc(u32x2_emul a, u32x2_emul b, f32x4_emul &acc) {
    acc = hipemul_mfma_bf16<16>(a, b, acc);
}
// ds_read_b64_tr_b16 (tile_bf16.h documents the lane map)
inline u32x2_emul lds_read_tr16(const uint16_t *p) {
    hipemul::Wave &w = hipemul::wave();
    const int l = hipemul::lane();
    if (((uintptr_t)p & 7) != 0) {
        fprintf(stderr, "hipemul: ds_read_b64_tr_b16 address not 8-byte aligned\n");
        abort();
    }
    uint64_t mine;
    memcpy(&mine, p, 8);
    w.xch[l] = mine;
    w.bar.wait("tr16");
    const int base = l & ~15, pcol = l & 15;
    uint16_t out[4];
    for (int j = 0; j < 4; ++j) {
        const uint64_t chunk = w.xch[base + 4 * j + (pcol >> 2)];
        out[j] = (uint16_t)(chunk >> (16 * (pcol & 3)));
    }
    w.bar.wait("tr16");
    u32x2_emul r;
    r[0] = (uint32_t)out[0] | ((uint32_t)out[1] << 16);
    r[1] = (uint32_t)out[2] | ((uint32_t)out[3] << 16);
    return r;
}
// buffer addressing (tile_bf16.h): offset = voffset + soffset, out of range -> 0 / dropped, per dword
struct buf_rsrc_t {
    char *base;
    uint32_t n;
};
inline buf_rsrc_t buf_make(const void *base, uint32_t num_bytes) { return {(char *)base, num_bytes}; }
inline uint32_t hipemul_buf_dword(buf_rsrc_t r, uint64_t off) {
    if (off + 4 > r.n) return 0u;
    uint32_t v;
    memcpy(&v, r.base + off, 4);
    return v;
}
inline uint32_t buf_load_u32(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    return hipemul_buf_dword(r, (uint64_t)voff + soff);
}
inline u32x2_emul buf_load_u32x2(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    u32x2_emul v;
    for (int i = 0; i < 2; ++i) v[i] = hipemul_buf_dword(r, (uint64_t)voff + soff + 4 * i);
    return v;
}
inline u32x4_emul buf_load_u32x4(buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    u32x4_emul v;
    for (int i = 0; i < 4; ++i) v[i] = hipemul_buf_dword(r, (uint64_t)voff + soff + 4 * i);
    return v;
}
inline void buf_store_u32(uint32_t v, buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    const uint64_t off = (uint64_t)voff + soff;
    if (off + 4 <= r.n) memcpy(r.base + off, &v, 4);
}
inline void buf_store_u32x2(u32x2_emul v, buf_rsrc_t r, uint32_t voff, uint32_t soff) {
    const uint64_t off = (uint64_t)voff + soff;
    for (int i = 0; i < 2; ++i)
        if (off + 4 * i + 4 <= r.n) {
            const uint32_t w = v[i];
            memcpy(r.base + off + 4 * i, &w, 4);
        }
}
}  // namespace gnntrk

template <class T>
inline T hipemul_shfl(T v, int srclane) {
    static_assert(sizeof(T) <= 8, "shfl width");
    hipemul::Wave &w = hipemul::wave();
    const int l = hipemul::lane();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.xch[l] = raw;
    w.bar.wait("shfl");
    raw = w.xch[srclane & 63];
    w.bar.wait("shfl");
    T out;
    memcpy(&out, &raw, sizeof(T));
    return out;
}
template <class T>
inline T __shfl_xor(T v, int mask, int = 64) {
    return hipemul_shfl(v, hipemul::lane() ^ mask);
}
template <class T>
inline T __shfl(T v, int src, int = 64) {
    return hipemul_shfl(v, src);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int = 64) {
    const int l = hipemul::lane();
    return hipemul_shfl(v, (l + (int)d < 64) ? l + (int)d : l);
}
inline unsigned long long __ballot(int pred) {
    hipemul::Wave &w = hipemul::wave();
    const int l = hipemul::lane();
    w.xch[l] = pred ? 1 : 0;
    w.bar.wait("ballot");
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i)
        if (w.xch[i]) m |= 1ull << i;
    w.bar.wait("ballot");
    return m;
}

inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned *p, unsigned v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicMin(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED)) {
    }
    return old;
}
inline int atomicMax(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED)) {
    }
    return old;
}
inline unsigned atomicMax(unsigned *p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
inline float atomicAdd(float *p, float v) {
    float old = *p, nw;
    do {
        nw = old + v;
    } while (!__atomic_compare_exchange(p, &old, &nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}

inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned int v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline unsigned __float_as_uint(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(unsigned u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return v; }  // callers pass uniform values
inline unsigned __builtin_amdgcn_readlane(unsigned v, int srclane) { return hipemul_shfl(v, srclane); }   // (all lanes call it)
