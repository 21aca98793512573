// gfx950 wave64 execution-model emulator for CPU-side runs of the kernel sources.
//
// TEST INFRASTRUCTURE ONLY.  The product kernels in gnn_tracking_amd/csrc/*.hip are
// written for hipcc --offload-arch=gfx950 and use HIP / __builtin_amdgcn_* directly.
// There is no GPU in the build container, so tests/emul/build_emul.py compiles the
// SAME source files with g++ (optionally -fsanitize=address,undefined) against this
// header, which stands in for <hip/hip_runtime.h>: every "lane" is a host thread,
// a workgroup is blockDim.x threads, wave-level builtins (MFMA 16x16x4 f32,
// shuffles, wave barrier) rendezvous the 64 threads of a wave and reproduce the
// documented gfx950 lane layouts bit-for-bit (MFMA = k-ordered fmaf chain,
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]).
// A divergent or out-of-bounds kernel deadlocks (-> timeout abort) or trips ASan
// here instead of on the GPU box.  Nothing in gnn_tracking_amd/ references this.
#pragma once

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }

// ------------------------------------------------------------- execution model
namespace hipemul {

struct Barrier {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    unsigned n = 0, count = 0, gen = 0;
    void init(unsigned n_) {
        n = n_;
        count = 0;
        gen = 0;
        pthread_mutex_init(&mu, nullptr);
        pthread_cond_init(&cv, nullptr);
    }
    void destroy() {
        pthread_mutex_destroy(&mu);
        pthread_cond_destroy(&cv);
    }
    void wait(const char *what) {
        pthread_mutex_lock(&mu);
        unsigned g = gen;
        if (++count == n) {
            count = 0;
            ++gen;
            pthread_cond_broadcast(&cv);
        } else {
            while (g == gen) {
                timespec ts;
                clock_gettime(CLOCK_REALTIME, &ts);
                ts.tv_sec += 60;
                if (pthread_cond_timedwait(&cv, &mu, &ts) != 0 && g == gen) {
                    fprintf(stderr,
                            "hipemul: DEADLOCK at %s (block %u thread %u): lanes diverged "
                            "around a wave/block collective\n",
                            what, blockIdx.x, threadIdx.x);
                    abort();
                }
            }
        }
        pthread_mutex_unlock(&mu);
    }
};

struct Wave {
    Barrier bar;
    float A[16][4];
    float B[4][16];
    uint64_t xch[64];
};

struct Block {
    Barrier bar;
    std::vector<Wave> waves;
};

extern thread_local Block *cur_block;

inline Wave &wave() { return cur_block->waves[threadIdx.x >> 6]; }
inline int lane() { return threadIdx.x & 63; }

template <class F>
void launch(dim3 grid, dim3 block, F fn) {
    if (block.x % 64 != 0 || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "hipemul: unsupported launch geometry\n");
        abort();
    }
    for (unsigned b = 0; b < grid.x; ++b) {
        Block blk;
        blk.bar.init(block.x);
        blk.waves.resize(block.x / 64);
        for (auto &w : blk.waves) w.bar.init(64);
        std::vector<std::thread> th;
        th.reserve(block.x);
        for (unsigned t = 0; t < block.x; ++t) {
            th.emplace_back([&, t, b]() {
                threadIdx = {t, 0, 0};
                blockIdx = {b, 0, 0};
                blockDim = {block.x, 1, 1};
                gridDim = {grid.x, 1, 1};
                cur_block = &blk;
                fn();
            });
        }
        for (auto &x : th) x.join();
        for (auto &w : blk.waves) w.bar.destroy();
        blk.bar.destroy();
    }
}
}  // namespace hipemul

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ::hipemul::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

// -------------------------------------------------------------------- builtins
inline void __syncthreads() { hipemul::cur_block->bar.wait("__syncthreads"); }
inline void __builtin_amdgcn_wave_barrier() { hipemul::wave().bar.wait("wave_barrier"); }
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
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
inline int atomicMax(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED)) {
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
