// Issue rate of v_mfma_f32_16x16x16_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950 (one wave per SIMD, four
// independent accumulators):  hipcc --offload-arch=gfx950 -O3 tools/probe_mfma16.hip -o tools/_bin/probe_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int K32>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) {
        a8[i] = (__bf16)(float)(threadIdx.x + i);
        b8[i] = (__bf16)(float)(i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (K32)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[u], 0, 0, 0);
            else
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
    float *out;
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int k32 = 0; k32 < 2; ++k32) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (k32) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA and SIMD\n", k32 ? "16x16x32" : "16x16x16", ms, iters * 4, ms * 1e6 / (iters * 4));
        }
    }
    return 0;
}
