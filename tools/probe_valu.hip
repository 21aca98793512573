// Issue cost of the vector instructions the bf16 backward kernels are made of, alone and beside MFMAs (gfx950):
//   hipcc --offload-arch=gfx950 -O3 tools/probe_valu.hip -o tools/_bin/probe_valu
// One workgroup per CU, 1 or 2 waves per SIMD; every wave issues a stream of INDEPENDENT instructions (eight
// destination registers in rotation).  Prints ns per instruction and SIMD, and - for the mixed streams - per group of
// one v_mfma_f32_16x16x32_bf16 + k vector instructions: how many of them hide in the shadow of an MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define OPS8(STR)                                                                                  \
    asm volatile(STR " %0, %8, %9\n" STR " %1, %8, %9\n" STR " %2, %8, %9\n" STR " %3, %8, %9\n"   \
                 STR " %4, %8, %9\n" STR " %5, %8, %9\n" STR " %6, %8, %9\n" STR " %7, %8, %9\n"   \
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) \
                 : "v"(a), "v"(b))

template <int OP>
__device__ __forceinline__ void ops8(uint32_t (&r)[8], uint32_t a, uint32_t b) {
    if (OP == 0) OPS8("v_add_f32");
    if (OP == 1) OPS8("v_cvt_pk_bf16_f32");
    if (OP == 2) OPS8("v_pk_max_i16");
    if (OP == 3) OPS8("v_pk_min_i16");
    if (OP == 4) OPS8("v_pk_mul_lo_u16");
    if (OP == 5) OPS8("v_and_b32");
    if (OP == 6) OPS8("v_lshlrev_b32");
    if (OP == 7) OPS8("v_pk_add_f32");   // (64-bit operands: registers pairs - measured separately below)
    if (OP == 8) OPS8("v_mul_lo_u32");
    if (OP == 9) OPS8("v_pk_add_u16");
    if (OP == 10) OPS8("v_bfi_b32 %0, %8, %9, %9\n s_nop 0\n//");
}
static const char *kNames[] = {"v_add_f32", "v_cvt_pk_bf16_f32", "v_pk_max_i16", "v_pk_min_i16", "v_pk_mul_lo_u16",
                               "v_and_b32", "v_lshlrev_b32", "-", "v_mul_lo_u32", "v_pk_add_u16"};

template <int OP>
__global__ void valu_only(uint32_t *out, int iters) {
    uint32_t r[8] = {1, 2, 3, 4, 5, 6, 7, 8}, a = threadIdx.x | 0x3f800000u, b = 0x3f803f80u;
    for (int it = 0; it < iters; ++it) {
        ops8<OP>(r, a, b);
        ops8<OP>(r, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5] ^ r[6] ^ r[7];
}

// one MFMA (four accumulators in rotation) + K independent vector instructions per group
template <int OP, int K>
__global__ void mixed(uint32_t *out, int iters) {
    uint32_t r[8] = {1, 2, 3, 4, 5, 6, 7, 8}, a = threadIdx.x | 0x3f800000u, b = 0x3f803f80u;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) {
        a8[i] = (__bf16)(float)(threadIdx.x + i);
        b8[i] = (__bf16)(float)(i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a8), "v"(b8));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (OP == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[(u * K + k) & 7]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[(u * K + k) & 7]) : "v"(a), "v"(b));
                if (OP == 4) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r[(u * K + k) & 7]) : "v"(a), "v"(b));
                if (OP == 2) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(r[(u * K + k) & 7]) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] =
        r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5] ^ r[6] ^ r[7] ^ __float_as_uint(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
}

template <class F>
static float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    for (int tpb : {256, 512}) {
        printf("== %d waves per SIMD\n", tpb / 256);
#define RUN(OP)                                                                                                     \
    {                                                                                                               \
        const float ms = timed([&] { hipLaunchKernelGGL(valu_only<OP>, dim3(256), dim3(tpb), 0, 0, out, iters); }); \
        printf("%-20s %.2f ns per instruction and SIMD\n", kNames[OP], ms * 1e6 / (iters * 16.0 * (tpb / 256)));     \
    }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(8) RUN(9)
#undef RUN
#define MIX(OP, K)                                                                                                 \
    {                                                                                                              \
        const float ms = timed([&] { hipLaunchKernelGGL((mixed<OP, K>), dim3(256), dim3(tpb), 0, 0, out, iters / 4); }); \
        printf("mfma + %d x %-18s %.2f ns per group and SIMD\n", K, kNames[OP], ms * 1e6 / (iters / 4 * 8.0 * (tpb / 256))); \
    }
        MIX(0, 0) MIX(0, 1) MIX(0, 2) MIX(0, 3) MIX(0, 4) MIX(0, 6) MIX(1, 2) MIX(1, 3) MIX(1, 4) MIX(4, 2) MIX(4, 3) MIX(4, 4)
        MIX(2, 3) MIX(2, 4)
#undef MIX
    }
    return 0;
}
