// Hardware check of the lane maps and rounding behaviour csrc/tile_bf16.h builds on
// (gfx950).  Stand-alone:  hipcc --offload-arch=gfx950 -O2 -o tools/_bin/probe_bf16 tools/probe_bf16.hip
// Each probe runs ONE wavefront, copies the raw registers back and compares them with a
// host model of the documented layout; prints PASS / FAIL per probe, exit code = failures.
//
//   1. v_mfma_f32_16x16x32_bf16:  A[i = l&15][k = 8(l>>4) + e], B[k = 8(l>>4) + e][j = l&15],
//      D[4(l>>4) + r][l&15]  (e = 0..7: four VGPRs of packed bf16, r = 0..3)
//   2. v_mfma_f32_16x16x16_bf16 (_1k): the same with k = 4(l>>4) + e, e = 0..3
//   3. v_cvt_pk_bf16_f32 rounds to nearest even (ties, carries into the exponent)
//   4. v_pk_max_i16(x, 0) is ReLU on packed bf16 (sign bit = int16 sign; -0.0 -> +0.0)
//   5. ds_read_b64_tr_b16: in each 16-lane group lane p supplies row p>>2, columns 4(p&3)..+3 of
//      a 4 x 16 block and receives column p of the four rows
//   6. x * min(p, 1) per 16-bit half (v_pk_min_i16 + v_pk_mul_lo_u16) gates x by "p is a
//      positive bf16" for ReLU outputs p (>= +0)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef short i16x4_hw __attribute__((ext_vector_type(4)));
typedef short i16x2_hw __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 99;                                                                 \
        }                                                                              \
    } while (0)

static uint16_t host_bf16(float f) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float host_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// A and B are handed over as [16][K] / [K][16] bf16 matrices in memory; the kernel loads
// them with the CLAIMED lane map, so a correct D = A x B confirms the map.
template <int K>
__global__ void mfma_probe(const uint16_t *A, const uint16_t *B, float *D) {
    const int l = threadIdx.x, g = l >> 4, c = l & 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (K == 32) {
        u32x4 a, b;
        for (int e = 0; e < 8; e += 2) {
            const int k = 8 * g + e;
            a[e / 2] = (uint32_t)A[c * K + k] | ((uint32_t)A[c * K + k + 1] << 16);
            b[e / 2] = (uint32_t)B[k * 16 + c] | ((uint32_t)B[(k + 1) * 16 + c] << 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                                      __builtin_bit_cast(bf16x8_hw, b), acc, 0, 0, 0);
    } else {
        u32x2 a, b;
        for (int e = 0; e < 4; e += 2) {
            const int k = 4 * g + e;
            a[e / 2] = (uint32_t)A[c * K + k] | ((uint32_t)A[c * K + k + 1] << 16);
            b[e / 2] = (uint32_t)B[k * 16 + c] | ((uint32_t)B[(k + 1) * 16 + c] << 16);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(i16x4_hw, a),
                                                        __builtin_bit_cast(i16x4_hw, b), acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}

__global__ void cvt_probe(const float *in, uint32_t *packed, uint32_t *relu, int n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const f32x2_hw v = {in[2 * i], in[2 * i + 1]};
    const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
    packed[i] = p;
    const i16x2_hw z = {0, 0};
    relu[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_hw, p), z));
}

__global__ void gate_probe(const uint32_t *x, const uint32_t *p, uint32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t one = 0x00010001u;
    asm volatile("" : "+v"(one));
    const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(i16x2_hw, p[i]),
                                                                              __builtin_bit_cast(i16x2_hw, one)));
    out[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_hw, x[i]) * __builtin_bit_cast(u16x2_hw, m));
}

// image X[16 rows][16 features] of distinct values in LDS; lane (g, p) points at
// X[4g + (p>>2)][4(p&3)] and must receive X[4g + 0..3][p]
__global__ void tr_probe(uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t X[16 * 16];
    const int l = threadIdx.x, g = l >> 4, p = l & 15;
    for (int i = l; i < 256; i += 64) X[i] = (uint16_t)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) i16x4_hw *lds_ptr;
    const uint16_t *src = &X[(4 * g + (p >> 2)) * 16 + 4 * (p & 3)];
    const i16x4_hw v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(__attribute__((address_space(3))) void *)src);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = (uint16_t)v[r];
}

template <int K>
static int run_mfma() {
    std::vector<uint16_t> A(16 * K), B(K * 16);
    uint32_t s = 12345u + K;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return host_bf16((float)((int)(s >> 20) % 17 - 8) * 0.25f);   // small exact values
    };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    uint16_t *dA, *dB;
    float *dD;
    CHECK(hipMalloc(&dA, A.size() * 2));
    CHECK(hipMalloc(&dB, B.size() * 2));
    CHECK(hipMalloc(&dD, 256 * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_probe<K>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(256);
    CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float ref = 0.f;
            for (int k = 0; k < K; ++k) ref += host_f32(A[i * K + k]) * host_f32(B[k * 16 + j]);
            if (D[i * 16 + j] != ref) ++bad;
        }
    printf("%s mfma_f32_16x16x%d_bf16 lane map (A[l&15][K/4*(l>>4)+e], D[4(l>>4)+r][l&15]): %d mismatches\n",
           bad ? "FAIL" : "PASS", K, bad);
    hipFree(dA), hipFree(dB), hipFree(dD);
    return bad ? 1 : 0;
}

int main() {
    int fails = 0;
    fails += run_mfma<32>();
    fails += run_mfma<16>();

    {   // rounding + ReLU
        std::vector<float> in;
        const float specials[] = {0.f, -0.f, 1.f, -1.f, 1.00390625f /* tie, even below */, 1.01171875f /* tie, odd below */,
                                  3.3895314e38f /* rounds to inf */, 1e-40f, -2.5f, 65504.f, 0.1f, -0.1f};
        for (float v : specials) in.push_back(v);
        uint32_t s = 99u;
        while (in.size() < 4096) {
            s = s * 1664525u + 1013904223u;
            uint32_t u = s & 0xbfffffffu;   // keep the exponent finite
            float f;
            memcpy(&f, &u, 4);
            if (std::isfinite(f)) in.push_back(f);
        }
        const int n_pairs = (int)in.size() / 2;
        float *dIn;
        uint32_t *dP, *dR;
        CHECK(hipMalloc(&dIn, in.size() * 4));
        CHECK(hipMalloc(&dP, n_pairs * 4));
        CHECK(hipMalloc(&dR, n_pairs * 4));
        CHECK(hipMemcpy(dIn, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_probe, dim3((n_pairs + 63) / 64), dim3(64), 0, 0, dIn, dP, dR, n_pairs);
        std::vector<uint32_t> P(n_pairs), R(n_pairs);
        CHECK(hipMemcpy(P.data(), dP, n_pairs * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(R.data(), dR, n_pairs * 4, hipMemcpyDeviceToHost));
        int bad_c = 0, bad_r = 0;
        for (int i = 0; i < n_pairs; ++i) {
            const uint16_t lo = host_bf16(in[2 * i]), hi = host_bf16(in[2 * i + 1]);
            if (P[i] != ((uint32_t)lo | ((uint32_t)hi << 16))) ++bad_c;
            auto relu = [](uint16_t h) { return (uint16_t)((h & 0x8000) ? 0 : h); };
            if (R[i] != ((uint32_t)relu(lo) | ((uint32_t)relu(hi) << 16))) ++bad_r;
        }
        printf("%s v_cvt_pk_bf16_f32 = round to nearest even, lo half first: %d mismatches of %d pairs\n",
               bad_c ? "FAIL" : "PASS", bad_c, n_pairs);
        printf("%s v_pk_max_i16(x, 0) = ReLU on packed bf16: %d mismatches\n", bad_r ? "FAIL" : "PASS", bad_r);
        fails += (bad_c != 0) + (bad_r != 0);

        // gate: x * min(p, 1) with p = ReLU outputs
        uint32_t *dO;
        CHECK(hipMalloc(&dO, n_pairs * 4));
        hipLaunchKernelGGL(gate_probe, dim3((n_pairs + 63) / 64), dim3(64), 0, 0, dP, dR, dO, n_pairs);
        std::vector<uint32_t> O(n_pairs);
        CHECK(hipMemcpy(O.data(), dO, n_pairs * 4, hipMemcpyDeviceToHost));
        int bad_g = 0;
        for (int i = 0; i < n_pairs; ++i) {
            auto half = [&](int sh) {
                const uint16_t x = (uint16_t)(P[i] >> sh), p = (uint16_t)(R[i] >> sh);
                return (uint16_t)(p != 0 ? x : 0);   // p is +0 or a positive bf16
            };
            if (O[i] != ((uint32_t)half(0) | ((uint32_t)half(16) << 16))) ++bad_g;
        }
        printf("%s x * min(p, 1) per half gates x by p > 0 (p a ReLU output): %d mismatches\n",
               bad_g ? "FAIL" : "PASS", bad_g);
        fails += bad_g != 0;
        hipFree(dIn), hipFree(dP), hipFree(dR), hipFree(dO);
    }

    {   // transpose read
        uint16_t *dO;
        CHECK(hipMalloc(&dO, 256 * 2));
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dO);
        std::vector<uint16_t> O(256);
        CHECK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (O[l * 4 + r] != (uint16_t)((4 * (l >> 4) + r) * 16 + (l & 15))) ++bad;
        printf("%s ds_read_b64_tr_b16: lane (g, p) <- X[4g + 0..3][p] from pointers at X[4g + (p>>2)][4(p&3)]: %d mismatches\n",
               bad ? "FAIL" : "PASS", bad);
        fails += bad != 0;
        hipFree(dO);
    }
    printf("probe_bf16: %d failing probe(s)\n", fails);
    return fails;
}
