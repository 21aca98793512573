#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// out[0..63]: load with voffset in range; [64..127]: voffset 0xF0000000 (OOB); [128..191]: soffset pushes past num_records
__global__ void probe(const uint32_t* src, uint32_t nbytes, uint32_t* out, uint32_t* st, uint32_t st_bytes) {
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, 0, 0);
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 0xF0000000u + l * 4, 0, 0);
    out[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, nbytes - 128, 0);   // lanes >= 32 past the end via soffset
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, l * 16, 0, 0);                   // 1024 bytes: partially OOB if nbytes < 1024
    out[192 + l] = v[0] + v[3];
    u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (l & 1) ? 0xF0000000u : l * 8, 0, 0);
    out[256 + l] = w[0] ^ w[1];
    __amdgpu_buffer_rsrc_t s = __builtin_amdgcn_make_buffer_rsrc((void*)st, 0, st_bytes, 0x00020000);
    u32x2 d = {0xAAAA0000u + l, 0xBBBB0000u + l};
    __builtin_amdgcn_raw_buffer_store_b64(d, s, (l & 1) ? 0xF0000000u + l * 8 : l * 8, 0, 0);   // odd lanes dropped
    __builtin_amdgcn_raw_buffer_store_b32(0xCCCC0000u + l, s, l * 4, st_bytes - 64, 0);        // soffset: lanes >= 16 past the end
}
int main() {
    uint32_t *src, *out, *st;
    const uint32_t n = 200;  // 800 bytes
    hipMalloc(&src, 4096); hipMalloc(&out, 4096); hipMalloc(&st, 4096);
    uint32_t h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    hipMemset(out, 0xff, 4096); hipMemset(st, 0, 4096);
    probe<<<1, 64>>>(src, n * 4, out, st, 1024);
    hipDeviceSynchronize();
    uint32_t o[1024], s[1024];
    hipMemcpy(o, out, 4096, hipMemcpyDeviceToHost); hipMemcpy(s, st, 4096, hipMemcpyDeviceToHost);
    printf("in range: %u %u | OOB voffset: %u %u | soffset: lane0 %u (want %u) lane31 %u lane32 %u lane63 %u\n", o[0], o[63], o[64], o[127],
           o[128], 1000 + (n * 4 - 128) / 4, o[128 + 31], o[128 + 32], o[128 + 63]);
    printf("b128: lane0 %u (want %u) lane49 %u (want %u) lane50 %u lane63 %u (OOB -> 0)\n", o[192], 1000 + 1003, o[192 + 49], 1000 + 196 + 1000 + 199, o[192 + 50], o[192 + 63]);
    printf("b64 mixed: lane0 %u lane1 %u lane2 %u\n", o[256], o[257], o[258]);
    printf("store b64: [0]=%x [1]=%x [2]=%x [3]=%x [4]=%x  (odd lanes dropped -> 0)\n", s[0], s[1], s[2], s[3], s[4]);
    printf("store b32 soffset: [240]=%x [255]=%x [256]=%x (past the end -> dropped, 0)\n", s[240], s[255], s[256]);
    return 0;
}
