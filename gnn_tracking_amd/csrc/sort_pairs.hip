// Stable device radix sort of (u32 key, u32 value) pairs: rocPRIM's LSD radix sort
// (AMD's own device primitive library; a plain library call like hipBLASLt for a
// plain GEMM - the hand-written kernels of this library are the gather/MLP/segment
// ones).  Kept in its own translation unit: it is the only rocPRIM user.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "host_util.h"

namespace gnntrk {

size_t sort_pairs_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    size_t bytes = 0;
    uint32_t *nil = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nil, nil, nil, nil, (size_t)n, 0u, 32u,
                                    (hipStream_t)0, false);
    return bytes;
}

int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, int64_t n, int end_bit, void *temp, size_t temp_bytes,
                   hipStream_t stream) {
    if (n <= 0) return GNNTRK_OK;
    size_t need = temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)n, 0u, (unsigned)end_bit, stream, false);
    return check_hip(e, "radix_sort_pairs");
}

size_t sort_pairs_u64_temp_bytes(int64_t n) {
    if (n <= 0) return 0;
    size_t bytes = 0;
    unsigned long long *nk = nullptr;
    uint32_t *nv = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nk, nk, nv, nv, (size_t)n, 0u, 64u,
                                    (hipStream_t)0, false);
    return bytes;
}

int sort_pairs_u64_bits(const unsigned long long *keys_in, unsigned long long *keys_out,
                        const uint32_t *vals_in, uint32_t *vals_out, int64_t n, int end_bit, void *temp,
                        size_t temp_bytes, hipStream_t stream) {
    if (n <= 0) return GNNTRK_OK;
    size_t need = temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)n, 0u, (unsigned)end_bit, stream, false);
    return check_hip(e, "radix_sort_pairs(u64)");
}
int sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, int64_t n, void *temp,
                   size_t temp_bytes, hipStream_t stream) {
    return sort_pairs_u64_bits(keys_in, keys_out, vals_in, vals_out, n, 64, temp, temp_bytes, stream);
}

}  // namespace gnntrk
