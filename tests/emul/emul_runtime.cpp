// Emulator runtime: thread-local HIP index variables, host stable sort standing in
// for the rocPRIM translation unit (sort_pairs.hip).  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "host_util.h"

thread_local hipemul_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local hipemul::Block *hipemul::cur_block = nullptr;

namespace gnntrk {
size_t sort_pairs_temp_bytes(int64_t) { return 16; }
int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, int64_t n, int, void *, size_t, hipStream_t) {
    std::vector<int64_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return keys_in[a] < keys_in[b]; });
    std::vector<uint32_t> k(n), v(n);
    for (int64_t i = 0; i < n; ++i) {
        k[i] = keys_in[o[i]];
        v[i] = vals_in[o[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), vals_out);
    return 0;
}
size_t sort_pairs_u64_temp_bytes(int64_t) { return 16; }
int sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out,
                   const uint32_t *vals_in, uint32_t *vals_out, int64_t n, void *, size_t,
                   hipStream_t) {
    std::vector<int64_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return keys_in[a] < keys_in[b]; });
    std::vector<unsigned long long> k(n);
    std::vector<uint32_t> v(n);
    for (int64_t i = 0; i < n; ++i) {
        k[i] = keys_in[o[i]];
        v[i] = vals_in[o[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), vals_out);
    return 0;
}
}  // namespace gnntrk
