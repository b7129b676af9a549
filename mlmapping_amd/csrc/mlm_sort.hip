// mlm_sort.hip — device radix sort used only on frames where the emulated hit container rehashes
// (at most ~20 times in a stream's life).  Kept in its own TU: rocPRIM is header-only and slow
// to compile.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

extern "C" size_t mlm_sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                              (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    return bytes;
}

extern "C" int mlm_sort_pairs_u64_u32(void *temp, size_t temp_bytes, const unsigned long long *kin,
                                      unsigned long long *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                                      hipStream_t stream) {
    return (int)rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, 64, stream);
}
