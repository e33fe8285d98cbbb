// Device radix sort of (uint32 key, uint32 val) pairs used to build the per-step inverted index of
// touched embedding rows (stable, so contributions are summed in sample order => deterministic).
// rocPRIM is the ROCm-native primitive library (not a CUDA shim); the sort is plumbing around the
// hand-written gather / segmented-reduce kernels, which carry the HBM traffic.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "drx_common.hpp"

namespace drx {

size_t sort_pairs_temp_bytes(size_t n, int end_bit) {
  size_t bytes = 0;
  const uint32_t *k = nullptr;
  uint32_t *ko = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, ko, k, ko, n, 0, end_bit, (hipStream_t)0);
  return bytes;
}

int sort_pairs(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
               uint32_t *vout, size_t n, int end_bit, hipStream_t stream) {
  if (n == 0) return 0;
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, end_bit, stream);
  return (int)e;
}

}  // namespace drx
