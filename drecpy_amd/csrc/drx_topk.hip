// Per-row masked top-k with the ordering of heapq.nlargest over (score, iid) tuples
// (DRecPy/Recommender/cdae.py:102-103, recommender_abc.py:460-461): descending score, ties by LARGER index.
// One workgroup per row; the row's candidates become 64-bit keys (ordered score bits << 32 | index) sorted by a
// bitonic network in LDS (up to 16384 keys = 128 KiB of the CU's 160 KiB).
#include "drx_common.hpp"

namespace drx {

__device__ __forceinline__ uint32_t ordered_bits(float f) {
  if (f == 0.0f) f = 0.0f;   // -0.0 == 0.0 in Python comparisons
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kBlock) void k_topk_lds(const float *__restrict__ scores, const uint32_t *__restrict__ mask,
                                                     int n, int k, int npad, int32_t *__restrict__ out_idx,
                                                     float *__restrict__ out_val) {
  extern __shared__ __align__(16) unsigned long long keys[];
  const size_t r = blockIdx.x;
  for (int i = threadIdx.x; i < npad; i += kBlock) {
    unsigned long long key = 0ull;
    if (i < n) {
      const size_t bit = r * (size_t)n + i;
      const bool ok = !mask || ((mask[bit >> 5] >> (bit & 31)) & 1u);
      if (ok) key = ((unsigned long long)ordered_bits(scores[bit]) << 32) | (unsigned)i;
    }
    keys[i] = key;
  }
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (npad >> 1); t += kBlock) {
        const int a = 2 * t - (t & (stride - 1));
        const int b = a + stride;
        const unsigned long long ka = keys[a], kb = keys[b];
        const bool desc = (a & size) == 0;
        if (desc ? (ka < kb) : (ka > kb)) { keys[a] = kb; keys[b] = ka; }
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += kBlock) {
    const unsigned long long key = j < npad ? keys[j] : 0ull;
    if (key == 0ull) {
      out_idx[r * (size_t)k + j] = -1;
      out_val[r * (size_t)k + j] = -INFINITY;
    } else {
      const int idx = (int)(key & 0xFFFFFFFFull);
      out_idx[r * (size_t)k + j] = idx;
      out_val[r * (size_t)k + j] = scores[r * (size_t)n + idx];
    }
  }
}

}  // namespace drx

extern "C" int drx_topk(const float *scores, const uint32_t *cand_mask, int32_t R, int32_t n, int32_t k, int32_t *out_idx,
                        float *out_val, void *stream) {
  if (!scores || !out_idx || !out_val || R < 1 || n < 1 || k < 1) return DRX_EINVAL;
  int npad = 2;
  while (npad < n) npad <<= 1;
  if (npad > 16384) return DRX_ENOTIMPL;   // larger rows: segmented device sort (planned)
  const size_t lds = (size_t)npad * sizeof(unsigned long long);
  DRX_HIP(hipFuncSetAttribute((const void *)drx::k_topk_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(drx::k_topk_lds, dim3(R), dim3(drx::kBlock), lds, (hipStream_t)stream, scores, cand_mask, n, k, npad,
                     out_idx, out_val);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}
