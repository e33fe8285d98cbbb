// Per-row masked top-k with the ordering of heapq.nlargest over (score, iid) tuples
// (DRecPy/Recommender/cdae.py:102-103, recommender_abc.py:460-461): descending score, ties by LARGER index.
// One workgroup per row; the row's candidates become 64-bit keys (ordered score bits << 32 | index) sorted by a
// bitonic network in LDS (up to 16384 keys = 128 KiB of the CU's 160 KiB).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>
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

// ---- rows longer than the LDS path: one device-wide segmented radix sort of the 64-bit keys (descending) ----------
__global__ void k_topk_keys(const float *__restrict__ scores, const uint32_t *__restrict__ mask, size_t total, int n,
                            unsigned long long *keys) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const bool ok = !mask || ((mask[i >> 5] >> (i & 31)) & 1u);
    keys[i] = ok ? (((unsigned long long)ordered_bits(scores[i]) << 32) | (unsigned)(i % n)) : 0ull;
  }
}

__global__ void k_topk_offsets(int R, int n, int *off) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= R; i += gridDim.x * blockDim.x) off[i] = i * n;
}

__global__ void k_topk_emit(const unsigned long long *__restrict__ sorted, const float *__restrict__ scores, int R, int n, int k,
                            int32_t *out_idx, float *out_val) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)R * k; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / k;
    const int j = (int)(i % k);
    const unsigned long long key = j < n ? sorted[r * n + j] : 0ull;
    if (key == 0ull) { out_idx[i] = -1; out_val[i] = -INFINITY; }
    else { const int idx = (int)(key & 0xFFFFFFFFull); out_idx[i] = idx; out_val[i] = scores[r * n + idx]; }
  }
}

struct TopkLayout { unsigned long long *keys, *sorted; int *off; void *temp; size_t temp_bytes; };

static TopkLayout topk_layout(Carver &cv, int R, int n) {
  TopkLayout L{};
  const size_t total = (size_t)R * n;
  L.keys = cv.take<unsigned long long>(total);
  L.sorted = cv.take<unsigned long long>(total);
  L.off = cv.take<int>(R + 1);
  L.temp_bytes = 0;
  unsigned long long *d = nullptr;
  int *o = nullptr;
  (void)rocprim::segmented_radix_sort_keys_desc(nullptr, L.temp_bytes, d, d, total, R, o, o, 0, 64, (hipStream_t)0);
  L.temp = cv.take<char>(L.temp_bytes);
  return L;
}

}  // namespace drx

extern "C" size_t drx_topk_scratch_bytes(int32_t R, int32_t n) {
  if (R < 1 || n <= 16384) return 0;
  drx::Carver cv(nullptr, 0);
  (void)drx::topk_layout(cv, R, n);
  return drx::align_up(cv.off, 256) + 256;
}

extern "C" int drx_topk(const float *scores, const uint32_t *cand_mask, int32_t R, int32_t n, int32_t k, int32_t *out_idx,
                        float *out_val, void *scratch, size_t scratch_bytes, void *stream) {
  if (!scores || !out_idx || !out_val || R < 1 || n < 1 || k < 1) return DRX_EINVAL;
  int npad = 2;
  while (npad < n) npad <<= 1;
  if (npad > 16384) {
    if ((long long)R * n > 0x7FFFFFFFll || !scratch) return DRX_EINVAL;
    drx::Carver cv(scratch, scratch_bytes);
    drx::TopkLayout L = drx::topk_layout(cv, R, n);
    if (!cv.ok()) return DRX_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)R * n;
    hipLaunchKernelGGL(drx::k_topk_keys, dim3(2048), dim3(256), 0, st, scores, cand_mask, total, n, L.keys);
    hipLaunchKernelGGL(drx::k_topk_offsets, dim3(64), dim3(256), 0, st, R, n, L.off);
    hipError_t e = rocprim::segmented_radix_sort_keys_desc(L.temp, L.temp_bytes, L.keys, L.sorted, total, R, L.off, L.off + 1, 0,
                                                           64, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(drx::k_topk_emit, dim3(1024), dim3(256), 0, st, L.sorted, scores, R, n, k, out_idx, out_val);
    DRX_LAUNCH_CHECK();
    return DRX_OK;
  }
  const size_t lds = (size_t)npad * sizeof(unsigned long long);
  DRX_HIP(hipFuncSetAttribute((const void *)drx::k_topk_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(drx::k_topk_lds, dim3(R), dim3(drx::kBlock), lds, (hipStream_t)stream, scores, cand_mask, n, k, npad,
                     out_idx, out_val);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}
