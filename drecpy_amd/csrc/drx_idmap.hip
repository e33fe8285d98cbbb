// raw -> internal id map in first-appearance order (DRecPy/Dataset/mem_dataset.py:309-330:
// `pd.Categorical(col, categories=col.unique()).codes`), for int64 raw ids, bit-exact.
//   1. insert every row into an open-addressing table (atomicCAS on the key, atomicMin on the first row index)
//   2. flag rows that are the first appearance of their id; exclusive scan of the flags = category code
//   3. codes[r] = scan[first_row(raw[r])];  uniques[code] = raw id
// Integer-only, HBM/latency bound; the table has >= 2n slots (power of two).
#include <cstring>
#include <hip/hip_runtime.h>
#include "drx_scan.hpp"
#include "drx_common.hpp"

namespace drx {

constexpr long long kEmpty = (long long)0x8000000000000000ull;   // INT64_MIN cannot be a raw id

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
  return x;
}

__global__ void k_idmap_fill(long long *tkeys, int *tmin, size_t cap) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) {
    tkeys[i] = kEmpty;
    tmin[i] = 0x7FFFFFFF;
  }
}

__global__ void k_idmap_insert(const long long *__restrict__ raw, long long n, long long *tkeys, int *tmin, size_t cap_mask,
                               unsigned *slot_of) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    const long long key = raw[r];
    size_t s = mix64((uint64_t)key) & cap_mask;
    for (;;) {
      long long cur = tkeys[s];
      if (cur == kEmpty) cur = (long long)atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)kEmpty, (unsigned long long)key);
      if (cur == kEmpty || cur == key) break;
      s = (s + 1) & cap_mask;
    }
    atomicMin(&tmin[s], (int)r);
    slot_of[r] = (unsigned)s;
  }
}

__global__ void k_idmap_flag(long long n, const int *__restrict__ tmin, const unsigned *__restrict__ slot_of, int *flag) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x)
    flag[r] = tmin[slot_of[r]] == (int)r ? 1 : 0;
}

__global__ void k_idmap_codes(const long long *__restrict__ raw, long long n, const int *__restrict__ tmin,
                              const unsigned *__restrict__ slot_of, const int *__restrict__ flag, const int *__restrict__ scan,
                              int *codes, long long *uniques, int *n_unique) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    const int first = tmin[slot_of[r]];
    const int code = scan[first];
    codes[r] = code;
    if (flag[r]) uniques[code] = raw[r];
    if (r == n - 1) *n_unique = scan[r] + flag[r];
  }
}

static size_t table_cap(int64_t n) {
  size_t cap = 1024;
  while (cap < (size_t)n * 2) cap <<= 1;
  return cap;
}

struct IdmapLayout {
  long long *tkeys; int *tmin; unsigned *slot_of; int *flag; int *scan; void *temp; size_t temp_bytes; size_t cap;
};

static IdmapLayout idmap_layout(Carver &cv, int64_t n) {
  IdmapLayout L{};
  L.cap = table_cap(n);
  L.tkeys = cv.take<long long>(L.cap);
  L.tmin = cv.take<int>(L.cap);
  L.slot_of = cv.take<unsigned>((size_t)n);
  L.flag = cv.take<int>((size_t)n);
  L.scan = cv.take<int>((size_t)n);
  L.temp_bytes = scan_i32_temp_bytes((size_t)n);
  L.temp = cv.take<char>(L.temp_bytes);
  return L;
}

}  // namespace drx

extern "C" size_t drx_idmap_scratch_bytes(int64_t n) {
  if (n < 1) return 0;
  drx::Carver cv(nullptr, 0);
  (void)drx::idmap_layout(cv, n);
  return drx::align_up(cv.off, 256) + 256;
}

extern "C" int drx_idmap_build(const int64_t *raw, int64_t n, int32_t *codes, int64_t *uniques, int32_t *n_unique,
                               void *scratch, size_t scratch_bytes, void *stream) {
  using namespace drx;
  if (!raw || !codes || !uniques || !n_unique || !scratch || n < 1 || n > 0x7FFFFFF0ll) return DRX_EINVAL;
  Carver cv(scratch, scratch_bytes);
  IdmapLayout L = idmap_layout(cv, n);
  if (!cv.ok()) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_idmap_fill, dim3(2048), dim3(256), 0, st, L.tkeys, L.tmin, L.cap);
  hipLaunchKernelGGL(k_idmap_insert, dim3(blocks), dim3(256), 0, st, (const long long *)raw, (long long)n, L.tkeys, L.tmin,
                     L.cap - 1, L.slot_of);
  hipLaunchKernelGGL(k_idmap_flag, dim3(blocks), dim3(256), 0, st, (long long)n, L.tmin, L.slot_of, L.flag);
  const int rc = scan_i32(L.temp, L.temp_bytes, L.flag, L.scan, (size_t)n, false, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_idmap_codes, dim3(blocks), dim3(256), 0, st, (const long long *)raw, (long long)n, L.tmin, L.slot_of,
                     L.flag, L.scan, (int *)codes, (long long *)uniques, (int *)n_unique);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}
