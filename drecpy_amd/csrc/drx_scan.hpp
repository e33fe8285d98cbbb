// Device prefix sums of int32 lists (gfx950), written for the places that used rocprim::inclusive_scan / exclusive_scan: the id map
// (first-appearance ranks), the row-sharded index (distinct keys -> slots) and the touch list prepared in parts.  Three small launches:
// tile sums -> the spine (one workgroup scans the tile sums) -> tiles scanned again with their offsets.  The lists are a few hundred
// thousand to ten million entries, read twice and written once: a few microseconds of HBM time next to the sorts they sit beside.
// Integer sums: any order gives the same bits.  (The kernels are `static`: each translation unit that includes this carries its own.)
#pragma once
#include "drx_common.hpp"

namespace drx {

constexpr int kScanThreads = 256;
constexpr int kScanSub = 4;                                   // sub-tiles of kScanThreads x int4 per workgroup
constexpr int kScanTile = kScanThreads * 4 * kScanSub;        // 4096 items per workgroup

// inclusive scan of one int per thread over the workgroup (kScanThreads threads); returns this thread's inclusive value and the total
__device__ __forceinline__ int block_scan_incl(int v, int *lds /* [kScanThreads / 64] */, int &total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  __syncthreads();                                            // (lds may still be read from the previous call)
  if (lane == 63) lds[w] = v;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    const int s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  total = tot;
  return v + base;
}

static __global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const int *__restrict__ in, size_t n, int *__restrict__ tile_sum) {
  __shared__ int lds[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kScanSub; ++k) {
    const size_t i = base + (size_t)k * kScanThreads * 4 + (size_t)threadIdx.x * 4;
    if (i + 3 < n) { const int4 v = *reinterpret_cast<const int4 *>(in + i); s += v.x + v.y + v.z + v.w; }
    else for (size_t j = i; j < n && j < i + 4; ++j) s += in[j];
  }
  int total;
  (void)block_scan_incl(s, lds, total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// exclusive scan of the tile sums by ONE workgroup (in place)
static __global__ __launch_bounds__(1024) void k_scan_spine(int *__restrict__ tile_sum, int n_tiles) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_tiles; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int x = i < n_tiles ? tile_sum[i] : 0;
    int v = x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if (lane >= d) v += o;
    }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int off = carry_s;
    for (int k = 0; k < w; ++k) off += wsum[k];
    if (i < n_tiles) tile_sum[i] = off + v - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + v;
    __syncthreads();
  }
}

template <bool INCLUSIVE>
// (in / out carry no __restrict__: scan_i32 is called in place — every thread reads its int4 before it writes the same addresses)
static __global__ __launch_bounds__(kScanThreads) void k_scan_apply(const int *in, int *out, size_t n, const int *__restrict__ tile_off) {
  __shared__ int lds[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  int carry = tile_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanSub; ++k) {
    const size_t i = base + (size_t)k * kScanThreads * 4 + (size_t)threadIdx.x * 4;
    int x[4] = {0, 0, 0, 0};
    if (i + 3 < n) { const int4 v = *reinterpret_cast<const int4 *>(in + i); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
    else for (int j = 0; j < 4; ++j) if (i + j < n) x[j] = in[i + j];
    const int mine = x[0] + x[1] + x[2] + x[3];
    int total;
    const int incl = block_scan_incl(mine, lds, total);
    int run = carry + incl - mine;                            // exclusive prefix of this thread's first item
    int y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { y[j] = INCLUSIVE ? run + x[j] : run; run += x[j]; }
    if (i + 3 < n) *reinterpret_cast<int4 *>(out + i) = make_int4(y[0], y[1], y[2], y[3]);
    else for (int j = 0; j < 4; ++j) if (i + j < n) out[i + j] = y[j];
    carry += total;
  }
}

inline size_t scan_i32_temp_bytes(size_t n) { return align_up(((n + kScanTile - 1) / kScanTile + 1) * sizeof(int), 256); }

// out may alias in.  in / out must be 16-byte aligned (they are Carver allocations).
inline int scan_i32(void *temp, size_t temp_bytes, const int *in, int *out, size_t n, bool inclusive, hipStream_t st) {
  if (n == 0) return DRX_OK;
  if (!temp || temp_bytes < scan_i32_temp_bytes(n)) return DRX_ESCRATCH;
  if (((uintptr_t)in | (uintptr_t)out) & 15) return DRX_EINVAL;          // the kernels move int4
  const int n_tiles = (int)((n + kScanTile - 1) / kScanTile);
  int *ts = (int *)temp;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(kScanThreads), 0, st, in, n, ts);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, ts, n_tiles);
  if (inclusive) hipLaunchKernelGGL(k_scan_apply<true>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, out, n, ts);
  else hipLaunchKernelGGL(k_scan_apply<false>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, out, n, ts);
  return DRX_OK;
}

}  // namespace drx
