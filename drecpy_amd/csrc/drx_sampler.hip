// Device-side PointSampler for the throughput mode (counter-based draws instead of the MT19937 streams of
// DRecPy/Sampler/point_sampler.py:44-61 — same distribution, different stream; the bit-exact stream is the host
// sampler in drx_host.cpp):
//   with probability neg_ratio/(neg_ratio+1) a negative: uniform (u, i) with i NOT among the pairs the frame records for u
//   (mem_dataset.py:154-163; the CSR of recorded pairs, or the positives CSR where every recorded pair is a positive), else a positive:
//   uniform user with >= 1 positive, then a uniform positive of that user (mem_dataset.py:119-129).  Also emits keep_off = exclusive
//   scan of deg(uid[b]) for the step kernels — the sampler's workgroups sum their degrees, k_deg_apply_sub turns the sums into offsets,
//   any batch size — and, when the host passes a pinned mailbox, posts the total, tagged, straight into host memory with one
//   system-scope store: no copy kernel, no event, and no L2 write-back for the host's sake between training kernels.
#include <cstring>
#include <hip/hip_runtime.h>
#include "drx_common.hpp"

namespace drx {

__device__ __forceinline__ uint32_t bounded(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

// PR: the pairs the training set RECORDS, positives or not (point_sampler.py:56: a negative is a pair absent from the frame,
// whatever value a recorded pair carries) — the positives' CSR itself where every recorded pair is a positive
__global__ __launch_bounds__(kBlock) void k_point_sample(DrxHistory H, DrxHistory PR, int n_users, int n_items, int B, int neg_ratio,
                                                         uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *deg,
                                                         int32_t *keep_off, int *sub_sums, const float *__restrict__ values, float vmin,
                                                         float vrange) {
  __shared__ int wsum[kBlock / 64];
  const int b = blockIdx.x * kBlock + threadIdx.x;
  int my_deg = 0;
  if (b < B) {
  if (b == 0 && keep_off) keep_off[0] = 0;     // the scan below fills keep_off[1..B]
  const uint32_t r0 = hash_u32(seed, (uint32_t)b, 0u);
  const bool null_pair = ((double)r0 * (1.0 / 4294967296.0)) * (double)(neg_ratio + 1) > 1.0;
  int u = 0, i = 0;
  float val = 1.0f;
  uint32_t c = 1;
  for (int tries = 0; tries < 4096; ++tries) {
    u = (int)bounded(hash_u32(seed, (uint32_t)b, c++), (uint32_t)n_users);
    const int64_t s = H.indptr[u], e = H.indptr[u + 1];
    if (null_pair) {
      i = (int)bounded(hash_u32(seed, (uint32_t)b, c++), (uint32_t)n_items);
      const int64_t pe = PR.indptr[u + 1];
      int64_t lo = PR.indptr[u], hi = pe;      // lower_bound in the sorted row of recorded pairs
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (PR.indices[mid] < i) lo = mid + 1; else hi = mid;
      }
      if (lo == pe || PR.indices[lo] != i) break;
    } else {
      if (e == s) continue;
      const int64_t at = s + bounded(hash_u32(seed, (uint32_t)b, c++), (uint32_t)(e - s));
      i = H.indices[at];
      // value-carrying draw (DMF: the target is the pair's interaction value, standardised as recommender_abc.py:463-465)
      if (values) val = vrange > 0.f ? (values[at] - vmin) / vrange : values[at];
      break;
    }
  }
  uid[b] = u;
  iid[b] = i;
  // (a negative carries the interaction value 0, standardised like every other target when use_nce is on: dmf.py:68 with
  // recommender_abc.py:463-465 — (0 - min) / (max - min), non-zero whenever min_interaction is)
  y[b] = null_pair ? ((values && vrange > 0.f) ? (0.0f - vmin) / vrange : 0.0f) : val;
  my_deg = (int32_t)(H.indptr[u + 1] - H.indptr[u]);
  if (deg) deg[b] = my_deg;
  }
  // the workgroup's degree sum (the scan's first level: one launch less on the preparation's stream)
  if (sub_sums) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) my_deg += __shfl_xor(my_deg, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = my_deg;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
#pragma unroll
      for (int i2 = 0; i2 < kBlock / 64; ++i2) t += wsum[i2];
      sub_sums[blockIdx.x] = t;
    }
  }
}

// keep_off[1..B] = inclusive scan of deg[0..B) in three small launches (r03; r02 used ONE workgroup walking the batch in slabs: 54 - 80 us
// of side-stream time per step beside the training kernels, where the preparation had become the pipeline's bound): sums of tiles of 4096
// degrees, the spine (one wave scans the <= 64 tile sums and — when the host passes a pinned mailbox — posts the total, tagged, straight
// into host memory with one system-scope store: no copy kernel, no event, no L2 write-back for the host's sake), the tiles again with
// their offsets.  Batches of more than 262 144 triples fall back to the one-workgroup form.
constexpr int kDegTile = 4096;

__global__ __launch_bounds__(1024) void k_deg_tile_sums(const int32_t *__restrict__ deg, int B, int *__restrict__ tsum) {
  __shared__ int wsum[16];
  const int base = blockIdx.x * kDegTile + (int)threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += base + q < B ? deg[base + q] : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < 16; ++i) t += wsum[i];
    tsum[blockIdx.x] = t;
  }
}

// sub (or nullptr): n_sub sums of kBlock degrees each (k_point_sample's workgroups) — a tile's sum is then taken from its kDegTile / kBlock of them
__global__ __launch_bounds__(64) void k_deg_spine(int *__restrict__ tsum, int n_tiles, unsigned long long *mailbox, uint32_t tag,
                                                  const int *__restrict__ sub, int n_sub) {
  const int lane = threadIdx.x;
  int x = 0;
  if (sub) {
    constexpr int PER = kDegTile / kBlock;
#pragma unroll
    for (int q = 0; q < PER; ++q) x += (lane * PER + q < n_sub) ? sub[lane * PER + q] : 0;
  } else {
    x = lane < n_tiles ? tsum[lane] : 0;
  }
  int inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane < n_tiles) tsum[lane] = inc - x;                  // exclusive offsets of the tiles
  const int total = __shfl(inc, 63);
  if (lane == 0 && mailbox)
    __hip_atomic_store(mailbox, ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)total, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(1024) void k_deg_apply(const int32_t *__restrict__ deg, int B, const int *__restrict__ toff,
                                                    int32_t *__restrict__ keep_off) {
  __shared__ int wsum[16];
  const int base = blockIdx.x * kDegTile + (int)threadIdx.x * 4, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { sum += base + q < B ? deg[base + q] : 0; v[q] = sum; }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = toff[blockIdx.x] + inc - sum;
  for (int ww = 0; ww < w; ++ww) off += wsum[ww];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (base + q < B) keep_off[base + q + 1] = v[q] + off;
}

// k_deg_apply with the spine folded in (r03: a launch on the preparation's stream waits 10 - 30 us for room beside the training kernels
// whatever it computes): a tile's workgroup sums the workgroup sums of k_point_sample that lie before it itself, and the last tile
// posts the total.  r04: tiles of 1024 degrees in workgroups of 256 threads — a 1024-thread workgroup needs a whole CU's wave slots
// free at once, and beside the training kernels it waited for them (rocprofv3: 44 - 64 us for 13 us of work).
constexpr int kSubTile = 1024;
__global__ __launch_bounds__(256) void k_deg_apply_sub(const int32_t *__restrict__ deg, int B, const int *__restrict__ sub, int n_sub,
                                                       int32_t *__restrict__ keep_off, unsigned long long *mailbox, uint32_t tag) {
  __shared__ int wsum[4], wpre[4];
  const int base = blockIdx.x * kSubTile + (int)threadIdx.x * 4, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int PER = kSubTile / kBlock;
  const int n_before = min(n_sub, (int)blockIdx.x * PER);
  int pre = 0;
  for (int q = (int)threadIdx.x; q < n_before; q += 256) pre += sub[q];
  int v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { sum += base + q < B ? deg[base + q] : 0; v[q] = sum; }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) pre += __shfl_xor(pre, m, 64);
  if (lane == 63) wsum[w] = inc;
  if (lane == 0) wpre[w] = pre;
  __syncthreads();
  int off = inc - sum;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) off += wpre[ww] + (ww < w ? wsum[ww] : 0);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (base + q < B) keep_off[base + q + 1] = v[q] + off;
  if (mailbox && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255)
    __hip_atomic_store(mailbox, ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)(off + sum), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// (the one-workgroup form: any batch size)
__global__ __launch_bounds__(1024) void k_scan_degrees(const int32_t *__restrict__ deg, int B, int32_t *__restrict__ keep_off,
                                                       unsigned long long *mailbox, uint32_t tag) {
  constexpr int PER = 8, SLAB = 1024 * PER;
  __shared__ int sl[SLAB + SLAB / PER];          // padded: thread t starts at t * (PER + 1)
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int carry = 0;
  for (int base = 0; base < B; base += SLAB) {
    const int n = B - base < SLAB ? B - base : SLAB;
    for (int i = tid; i < SLAB; i += 1024) sl[i + (i >> 3)] = i < n ? deg[base + i] : 0;
    __syncthreads();
    int v[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { sum += sl[tid * (PER + 1) + q]; v[q] = sum; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int off = carry + inc - sum, total = 0;
    for (int ww = 0; ww < 16; ++ww) { if (ww < w) off += wsum[ww]; total += wsum[ww]; }
#pragma unroll
    for (int q = 0; q < PER; ++q) sl[tid * (PER + 1) + q] = v[q] + off;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) keep_off[base + i + 1] = sl[i + (i >> 3)];
    carry += total;
    __syncthreads();
  }
  if (tid == 0 && mailbox)
    __hip_atomic_store(mailbox, ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)carry, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlock) void k_row_lengths(const int64_t *__restrict__ indptr, const int32_t *__restrict__ ids, int B,
                                                        int32_t *__restrict__ deg, int32_t *__restrict__ off) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b == 0) off[0] = 0;
  if (b < B) deg[b] = (int32_t)(indptr[ids[b] + 1] - indptr[ids[b]]);
}

}  // namespace drx

extern "C" size_t drx_point_sample_scratch_bytes(int32_t B) {
  if (B < 1) return 0;
  const size_t n_sub = ((size_t)B + drx::kBlock - 1) / drx::kBlock;
  return drx::align_up((size_t)B * 4, 256) + 256 + 256 + drx::align_up(n_sub * 4, 4096);        // degrees, 64 tile sums, the sampler's workgroup sums
}

extern "C" int drx_point_sample(const DrxHistory *hist, int32_t n_users, int32_t n_items, int32_t B, int32_t neg_ratio,
                                uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off, void *scratch,
                                size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream) {
  return drx_point_sample_recorded(hist, nullptr, n_users, n_items, B, neg_ratio, seed, uid, iid, y, keep_off, scratch, scratch_bytes,
                                   host_mailbox, tag, stream);
}

extern "C" int drx_point_sample_recorded(const DrxHistory *hist, const DrxHistory *recorded, int32_t n_users, int32_t n_items, int32_t B,
                                         int32_t neg_ratio, uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off,
                                         void *scratch, size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream) {
  using namespace drx;
  if (recorded && (!recorded->indptr || !recorded->indices)) return DRX_EINVAL;
  if (!hist || !hist->indptr || !hist->indices || !uid || !iid || !y || !keep_off || !scratch || B < 1 || n_users < 1 ||
      n_items < 1 || neg_ratio < 0)
    return DRX_EINVAL;
  if (scratch_bytes < drx_point_sample_scratch_bytes(B)) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  int32_t *deg = (int32_t *)scratch;
  const int n_sub = (B + kBlock - 1) / kBlock;
  int *tsum = (int *)((char *)scratch + align_up((size_t)B * 4, 256));
  int *sub = tsum + 128;
  hipLaunchKernelGGL(k_point_sample, dim3(n_sub), dim3(kBlock), 0, st, *hist, recorded ? *recorded : *hist, n_users, n_items, B, neg_ratio,
                     seed, uid, iid, y, deg, keep_off, sub, nullptr, 0.f, 0.f);
  hipLaunchKernelGGL(k_deg_apply_sub, dim3((B + kSubTile - 1) / kSubTile), dim3(256), 0, st, deg, B, (const int *)sub, n_sub, keep_off,
                     (unsigned long long *)host_mailbox, tag);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

// ---- the same draws handed out in USER order (batches whose touch lists are prepared through the history's transpose) -------------
namespace drx {
static __global__ void k_by_user_keys(const int32_t *__restrict__ uid, int B, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B; j += gridDim.x * blockDim.x) { keys[j] = (uint32_t)uid[j]; vals[j] = (uint32_t)j; }
}
// triple p of the output = draw vs[p]; its degree, and the workgroup's degree sum (the scan's first level, as in k_point_sample)
static __global__ __launch_bounds__(kBlock) void k_by_user_gather(DrxHistory H, const uint32_t *__restrict__ ks, const uint32_t *__restrict__ vs,
                                                                  const int32_t *__restrict__ iid_d, const float *__restrict__ y_d, int B,
                                                                  int32_t *uid, int32_t *iid, float *y, int32_t *deg, int32_t *keep_off,
                                                                  int *sub_sums) {
  __shared__ int wsum[kBlock / 64];
  const int p = blockIdx.x * kBlock + threadIdx.x;
  int my_deg = 0;
  if (p < B) {
    if (p == 0) keep_off[0] = 0;
    const int u = (int)ks[p];
    const uint32_t d = vs[p];
    uid[p] = u; iid[p] = iid_d[d]; y[p] = y_d[d];
    my_deg = (int32_t)(H.indptr[u + 1] - H.indptr[u]);
    deg[p] = my_deg;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) my_deg += __shfl_xor(my_deg, m, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = my_deg;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i2 = 0; i2 < kBlock / 64; ++i2) t += wsum[i2];
    sub_sums[blockIdx.x] = t;
  }
}
}  // namespace drx

extern "C" size_t drx_point_sample_by_user_scratch_bytes(int32_t B, int32_t n_users) {
  if (B < 1 || n_users < 1) return 0;
  return drx_point_sample_scratch_bytes(B) + 7 * drx::align_up((size_t)B * 4, 256) +
         drx::sort_pairs_temp_bytes((size_t)B, drx::bits_for((uint64_t)n_users + 1)) + 512;
}

extern "C" int drx_point_sample_by_user(const DrxHistory *hist, const DrxHistory *recorded, int32_t n_users, int32_t n_items, int32_t B,
                                        int32_t neg_ratio, uint64_t seed, int32_t *uid, int32_t *iid, float *y, int32_t *keep_off,
                                        void *scratch, size_t scratch_bytes, uint64_t *host_mailbox, uint32_t tag, void *stream) {
  using namespace drx;
  if (recorded && (!recorded->indptr || !recorded->indices)) return DRX_EINVAL;
  if (!hist || !hist->indptr || !hist->indices || !uid || !iid || !y || !keep_off || !scratch || B < 1 || n_users < 1 || n_items < 1 ||
      neg_ratio < 0)
    return DRX_EINVAL;
  if (scratch_bytes < drx_point_sample_by_user_scratch_bytes(B, n_users)) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  int32_t *deg = cv.take<int32_t>(B);
  int *tsum = cv.take<int>(64);
  int *sub = cv.take<int>(((size_t)B + kBlock - 1) / kBlock + 64);
  (void)tsum;
  int32_t *uid_d = cv.take<int32_t>(B), *iid_d = cv.take<int32_t>(B);
  float *y_d = cv.take<float>(B);
  uint32_t *k0 = cv.take<uint32_t>(B), *v0 = cv.take<uint32_t>(B), *ks = cv.take<uint32_t>(B), *vs = cv.take<uint32_t>(B);
  const int bits = bits_for((uint64_t)n_users + 1);
  const size_t sb = sort_pairs_temp_bytes((size_t)B, bits);
  void *stemp = cv.take<char>(sb);
  if (!cv.ok()) return DRX_ESCRATCH;
  const int n_sub = (B + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_point_sample, dim3(n_sub), dim3(kBlock), 0, st, *hist, recorded ? *recorded : *hist, n_users, n_items, B, neg_ratio,
                     seed, uid_d, iid_d, y_d, (int32_t *)nullptr, (int32_t *)nullptr, (int *)nullptr, (const float *)nullptr, 0.f, 0.f);
  hipLaunchKernelGGL(k_by_user_keys, dim3(n_sub < 1024 ? n_sub : 1024), dim3(256), 0, st, uid_d, B, k0, v0);
  const int rc = sort_pairs(stemp, sb, k0, ks, v0, vs, (size_t)B, bits, st);          // stable: a user's draws keep their order
  if (rc) return rc;
  hipLaunchKernelGGL(k_by_user_gather, dim3(n_sub), dim3(kBlock), 0, st, *hist, ks, vs, iid_d, y_d, B, uid, iid, y, deg, keep_off, sub);
  hipLaunchKernelGGL(k_deg_apply_sub, dim3((B + kSubTile - 1) / kSubTile), dim3(256), 0, st, deg, B, (const int *)sub, n_sub, keep_off,
                     (unsigned long long *)host_mailbox, tag);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

extern "C" int drx_point_sample_valued(const DrxHistory *hist, const DrxHistory *recorded, const float *pos_values, float vmin,
                                       float vrange, int32_t n_users, int32_t n_items, int32_t B, int32_t neg_ratio, uint64_t seed,
                                       int32_t *uid, int32_t *iid, float *y, void *stream) {
  using namespace drx;
  if (recorded && (!recorded->indptr || !recorded->indices)) return DRX_EINVAL;
  if (!hist || !hist->indptr || !hist->indices || !pos_values || !uid || !iid || !y || B < 1 || n_users < 1 || n_items < 1 || neg_ratio < 0)
    return DRX_EINVAL;
  hipLaunchKernelGGL(k_point_sample, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, *hist,
                     recorded ? *recorded : *hist, n_users, n_items, B, neg_ratio, seed, uid, iid, y, (int32_t *)nullptr,
                     (int32_t *)nullptr, (int *)nullptr, pos_values, vmin, vrange);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

/* off[0] = 0, off[b + 1] = sum_{b' <= b} (indptr[ids[b'] + 1] - indptr[ids[b']]): the keep_off of a batch of users (DrxBatch) or the touch
 * offsets of a batch of CSR rows, from device ids — what the host code otherwise asked torch.cumsum for.  scratch: drx_point_sample_scratch_bytes(B). */
extern "C" int drx_batch_offsets(const int64_t *indptr, const int32_t *ids, int32_t B, int32_t *off, void *scratch, size_t scratch_bytes,
                                 void *stream) {
  using namespace drx;
  if (!indptr || !ids || !off || !scratch || B < 1) return DRX_EINVAL;
  if (scratch_bytes < drx_point_sample_scratch_bytes(B)) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  int32_t *deg = (int32_t *)scratch;
  hipLaunchKernelGGL(k_row_lengths, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, st, indptr, ids, B, deg, off);
  const int n_tiles = (B + kDegTile - 1) / kDegTile;
  if (n_tiles <= 64) {
    int *tsum = (int *)((char *)scratch + align_up((size_t)B * 4, 256));
    hipLaunchKernelGGL(k_deg_tile_sums, dim3(n_tiles), dim3(1024), 0, st, deg, B, tsum);
    hipLaunchKernelGGL(k_deg_spine, dim3(1), dim3(64), 0, st, tsum, n_tiles, (unsigned long long *)nullptr, 0u, (const int *)nullptr, 0);
    hipLaunchKernelGGL(k_deg_apply, dim3(n_tiles), dim3(1024), 0, st, deg, B, tsum, off);
  } else {
    hipLaunchKernelGGL(k_scan_degrees, dim3(1), dim3(1024), 0, st, deg, B, off, (unsigned long long *)nullptr, 0u);
  }
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

// ---- device-side list sampler (include/drx.h: drx_list_sample_device) ---------------------------------------------------------
namespace drx {

__device__ __forceinline__ uint32_t scale_u32(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * (uint64_t)n) >> 32); }

// j-th (0-based) id of the ascending complement of held[0..nh) in [0, n_ids): the smallest v with v - #(held <= v) == j
__device__ __forceinline__ int32_t complement_at(const int32_t *__restrict__ held, int nh, uint32_t j) {
  int lo = 0, hi = nh;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)held[mid] - (int64_t)mid <= (int64_t)j) lo = mid + 1; else hi = mid;
  }
  return (int32_t)(j + (uint32_t)lo);
}

__global__ __launch_bounds__(kBlock) void k_list_sample(DrxListGroups G, int B, int L, int T, int neg, uint64_t seed,
                                                        int32_t *__restrict__ grp, int32_t *__restrict__ before, int32_t *after) {
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= B) return;
  const int g = G.eligible[scale_u32(hash_u32(seed, (uint32_t)d, 0u), (uint32_t)G.n_eligible)];
  const int64_t r0 = G.indptr[g];
  const int n_rows = (int)(G.indptr[g + 1] - r0);
  const int start = (int)scale_u32(hash_u32(seed, (uint32_t)d, 1u), (uint32_t)(n_rows - L - T + 1));
  grp[d] = G.group_value[g];
  for (int t = 0; t < L; ++t) before[(size_t)d * L + t] = G.seq_ids[r0 + start + t];
  const int Tp = T * (1 + neg), n_neg = T * neg;
  int32_t *const out = after + (size_t)d * Tp;
  for (int t = 0; t < T; ++t) out[t] = G.seq_ids[r0 + start + L + t];
  const int64_t h0 = G.held_indptr[g];
  const int nh = (int)(G.held_indptr[g + 1] - h0);
  const uint32_t n_pop = (uint32_t)(G.n_ids - nh);
  for (int i = 0; i < n_neg; ++i) {
    int32_t id = 0;
    uint32_t j = 0;
    bool dup = true;
    for (int a = 0; a < 16 && dup; ++a) {
      j = scale_u32(hash_u32(seed, (uint32_t)d, (uint32_t)(2 + 16 * i + a)), n_pop);
      id = complement_at(G.held + h0, nh, j);
      dup = false;
      for (int p = 0; p < i; ++p) dup |= out[T + p] == id;
    }
    while (dup) {                                   // (sixteen collisions in a row: a catalogue barely larger than the draw)
      j = j + 1 == n_pop ? 0u : j + 1;
      id = complement_at(G.held + h0, nh, j);
      dup = false;
      for (int p = 0; p < i; ++p) dup |= out[T + p] == id;
    }
    out[T + i] = id;
  }
}

// The same draws with SIXTEEN LANES per window (r06): a window's negatives are independent until two of them collide, so lane l forms the
// first candidate of negatives l, l + 16, ... side by side (hash, complement search: the thread-per-window kernel's chain of ~10
// dependent loads per negative, nine negatives deep, was 36 us for 4096 windows); the candidates are then confirmed IN ORDER — negative i
// keeps the first attempt a = 0, 1, ... whose id none of the negatives before it took, exactly the rule above, so the two kernels return
// the same windows — through group-wide comparisons in registers.  Up to 64 negatives per window (4 per lane).
constexpr int kListMaxPerLane = 4;
__global__ __launch_bounds__(kBlock) void k_list_sample16(DrxListGroups G, int B, int L, int T, int neg, uint64_t seed,
                                                          int32_t *__restrict__ grp, int32_t *__restrict__ before, int32_t *after) {
  const int l = threadIdx.x & 15, gbase = threadIdx.x & 48;          // lane in the window's group, the group's first lane in the wave
  const int d_raw = (blockIdx.x * kBlock + threadIdx.x) >> 4;
  const bool live = d_raw < B;
  const int d = live ? d_raw : B - 1;                                 // (idle groups shadow the last window and write nothing)
  const int g = G.eligible[scale_u32(hash_u32(seed, (uint32_t)d, 0u), (uint32_t)G.n_eligible)];
  const int64_t r0 = G.indptr[g];
  const int n_rows = (int)(G.indptr[g + 1] - r0);
  const int start = (int)scale_u32(hash_u32(seed, (uint32_t)d, 1u), (uint32_t)(n_rows - L - T + 1));
  const int Tp = T * (1 + neg), n_neg = T * neg;
  int32_t *const out = after + (size_t)d * Tp;
  if (live) {
    if (l == 0) grp[d] = G.group_value[g];
    for (int t = l; t < L; t += 16) before[(size_t)d * L + t] = G.seq_ids[r0 + start + t];
    for (int t = l; t < T; t += 16) out[t] = G.seq_ids[r0 + start + L + t];
  }
  const int64_t h0 = G.held_indptr[g];
  const int nh = (int)(G.held_indptr[g + 1] - h0);
  const uint32_t n_pop = (uint32_t)(G.n_ids - nh);
  int32_t fin[kListMaxPerLane];
#pragma unroll
  for (int c = 0; c < kListMaxPerLane; ++c) fin[c] = -1;
  for (int c = 0; c * 16 < n_neg; ++c) {
    const int i = c * 16 + l;
    int a = 0;
    uint32_t j = 0;
    int32_t id = -1;
    if (i < n_neg) {
      j = scale_u32(hash_u32(seed, (uint32_t)d, (uint32_t)(2 + 16 * i)), n_pop);
      id = complement_at(G.held + h0, nh, j);
    }
    for (int k = 0; k < 16; ++k) {                                    // confirm the chunk's negatives in order
      bool need = c * 16 + k < n_neg;                                 // (uniform in the group; groups of one wave may differ)
      while (__any(need)) {
        const int32_t idk = __shfl(id, k, 16);
        bool mine = false;
        if (need) {
#pragma unroll
          for (int q = 0; q < kListMaxPerLane; ++q) mine |= fin[q] == idk;      // (fin holds confirmed ids only: all of them came before)
        }
        const unsigned long long b = __ballot(mine);
        const bool dup = ((b >> gbase) & 0xFFFFull) != 0;
        if (need && !dup) need = false;
        else if (need && l == k) {                                    // taken: the next attempt of this negative
          ++a;
          if (a < 16) j = scale_u32(hash_u32(seed, (uint32_t)d, (uint32_t)(2 + 16 * i + a)), n_pop);
          else j = j + 1 == n_pop ? 0u : j + 1;                       // (sixteen collisions in a row: walk the complement)
          id = complement_at(G.held + h0, nh, j);
        }
      }
      if (l == k && i < n_neg) {
#pragma unroll
        for (int q = 0; q < kListMaxPerLane; ++q) if (q == c) fin[q] = id;
        if (live) out[T + i] = id;
      }
    }
  }
}

}  // namespace drx

extern "C" int drx_list_sample_device(const DrxListGroups *g, int32_t B, int32_t n_inputs, int32_t n_targets, int32_t neg_ratio,
                                      uint64_t seed, int32_t *group_out, int32_t *before, int32_t *after, void *stream) {
  using namespace drx;
  if (!g || !g->indptr || !g->seq_ids || !g->held_indptr || !g->held || !g->group_value || !g->eligible || g->n_eligible < 1 ||
      g->n_ids < 1 || B < 1 || n_inputs < 1 || n_targets < 1 || neg_ratio < 0 || !group_out || !before || !after)
    return DRX_EINVAL;
  if (n_targets * neg_ratio <= 16 * kListMaxPerLane && kBlock % 64 == 0)
    hipLaunchKernelGGL(k_list_sample16, dim3((B * 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, *g, B, n_inputs,
                       n_targets, neg_ratio, seed, group_out, before, after);
  else
    hipLaunchKernelGGL(k_list_sample, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, *g, B, n_inputs, n_targets,
                       neg_ratio, seed, group_out, before, after);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}
