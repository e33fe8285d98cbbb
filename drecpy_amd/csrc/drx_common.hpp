// Internal helpers shared by the libdrx.so translation units (gfx950 only).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "drx.h"

#define DRX_HIP(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return (int)e_;              \
  } while (0)

#define DRX_LAUNCH_CHECK()                             \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) return (int)e_;              \
  } while (0)

namespace drx {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD

// splitmix64-style mix of (seed, a, b) -> 32 bits.  Same function on host and device.
__host__ __device__ inline uint32_t hash_u32(uint64_t seed, uint32_t a, uint32_t b) {
  uint64_t x = seed + (uint64_t)a * 0x9E3779B97F4A7C15ull + (uint64_t)b * 0xD1B54A32D192ED03ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

__host__ __device__ inline uint32_t q_threshold(float q) {
  double t = (double)q * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 0xFFFFFFFFu;
  return (uint32_t)t;
}

// Sum over an aligned group of G lanes (G power of two <= 64); every lane gets the total.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4 &a, float s, const float4 &x) {
  a.x = fmaf(s, x.x, a.x); a.y = fmaf(s, x.y, a.y); a.z = fmaf(s, x.z, a.z); a.w = fmaf(s, x.w, a.w);
}
__device__ __forceinline__ void f4_add(float4 &a, const float4 &x) {
  a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
}
__device__ __forceinline__ float f4_dot(const float4 &a, const float4 &b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Plain-division sigmoid (matches 1/(1+exp(-x)) of the oracle to ~1 ulp; no fast-math).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Keras binary_crossentropy element and its derivative wrt p.
__device__ __forceinline__ float bce_elem(float t, float p) {
  const float eps = 1e-7f;
  float pc = fminf(fmaxf(p, eps), 1.0f - eps);
  return -(t * logf(pc + eps) + (1.0f - t) * logf(1.0f - pc + eps));
}
__device__ __forceinline__ float bce_grad(float t, float p) {
  const float eps = 1e-7f;
  float pc = fminf(fmaxf(p, eps), 1.0f - eps);
  float g = -(t / (pc + eps) - (1.0f - t) / (1.0f - pc + eps));
  return (p >= eps && p <= 1.0f - eps) ? g : 0.0f;
}

// Row-group geometry: G lanes cooperate on one table row, each lane owns J float4 (cols
// 4*(lane + j*G) .. +3).  ld <= 4*G*J.
struct Geom { int G, J; };
inline Geom pick_geom(int ld) {
  if (ld <= 16) return {4, 1};      // e.g. K = 128 columns sharded over 8 GPUs: 16 wave64 rows per load instruction
  if (ld <= 32) return {8, 1};
  if (ld <= 64) return {16, 1};
  if (ld <= 128) return {32, 1};
  if (ld <= 256) return {64, 1};
  if (ld <= 512) return {64, 2};
  return {64, 4};
}

#define DRX_DISPATCH_GEOM(ld, CALL)                        \
  do {                                                     \
    drx::Geom g_ = drx::pick_geom(ld);                     \
    if (g_.G == 4) { CALL(4, 1); }                         \
    else if (g_.G == 8) { CALL(8, 1); }                    \
    else if (g_.G == 16) { CALL(16, 1); }                  \
    else if (g_.G == 32) { CALL(32, 1); }                  \
    else if (g_.J == 1) { CALL(64, 1); }                   \
    else if (g_.J == 2) { CALL(64, 2); }                   \
    else { CALL(64, 4); }                                  \
  } while (0)

// Orders the LDS traffic of ONE wave (its private scratch is written by some lanes and read by others): every LDS operation of the
// wave issued so far has completed.  A workgroup barrier is not needed for that and would tie unrelated waves together.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// In-kernel phase stamps (diagnostic builds only: -DDRX_STAMPS, scripts/build_variant.sh): lane 0 of a row group writes the
// constant-rate device clock (100 MHz) at a few points of its life into stamps[unit * 16 + i] — the buffer travels in the kernel
// arguments (SparseBufs / SegBufs, set from drx_debug_set_stamps); scripts/stamps.py turns them into mean phase durations.
#ifdef DRX_STAMPS
#define DRX_STAMP(buf, unit, i, lane)                                                             \
  do {                                                                                            \
    if ((lane) == 0 && (buf) && (unsigned)(unit) < 110000u) (buf)[(size_t)(unit) * 16 + (i)] = wall_clock64(); \
  } while (0)
#else
#define DRX_STAMP(buf, unit, i, lane) do { } while (0)
#endif

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's scratch buffer.
struct Carver {
  char *base; size_t off; size_t cap;
  Carver(void *p, size_t c) : base((char *)p), off(0), cap(c) {}
  template <typename T> T *take(size_t n) {
    off = align_up(off, 256);
    T *r = (T *)(base ? base + off : nullptr);
    off += n * sizeof(T);
    return r;
  }
  bool ok() const { return off <= cap; }
};

// Internal: stable radix sort of (key, val) pairs, implemented in drx_sort.hip.  Keys must be < 2^end_bit (or DRX_KEY_NONE with drop_none):
// the passes cover ceil(end_bit / digit) * digit bits, higher bits are not masked off.
size_t sort_pairs_temp_bytes(size_t n, int end_bit);
int sort_pairs(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
               uint32_t *vout, size_t n, int end_bit, hipStream_t stream);
// drop_none: pairs with key DRX_KEY_NONE take no part (kout = the sorted pairs, then DRX_KEY_NONE up to n; their vals are not written).
// Launch order of the sampled forward kernel (drx_cdae.hip: k_degree_counts): the bucket of a triple's history length, the lanes of a
// wave that share a bucket, and the counting half of the counting sort — here because the counts may ride in the sort's first launch.
__device__ __forceinline__ int degree_bucket(const int32_t *keep_off, int b) {
  const int d = (keep_off[b + 1] - keep_off[b]) >> 2;
  return 255 - (d > 255 ? 255 : d);                  // descending
}

__device__ __forceinline__ unsigned long long same_bucket_lanes(bool valid, int d) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const bool one = (d >> bit) & 1;
    const unsigned long long bl = __ballot(one);
    m &= one ? bl : ~bl;
  }
  return m;
}

template <int NT>
__device__ __forceinline__ void degree_counts_body(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work, int block_id,
                                                   unsigned int *cnt /* LDS [256] */) {
  for (int i = threadIdx.x; i < 256; i += NT) cnt[i] = 0;
  __syncthreads();
  const int b = block_id * NT + (int)threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = b < B;
  const int d = valid ? degree_bucket(keep_off, b) : 0;
  const unsigned long long m = same_bucket_lanes(valid, d);
  if (valid && lane == __ffsll((long long)m) - 1) atomicAdd(&cnt[d], (unsigned int)__popcll(m));
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += NT)
    if (cnt[i]) atomicAdd(&work[i], cnt[i]);
}

// Optional work that rides in the sort's first launch as extra workgroups (a launch on a stream that shares the chip with the training
// kernels waits 10 - 30 us for room whatever it computes): the degree-bucket counts of a batch.  keep_off == nullptr: none.
struct SortRider {
  const int32_t *keep_off;
  int B;
  unsigned int *counts;      // [256], zero on entry
};

// pre_zeroed: the caller has already cleared the words sort_pairs_zero_region names (a kernel of its own that runs before the sort
// anyway: one launch less on the stream).
int sort_pairs_ex(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, int end_bit,
                  bool drop_none, hipStream_t stream, bool pre_zeroed = false, SortRider rider = SortRider{nullptr, 0, nullptr});
void sort_pairs_zero_region(void *temp, size_t n, int end_bit, uint32_t **words, size_t *n_words);

inline int bits_for(uint64_t max_key_exclusive) {
  int b = 1;
  while (b < 32 && (1ull << b) < max_key_exclusive) ++b;
  return b;
}

}  // namespace drx
