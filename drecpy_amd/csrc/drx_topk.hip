// Per-row masked top-k with the ordering of heapq.nlargest over (score, iid) tuples
// (DRecPy/Recommender/cdae.py:102-103, recommender_abc.py:460-461): descending score, ties by LARGER index.
// One workgroup per row; the row's candidates become 64-bit keys (ordered score bits << 32 | index) sorted by a
// bitonic network in LDS (up to 16384 keys = 128 KiB of the CU's 160 KiB).
#include <cstring>
#include <hip/hip_runtime.h>
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

__device__ __forceinline__ unsigned long long topk_key(const float *scores, const uint32_t *mask, size_t row_off, int i) {
  const size_t bit = row_off + (size_t)i;
  const bool ok = !mask || ((mask[bit >> 5] >> (bit & 31)) & 1u);
  return ok ? (((unsigned long long)ordered_bits(scores[bit]) << 32) | (unsigned)i) : 0ull;
}

// ---- small k of a short row (the ranking protocols: k = 10 of a MovieLens catalogue): SELECTION by one wave, no sort (r06) ----------
// The LDS path above sorts every padded row in full whatever k is: 231 us for 2048 rows of 3706 scores at k = 10 (78 bitonic stages of
// 4096 keys behind 78 workgroup barriers) — 25 x the 9.4 us the DMF scorer needs to produce them.  Here ONE WAVE owns a row, holds its
// keys in registers (NPL per lane, row position lane + 64 j) and extracts the k largest one after another: the keys are unique (the
// index is part of them), so the next winner is the largest key BELOW the last one — a lane's local maximum under that bound, then a
// wave maximum by shuffles.  No LDS, no barrier, O(k NPL) instructions per wave; every row of the launch is resident at once.
// maximum of a 32-bit value over the wave, in every lane: the GCN data-parallel-primitive ladder (quad swaps, half-row and row mirrors,
// then the row broadcasts that gfx9 keeps) — six dependent v_max_u32_dpp, no LDS crossbar; the last lane holds the result
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define DRX_DPP_MAX(ctrl, rows) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, true))
  DRX_DPP_MAX(0xB1, 0xF);          // quad_perm [1, 0, 3, 2]
  DRX_DPP_MAX(0x4E, 0xF);          // quad_perm [2, 3, 0, 1]
  DRX_DPP_MAX(0x141, 0xF);         // row_half_mirror
  DRX_DPP_MAX(0x140, 0xF);         // row_mirror: every lane of a row of 16 holds the row's maximum
  DRX_DPP_MAX(0x142, 0xA);         // row_bcast15 into rows 1 and 3
  DRX_DPP_MAX(0x143, 0xC);         // row_bcast31 into rows 2 and 3
#undef DRX_DPP_MAX
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// a lane's two largest keys below `bound` (a >= b; 0 = none): 11 vector instructions per key
template <int NPL>
__device__ __forceinline__ void lane_top2(const unsigned long long (&key)[NPL], unsigned long long bound, unsigned long long &a,
                                          unsigned long long &b) {
  a = 0ull; b = 0ull;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const unsigned long long c = key[j] < bound ? key[j] : 0ull;
    const bool up = c > a;
    const unsigned long long lo = up ? a : c;
    a = up ? c : a;
    b = lo > b ? lo : b;
  }
}

template <int NPL>
__global__ __launch_bounds__(kBlock) void k_topk_wave(const float *__restrict__ scores, const uint32_t *__restrict__ mask, int R, int n, int k,
                                                      int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
  const int lane = threadIdx.x & 63;
  const size_t r = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= (size_t)R) return;
  const size_t row_off = r * (size_t)n;
  unsigned long long key[NPL];
  {
    // every load of the row issued before the first is waited for: clamped positions, no branch (a conditional load per key had the
    // compiler wait for each in turn: 58 round trips, 18 us of a wave's life whatever k)
    float v[NPL];
    uint32_t mw[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = min(lane + 64 * j, n - 1);
      v[j] = scores[row_off + (size_t)i];
    }
    if (mask) {
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const size_t bit = row_off + (size_t)min(lane + 64 * j, n - 1);
        mw[j] = mask[bit >> 5] >> (bit & 31);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPL; ++j) mw[j] = 1u;
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int i = lane + 64 * j;
      key[j] = (i < n && (mw[j] & 1u)) ? (((unsigned long long)ordered_bits(v[j]) << 32) | (unsigned)i) : 0ull;
    }
  }
  // Every lane keeps its TWO best keys not yet handed out (first version: the lane's maximum below the last winner recomputed over all
  // its registers in every round — 38 us for 2048 rows of 3706 at k = 10, bound by those 6 NPL instructions per round).  A round is then
  // a wave maximum of the lanes' heads and a pop on the winning lane; a lane that has handed out both refills from its registers below
  // its last key — rare: k winners fall on 64 lanes.
  unsigned long long a, b;
  lane_top2<NPL>(key, ~0ull, a, b);                    // (a valid key is never all ones: the index is below 2^31)
  unsigned long long last = ~0ull;                     // this lane's last winner: what a refill starts below
  unsigned long long mine = 0ull;
  for (int it = 0; it < k; ++it) {
    // the largest head: its score bits first, then the largest index among the lanes that hold those bits (two 32-bit wave maxima by
    // DPP; six shuffle rounds of a 64-bit value through the LDS crossbar had cost 0.65 us per round)
    const uint32_t mh = wave_max_u32((uint32_t)(a >> 32));
    const uint32_t ml = wave_max_u32((uint32_t)(a >> 32) == mh ? (uint32_t)a : 0u);
    const unsigned long long m = ((unsigned long long)mh << 32) | ml;
    if (lane == it) mine = m;                          // winner `it` waits on lane `it` (k <= 64): written out after the loop, all at once
    if (m == 0ull) continue;                           // (the row is exhausted: the rest comes out "missing"; wave-uniform)
    const bool won = a == m;                           // keys are unique: exactly one lane
    if (won) { last = a; a = b; b = 0ull; }
    if (__ballot(won && a == 0ull)) {                  // the winner has nothing left in hand: its next two below `last` (wave-uniform branch)
      unsigned long long na, nb;
      lane_top2<NPL>(key, last, na, nb);
      if (won && a == 0ull) { a = na; b = nb; }
    }
  }
  // (first version: lane 0 wrote winner `it` inside the loop — its score is a dependent load the store waits for: a round trip per
  // round, 29 us for the launch)
  if (lane < k) {
    if (mine == 0ull) { out_idx[r * (size_t)k + lane] = -1; out_val[r * (size_t)k + lane] = -INFINITY; }
    else {
      const int idx = (int)(mine & 0xFFFFFFFFull);
      out_idx[r * (size_t)k + lane] = idx;
      out_val[r * (size_t)k + lane] = scores[row_off + idx];
    }
  }
}

// ---- rows longer than the LDS path (the 1 M-item catalogue): RADIX SELECT, not a sort ------------------------------------------
// One workgroup per row finds the k-th largest 64-bit key by walking its digits from the top, 11 bits at a time: a 2048-bin LDS
// histogram of the digit among the keys that match the prefix found so far, the bin where the count from the top reaches k, next
// digit (6 passes over the row, which L2 / the Infinity Cache hold after the first).  Keys are unique (the index is part of them),
// so exactly k keys are >= the k-th one: they are collected (in any order) and a second launch orders those k in LDS with the
// bitonic network above.  r02 sorted every row in full with rocprim::segmented_radix_sort_keys_desc.
constexpr int kWaveTopkMaxK = 64;       // k_topk_wave: one winner per lane.  (Up to 512 winners in 8 registers per lane was built: k = 100 took 214 us
                                        // against the sort's 232 — with more winners than lanes the refills, 1 us each, take over.)
constexpr int kSelThreads = 1024;
constexpr int kSelBits = 11, kSelBins = 1 << kSelBits;

__global__ __launch_bounds__(kSelThreads) void k_topk_select(const float *__restrict__ scores, const uint32_t *__restrict__ mask, int n,
                                                             int k, unsigned long long *__restrict__ cand, int *__restrict__ n_cand) {
  __shared__ unsigned int hist[kSelBins];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned int s_need, s_count;
  __shared__ unsigned int wsum[kSelThreads / 64];
  const size_t r = blockIdx.x, row_off = r * (size_t)n;
  // valid candidates of the row
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  unsigned int mine = 0;
  for (int i = threadIdx.x; i < n; i += kSelThreads) mine += topk_key(scores, mask, row_off, i) != 0ull;
  atomicAdd(&s_count, mine);
  __syncthreads();
  const unsigned int want = min((unsigned int)k, s_count);
  if (threadIdx.x == 0) { s_prefix = 0ull; s_need = want; n_cand[r] = (int)want; }
  __syncthreads();
  if (want == 0) return;
  // digits from the top: after the pass at `shift`, s_prefix holds the k-th key's bits at and above it
  for (int shift = 64 - kSelBits + (6 * kSelBits - 64); shift >= 0; shift -= kSelBits) {      // 55, 44, 33, 22, 11, 0 (the first digit is 9 bits wide)
    for (int i = threadIdx.x; i < kSelBins; i += kSelThreads) hist[i] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int hi_shift = shift + kSelBits;               // bits at and above this position are decided
    for (int i = threadIdx.x; i < n; i += kSelThreads) {
      const unsigned long long key = topk_key(scores, mask, row_off, i);
      if (key == 0ull) continue;
      const bool match = hi_shift >= 64 || (key >> hi_shift) == (prefix >> hi_shift);
      if (match) atomicAdd(&hist[(unsigned)(key >> shift) & (kSelBins - 1)], 1u);
    }
    __syncthreads();
    // the bin, from the top, where the running count reaches what is still needed: thread t owns bins 2t and 2t + 1; an exclusive
    // SUFFIX sum over the threads (wave shuffles + one LDS word per wave) gives every thread the count above its bins
    {
      const unsigned int c0 = hist[2 * threadIdx.x], c1 = hist[2 * threadIdx.x + 1];
      const unsigned int pair = c0 + c1;
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      unsigned int v = pair;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const unsigned int o = __shfl_down(v, dd, 64);
        if (lane + dd < 64) v += o;
      }
      if (lane == 0) wsum[w] = v;                        // the wave's total
      __syncthreads();
      unsigned int above = v - pair;                     // pairs of higher lanes of this wave
      for (int ww = w + 1; ww < kSelThreads / 64; ++ww) above += wsum[ww];
      const unsigned int need = s_need;
      __syncthreads();                                   // (every thread has read s_need before one of them rewrites it)
      if (above < need && need <= above + pair) {
        const bool upper = above + c1 >= need;
        s_need = need - (upper ? above : above + c1);
        s_prefix = prefix | ((unsigned long long)(2 * threadIdx.x + (upper ? 1 : 0)) << shift);
      }
    }
    __syncthreads();
  }
  const unsigned long long kth = s_prefix;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kSelThreads) {
    const unsigned long long key = topk_key(scores, mask, row_off, i);
    if (key != 0ull && key >= kth) {
      const unsigned int at = atomicAdd(&s_count, 1u);
      if (at < (unsigned)k) cand[r * (size_t)k + at] = key;
    }
  }
}

// the k selected keys of every row, ordered in LDS (k <= 16384) and emitted
__global__ __launch_bounds__(kBlock) void k_topk_order(const unsigned long long *__restrict__ cand, const int *__restrict__ n_cand,
                                                       const float *__restrict__ scores, int n, int ks, int k, int kpad,
                                                       int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
  extern __shared__ __align__(16) unsigned long long keys[];       // ks = selected per row (stride of cand), k = output columns
  const size_t r = blockIdx.x;
  const int have = n_cand[r];
  for (int i = threadIdx.x; i < kpad; i += kBlock) keys[i] = i < have ? cand[r * (size_t)ks + i] : 0ull;
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (kpad >> 1); t += kBlock) {
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
    const unsigned long long key = j < kpad ? keys[j] : 0ull;
    if (key == 0ull) { out_idx[r * (size_t)k + j] = -1; out_val[r * (size_t)k + j] = -INFINITY; }
    else { const int idx = (int)(key & 0xFFFFFFFFull); out_idx[r * (size_t)k + j] = idx; out_val[r * (size_t)k + j] = scores[r * (size_t)n + idx]; }
  }
}

// k > 16384 of a long row (rank everything): the selected keys ordered by two stable passes of the library's own pair sort — by index
// descending, then by score descending — one row at a time (rare: a full ranking of a large catalogue)
__global__ void k_topk_split(const unsigned long long *__restrict__ cand, int have, uint32_t *__restrict__ kidx, uint32_t *__restrict__ vals) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < have; i += gridDim.x * blockDim.x) {
    const uint32_t idx = (uint32_t)(cand[i] & 0xFFFFFFFFull);
    kidx[i] = ~idx;                       // ascending ~idx = descending idx
    vals[i] = (uint32_t)i;                // position in cand
  }
}
__global__ void k_topk_score_keys(const unsigned long long *__restrict__ cand, const uint32_t *__restrict__ order, int have,
                                  uint32_t *__restrict__ kscore) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < have; i += gridDim.x * blockDim.x)
    kscore[i] = ~(uint32_t)(cand[order[i]] >> 32);      // ascending ~score = descending score
}
__global__ void k_topk_emit_sorted(const unsigned long long *__restrict__ cand, const uint32_t *__restrict__ order, int have,
                                   const float *__restrict__ scores_row, int k, int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
    if (j < have) { const int idx = (int)(cand[order[j]] & 0xFFFFFFFFull); out_idx[j] = idx; out_val[j] = scores_row[idx]; }
    else { out_idx[j] = -1; out_val[j] = -INFINITY; }
  }
}

struct TopkLayout {
  unsigned long long *cand;     // [R, k]
  int *n_cand;                  // [R]
  uint32_t *ka, *kb, *va, *vb;  // [k] each (k > 16384 only)
  void *sort_temp;
  size_t sort_bytes;
};

static TopkLayout topk_layout(Carver &cv, int R, int k) {
  TopkLayout L{};
  L.cand = cv.take<unsigned long long>((size_t)R * k);
  L.n_cand = cv.take<int>(R);
  if (k > 16384) {
    L.ka = cv.take<uint32_t>(k); L.kb = cv.take<uint32_t>(k); L.va = cv.take<uint32_t>(k); L.vb = cv.take<uint32_t>(k);
    L.sort_bytes = sort_pairs_temp_bytes((size_t)k, 32);
    L.sort_temp = cv.take<char>(L.sort_bytes);
  }
  return L;
}

}  // namespace drx

extern "C" size_t drx_topk_scratch_bytes_k(int32_t R, int32_t n, int32_t k) {
  if (R < 1 || n <= 16384 || k < 1) return 0;
  drx::Carver cv(nullptr, 0);
  (void)drx::topk_layout(cv, R, k < n ? k : n);
  return drx::align_up(cv.off, 256) + 256;
}

/* (kept for callers that size the scratch before they know k: the bound for k = n) */
extern "C" size_t drx_topk_scratch_bytes(int32_t R, int32_t n) { return drx_topk_scratch_bytes_k(R, n, n); }

extern "C" int drx_topk(const float *scores, const uint32_t *cand_mask, int32_t R, int32_t n, int32_t k, int32_t *out_idx,
                        float *out_val, void *scratch, size_t scratch_bytes, void *stream) {
  if (!scores || !out_idx || !out_val || R < 1 || n < 1 || k < 1) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int npad = 2;
  while (npad < n) npad <<= 1;
  if (npad > 16384) {
    if ((long long)R * n > 0x7FFFFFFFll || !scratch) return DRX_EINVAL;
    const int ks = k < n ? k : n;                         // keys selected per row; out columns beyond them are "missing"
    drx::Carver cv(scratch, scratch_bytes);
    drx::TopkLayout L = drx::topk_layout(cv, R, ks);
    if (!cv.ok()) return DRX_ESCRATCH;
    hipLaunchKernelGGL(drx::k_topk_select, dim3(R), dim3(drx::kSelThreads), 0, st, scores, cand_mask, n, ks, L.cand, L.n_cand);
    if (ks <= 16384) {
      int kpad = 2;
      while (kpad < ks) kpad <<= 1;
      const size_t lds = (size_t)kpad * sizeof(unsigned long long);
      DRX_HIP(hipFuncSetAttribute((const void *)drx::k_topk_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(drx::k_topk_order, dim3(R), dim3(drx::kBlock), lds, st, L.cand, L.n_cand, scores, n, ks, k, kpad, out_idx, out_val);
    } else {
      for (int r = 0; r < R; ++r) {
        int have = 0;
        DRX_HIP(hipMemcpyAsync(&have, L.n_cand + r, sizeof(int), hipMemcpyDeviceToHost, st));
        DRX_HIP(hipStreamSynchronize(st));
        const unsigned long long *cand = L.cand + (size_t)r * ks;
        if (have > 0) {
          hipLaunchKernelGGL(drx::k_topk_split, dim3(256), dim3(256), 0, st, cand, have, L.ka, L.va);
          int rc = drx::sort_pairs(L.sort_temp, L.sort_bytes, L.ka, L.kb, L.va, L.vb, (size_t)have, 32, st);
          if (rc) return rc;
          hipLaunchKernelGGL(drx::k_topk_score_keys, dim3(256), dim3(256), 0, st, cand, L.vb, have, L.ka);
          rc = drx::sort_pairs(L.sort_temp, L.sort_bytes, L.ka, L.kb, L.vb, L.va, (size_t)have, 32, st);
          if (rc) return rc;
        }
        hipLaunchKernelGGL(drx::k_topk_emit_sorted, dim3(256), dim3(256), 0, st, cand, L.va, have, scores + (size_t)r * n, k,
                           out_idx + (size_t)r * k, out_val + (size_t)r * k);
      }
    }
    DRX_LAUNCH_CHECK();
    return DRX_OK;
  }
  if (k <= drx::kWaveTopkMaxK && n <= 64 * 64) {            // few of a short row: selection by one wave per row
    const dim3 grid((unsigned)((R + drx::kBlock / 64 - 1) / (drx::kBlock / 64)));
    if (n <= 64 * 16) hipLaunchKernelGGL(drx::k_topk_wave<16>, grid, dim3(drx::kBlock), 0, st, scores, cand_mask, R, n, k, out_idx, out_val);
    else if (n <= 64 * 32) hipLaunchKernelGGL(drx::k_topk_wave<32>, grid, dim3(drx::kBlock), 0, st, scores, cand_mask, R, n, k, out_idx, out_val);
    else hipLaunchKernelGGL(drx::k_topk_wave<64>, grid, dim3(drx::kBlock), 0, st, scores, cand_mask, R, n, k, out_idx, out_val);
    DRX_LAUNCH_CHECK();
    return DRX_OK;
  }
  const size_t lds = (size_t)npad * sizeof(unsigned long long);
  DRX_HIP(hipFuncSetAttribute((const void *)drx::k_topk_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(drx::k_topk_lds, dim3(R), dim3(drx::kBlock), lds, st, scores, cand_mask, n, k, npad, out_idx, out_val);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}
