// Segmented reduction over a key-sorted touch list, shared by the single-GPU sparse step (drx_cdae.hip) and the
// row-sharded multi-GPU step (drx_shard.hip).  A POLICY supplies where a touch's contribution row comes from and
// what happens to a finished segment (apply the optimizer to a table row, or park the gradient row for an exchange):
//
//   struct Policy {
//     // contribution of touch (key, val): row (float4[J] of this lane), scalar side-value, multiplier
//     template <int G, int J> __device__ void load(uint32_t key, uint32_t val, int lane, float4 (&row)[J], float &sc, float &coef) const;
//     // segment `key` is complete: g = sum coef*row, gs = sum sc; `pos` is the sorted position of one of its touches
//     template <int G, int J> __device__ void finish(uint32_t key, int pos, int lane, const float4 (&g)[J], float gs) const;
//   };
//
// Work split: fixed chunks of kChunk touches per group (load-balanced under Zipf skew); segments inside one chunk are
// finished in place, chunk-crossing segments leave partial rows that the two fix-up tiers combine in chunk order.
// Every sum has a fixed order => bit-reproducible results, no float atomics.
#pragma once
#include "drx_rows.hpp"

namespace drx {

#ifndef DRX_CHUNK
#define DRX_CHUNK 32
#endif
constexpr int kChunk = DRX_CHUNK;      // touches per group in the segmented reduction
#ifdef DRX_SEG_WAVE_PER_CHUNK
#define SEG_GPB(G) (drx::kBlock / 64)
#else
#define SEG_GPB(G) (drx::kBlock / (G))
#endif
constexpr int kShortSpan = 64;     // chunk borders a segment may cross and still be combined by one group
constexpr int kFixBlock = 512;     // (1024: the long-span fix-up kernel hit the 128-VGPR cap of a 16-wave workgroup and spilled)

struct SegBufs {
  const uint32_t *keys_s, *vals_s;            // [T] sorted touches (padding keys DRX_KEY_NONE sort last)
  float *phead, *ptail;                       // [n_chunks, ld]
  float *phs, *pts;                           // [n_chunks] scalar partials
  uint32_t *span_list, *long_list;            // [n_chunks] each
  uint32_t *n_span;                           // [0] crossing segments, [1] long ones (zeroed by the caller)
  uint8_t *cflag;                             // [n_chunks] 0: chunk's first segment starts here; 2: the whole chunk is the
                                              //   middle of one crossing segment; 1: it starts with the END of one
  int T, n_chunks, ld;
};

// Segmented reduction over the sorted touch list in fixed chunks of kChunk touches per group.
// Segments that lie inside one chunk are updated here; segments crossing chunk borders leave
// partial rows that k_span_fixup combines in chunk order (deterministic).
// The chunk's (key, sample) pairs are fetched with one coalesced load per lane and broadcast by shuffles; the
// contribution rows are then loaded LB at a time (independent loads in flight) before they are folded in order.
// LB1 = contribution rows a group of one-float4-per-lane rows (J == 1) keeps in flight.  2 where segments are short (the 10M x 1M
// set: 0.09 touches per table row; 92 instead of 127 VGPRs, 5 waves per SIMD instead of 4: the reduction 0.215 -> 0.192 ms, the step
// 150 -> 160 M triples/s); 8 where a row collects hundreds of touches (the MovieLens shapes: +5 % there).
template <int G, int J, class Policy, int LB1 = 2>
__global__ __launch_bounds__(kBlock) void k_seg_reduce(SegBufs S, Policy pol) {
#ifdef DRX_SEG_WAVE_PER_CHUNK
  // experiment: one chunk per WAVE (lanes >= G idle) so that a group's flush never stalls a sibling group
  const int lane = threadIdx.x % 64;
  if (lane >= G) return;
  const int g = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
#else
  const int lane = threadIdx.x % G;
  const int g = blockIdx.x * (kBlock / G) + threadIdx.x / G;
#endif
  if (g >= S.n_chunks) return;
  const int start = g * kChunk, end = min(S.T, start + kChunk);
  const int n = end - start;
  const uint32_t prev_key = start > 0 ? S.keys_s[start - 1] : DRX_KEY_NONE;
  const uint32_t next_key = end < S.T ? S.keys_s[end] : DRX_KEY_NONE;
  constexpr int KPL = (kChunk + G - 1) / G;          // (key, val) registers per lane
  constexpr int LB = J == 1 ? LB1 : (J == 2 ? 4 : 2);  // rows in flight per group
  uint32_t kreg[KPL], vreg[KPL];
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const int t = r * G + lane;
    const bool ok = t < n && t < kChunk;
    kreg[r] = ok ? S.keys_s[start + t] : DRX_KEY_NONE;
    vreg[r] = ok ? S.vals_s[start + t] : 0u;
  }
  auto bcast = [&](const uint32_t (&reg)[KPL], int t) -> uint32_t {
    uint32_t sel = reg[0];
#pragma unroll
    for (int r = 1; r < KPL; ++r) sel = (t / G == r) ? reg[r] : sel;
    return (uint32_t)__shfl((int)sel, t % G, G);
  };
  float4 acc[J];
#pragma unroll
  for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
  float accs = 0.f;
  uint32_t cur = DRX_KEY_NONE;
  int cur_pos = 0;
  bool cur_from_start = false;
  auto flush = [&](bool at_end) {
    if (cur == DRX_KEY_NONE) return;
    const bool cont_left = cur_from_start && prev_key == cur;
    const bool cont_right = at_end && next_key == cur;
    if (!cont_left && !cont_right) {
      pol.template finish<G, J>(cur, cur_pos, lane, acc, accs);
    } else if (cont_left) {
      store_row<G, J>(S.phead, (size_t)g, S.ld, lane, acc);
      if (lane == 0) S.phs[g] = accs;
    } else {
      store_row<G, J>(S.ptail, (size_t)g, S.ld, lane, acc);
      if (lane == 0) {
        S.pts[g] = accs;
        const uint32_t slot = atomicAdd(S.n_span, 1u);
        S.span_list[slot] = (uint32_t)g;
      }
    }
  };
  for (int t0 = 0; t0 < n; t0 += LB) {
    uint32_t k8[LB];
    float s8[LB], c8[LB];
    float4 rows[LB][J];
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int t = t0 + u;
      k8[u] = t < n ? bcast(kreg, t) : DRX_KEY_NONE;    // padding (dropped inputs) sorts last
      const uint32_t b = bcast(vreg, t < n ? t : 0);
      s8[u] = 0.f;
      c8[u] = 1.f;
#pragma unroll
      for (int jx = 0; jx < J; ++jx) rows[u][jx] = f4_zero();
      if (k8[u] != DRX_KEY_NONE) pol.template load<G, J>(k8[u], b, lane, rows[u], s8[u], c8[u]);
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const uint32_t key = k8[u];
      if (key != DRX_KEY_NONE) {
        if (key != cur) {
          flush(false);
          cur = key;
          cur_from_start = (t0 + u == 0);
#pragma unroll
          for (int jx = 0; jx < J; ++jx) acc[jx] = f4_zero();
          accs = 0.f;
        }
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], c8[u], rows[u][jx]);
        accs += s8[u];
        cur_pos = start + t0 + u;
      }
    }
  }
  // the last segment ends at the chunk border iff the final touch of the chunk is a real key
  const bool ran_to_end = n > 0 && bcast(kreg, n - 1) != DRX_KEY_NONE;
  flush(ran_to_end);
  if (lane == 0) {
    const uint32_t first = n > 0 ? S.keys_s[start] : DRX_KEY_NONE, last = n > 0 ? S.keys_s[end - 1] : DRX_KEY_NONE;
    const bool cont = first != DRX_KEY_NONE && first == prev_key;
    const bool middle = cont && last == first && next_key == first;
    S.cflag[g] = middle ? 2 : (cont ? 1 : 0);
  }
}

// Fix-up of chunk-crossing segments, two tiers.
//   k_span_short : one GROUP per crossing segment: tail partial of its first chunk + head partials of the next chunks
//                  whose first key equals the segment key, in chunk order.  Segments that cross more than
//                  kShortSpan chunk borders (hot items) are queued for
//   k_span_long  : one 1024-thread workgroup per such segment; its R = 1024/G groups stride over the chunks and the
//                  R partial sums are combined in a fixed order.  Both tiers are deterministic.

template <int G, int J, class Policy>
__device__ __forceinline__ void span_short_body(const SegBufs &S, const Policy &pol, int block_id, int n_blocks) {
  const int lane = threadIdx.x % G;
  const int gshift = (threadIdx.x & 63) / G * G;          // position of this group's lanes in the wave's ballot
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const uint32_t n_span = S.n_span[0];
  const int gpb = kBlock / G;
  constexpr int UL = J == 1 ? 8 : 2;                       // partial rows in flight
  for (uint32_t si = block_id * gpb + threadIdx.x / G; si < n_span; si += n_blocks * gpb) {
    const int g0 = (int)S.span_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    // m = number of following chunks that continue this segment: a run of "middle" chunks (flag 2), closed by an
    // optional "end" chunk (flag 1); the lanes probe G one-byte chunk flags at a time
    int m = 0;
    bool is_long = false;
    for (int base = g0 + 1;; base += G) {
      const int c = base + lane;
      const int fl = c < S.n_chunks ? (int)S.cflag[c] : 0;
      const unsigned long long mid = (__ballot(fl == 2) >> gshift) & gmask;
      const int run = mid == gmask ? G : __builtin_ctzll(~mid);       // leading run of middle chunks
      m += run;
      if (run < G) {
        m += (__shfl(fl, run, G) == 1) ? 1 : 0;
        break;
      }
      if (m >= kShortSpan) { is_long = true; break; }
    }
    if (is_long || m > kShortSpan) {
      if (lane == 0) S.long_list[atomicAdd(&S.n_span[1], 1u)] = (uint32_t)g0;
      continue;
    }
    float4 t[J];
    load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
    float ts = S.pts[g0];
    for (int c0 = g0 + 1; c0 <= g0 + m; c0 += UL) {
      float4 v[UL][J];
      float sv[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        sv[u] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
        if (c0 + u <= g0 + m) { load_row<G, J>(S.phead, (size_t)(c0 + u), S.ld, lane, v[u]); sv[u] = S.phs[c0 + u]; }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {           // chunk order
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[u][j]);
        ts += sv[u];
      }
    }
    pol.template finish<G, J>(key, min(S.T, (g0 + 1) * kChunk) - 1, lane, t, ts);
  }
}

template <int G, int J, class Policy>
__global__ __launch_bounds__(kBlock) void k_span_short(SegBufs S, Policy pol) {
  span_short_body<G, J, Policy>(S, pol, (int)blockIdx.x, (int)gridDim.x);
}

// lds: [R, ld] + [R] floats, R = kFixBlock / G
template <int G, int J, class Policy>
__device__ __forceinline__ void span_long_body(const SegBufs &S, const Policy &pol, int block_id, int n_blocks, float *lds) {
  constexpr int R = kFixBlock / G;
  float *sc = lds + (size_t)R * S.ld;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const uint32_t n_long = S.n_span[1];
  for (uint32_t si = block_id; si < n_long; si += n_blocks) {
    const int g0 = (int)S.long_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    float accs = 0.f;
    // span length: all threads probe one-byte chunk flags, kFixBlock at a time (a run of "middle" chunks, closed by
    // an optional "end" chunk); positions beyond the last chunk read as 0, so the loop always terminates
    __shared__ int s_stop, s_len;
    int len = 0;
    for (int base = g0 + 1;; base += kFixBlock) {
      if (threadIdx.x == 0) s_stop = kFixBlock;
      __syncthreads();
      const int c = base + (int)threadIdx.x;
      const int fl = c < S.n_chunks ? (int)S.cflag[c] : 0;
      if (fl != 2) atomicMin(&s_stop, (int)threadIdx.x);
      __syncthreads();
      const int stop = s_stop;
      if (stop < kFixBlock) {
        if ((int)threadIdx.x == stop) s_len = len + stop + (fl == 1 ? 1 : 0);
        __syncthreads();
        break;
      }
      len += kFixBlock;
      __syncthreads();
    }
    const int m = s_len;
    constexpr int UL = J == 1 ? 8 : 2;        // partial rows in flight per group
    for (int c = g0 + 1 + r; c <= g0 + m; c += UL * R) {
      float4 v[UL][J];
      float sv[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int cu = c + u * R;
        sv[u] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
        if (cu <= g0 + m) { load_row<G, J>(S.phead, (size_t)cu, S.ld, lane, v[u]); sv[u] = S.phs[cu]; }
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(acc[j], v[u][j]);
        accs += sv[u];
      }
    }
    __syncthreads();
    store_row<G, J>(lds, (size_t)r, S.ld, lane, acc);
    if (lane == 0) sc[r] = accs;
    __syncthreads();
    if (r == 0) {
      float4 t[J];
      load_row<G, J>(S.ptail, (size_t)g0, S.ld, lane, t);
      float ts = S.pts[g0];
#pragma unroll 8
      for (int rr = 0; rr < R; ++rr) {
        float4 v[J];
        load_row<G, J>(lds, (size_t)rr, S.ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
        ts += sc[rr];
      }
      pol.template finish<G, J>(key, min(S.T, (g0 + 1) * kChunk) - 1, lane, t, ts);
    }
  }
}

template <int G, int J, class Policy>
__global__ __launch_bounds__(kFixBlock) void k_span_long(SegBufs S, Policy pol) {
  extern __shared__ __align__(16) float lds[];   // [R, ld] + [R]
  span_long_body<G, J, Policy>(S, pol, (int)blockIdx.x, (int)gridDim.x, lds);
}

}  // namespace drx
