// Row-sharded (multi-GPU) CDAE sampled step: the per-rank device work around the two RCCL all-to-all exchanges.
//
// Layout: users — V rows, their histories and the triples sampled for them — are sharded by contiguous
// uid range and never move.  Item-side rows (W, W2T, b2 and their optimizer slots) are sharded by contiguous item
// range; a step requests each DISTINCT item row it touches from its owner (all-to-all of ids, then of rows), runs
// forward/backward against that row cache, reduces its contributions per distinct row, and returns one gradient row
// per distinct row to the owner (all-to-all), which sums duplicates across ranks in rank order and applies the sparse
// optimizer.  Everything here is per-rank and collective-free; drecpy_amd/dist.py drives the exchanges.
//
// Key space (owner-major so that a rank's sorted distinct keys are contiguous per owner):
//     item n -> owner o = n / ipr, l = n - o*ipr ;  W row: o*2ipr + l ;  W2T row: o*2ipr + ipr + l
//     local user u -> world*2ipr + u
#include "drx_common.hpp"
#include "drx_rows.hpp"
#include "drx_segreduce.hpp"
#include "drx_scan.hpp"

namespace drx {

__device__ __forceinline__ uint32_t item_key(const DrxShard &sh, int n, int is_out) {
  const int o = n / sh.items_per_rank;
  return (uint32_t)(o * 2 * sh.items_per_rank + (is_out ? sh.items_per_rank : 0) + (n - o * sh.items_per_rank));
}
__device__ __forceinline__ uint32_t user_key0(const DrxShard &sh) { return (uint32_t)(sh.world * 2 * sh.items_per_rank); }

// ---- 1. touches of the local batch -----------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_shard_touches(DrxShard sh, DrxHistory H, DrxBatch bt, uint32_t qthr,
                                                          uint32_t *keys, uint32_t *vals, uint32_t *b_of_pos) {
  // one 16-lane group per sample
  constexpr int G = 16;
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const int base = bt.keep_off[b] + 2 * b;
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  for (int64_t j = s + lane; j < e; j += G) {
    const uint32_t jj = (uint32_t)(j - s);
    const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
    keys[base + jj] = kf ? item_key(sh, H.indices[j], 0) : DRX_KEY_NONE;
    vals[base + jj] = (uint32_t)(base + jj);
    b_of_pos[base + jj] = (uint32_t)b;
  }
  if (lane == 0) {
    const int deg = (int)(e - s);
    keys[base + deg] = item_key(sh, bt.iid[b], 1);
    keys[base + deg + 1] = user_key0(sh) + (uint32_t)u;
    vals[base + deg] = (uint32_t)(base + deg);
    vals[base + deg + 1] = (uint32_t)(base + deg + 1);
    b_of_pos[base + deg] = b_of_pos[base + deg + 1] = (uint32_t)b;
  }
}

// ---- 2. distinct keys, slots, per-owner bounds -----------------------------------------------------------------------
__global__ void k_head_flags(const uint32_t *__restrict__ keys_s, int T, int *flag) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    const uint32_t k = keys_s[j];
    flag[j] = (k != DRX_KEY_NONE && (j == 0 || keys_s[j - 1] != k)) ? 1 : 0;
  }
}

// slot_sorted holds the INCLUSIVE scan of the head flags; slot = scan - 1
__global__ void k_slots(const uint32_t *__restrict__ keys_s, const uint32_t *__restrict__ vals_s, int T, int *slot_sorted,
                        uint32_t *slot_of_pos, uint32_t *uniq_keys) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    const uint32_t k = keys_s[j];
    if (k == DRX_KEY_NONE) { slot_of_pos[vals_s[j]] = DRX_KEY_NONE; slot_sorted[j] = -1; continue; }
    const int slot = slot_sorted[j] - 1;
    slot_sorted[j] = slot;
    slot_of_pos[vals_s[j]] = (uint32_t)slot;
    if (j == 0 || keys_s[j - 1] != k) uniq_keys[slot] = k;
  }
}

// bounds[o] = number of distinct keys < o*2ipr for o = 0..world (bounds[world] = first user key); bounds[world+1] = Q
__global__ void k_owner_bounds(const uint32_t *__restrict__ keys_s, const int *__restrict__ slot_sorted, int T, DrxShard sh,
                               int32_t *bounds) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > sh.world + 1) return;
  const uint32_t target = o <= sh.world ? (uint32_t)(o * 2 * sh.items_per_rank) : DRX_KEY_NONE;
  int lo = 0, hi = T;                         // lower_bound of target over the sorted touches
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys_s[mid] < target) lo = mid + 1; else hi = mid;
  }
  // distinct keys before position lo = slot of the last real touch before lo, + 1
  bounds[o] = lo == 0 ? 0 : slot_sorted[lo - 1] + 1;
}

// ---- 3. owner side: fetch requested rows -------------------------------------------------------------------------
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_shard_gather_rows(DrxCdaeParams P, DrxShard sh, const uint32_t *__restrict__ req, int n,
                                                              float *__restrict__ rows, float *__restrict__ b2out) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  for (int i = blockIdx.x * gpb + threadIdx.x / G; i < n; i += gridDim.x * gpb) {
    const uint32_t t = req[i] - (uint32_t)(sh.rank * 2 * sh.items_per_rank);     // local key in [0, 2*ipr)
    const bool is_out = t >= (uint32_t)sh.items_per_rank;
    const size_t row = is_out ? t - sh.items_per_rank : t;
    float4 v[J];
    load_row<G, J>(is_out ? P.W2T : P.W, row, P.ld, lane, v);
    store_row<G, J>(rows, (size_t)i, P.ld, lane, v);
    if (lane == 0) b2out[i] = is_out ? P.b2[row] : 0.f;
  }
}

// ---- 4. forward/backward against the row cache -----------------------------------------------------------------------
struct ShardFwd {
  const uint32_t *slot_of_pos;   // [T]
  const float *rows;             // [Q_item, ld] requested rows, in distinct-key order
  const float *b2c;              // [Q_item]
  float *dz1, *g2, *dz2, *lossb;
  float inv_b_norm;              // 1 / global batch
};

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_shard_fwd_bwd(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale,
                                                          int loss_kind, ShardFwd F) {
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  const int u = bt.uid[b];
  const float y = bt.y[b];
  const int base = bt.keep_off[b] + 2 * b;
  const int deg = bt.keep_off[b + 1] - bt.keep_off[b];
  float4 acc[J], h[J], w2[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int c = 0; c < deg; c += G) {
    const int jj = c + lane;
    uint32_t slot = DRX_KEY_NONE;
    if (jj < deg) slot = F.slot_of_pos[base + jj];
    const int n_here = min(G, deg - c);
    for (int t = 0; t < n_here; t += 4) {
      uint32_t s0 = (uint32_t)__shfl((int)slot, t, G), s1 = (uint32_t)__shfl((int)slot, t + 1, G);
      uint32_t s2 = (uint32_t)__shfl((int)slot, t + 2, G), s3 = (uint32_t)__shfl((int)slot, t + 3, G);
      if (t + 1 >= n_here) s1 = DRX_KEY_NONE;
      if (t + 2 >= n_here) s2 = DRX_KEY_NONE;
      if (t + 3 >= n_here) s3 = DRX_KEY_NONE;
      float4 r0[J], r1[J], r2[J], r3[J];
#pragma unroll
      for (int jx = 0; jx < J; ++jx) r0[jx] = r1[jx] = r2[jx] = r3[jx] = f4_zero();
      if (s0 != DRX_KEY_NONE) load_row<G, J>(F.rows, (size_t)s0, P.ld, lane, r0);
      if (s1 != DRX_KEY_NONE) load_row<G, J>(F.rows, (size_t)s1, P.ld, lane, r1);
      if (s2 != DRX_KEY_NONE) load_row<G, J>(F.rows, (size_t)s2, P.ld, lane, r2);
      if (s3 != DRX_KEY_NONE) load_row<G, J>(F.rows, (size_t)s3, P.ld, lane, r3);
#pragma unroll
      for (int jx = 0; jx < J; ++jx) {
        f4_add(acc[jx], r0[jx]); f4_add(acc[jx], r1[jx]); f4_add(acc[jx], r2[jx]); f4_add(acc[jx], r3[jx]);
      }
    }
  }
  hidden_act<G, J>(P, u, scale, lane, acc, h);
  const uint32_t so = F.slot_of_pos[base + deg];
  load_row<G, J>(F.rows, (size_t)so, P.ld, lane, w2);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) d += f4_dot(w2[j], h[j]);
  d = group_sum<G>(d);
  const float p = sigmoidf_(d + F.b2c[so]);
  float lval, dp;
  if (loss_kind == DRX_LOSS_BCE) { lval = bce_elem(y, p); dp = bce_grad(y, p) * F.inv_b_norm; }
  else { lval = (p - y) * (p - y); dp = 2.0f * (p - y) * F.inv_b_norm; }
  const float dz2 = dp * p * (1.0f - p);
  float4 dz1[J], g2[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    dz1[j].x = dz2 * w2[j].x * h[j].x * (1.0f - h[j].x); dz1[j].y = dz2 * w2[j].y * h[j].y * (1.0f - h[j].y);
    dz1[j].z = dz2 * w2[j].z * h[j].z * (1.0f - h[j].z); dz1[j].w = dz2 * w2[j].w * h[j].w * (1.0f - h[j].w);
    g2[j].x = dz2 * h[j].x; g2[j].y = dz2 * h[j].y; g2[j].z = dz2 * h[j].z; g2[j].w = dz2 * h[j].w;
  }
  store_row<G, J>(F.dz1, (size_t)b, P.ld, lane, dz1);
  store_row<G, J>(F.g2, (size_t)b, P.ld, lane, g2);
  if (lane == 0) { F.dz2[b] = dz2; F.lossb[b] = lval; }
}

// ---- 5. local reduction: one gradient row per distinct item row, V rows updated in place -----------------------------------
struct LocalPolicy {
  DrxCdaeParams P;               // local tables (V used here)
  DrxOptim opt;
  DrxShard sh;
  int b_norm;
  float scale;
  const float *dz1;              // g2 = dz1 + g2_off (value select, see DirectPolicy in drx_cdae.hip)
  long long g2_off;
  const float *dz2;
  const uint32_t *b_of_pos;
  const int *slot_sorted;
  float *gc, *gb2c;              // [Q_item, ld], [Q_item]
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t key, uint32_t pos, int lane, float4 (&row)[J], float &sc, float &coef) const {
    const uint32_t b = b_of_pos[pos];
    const uint32_t uk0 = user_key0(sh);
    const bool is_user = key >= uk0;
    const bool is_out = !is_user && (key % (2u * sh.items_per_rank)) >= (uint32_t)sh.items_per_rank;
    load_row<G, J>(dz1 + (is_out ? g2_off : 0ll), (size_t)b, P.ld, lane, row);
    if (is_out) sc = dz2[b];
    coef = (!is_user && !is_out) ? scale : 1.0f;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int pos, int lane, const float4 (&g)[J], float gs) const {
    const uint32_t uk0 = user_key0(sh);
    if (key >= uk0) {
      const size_t row = key - uk0;
      OptScalars o = opt_for(opt, 0, b_norm);
      o.inv_k = 1.0f / (float)P.k;
      float4 w[J];
      load_row<G, J>(P.V, row, P.ld, lane, w);
      row_update<G, J>(o, P.V, opt.s1[2], opt.s2[2], row, P.ld, lane, w, g);
    } else {
      const int slot = slot_sorted[pos];
      store_row<G, J>(gc, (size_t)slot, P.ld, lane, g);
      if (lane == 0) gb2c[slot] = gs;
    }
  }
};

// ---- 6. owner side: sum the gradient rows received for each owned row (rank order) and apply the optimizer ----------
struct OwnerPolicy {
  DrxCdaeParams P;               // local (owned) item tables
  DrxOptim opt;
  DrxShard sh;
  int b_norm;
  const float *recv_rows, *recv_b2;
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t, uint32_t idx, int lane, float4 (&row)[J], float &sc, float &coef) const {
    load_row<G, J>(recv_rows, (size_t)idx, P.ld, lane, row);
    sc = recv_b2[idx];
    coef = 1.0f;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int, int lane, const float4 (&g)[J], float gs) const {
    const uint32_t t = key - (uint32_t)(sh.rank * 2 * sh.items_per_rank);
    const bool is_out = t >= (uint32_t)sh.items_per_rank;
    const size_t row = is_out ? t - sh.items_per_rank : t;
    OptScalars o = opt_for(opt, 0, b_norm);
    o.inv_k = 1.0f / (float)P.k;
    float4 w[J];
    float *wt = P.W, *w2 = P.W2T, *a0 = opt.s1[0], *a1 = opt.s1[1], *c0 = opt.s2[0], *c1 = opt.s2[1];   // scalar loads first
    float *tab = is_out ? w2 : wt;
    load_row<G, J>(tab, row, P.ld, lane, w);
    row_update<G, J>(o, tab, is_out ? a1 : a0, is_out ? c1 : c0, row, P.ld, lane, w, g);
    if (is_out && lane == 0) {
      float pb = P.b2[row], m = opt.s1[4][row], v = o.kind == DRX_OPT_ADAM ? opt.s2[4][row] : 0.f;
      o.rb = 0.f;
      opt_update1(o, gs, pb, m, v);
      P.b2[row] = pb; opt.s1[4][row] = m;
      if (o.kind == DRX_OPT_ADAM) opt.s2[4][row] = v;
    }
  }
};

// The rows a rank receives are W segments (one per source rank, in rank order), each with ascending DISTINCT keys.  Instead
// of sorting them again, every received row is entered in a direct-address table tab[local key][source] = row index
// (no two writers per entry); then the row of the LOWEST source holding a key sums all holders in source order and
// applies the optimizer — one pass, fixed summation order, no sort, no atomics.
struct SegOff { int off[DRX_MAX_WORLD * DRX_MAX_MICRO + 1]; int n_seg; };   // segments: micro-batch-major, then source rank

__device__ __forceinline__ int source_of(const SegOff &so, int j) {
  int s = 0;
  while (s + 1 < so.n_seg && j >= so.off[s + 1]) ++s;
  return s;
}

__global__ void k_owner_scatter(DrxShard sh, SegOff so, const uint32_t *__restrict__ recv_keys, int n, uint32_t *tab) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t lk = recv_keys[j] - (uint32_t)(sh.rank * 2 * sh.items_per_rank);
    tab[(size_t)lk * so.n_seg + source_of(so, j)] = (uint32_t)j;
  }
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_owner_apply(OwnerPolicy pol, SegOff so, const uint32_t *__restrict__ recv_keys, int n,
                                                        const uint32_t *__restrict__ tab) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  const int W = so.n_seg;                          // holders of a key are looked up per (micro-batch, source rank) segment
  for (int j = blockIdx.x * gpb + threadIdx.x / G; j < n; j += gridDim.x * gpb) {
    const uint32_t key = recv_keys[j];
    const uint32_t lk = key - (uint32_t)(pol.sh.rank * 2 * pol.sh.items_per_rank);
    const uint32_t *row_tab = tab + (size_t)lk * W;
    float4 g[J];
#pragma unroll
    for (int jx = 0; jx < J; ++jx) g[jx] = f4_zero();
    float gs = 0.f;
    bool leader = true, first = true;
    for (int s0 = 0; s0 < W && leader; s0 += G) {                 // G sources at a time, one table entry per lane
      const uint32_t e = (s0 + lane < W) ? row_tab[s0 + lane] : DRX_KEY_NONE;
      const int n_here = min(G, W - s0);
      for (int t = 0; t < n_here; t += 4) {
        uint32_t r[4];
        float4 v[4][J];
        float sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          r[q] = (t + q < n_here) ? (uint32_t)__shfl((int)e, t + q, G) : DRX_KEY_NONE;
          sc[q] = 0.f;
#pragma unroll
          for (int jx = 0; jx < J; ++jx) v[q][jx] = f4_zero();
        }
        if (first) {                                               // the first holder must be this very row
          uint32_t r0 = DRX_KEY_NONE;
#pragma unroll
          for (int q = 3; q >= 0; --q) if (r[q] != DRX_KEY_NONE) r0 = r[q];
          if (r0 != DRX_KEY_NONE) { first = false; if (r0 != (uint32_t)j) { leader = false; break; } }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (r[q] != DRX_KEY_NONE) { load_row<G, J>(pol.recv_rows, (size_t)r[q], pol.P.ld, lane, v[q]); sc[q] = pol.recv_b2[r[q]]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int jx = 0; jx < J; ++jx) f4_add(g[jx], v[q][jx]);
          gs += sc[q];
        }
      }
    }
    if (leader && !first) pol.template finish<G, J>(key, 0, lane, g, gs);
  }
}

__global__ void k_iota(uint32_t *v, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

// ---- 7. hidden bias --------------------------------------------------------------------------------------------
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_colsum_partial(int ld, int B, const float *__restrict__ x, float *__restrict__ part,
                                                           int rows_per_block, const float *__restrict__ lossb,
                                                           float *__restrict__ loss_part) {
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[kBlock / 64];
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b0 = blockIdx.x * rows_per_block, b1 = min(B, b0 + rows_per_block);
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int b = b0 + r; b < b1; b += R) {
    float4 v[J];
    load_row<G, J>(x, (size_t)b, ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
    }
    store_row<G, J>(part, (size_t)blockIdx.x, ld, lane, t);
  }
  float a = 0.f;                                   // this block's slice of the per-sample losses
  for (int b = b0 + (int)threadIdx.x; b < b1; b += kBlock) a += lossb[b];
  const float tl = block_sum(a, red);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = tl;
}

// out[0..ld) = sum of the n_part partial rows; out[ld] = sum of the n_part loss partials   (one workgroup, fixed order)
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_colsum_final(int ld, const float *__restrict__ part, int n_part,
                                                         const float *__restrict__ lossb, int B, float *__restrict__ out) {
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[kBlock / 64];
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int i = r; i < n_part; i += R) {
    float4 v[J];
    load_row<G, J>(part, (size_t)i, ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
    }
    store_row<G, J>(out, 0, ld, lane, t);
  }
  float a = 0.f;
  for (int i = threadIdx.x; i < n_part; i += kBlock) a += lossb[i];      // lossb = per-block loss partials here
  const float tl = block_sum(a, red);
  if (threadIdx.x == 0) out[ld] = tl;
  (void)B;
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_bias_apply(DrxCdaeParams P, DrxOptim opt, int b_norm, const float *__restrict__ grad) {
  const int lane = threadIdx.x % G;
  if (threadIdx.x < G) {
    float4 g[J], w[J];
    load_row<G, J>(grad, 0, P.ld, lane, g);
    load_row<G, J>(P.b, 0, P.ld, lane, w);
    OptScalars o = opt_for(opt, 0, b_norm);
    if (o.kind == DRX_OPT_ROWWISE_ADAGRAD) o.kind = DRX_OPT_ADAGRAD;
    o.rb = 0.f;
    row_update<G, J>(o, P.b, opt.s1[3], opt.s2[3], 0, P.ld, lane, w, g);
  }
}

struct SegLayout {
  SegBufs sb;
  uint32_t *idx, *keys_s, *vals_s;
  void *sort_temp;
  size_t sort_bytes;
  int *flags;
  void *scan_temp;
  size_t scan_bytes;
  float *part;
};

static SegLayout seg_layout(Carver &cv, int ld, int T, int sort_bits) {
  SegLayout L{};
  const int n_chunks = (T + kChunk - 1) / kChunk;
  L.sb.T = T; L.sb.n_chunks = n_chunks; L.sb.ld = ld;
  L.sb.phead = cv.take<float>((size_t)n_chunks * ld);
  L.sb.ptail = cv.take<float>((size_t)n_chunks * ld);
  L.sb.phs = cv.take<float>(n_chunks);
  L.sb.pts = cv.take<float>(n_chunks);
  L.sb.span_list = cv.take<uint32_t>(n_chunks);
  L.sb.long_list = cv.take<uint32_t>(n_chunks);
  L.sb.n_span = cv.take<uint32_t>(64);
  L.sb.cflag = cv.take<uint8_t>(n_chunks);
  L.idx = cv.take<uint32_t>(T);
  L.keys_s = cv.take<uint32_t>(T);
  L.vals_s = cv.take<uint32_t>(T);
  L.sort_bytes = sort_pairs_temp_bytes((size_t)T, sort_bits);
  L.sort_temp = cv.take<char>(L.sort_bytes);
  L.flags = cv.take<int>(T);
  L.scan_bytes = scan_i32_temp_bytes((size_t)(T > 0 ? T : 1));
  L.scan_temp = cv.take<char>(L.scan_bytes);
  L.part = cv.take<float>((size_t)256 * ld);
  return L;
}

static uint32_t *owner_table(Carver &cv, const DrxShard &sh) {
  return cv.take<uint32_t>((size_t)2 * sh.items_per_rank * sh.world * DRX_MAX_MICRO);
}

static int key_bits(const DrxShard &sh) {
  return bits_for((uint64_t)sh.world * 2 * sh.items_per_rank + (uint64_t)sh.n_users_local + 1);
}

static int check_shard(const DrxShard *sh) {
  if (!sh || sh->world < 1 || sh->world > DRX_MAX_WORLD || sh->rank < 0 || sh->rank >= sh->world || sh->items_per_rank < 1 || sh->n_items < 1 ||
      sh->n_users_local < 1)
    return DRX_EINVAL;
  if ((uint64_t)sh->world * 2 * sh->items_per_rank + (uint64_t)sh->n_users_local + 1 >= 0xFFFFFFFFull) return DRX_EINVAL;
  return DRX_OK;
}

template <class Policy>
static int run_segreduce(const DrxCdaeParams &P, const SegBufs &sb, const Policy &pol, hipStream_t st) {
  if (sb.T == 0) return DRX_OK;
  DRX_HIP(hipMemsetAsync(sb.n_span, 0, 2 * sizeof(uint32_t), st));
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    hipLaunchKernelGGL((k_seg_reduce<G, J, Policy>), dim3((sb.n_chunks + SEG_GPB(G) - 1) / SEG_GPB(G)), dim3(kBlock), 0, st, sb, pol); \
    hipLaunchKernelGGL((k_span_short<G, J, Policy>), dim3(1024), dim3(kBlock), 0, st, sb, pol);                        \
    const size_t lds = ((size_t)(kFixBlock / G) * (P.ld + 1)) * 4;                                                     \
    if (lds > 48 * 1024)                                                                                               \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_long<G, J, Policy>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)lds));                                                                          \
    hipLaunchKernelGGL((k_span_long<G, J, Policy>), dim3(256), dim3(kFixBlock), lds, st, sb, pol);                     \
  }
  DRX_DISPATCH_GEOM(P.ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // namespace drx

using namespace drx;

extern "C" {

size_t drx_shard_scratch_bytes(const DrxCdaeParams *p, const DrxShard *sh, int32_t n_touches) {
  if (!p || check_shard(sh) || n_touches < 0) return 0;
  Carver cv(nullptr, 0);
  (void)seg_layout(cv, p->ld, n_touches, key_bits(*sh));
  (void)owner_table(cv, *sh);
  return align_up(cv.off, 256) + 256;
}

int drx_shard_touches(const DrxShard *sh, const DrxHistory *hist, const DrxBatch *bt, uint32_t *keys, uint32_t *vals,
                      uint32_t *b_of_pos, void *stream) {
  if (check_shard(sh) || !hist || !hist->indptr || !hist->indices || !bt || !bt->uid || !bt->iid || !bt->keep_off ||
      !keys || !vals || !b_of_pos || bt->B < 1)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T = bt->n_touch_slots + 2 * bt->B;
  DRX_HIP(hipMemsetAsync(keys, 0xFF, (size_t)T * 4, st));
  DRX_HIP(hipMemsetAsync(b_of_pos, 0, (size_t)T * 4, st));
  hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, st, vals, T);
  const int gpb = kBlock / 16;
  hipLaunchKernelGGL(k_shard_touches, dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *sh, *hist, *bt,
                     q_threshold(bt->q), keys, vals, b_of_pos);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_index(const DrxCdaeParams *p, const DrxShard *sh, const uint32_t *keys, const uint32_t *vals, int32_t T,
                    uint32_t *keys_s, uint32_t *vals_s, int32_t *slot_sorted, uint32_t *slot_of_pos, uint32_t *uniq_keys,
                    int32_t *bounds, void *scratch, size_t scratch_bytes, void *stream) {
  if (!p || check_shard(sh) || !keys || !vals || !keys_s || !vals_s || !slot_sorted || !slot_of_pos || !uniq_keys ||
      !bounds || !scratch || T < 1)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  SegLayout L = seg_layout(cv, p->ld, T, key_bits(*sh));
  if (!cv.ok()) return DRX_ESCRATCH;
  int rc = sort_pairs(L.sort_temp, L.sort_bytes, keys, keys_s, vals, vals_s, (size_t)T, key_bits(*sh), st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_head_flags, dim3(1024), dim3(256), 0, st, keys_s, T, L.flags);
  const int scan_rc = scan_i32(L.scan_temp, L.scan_bytes, L.flags, slot_sorted, (size_t)T, true, st);
  if (scan_rc) return scan_rc;
  hipLaunchKernelGGL(k_slots, dim3(1024), dim3(256), 0, st, keys_s, vals_s, T, slot_sorted, slot_of_pos, uniq_keys);
  hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(256), 0, st, keys_s, slot_sorted, T, *sh, bounds);
  DRX_LAUNCH_CHECK();
  return sh->world + 2 <= 256 ? DRX_OK : DRX_EINVAL;
}

int drx_shard_gather_rows(const DrxCdaeParams *p, const DrxShard *sh, const uint32_t *req_keys, int32_t n, float *rows,
                          float *b2_out, void *stream) {
  if (!p || check_shard(sh) || n < 0 || (n > 0 && (!req_keys || !rows || !b2_out))) return DRX_EINVAL;
  if (n == 0) return DRX_OK;
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J)                                                                                              \
  {                                                                                                             \
    const int gpb = kBlock / G;                                                                                 \
    int blocks = (n + gpb - 1) / gpb;                                                                           \
    if (blocks > 4096) blocks = 4096;                                                                           \
    hipLaunchKernelGGL((k_shard_gather_rows<G, J>), dim3(blocks), dim3(kBlock), 0, st, *p, *sh, req_keys, n, rows, b2_out); \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_fwd_bwd(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const uint32_t *slot_of_pos,
                      const float *rows_cache, const float *b2_cache, int32_t b_norm, int32_t loss_kind, float *dz1, float *g2,
                      float *dz2, float *lossb, void *stream) {
  if (!p || !hist || !bt || !bt->uid || !bt->iid || !bt->y || !bt->keep_off || !slot_of_pos || !rows_cache || !b2_cache ||
      !dz1 || !g2 || !dz2 || !lossb || b_norm < 1)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ShardFwd F{slot_of_pos, rows_cache, b2_cache, dz1, g2, dz2, lossb, 1.0f / (float)b_norm};
  const float scale = 1.0f / (1.0f - bt->q);
#define CALL(G, J)                                                                                               \
  {                                                                                                              \
    const int gpb = kBlock / G;                                                                                  \
    hipLaunchKernelGGL((k_shard_fwd_bwd<G, J>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt, scale, \
                       loss_kind, F);                                                                            \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_reduce(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, int32_t b_norm, float q,
                     const uint32_t *keys_s, const uint32_t *vals_s, const int32_t *slot_sorted, const uint32_t *b_of_pos,
                     int32_t T, const float *dz1, const float *g2, const float *dz2, float *gc, float *gb2c, void *scratch,
                     size_t scratch_bytes, void *stream) {
  if (!p || !opt || check_shard(sh) || !keys_s || !vals_s || !slot_sorted || !b_of_pos || !dz1 || !g2 || !dz2 || !gc ||
      !gb2c || !scratch || T < 1 || b_norm < 1)
    return DRX_EINVAL;
  Carver cv(scratch, scratch_bytes);
  SegLayout L = seg_layout(cv, p->ld, T, key_bits(*sh));
  if (!cv.ok()) return DRX_ESCRATCH;
  L.sb.keys_s = keys_s; L.sb.vals_s = vals_s;
  LocalPolicy pol{*p, *opt, *sh, b_norm, 1.0f / (1.0f - q), dz1, (long long)(g2 - dz1), dz2, b_of_pos, slot_sorted, gc, gb2c};
  return run_segreduce(*p, L.sb, pol, (hipStream_t)stream);
}

int drx_shard_apply(const DrxCdaeParams *p, const DrxOptim *opt, const DrxShard *sh, int32_t b_norm, const uint32_t *recv_keys,
                    const float *recv_rows, const float *recv_b2, int32_t n, const int32_t *recv_counts, int32_t n_segments,
                    void *scratch, size_t scratch_bytes, void *stream) {
  if (!p || !opt || check_shard(sh) || n < 0 || b_norm < 1 || !scratch || !recv_counts) return DRX_EINVAL;
  if (n_segments < sh->world || n_segments % sh->world || n_segments > sh->world * DRX_MAX_MICRO) return DRX_EINVAL;
  if (n == 0) return DRX_OK;
  if (!recv_keys || !recv_rows || !recv_b2) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SegOff so{};
  so.n_seg = n_segments;
  for (int s = 0; s < n_segments; ++s) {
    if (recv_counts[s] < 0) return DRX_EINVAL;
    so.off[s + 1] = so.off[s] + recv_counts[s];
  }
  if (so.off[n_segments] != n) return DRX_EINVAL;
  Carver cv(scratch, scratch_bytes);
  uint32_t *tab = owner_table(cv, *sh);
  if (!cv.ok()) return DRX_ESCRATCH;
  DRX_HIP(hipMemsetAsync(tab, 0xFF, (size_t)2 * sh->items_per_rank * n_segments * 4, st));
  hipLaunchKernelGGL(k_owner_scatter, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048), dim3(256), 0, st, *sh, so, recv_keys, n, tab);
  OwnerPolicy pol{*p, *opt, *sh, b_norm, recv_rows, recv_b2};
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    int blocks = (n + gpb - 1) / gpb;                                                                                  \
    if (blocks > 8192) blocks = 8192;                                                                                  \
    hipLaunchKernelGGL((k_owner_apply<G, J>), dim3(blocks), dim3(kBlock), 0, st, pol, so, recv_keys, n, tab);          \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_bias_grad(const DrxCdaeParams *p, const float *dz1, const float *lossb, int32_t B, float *out, void *scratch,
                        size_t scratch_bytes, void *stream) {
  if (!p || !dz1 || !lossb || !out || !scratch || B < 1) return DRX_EINVAL;
  if (scratch_bytes < (size_t)256 * (p->ld + 1) * 4) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  float *part = (float *)scratch;
  float *loss_part = part + (size_t)256 * p->ld;
  const int rows_per_block = (B + 255) / 256;
  const int n_part = (B + rows_per_block - 1) / rows_per_block;
#define CALL(G, J)                                                                                                  \
  {                                                                                                                 \
    const int gpb = kBlock / G;                                                                                     \
    hipLaunchKernelGGL((k_colsum_partial<G, J>), dim3(n_part), dim3(kBlock), (size_t)gpb * p->ld * 4, st, p->ld, B, dz1, \
                       part, rows_per_block, lossb, loss_part);                                                     \
    hipLaunchKernelGGL((k_colsum_final<G, J>), dim3(1), dim3(kBlock), (size_t)gpb * p->ld * 4, st, p->ld, part, n_part,  \
                       loss_part, B, out);                                                                          \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_shard_bias_apply(const DrxCdaeParams *p, const DrxOptim *opt, int32_t b_norm, const float *grad, void *stream) {
  if (!p || !opt || !grad || b_norm < 1) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J) { hipLaunchKernelGGL((k_bias_apply<G, J>), dim3(1), dim3(kBlock), 0, st, *p, *opt, b_norm, grad); }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // extern "C"
