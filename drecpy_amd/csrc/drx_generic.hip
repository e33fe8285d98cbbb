// Model-independent pieces of the reference-mode ("dense Keras Adam") steps of DMF and Caser:
//   drx_adam_dense   : fused L2 + Keras-Adam sweep over a flat parameter array (replaces one
//                      optimizer.apply_gradients of recommender_abc.py:328-334): p, m, v read once / written once
//   drx_scatter_rows : deterministic scatter-add of per-touch gradient rows into a dense gradient table
//                      (what tape.gradient does for tf.nn.embedding_lookup / Keras Embedding inputs): stable sort of
//                      the touches by destination row + the segmented reduction of drx_segreduce.hpp
//   drx_rows_dot     : scores[b, n] = x_b . T[n] (+ bias[n]) for all rows n (all-item scoring of _rank)
#include "drx_common.hpp"
#include "drx_rows.hpp"
#include "drx_segreduce.hpp"

namespace drx {

__global__ __launch_bounds__(kBlock) void k_adam_dense(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                                       const float *__restrict__ g, size_t n, float alpha, float l2c, float b1,
                                                       float b2, float eps) {
  const size_t n4 = n / 4;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
    float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
    float4 gg = g ? reinterpret_cast<const float4 *>(g)[i] : f4_zero();
    OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, alpha};
    opt_update1(o, fmaf(l2c, pp.x, gg.x), pp.x, mm.x, vv.x);
    opt_update1(o, fmaf(l2c, pp.y, gg.y), pp.y, mm.y, vv.y);
    opt_update1(o, fmaf(l2c, pp.z, gg.z), pp.z, mm.z, vv.z);
    opt_update1(o, fmaf(l2c, pp.w, gg.w), pp.w, mm.w, vv.w);
    reinterpret_cast<float4 *>(p)[i] = pp; reinterpret_cast<float4 *>(m)[i] = mm; reinterpret_cast<float4 *>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float pp = p[i], mm = m[i], vv = v[i];
    OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, alpha};
    opt_update1(o, fmaf(l2c, pp, g ? g[i] : 0.f), pp, mm, vv);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// A table whose gradient is a sum of lookup rows AND whose optimizer is dense (Keras Adam on an Embedding / Dense kernel: the
// moments of every row decay every step): gradient and update in one pass over the table, from the lookups grouped by row on the
// host (drx_batch_csr: counting sort, lookups of a row in batch order).  One group of G lanes per row:
//   g = sum over the row's lookups q (ascending) of src[order[q]];  p, m, v <- Adam(g + l2c * p)     (+ the same for one scalar per row)
// No touch keys on the device, no sort, no zeroed gradient table, no separate optimizer launch.
// OUTER: the gradient row of lookup o is src_s[o] * src[o / group] — an outer product the producer does not write out (Caser's dense_1
// rows: the score's gradient times the sample's hidden state, drx.h drx_rows_csr_adam_outer); src_s doubles as the per-lookup scalar.
// blk of n_blk: this workgroup's share of the table's rows (a launch of its own, or a table's part of drx_rows_csr_adam_multi's).
constexpr int kCsrSplit = 64;     // lookups from which a row's sum is split over the groups of its workgroup

template <int G, int J, bool OUTER>
__device__ __forceinline__ void rows_csr_adam_body(const int32_t *__restrict__ ptr, const int32_t *__restrict__ order,
                                                   const float *__restrict__ src, const float *__restrict__ src_s, int ld, int n_rows,
                                                   float *p, float *m, float *v, float *ps, float *ms, float *vs, float alpha,
                                                   float alpha_s, float l2c, float b1, float b2, float eps, int group, int blk, int n_blk,
                                                   float *part /* LDS [kBlock / G][4 * G * J] */, float *part_s /* LDS [kBlock / G] */,
                                                   int *cnt /* LDS [kBlock / G] */) {
  const int lane = threadIdx.x % G, gid = threadIdx.x / G;
  constexpr int gpb = kBlock / G;
  // g, gs += the gradient rows / scalars of the lookups [qa, qb) of the list, in list order, four rows in flight
  auto sum_range = [&](int qa, int qb, float4 (&g)[J], float &gs) __attribute__((always_inline)) {
    int q = qa;
    for (; q + 4 <= qb; q += 4) {
      const int o0 = order[q], o1 = order[q + 1], o2 = order[q + 2], o3 = order[q + 3];
      float4 r0[J], r1[J], r2[J], r3[J];
      load_row<G, J>(src, (size_t)(OUTER ? o0 / group : o0), ld, lane, r0);
      load_row<G, J>(src, (size_t)(OUTER ? o1 / group : o1), ld, lane, r1);
      load_row<G, J>(src, (size_t)(OUTER ? o2 / group : o2), ld, lane, r2);
      load_row<G, J>(src, (size_t)(OUTER ? o3 / group : o3), ld, lane, r3);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (OUTER || src_s) { s0 = src_s[o0]; s1 = src_s[o1]; s2 = src_s[o2]; s3 = src_s[o3]; }
      gs = ((gs + s0) + s1) + s2 + s3;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (OUTER) { f4_fma(g[j], s0, r0[j]); f4_fma(g[j], s1, r1[j]); f4_fma(g[j], s2, r2[j]); f4_fma(g[j], s3, r3[j]); }
        else { f4_add(g[j], r0[j]); f4_add(g[j], r1[j]); f4_add(g[j], r2[j]); f4_add(g[j], r3[j]); }
      }
    }
    for (; q < qb; ++q) {
      const int o0 = order[q];
      float4 r0[J];
      load_row<G, J>(src, (size_t)(OUTER ? o0 / group : o0), ld, lane, r0);
      const float s0 = (OUTER || src_s) ? src_s[o0] : 0.f;
      gs += s0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (OUTER) f4_fma(g[j], s0, r0[j]);
        else f4_add(g[j], r0[j]);
      }
    }
  };
  // l2 + Keras Adam of table row `row` (and of its scalar) with the gradient g / gs
  auto apply = [&](int row, const float4 (&w)[J], float4 (&mm)[J], float4 (&vv)[J], const float4 (&g)[J], float gs) __attribute__((always_inline)) {
    const OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, alpha};
    float4 *pr = reinterpret_cast<float4 *>(p + (size_t)row * ld), *mr = reinterpret_cast<float4 *>(m + (size_t)row * ld),
           *vr = reinterpret_cast<float4 *>(v + (size_t)row * ld);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = lane + j * G;
      if (4 * c < ld) {
        float4 ww = w[j];
        opt_update1(o, fmaf(l2c, ww.x, g[j].x), ww.x, mm[j].x, vv[j].x);
        opt_update1(o, fmaf(l2c, ww.y, g[j].y), ww.y, mm[j].y, vv[j].y);
        opt_update1(o, fmaf(l2c, ww.z, g[j].z), ww.z, mm[j].z, vv[j].z);
        opt_update1(o, fmaf(l2c, ww.w, g[j].w), ww.w, mm[j].w, vv[j].w);
        pr[c] = ww; mr[c] = mm[j]; vr[c] = vv[j];
      }
    }
    if (ps && lane == 0) {
      const OptScalars os{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, alpha_s};
      float pp = ps[row], pm = ms[row], pv = vs[row];
      opt_update1(os, gs, pp, pm, pv);
      ps[row] = pp; ms[row] = pm; vs[row] = pv;
    }
  };
  // A workgroup takes gpb rows at a time, a group each.  A row named by kCsrSplit lookups or more (a popular item: thousands at a
  // batch of 4096 windows) would keep ONE group busy long after the others are done: its list is cut into gpb slices, every group sums
  // one, the partial sums are added in slice order by group 0 — a fixed order, like the single walk.
  for (int row0 = blk * gpb; row0 < n_rows; row0 += n_blk * gpb) {
    const int row = row0 + gid;
    const bool valid = row < n_rows;
    const int q0 = valid ? ptr[row] : 0, q1 = valid ? ptr[row + 1] : 0;
    if (lane == 0) cnt[gid] = q1 - q0;
    __syncthreads();
    if (valid && q1 - q0 < kCsrSplit) {
      float4 w[J], mm[J], vv[J], g[J];
      load_row<G, J>(p, (size_t)row, ld, lane, w);
      load_row<G, J>(m, (size_t)row, ld, lane, mm);
      load_row<G, J>(v, (size_t)row, ld, lane, vv);
      float gs = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) g[j] = f4_zero();
      sum_range(q0, q1, g, gs);
      apply(row, w, mm, vv, g, gs);
    }
    for (int k = 0; k < gpb; ++k) {                    // (uniform over the workgroup: cnt is shared)
      const int n = cnt[k];
      if (n < kCsrSplit) continue;
      const int hrow = row0 + k, h0 = ptr[hrow];
      const int per = (((n + gpb - 1) / gpb) + 3) & ~3;
      float4 g[J];
      float gs = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) g[j] = f4_zero();
      sum_range(min(h0 + gid * per, h0 + n), min(h0 + (gid + 1) * per, h0 + n), g, gs);
#pragma unroll
      for (int j = 0; j < J; ++j) *reinterpret_cast<float4 *>(part + (size_t)gid * (4 * G * J) + 4 * (lane + j * G)) = g[j];
      if (lane == 0) part_s[gid] = gs;
      __syncthreads();
      if (gid == 0) {
        float4 w[J], mm[J], vv[J];
        load_row<G, J>(p, (size_t)hrow, ld, lane, w);
        load_row<G, J>(m, (size_t)hrow, ld, lane, mm);
        load_row<G, J>(v, (size_t)hrow, ld, lane, vv);
        gs = part_s[0];
        for (int kk = 1; kk < gpb; ++kk) {
          gs += part_s[kk];
#pragma unroll
          for (int j = 0; j < J; ++j) f4_add(g[j], *reinterpret_cast<const float4 *>(part + (size_t)kk * (4 * G * J) + 4 * (lane + j * G)));
        }
        apply(hrow, w, mm, vv, g, gs);
      }
      __syncthreads();
    }
    __syncthreads();                                    // (cnt is rewritten by the next round)
  }
}

template <int G, int J, bool OUTER>
__global__ __launch_bounds__(kBlock) void k_rows_csr_adam(const int32_t *__restrict__ ptr, const int32_t *__restrict__ order,
                                                          const float *__restrict__ src, const float *__restrict__ src_s, int ld,
                                                          int n_rows, float *p, float *m, float *v, float *ps, float *ms, float *vs,
                                                          float alpha, float alpha_s, float l2c, float b1, float b2, float eps, int group) {
  __shared__ __align__(16) float part[(kBlock / G) * 4 * G * J];
  __shared__ float part_s[kBlock / G];
  __shared__ int cnt[kBlock / G];
  rows_csr_adam_body<G, J, OUTER>(ptr, order, src, src_s, ld, n_rows, p, m, v, ps, ms, vs, alpha, alpha_s, l2c, b1, b2, eps, group,
                                  (int)blockIdx.x, (int)gridDim.x, part, part_s, cnt);
}

// Several tables in one launch (Caser's three lookup tables: three launches of a few microseconds of work each spent most of their
// time starting and draining): consecutive ranges of workgroups take a table each.
struct CsrMulti { DrxCsrAdamTable t[DRX_MAX_CSR_TABLES]; int blocks[DRX_MAX_CSR_TABLES]; int n; };
__global__ __launch_bounds__(kBlock) void k_rows_csr_adam_multi(CsrMulti M, float b1, float b2, float eps) {
  int blk = blockIdx.x, ti = 0;
  while (ti + 1 < M.n && blk >= M.blocks[ti]) { blk -= M.blocks[ti]; ++ti; }
  const DrxCsrAdamTable &T = M.t[ti];
  const int nb = M.blocks[ti];
  __shared__ __align__(16) float part[kBlock * 4];           // (J = 1: (kBlock / G) * 4 G floats whatever G)
  __shared__ float part_s[kBlock / 4];
  __shared__ int cnt[kBlock / 4];
#define BODY(G, OUTER)                                                                                                             \
  rows_csr_adam_body<G, 1, OUTER>(T.row_ptr, T.order, T.src, T.scale, T.ld, T.n_rows, T.p, T.m, T.v, T.p_s, T.m_s, T.v_s, T.alpha,    \
                                  T.alpha_s, T.l2_coef, b1, b2, eps, T.group, blk, nb, part, part_s, cnt)
#define GEOM(OUTER)                                         \
  do {                                                      \
    if (T.ld <= 16) BODY(4, OUTER);                         \
    else if (T.ld <= 32) BODY(8, OUTER);                    \
    else if (T.ld <= 64) BODY(16, OUTER);                   \
    else if (T.ld <= 128) BODY(32, OUTER);                  \
    else BODY(64, OUTER);                                   \
  } while (0)
  if (T.group > 0) GEOM(true);
  else GEOM(false);
#undef GEOM
#undef BODY
}

struct ScatterPolicy {
  const float *src;          // [n_src, ld]
  const uint32_t *src_index; // [T] row of src contributed by touch `pos` (nullptr: pos itself)
  const float *coef;         // [T] multiplier (nullptr: 1)
  const float *src_s;        // [n_src] scalar side channel (nullptr: none)
  float *out;                // [n_rows, ld]
  float *out_s;              // [n_rows] or nullptr
  int ld;
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t, uint32_t pos, int lane, float4 (&row)[J], float &sc, float &c) const {
    const uint32_t r = src_index ? src_index[pos] : pos;
    load_row<G, J>(src, (size_t)r, ld, lane, row);
    c = coef ? coef[pos] : 1.0f;
    if (src_s) sc = src_s[r] * c;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int, int lane, const float4 (&g)[J], float gs) const {
    store_row<G, J>(out, (size_t)key, ld, lane, g);
    if (out_s && lane == 0) out_s[key] = gs;
  }
};

__global__ void k_iota2(uint32_t *v, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_rows_dot(const float *__restrict__ x, int B, const float *__restrict__ tab, int n_rows,
                                                     int ld, const float *__restrict__ bias, float *__restrict__ out) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  for (int n = blockIdx.x * gpb + threadIdx.x / G; n < n_rows; n += gridDim.x * gpb) {
    float4 w[J];
    load_row<G, J>(tab, (size_t)n, ld, lane, w);
    const float bb = bias ? bias[n] : 0.f;
    for (int b = 0; b < B; ++b) {
      float4 xv[J];
      load_row<G, J>(x, (size_t)b, ld, lane, xv);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) d += f4_dot(w[j], xv[j]);
      d = group_sum<G>(d);
      if (lane == 0) out[(size_t)b * n_rows + n] = d + bb;
    }
  }
}

// ---- small problems: no sort ------------------------------------------------------------------------------------------------
// A bit per (destination row, touch) — set with integer atomics, so deterministic — and one lane group per destination row that
// walks its bits in ascending touch order (the order the stable sort would give).  The mask costs n_rows * T / 8 bytes to clear
// and to read, so this is for small problems only: the three scatters of a Caser step of 512 windows (2.5 k / 6 k / 0.5 k touches
// into 3.7 k - 6 k rows: masks of 0.4 - 2.8 MB) each cost a rocPRIM merge sort, the reduction and the span fix-ups — 8 - 10
// launches, ~60 us; here it is 3 (step 0.35 -> 0.28 ms).  With masks of 18 + 48 MB (a DMF step of 256 pairs) it LOSES
// (0.29 -> 0.42 ms): hence the budget.
constexpr size_t kMaskBudget = (size_t)8 << 20;
inline bool scatter_by_mask(int T, int n_rows) { return (size_t)n_rows * (size_t)((T + 31) / 32) * 4 <= kMaskBudget; }

__global__ void k_mask_mark(const uint32_t *__restrict__ keys, int T, int W, int n_rows, uint32_t *__restrict__ mask) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
    const uint32_t key = keys[t];
    if (key != DRX_KEY_NONE && key < (uint32_t)n_rows) atomicOr(&mask[(size_t)key * W + (t >> 5)], 1u << (t & 31));
  }
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_mask_sweep(const uint32_t *__restrict__ mask, int W, int n_rows, ScatterPolicy pol) {
  const int lane = threadIdx.x % G;
  const int gshift = (G == 64) ? 0 : (((int)threadIdx.x % 64) / G) * G;
  const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
  const int gpb = kBlock / G;
  for (int row = blockIdx.x * gpb + threadIdx.x / G; row < n_rows; row += gridDim.x * gpb) {
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    float accs = 0.f;
    bool any = false;
    const uint32_t *mr = mask + (size_t)row * W;
    for (int w0 = 0; w0 < W; w0 += G) {
      const uint32_t word = (w0 + lane < W) ? mr[w0 + lane] : 0u;
      unsigned long long nz = (__ballot(word != 0u) >> gshift) & gmask;       // lanes of this group holding set bits
      while (nz) {
        const int l = __ffsll((long long)nz) - 1;
        nz &= nz - 1;
        uint32_t wv = (uint32_t)__shfl((int)word, l, G);
        any = true;
        while (wv) {
          const uint32_t t = (uint32_t)(w0 + l) * 32u + (uint32_t)(__ffs((int)wv) - 1);
          wv &= wv - 1;
          float4 r[J];
          float sc = 0.f, c = 1.f;
          pol.template load<G, J>(0u, t, lane, r, sc, c);
#pragma unroll
          for (int j = 0; j < J; ++j) f4_fma(acc[j], c, r[j]);
          accs += sc;
        }
      }
    }
    if (any) pol.template finish<G, J>((uint32_t)row, 0, lane, acc, accs);
  }
}

struct ScatterLayout {
  SegBufs sb;
  uint32_t *idx, *keys_s, *vals_s;
  void *sort_temp;
  size_t sort_bytes;
};

static ScatterLayout scatter_layout(Carver &cv, int ld, int T, int bits) {
  ScatterLayout L{};
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
  L.sort_bytes = sort_pairs_temp_bytes((size_t)T, bits);
  L.sort_temp = cv.take<char>(L.sort_bytes);
  return L;
}

}  // namespace drx

using namespace drx;

// Sum of squares of a float array in double, two deterministic stages (the value of an L2 term for the loss log — cdae.py:82,
// Keras `l2` regularizers: never on the training path, which applies reg * p inside the update kernels).
constexpr int kSumsqBlocks = 1024;
__global__ __launch_bounds__(kBlock) void k_sumsq_partial(const float *__restrict__ x, size_t n, double *__restrict__ part) {
  __shared__ double red[kBlock / 64];
  double a = 0.0;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a += (double)x[i] * (double)x[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) t += red[i];
    part[blockIdx.x] = t;
  }
}
__global__ void k_sumsq_final(const double *__restrict__ part, int n_part, double *__restrict__ out, int accumulate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = accumulate ? out[0] : 0.0;
    for (int i = 0; i < n_part; ++i) t += part[i];
    out[0] = t;
  }
}

// Streaming copy, 16 bytes per lane: what the library uses to snapshot tables and what bench.py times as "the HBM rate a plain kernel of
// this library reaches on this box" (report the achievable next to the nominal 8 TB/s).  Forms (drx_copy_f4_variant;
// scripts/copy_bench.py times them side by side): 0 grid-stride, four loads in flight per thread, 16 workgroups per CU; 1 one float4 per
// thread, as many workgroups as that takes; 2 grid-stride, eight in flight, 8 workgroups per CU; 3 = 1 with non-temporal loads / stores;
// 4 = 0 with non-temporal loads / stores.
typedef float copy_v4 __attribute__((ext_vector_type(4)));       // (the non-temporal builtins take native vectors, not HIP's float4 struct)
template <int NF, bool NT>
__global__ __launch_bounds__(kBlock) void k_copy_f4(copy_v4 *__restrict__ dst, const copy_v4 *__restrict__ src, size_t n4) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x;
  for (; i + (NF - 1) * stride < n4; i += NF * stride) {
    copy_v4 v[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) v[j] = NT ? __builtin_nontemporal_load(src + i + j * stride) : src[i + j * stride];
#pragma unroll
    for (int j = 0; j < NF; ++j) { if (NT) __builtin_nontemporal_store(v[j], dst + i + j * stride); else dst[i + j * stride] = v[j]; }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

extern "C" {

int drx_copy_f4_variant(void *dst, const void *src, size_t n_bytes, int32_t variant, void *stream) {
  if (!dst || !src || n_bytes < 16 || (n_bytes & 15) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return DRX_EINVAL;
  const size_t n4 = n_bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  copy_v4 *d = (copy_v4 *)dst;
  const copy_v4 *sp = (const copy_v4 *)src;
  const size_t one = (n4 + kBlock - 1) / kBlock;
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_copy_f4<4, false>), dim3((unsigned)std::min<size_t>((one + 3) / 4, 256 * 16)), dim3(kBlock), 0, st, d, sp, n4); break;
    case 1: hipLaunchKernelGGL((k_copy_f4<1, false>), dim3((unsigned)std::min<size_t>(one, 0x7FFFFFFFu)), dim3(kBlock), 0, st, d, sp, n4); break;
    case 2: hipLaunchKernelGGL((k_copy_f4<8, false>), dim3((unsigned)std::min<size_t>((one + 7) / 8, 256 * 8)), dim3(kBlock), 0, st, d, sp, n4); break;
    case 3: hipLaunchKernelGGL((k_copy_f4<1, true>), dim3((unsigned)std::min<size_t>(one, 0x7FFFFFFFu)), dim3(kBlock), 0, st, d, sp, n4); break;
    case 4: hipLaunchKernelGGL((k_copy_f4<4, true>), dim3((unsigned)std::min<size_t>((one + 3) / 4, 256 * 16)), dim3(kBlock), 0, st, d, sp, n4); break;
    default: return DRX_EINVAL;
  }
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

#ifndef DRX_COPY_VARIANT
#define DRX_COPY_VARIANT 3          // (what scripts/copy_bench.py measured fastest on an MI355X: profiles/r06_copy_variants.log)
#endif
int drx_copy_f4(void *dst, const void *src, size_t n_bytes, void *stream) { return drx_copy_f4_variant(dst, src, n_bytes, DRX_COPY_VARIANT, stream); }

int drx_adam_dense(float *p, float *m, float *v, const float *g, int64_t n, float alpha, float l2_coef, float beta1, float beta2,
                   float eps, void *stream) {
  if (!p || !m || !v || n < 1) return DRX_EINVAL;
  if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) return DRX_EINVAL;
  int blocks = (int)std::min<int64_t>((n / 4 + kBlock - 1) / kBlock + 1, 4096);
  hipLaunchKernelGGL(k_adam_dense, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, p, m, v, g, (size_t)n, alpha, l2_coef,
                     beta1, beta2, eps);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

// the two entry points below: lookups' rows as they are / as scale[o] * src[o / group]
static int rows_csr_adam(const int32_t *row_ptr, const int32_t *order, const float *src, const float *src_s, int32_t group, int32_t ld,
                         int32_t n_rows, float *p, float *m, float *v, float *p_s, float *m_s, float *v_s, float alpha, float alpha_s,
                         float l2_coef, float beta1, float beta2, float eps, void *stream) {
  if (!row_ptr || !order || !src || !p || !m || !v || n_rows < 1 || ld < 4 || (ld & 3) || ld > DRX_MAX_K) return DRX_EINVAL;
  if ((p_s != nullptr) != (src_s != nullptr) || (p_s && (!m_s || !v_s))) return DRX_EINVAL;
  if (((uintptr_t)src | (uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J)                                                                                                          \
  {                                                                                                                         \
    const int gpb = kBlock / G;                                                                                             \
    int blocks = (n_rows + gpb - 1) / gpb;                                                                                  \
    if (blocks > 8192) blocks = 8192;                                                                                       \
    if (group > 0)                                                                                                          \
      hipLaunchKernelGGL((k_rows_csr_adam<G, J, true>), dim3(blocks), dim3(kBlock), 0, st, row_ptr, order, src, src_s, ld, n_rows, p, m, \
                         v, p_s, m_s, v_s, alpha, alpha_s, l2_coef, beta1, beta2, eps, group);                              \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_rows_csr_adam<G, J, false>), dim3(blocks), dim3(kBlock), 0, st, row_ptr, order, src, src_s, ld, n_rows, p, m, \
                         v, p_s, m_s, v_s, alpha, alpha_s, l2_coef, beta1, beta2, eps, 1);                                  \
  }
  DRX_DISPATCH_GEOM(ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_rows_csr_adam(const int32_t *row_ptr, const int32_t *order, const float *src, const float *src_s, int32_t ld, int32_t n_rows,
                      float *p, float *m, float *v, float *p_s, float *m_s, float *v_s, float alpha, float alpha_s, float l2_coef,
                      float beta1, float beta2, float eps, void *stream) {
  return rows_csr_adam(row_ptr, order, src, src_s, 0, ld, n_rows, p, m, v, p_s, m_s, v_s, alpha, alpha_s, l2_coef, beta1, beta2, eps, stream);
}

int drx_rows_csr_adam_outer(const int32_t *row_ptr, const int32_t *order, const float *scale, const float *src, int32_t group, int32_t ld,
                            int32_t n_rows, float *p, float *m, float *v, float *p_s, float *m_s, float *v_s, float alpha, float alpha_s,
                            float l2_coef, float beta1, float beta2, float eps, void *stream) {
  if (!scale || group < 1 || !p_s) return DRX_EINVAL;
  return rows_csr_adam(row_ptr, order, src, scale, group, ld, n_rows, p, m, v, p_s, m_s, v_s, alpha, alpha_s, l2_coef, beta1, beta2, eps, stream);
}

// ---- drx_batch_csr on the device: the lookups of several key lists grouped by the table row they name -----------------------------
// One stable sort of all lists' (row, lookup) pairs — list l's rows offset by the rows of the lists before it, so the lists come out
// one after the other and a row's lookups ascending — then row_ptr by one binary search per row.
struct CsrLists {
  int n;
  const int32_t *keys[DRX_MAX_CSR_TABLES];
  int32_t *row_ptr[DRX_MAX_CSR_TABLES], *order[DRX_MAX_CSR_TABLES];
  uint32_t t_base[DRX_MAX_CSR_TABLES + 1], r_base[DRX_MAX_CSR_TABLES + 1];   // first lookup / first row of list l in the concatenation
};

static __global__ void k_csr_keys(CsrLists Ls, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t T = Ls.t_base[Ls.n];
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < Ls.n && j >= Ls.t_base[l + 1]) ++l;
    const uint32_t q = j - Ls.t_base[l];
    keys[j] = Ls.r_base[l] + (uint32_t)Ls.keys[l][q];
    vals[j] = q;
  }
}

static __global__ void k_csr_finish(CsrLists Ls, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals) {
  const uint32_t T = Ls.t_base[Ls.n], R = Ls.r_base[Ls.n] + (uint32_t)Ls.n;       // (every list: n_rows + 1 row_ptr entries)
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < T + R; j += gridDim.x * blockDim.x) {
    if (j < T) {
      int l = 0;
      while (l + 1 < Ls.n && j >= Ls.t_base[l + 1]) ++l;
      Ls.order[l][j - Ls.t_base[l]] = (int32_t)vals[j];
    } else {
      uint32_t e = j - T;
      int l = 0;
      while (l + 1 < Ls.n && e >= Ls.r_base[l + 1] + (uint32_t)(l + 1)) ++l;
      const uint32_t r = e - (Ls.r_base[l] + (uint32_t)l);                          // 0 .. n_rows of list l
      const uint32_t want = Ls.r_base[l] + r;
      uint32_t lo = Ls.t_base[l], hi = Ls.t_base[l + 1];                            // first pair of the list whose key is >= want
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
      }
      Ls.row_ptr[l][r] = (int32_t)(lo - Ls.t_base[l]);
    }
  }
}

static int csr_lists(const DrxCsrList *lists, int32_t n_lists, CsrLists &Ls, int &bits) {
  if (!lists || n_lists < 1 || n_lists > DRX_MAX_CSR_TABLES) return DRX_EINVAL;
  Ls.n = n_lists;
  uint64_t t = 0, r = 0;
  for (int l = 0; l < n_lists; ++l) {
    if (!lists[l].keys || !lists[l].row_ptr || !lists[l].order || lists[l].T < 1 || lists[l].n_rows < 1) return DRX_EINVAL;
    Ls.keys[l] = lists[l].keys; Ls.row_ptr[l] = lists[l].row_ptr; Ls.order[l] = lists[l].order;
    Ls.t_base[l] = (uint32_t)t; Ls.r_base[l] = (uint32_t)r;
    t += (uint64_t)lists[l].T; r += (uint64_t)lists[l].n_rows;
    if (t >= 0x7FFFFFFFull || r >= 0x7FFFFFFFull) return DRX_EINVAL;
  }
  Ls.t_base[n_lists] = (uint32_t)t; Ls.r_base[n_lists] = (uint32_t)r;
  bits = 1;
  while ((1ull << bits) < r) ++bits;
  return DRX_OK;
}

size_t drx_batch_csr_device_bytes(const DrxCsrList *lists, int32_t n_lists) {
  CsrLists Ls;
  int bits;
  if (csr_lists(lists, n_lists, Ls, bits)) return 0;
  const size_t T = Ls.t_base[n_lists];
  return 4 * ((T * 4 + 255) & ~size_t(255)) + sort_pairs_temp_bytes(T, bits) + 512;
}

int drx_batch_csr_device(const DrxCsrList *lists, int32_t n_lists, void *scratch, size_t scratch_bytes, void *stream) {
  CsrLists Ls;
  int bits;
  int rc = csr_lists(lists, n_lists, Ls, bits);
  if (rc) return rc;
  if (!scratch || scratch_bytes < drx_batch_csr_device_bytes(lists, n_lists)) return DRX_EINVAL;
  const size_t T = Ls.t_base[n_lists], arr = (T * 4 + 255) & ~size_t(255);
  char *base = (char *)(((uintptr_t)scratch + 255) & ~uintptr_t(255));
  uint32_t *k_in = (uint32_t *)base, *v_in = (uint32_t *)(base + arr), *k_out = (uint32_t *)(base + 2 * arr), *v_out = (uint32_t *)(base + 3 * arr);
  char *temp = base + 4 * arr;
  const size_t temp_bytes = sort_pairs_temp_bytes(T, bits);
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((T + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_csr_keys, dim3(blocks > 2048 ? 2048 : blocks), dim3(kBlock), 0, st, Ls, k_in, v_in);
  rc = sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, T, bits, st);
  if (rc) return rc;
  const size_t work = T + Ls.r_base[n_lists] + (size_t)n_lists;
  blocks = (int)((work + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_csr_finish, dim3(blocks > 4096 ? 4096 : blocks), dim3(kBlock), 0, st, Ls, k_out, v_out);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_rows_csr_adam_multi(const DrxCsrAdamTable *tables, int32_t n_tables, float beta1, float beta2, float eps, void *stream) {
  if (!tables || n_tables < 1 || n_tables > DRX_MAX_CSR_TABLES) return DRX_EINVAL;
  CsrMulti M;
  M.n = n_tables;
  int total = 0;
  for (int i = 0; i < n_tables; ++i) {
    const DrxCsrAdamTable &T = tables[i];
    if (!T.row_ptr || !T.order || !T.src || !T.p || !T.m || !T.v || T.n_rows < 1 || T.ld < 4 || (T.ld & 3) || T.ld > 256 || T.group < 0)
      return DRX_EINVAL;
    if ((T.p_s != nullptr) != (T.scale != nullptr) || (T.p_s && (!T.m_s || !T.v_s)) || (T.group > 0 && !T.scale)) return DRX_EINVAL;
    if (((uintptr_t)T.src | (uintptr_t)T.p | (uintptr_t)T.m | (uintptr_t)T.v) & 15) return DRX_EINVAL;
    const int gpb = kBlock / pick_geom(T.ld).G;
    int blocks = (T.n_rows + gpb - 1) / gpb;
    if (blocks > 4096) blocks = 4096;
    M.t[i] = T;
    M.blocks[i] = blocks;
    total += blocks;
  }
  hipLaunchKernelGGL(k_rows_csr_adam_multi, dim3(total), dim3(kBlock), 0, (hipStream_t)stream, M, beta1, beta2, eps);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_scatter_scratch_bytes(int32_t ld, int32_t n_touches, int32_t n_rows) {
  if (ld < 4 || (ld & 3) || n_touches < 1 || n_rows < 1) return 0;
  Carver cv(nullptr, 0);
  (void)scatter_layout(cv, ld, n_touches, bits_for((uint64_t)n_rows + 1));
  size_t need = align_up(cv.off, 256) + 256;
  if (scatter_by_mask(n_touches, n_rows)) need = std::max(need, (size_t)n_rows * (size_t)((n_touches + 31) / 32) * 4 + 512);
  return need;
}

int drx_scatter_rows(const uint32_t *keys, int32_t T, const float *src, const uint32_t *src_index, const float *coef,
                     const float *src_s, int32_t ld, int32_t n_rows, float *out, float *out_s, void *scratch,
                     size_t scratch_bytes, void *stream) {
  if (!keys || !src || !out || !scratch || T < 1 || n_rows < 1 || ld < 4 || (ld & 3) || ld > DRX_MAX_K) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (scatter_by_mask(T, n_rows)) {
    const int W = (T + 31) / 32;
    const size_t mbytes = (size_t)n_rows * W * 4;
    uint32_t *mask = (uint32_t *)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    if ((size_t)((char *)mask - (char *)scratch) + mbytes > scratch_bytes) return DRX_ESCRATCH;
    DRX_HIP(hipMemsetAsync(mask, 0, mbytes, st));
    hipLaunchKernelGGL(k_mask_mark, dim3((T + 255) / 256 < 2048 ? (T + 255) / 256 : 2048), dim3(256), 0, st, keys, T, W, n_rows, mask);
    ScatterPolicy mpol{src, src_index, coef, src_s, out, out_s, ld};
#define CALLM(G, J)                                                                                                     \
  {                                                                                                                     \
    const int gpb = kBlock / G;                                                                                         \
    int blocks = (n_rows + gpb - 1) / gpb;                                                                              \
    if (blocks > 8192) blocks = 8192;                                                                                   \
    hipLaunchKernelGGL((k_mask_sweep<G, J>), dim3(blocks), dim3(kBlock), 0, st, mask, W, n_rows, mpol);                 \
  }
    DRX_DISPATCH_GEOM(ld, CALLM);
#undef CALLM
    DRX_LAUNCH_CHECK();
    return DRX_OK;
  }
  const int bits = bits_for((uint64_t)n_rows + 1);
  Carver cv(scratch, scratch_bytes);
  ScatterLayout L = scatter_layout(cv, ld, T, bits);
  if (!cv.ok()) return DRX_ESCRATCH;
  hipLaunchKernelGGL(k_iota2, dim3(1024), dim3(256), 0, st, L.idx, T);
  int rc = sort_pairs(L.sort_temp, L.sort_bytes, keys, L.keys_s, L.idx, L.vals_s, (size_t)T, bits, st);
  if (rc) return rc;
  L.sb.keys_s = L.keys_s; L.sb.vals_s = L.vals_s;
  ScatterPolicy pol{src, src_index, coef, src_s, out, out_s, ld};
  DRX_HIP(hipMemsetAsync(L.sb.n_span, 0, 2 * sizeof(uint32_t), st));
#define CALL(G, J)                                                                                                      \
  {                                                                                                                     \
    hipLaunchKernelGGL((k_seg_reduce<G, J, ScatterPolicy, 8>), dim3((L.sb.n_chunks + SEG_GPB(G) - 1) / SEG_GPB(G)),     \
                       dim3(kBlock), 0, st, L.sb, pol);                                                                 \
    hipLaunchKernelGGL((k_span_short<G, J, ScatterPolicy>), dim3(256), dim3(kBlock), 0, st, L.sb, pol);                 \
    const size_t lds = ((size_t)(kFixBlock / G) * (ld + 1)) * 4;                                                        \
    if (lds > 48 * 1024)                                                                                                \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_long<G, J, ScatterPolicy>,                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((k_span_long<G, J, ScatterPolicy>), dim3(64), dim3(kFixBlock), lds, st, L.sb, pol);              \
  }
  DRX_DISPATCH_GEOM(ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_rows_dot(const float *x, int32_t B, const float *table, int32_t n_rows, int32_t ld, const float *bias, float *out,
                 void *stream) {
  if (!x || !table || !out || B < 1 || n_rows < 1 || ld < 4 || (ld & 3) || ld > DRX_MAX_K) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, J)                                                                                       \
  {                                                                                                      \
    const int gpb = kBlock / G;                                                                          \
    int blocks = (n_rows + gpb - 1) / gpb;                                                               \
    if (blocks > 2048) blocks = 2048;                                                                    \
    hipLaunchKernelGGL((k_rows_dot<G, J>), dim3(blocks), dim3(kBlock), 0, st, x, B, table, n_rows, ld, bias, out); \
  }
  DRX_DISPATCH_GEOM(ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

/* out[0] (+)= sum x[i]^2 in double; out: device double [1 + 1024] (out[1..] is scratch for the block partials) */
int drx_sumsq(const float *x, int64_t n, double *out, int32_t accumulate, void *stream) {
  if (!x || !out || n < 0) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sumsq_partial, dim3(kSumsqBlocks), dim3(kBlock), 0, st, x, (size_t)n, out + 1);
  hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(64), 0, st, out + 1, kSumsqBlocks, out, accumulate);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // extern "C"
