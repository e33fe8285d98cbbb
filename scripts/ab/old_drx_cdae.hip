// CDAE hot path for MI355X (gfx950): embedding-bag hidden layer, all-unit / sampled output layer,
// Keras BCE/MSE, backward and fused dense-Adam / sparse-Adagrad updates.
//
// Replaces, per fit() iteration, the TensorFlow eager ops issued by DRecPy/Recommender/cdae.py:50-82 and
// DRecPy/Recommender/recommender_abc.py:190-204,328-334 (see include/drx.h for the per-entry mapping).
//
// Thread geometry: a table row (ld floats) is owned by a GROUP of G lanes (G = 8..64, power of two,
// sub-wave), each lane holding J float4 -> one coalesced 16 B/lane access per row; K=128 is a half-wave
// (two rows per wave64), K=50 (ld 52) a 16-lane group (four rows per wave).
#include "drx_common.hpp"

namespace drx {

// ------------------------------------------------------------------------------------------------
// auxiliary per-step index built by the hidden-layer kernel in reference ("dense") mode
// ------------------------------------------------------------------------------------------------
struct DenseAux {
  int32_t *cnt;    // [N]       #batch rows having item n as a positive  -> batch-mean target
  uint32_t *km;    // [N, Bw]   bit b set: item n is a surviving (kept) input of batch row b
  uint32_t *vm;    // [U, Bw]   bit b set: batch row b belongs to user u
  uint32_t *tb;    // [B, Nw]   per-row target bits (DRX_TARGETS_PER_ROW) or nullptr
  int32_t Bw, Nw;
};

struct RowRef {
  const float4 *p;
};

template <int G, int J>
__device__ __forceinline__ void load_row(const float *base, size_t row, int ld, int lane, float4 (&v)[J]) {
  const float4 *r = reinterpret_cast<const float4 *>(base + row * (size_t)ld);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int c = lane + j * G;
    v[j] = (4 * c < ld) ? r[c] : f4_zero();
  }
}

template <int G, int J>
__device__ __forceinline__ void store_row(float *base, size_t row, int ld, int lane, const float4 (&v)[J]) {
  float4 *r = reinterpret_cast<float4 *>(base + row * (size_t)ld);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int c = lane + j * G;
    if (4 * c < ld) r[c] = v[j];
  }
}

__device__ __forceinline__ float colmask(int col, int k, float v) { return col < k ? v : 0.0f; }

// Gathers scale * sum_{kept} W[n] for one batch row.  MODE 0: plain; 1: also builds DenseAux;
// 2: also emits the (key,val) touch list of the sampled mode.
template <int G, int J, int MODE>
__device__ __forceinline__ void gather_bag(const DrxCdaeParams &P, const DrxHistory &H, const DrxBatch &bt,
                                           uint32_t qthr, int b, int lane, float4 (&acc)[J],
                                           const DenseAux &aux, uint32_t *tkeys, uint32_t *tvals,
                                           int touch_base) {
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  for (int64_t c = s; c < e; c += G) {
    const int64_t j = c + lane;
    int idx = -1, kf = 0;
    if (j < e) {
      idx = H.indices[j];
      const uint32_t jj = (uint32_t)(j - s);
      kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
      if (MODE == 1) {
        atomicAdd(&aux.cnt[idx], 1);
        if (aux.tb) atomicOr(&aux.tb[(size_t)b * aux.Nw + (idx >> 5)], 1u << (idx & 31));
        if (kf) atomicOr(&aux.km[(size_t)idx * aux.Bw + (b >> 5)], 1u << (b & 31));
      }
      if (MODE == 2) {
        tkeys[touch_base + jj] = kf ? (uint32_t)idx : DRX_KEY_NONE;
        tvals[touch_base + jj] = (uint32_t)b;
      }
    }
    const int n_here = (int)((e - c) < (int64_t)G ? (e - c) : (int64_t)G);
    for (int t = 0; t < n_here; t += 4) {
      int i0 = __shfl(idx, t, G), i1 = __shfl(idx, t + 1, G), i2 = __shfl(idx, t + 2, G), i3 = __shfl(idx, t + 3, G);
      int k0 = __shfl(kf, t, G), k1 = __shfl(kf, t + 1, G), k2 = __shfl(kf, t + 2, G), k3 = __shfl(kf, t + 3, G);
      k1 = (t + 1 < n_here) ? k1 : 0;
      k2 = (t + 2 < n_here) ? k2 : 0;
      k3 = (t + 3 < n_here) ? k3 : 0;
      float4 r0[J], r1[J], r2[J], r3[J];
#pragma unroll
      for (int jx = 0; jx < J; ++jx) r0[jx] = r1[jx] = r2[jx] = r3[jx] = f4_zero();
      if (k0) load_row<G, J>(P.W, (size_t)i0, P.ld, lane, r0);
      if (k1) load_row<G, J>(P.W, (size_t)i1, P.ld, lane, r1);
      if (k2) load_row<G, J>(P.W, (size_t)i2, P.ld, lane, r2);
      if (k3) load_row<G, J>(P.W, (size_t)i3, P.ld, lane, r3);
#pragma unroll
      for (int jx = 0; jx < J; ++jx) {
        f4_add(acc[jx], r0[jx]); f4_add(acc[jx], r1[jx]); f4_add(acc[jx], r2[jx]); f4_add(acc[jx], r3[jx]);
      }
    }
  }
}

// h = sigmoid(scale*bag + V[u] + b), zero in the padding columns.
template <int G, int J>
__device__ __forceinline__ void hidden_act(const DrxCdaeParams &P, int u, float scale, int lane,
                                           const float4 (&acc)[J], float4 (&h)[J]) {
  float4 v[J], bb[J];
  load_row<G, J>(P.V, (size_t)u, P.ld, lane, v);
  load_row<G, J>(P.b, 0, P.ld, lane, bb);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int col = 4 * (lane + j * G);
    h[j].x = colmask(col + 0, P.k, sigmoidf_(fmaf(scale, acc[j].x, v[j].x + bb[j].x)));
    h[j].y = colmask(col + 1, P.k, sigmoidf_(fmaf(scale, acc[j].y, v[j].y + bb[j].y)));
    h[j].z = colmask(col + 2, P.k, sigmoidf_(fmaf(scale, acc[j].z, v[j].z + bb[j].z)));
    h[j].w = colmask(col + 3, P.k, sigmoidf_(fmaf(scale, acc[j].w, v[j].w + bb[j].w)));
  }
}

template <int G, int J, int MODE>
__global__ __launch_bounds__(kBlock) void k_hidden_fwd(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale,
                                                       uint32_t qthr, float *__restrict__ hout, DenseAux aux) {
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  float4 acc[J], h[J];
  gather_bag<G, J, MODE>(P, H, bt, qthr, b, lane, acc, aux, nullptr, nullptr, 0);
  const int u = bt.uid[b];
  if (MODE == 1 && lane == 0) atomicOr(&aux.vm[(size_t)u * aux.Bw + (b >> 5)], 1u << (b & 31));
  hidden_act<G, J>(P, u, scale, lane, acc, h);
  store_row<G, J>(hout, (size_t)b, P.ld, lane, h);
}

// pred[b,n] = sigmoid(h_b . W2T[n] + b2[n]) for all b, n  (inference; cdae.py:76)
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_out_fwd(DrxCdaeParams P, const float *__restrict__ h, int B,
                                                    float *__restrict__ pred) {
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  for (int n = blockIdx.x * gpb + threadIdx.x / G; n < P.n_items; n += gridDim.x * gpb) {
    float4 w[J];
    load_row<G, J>(P.W2T, (size_t)n, P.ld, lane, w);
    const float bias = P.b2[n];
    for (int b = 0; b < B; ++b) {
      float4 hv[J];
      load_row<G, J>(h, (size_t)b, P.ld, lane, hv);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) d += f4_dot(w[j], hv[j]);
      d = group_sum<G>(d);
      if (lane == 0) pred[(size_t)b * P.n_items + n] = sigmoidf_(d + bias);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// optimizer row updates (fp32, Keras formulas)
// ------------------------------------------------------------------------------------------------
struct OptScalars {
  int kind;
  float lr, rb;              // rb = reg_rate / B
  float b1, b2, eps, alpha;  // alpha = Keras-Adam lr_t of the variable being updated
};

__device__ __forceinline__ void opt_update1(const OptScalars &o, float g, float &p, float &s1, float &s2) {
  if (o.kind == DRX_OPT_ADAM) {
    // TF's ApplyAdam functor, operation for operation (1 - beta is formed in fp32 there too)
    s1 = s1 + (g - s1) * (1.0f - o.b1);
    s2 = s2 + (g * g - s2) * (1.0f - o.b2);
    p = p - (s1 * o.alpha) / (sqrtf(s2) + o.eps);
  } else {
    s1 = s1 + g * g;
    p = p - o.lr * g / (sqrtf(s1) + o.eps);
  }
}

// Applies g (data gradient, already complete) + rb*p to one row of `tab` with slots s1/s2.
template <int G, int J>
__device__ __forceinline__ float row_update(const OptScalars &o, float *tab, float *s1, float *s2, size_t row, int ld,
                                            int lane, const float4 (&w)[J], const float4 (&g)[J]) {
  float sq = 0.f;
  float4 *pr = reinterpret_cast<float4 *>(tab + row * (size_t)ld);
  float4 *a1 = reinterpret_cast<float4 *>(s1 + row * (size_t)ld);
  float4 *a2 = (o.kind == DRX_OPT_ADAM) ? reinterpret_cast<float4 *>(s2 + row * (size_t)ld) : nullptr;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = lane + j * G;
    if (4 * c < ld) {
      float4 p = w[j];
      float4 m = a1[c];
      float4 v = a2 ? a2[c] : f4_zero();
      sq += f4_dot(p, p);
      opt_update1(o, fmaf(o.rb, p.x, g[j].x), p.x, m.x, v.x);
      opt_update1(o, fmaf(o.rb, p.y, g[j].y), p.y, m.y, v.y);
      opt_update1(o, fmaf(o.rb, p.z, g[j].z), p.z, m.z, v.z);
      opt_update1(o, fmaf(o.rb, p.w, g[j].w), p.w, m.w, v.w);
      pr[c] = p;
      a1[c] = m;
      if (a2) a2[c] = v;
    }
  }
  return sq;   // partial |row|^2 of this lane (pre-update), for the L2 loss value
}

__device__ __forceinline__ OptScalars opt_for(const DrxOptim &opt, int var, int B) {
  OptScalars o;
  o.kind = opt.kind; o.lr = opt.lr; o.rb = opt.reg_rate / (float)B;
  o.b1 = opt.beta1; o.b2 = opt.beta2; o.eps = opt.eps; o.alpha = opt.alpha[var];
  return o;
}

// Deterministic block reduction of one float per thread -> thread 0 holds the sum.
__device__ __forceinline__ float block_sum(float v, float *red /* [kBlock/64] in LDS */) {
  v = group_sum<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < kBlock / 64; ++i) t += red[i];
  return t;
}

// ------------------------------------------------------------------------------------------------
// reference mode, output layer: forward over ALL units, loss vs batch-mean / per-row target, dz2,
// dW2T/db2 (+L2) with fused Adam, and the per-workgroup partial of dh = dz2 . W_^T.
// One GROUP owns one output unit (its W2T row stays in registers), the sub-batch's hidden rows live
// in LDS; workgroups are persistent over tiles of R = 256/G units.
// ------------------------------------------------------------------------------------------------
struct OutDenseArgs {
  const float *h;         // [B, ld]
  const int32_t *cnt;     // [N]
  const uint32_t *tb;     // [B, Nw] or null
  int Nw;
  int B, Bs, n_sub;       // sub-batch rows resident in LDS, number of sub-batches
  float *gbuf;            // [N, ld] gradient accumulator across sub-batches (n_sub > 1)
  float *gb2buf;          // [N]
  float *dh_slab;         // [grid, B, ld]
  float *loss_part;       // [grid] prediction-loss partials
  float *reg_part;        // [grid] sum w^2 partials of W2T
  int loss_kind;
};

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_out_dense(DrxCdaeParams P, DrxOptim opt, OutDenseArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int R = kBlock / G;
  const int ld = P.ld;
  float *h_s = lds;                        // [Bs, ld]
  float *dh_s = h_s + (size_t)A.Bs * ld;   // [Bs, ld]
  float *w_s = dh_s + (size_t)A.Bs * ld;   // [R, ld]
  float *dz_s = w_s + (size_t)R * ld;      // [Bs, R]
  __shared__ float red[kBlock / 64];
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int n_tiles = (P.n_items + R - 1) / R;
  const OptScalars oW = opt_for(opt, 1, A.B), oB = opt_for(opt, 4, A.B);
  const float invBN = 1.0f / ((float)A.B * (float)P.n_items);
  const float invB = 1.0f / (float)A.B;
  float loss_acc = 0.f, reg_acc = 0.f;

  for (int sb = 0; sb < A.n_sub; ++sb) {
    const int b0 = sb * A.Bs;
    const int nb = min(A.Bs, A.B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * ld / 4; i += kBlock) {
      reinterpret_cast<float4 *>(h_s)[i] = reinterpret_cast<const float4 *>(A.h + (size_t)b0 * ld)[i];
      reinterpret_cast<float4 *>(dh_s)[i] = f4_zero();
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const int n = tile * R + r;
      const bool live = n < P.n_items;
      float4 w[J], gw[J];
#pragma unroll
      for (int j = 0; j < J; ++j) { w[j] = f4_zero(); gw[j] = f4_zero(); }
      float bias = 0.f, tbar = 0.f, gb2 = 0.f;
      if (live) {
        load_row<G, J>(P.W2T, (size_t)n, ld, lane, w);
        bias = P.b2[n];
        tbar = (float)A.cnt[n] * invB;
      }
      store_row<G, J>(w_s, (size_t)r, ld, lane, w);
      for (int b = 0; b < nb; ++b) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int c = lane + j * G;
          if (4 * c < ld) d += f4_dot(w[j], reinterpret_cast<const float4 *>(h_s + (size_t)b * ld)[c]);
        }
        d = group_sum<G>(d);
        float dz = 0.f;
        if (live) {
          const float p = sigmoidf_(d + bias);
          float t = tbar;
          if (A.tb) t = (A.tb[(size_t)(b0 + b) * A.Nw + (n >> 5)] >> (n & 31)) & 1u ? 1.0f : 0.0f;
          float dp;
          if (A.loss_kind == DRX_LOSS_BCE) {
            loss_acc += bce_elem(t, p);
            dp = bce_grad(t, p) * invBN;
          } else {
            const float df = p - t;
            // (B,B,N) broadcast of squared error: (p - tbar)^2 + var(t) for binary targets
            loss_acc += df * df + (A.tb ? 0.f : t * (1.0f - t));
            dp = 2.0f * df * invBN;
          }
          dz = dp * p * (1.0f - p);
          gb2 += dz;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int c = lane + j * G;
            if (4 * c < ld) f4_fma(gw[j], dz, reinterpret_cast<const float4 *>(h_s + (size_t)b * ld)[c]);
          }
        }
        if (lane == 0) dz_s[b * R + r] = dz;
      }
      __syncthreads();
      // dh_s[b,:] += sum_r dz_s[b,r] * w_s[r,:]   (each thread owns fixed (b, col) cells)
      for (int i = threadIdx.x; i < nb * (ld / 4); i += kBlock) {
        const int b = i / (ld / 4), c = i % (ld / 4);
        float4 a = reinterpret_cast<float4 *>(dh_s + (size_t)b * ld)[c];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) f4_fma(a, dz_s[b * R + rr], reinterpret_cast<const float4 *>(w_s + (size_t)rr * ld)[c]);
        reinterpret_cast<float4 *>(dh_s + (size_t)b * ld)[c] = a;
      }
      // weight update of unit n (gradient complete after the last sub-batch)
      if (live) {
        if (A.n_sub > 1) {
          float4 acc[J];
          if (sb > 0) load_row<G, J>(A.gbuf, (size_t)n, ld, lane, acc);
          if (sb > 0) {
#pragma unroll
            for (int j = 0; j < J; ++j) f4_add(gw[j], acc[j]);
            gb2 += A.gb2buf[n];
          }
          if (sb + 1 < A.n_sub) {
            store_row<G, J>(A.gbuf, (size_t)n, ld, lane, gw);
            if (lane == 0) A.gb2buf[n] = gb2;
          }
        }
        if (sb + 1 == A.n_sub) {
          reg_acc += row_update<G, J>(oW, P.W2T, opt.s1[1], opt.s2[1], (size_t)n, ld, lane, w, gw);
          if (lane == 0) {
            float pb = bias, m = opt.s1[4][n], v = oB.kind == DRX_OPT_ADAM ? opt.s2[4][n] : 0.f;
            OptScalars ob = oB; ob.rb = 0.f;
            opt_update1(ob, gb2, pb, m, v);
            P.b2[n] = pb; opt.s1[4][n] = m;
            if (oB.kind == DRX_OPT_ADAM) opt.s2[4][n] = v;
          }
        }
      }
      __syncthreads();
    }
    // this workgroup's partial of dh for the sub-batch rows
    for (int i = threadIdx.x; i < nb * ld / 4; i += kBlock)
      reinterpret_cast<float4 *>(A.dh_slab + ((size_t)blockIdx.x * A.B + b0) * ld)[i] = reinterpret_cast<float4 *>(dh_s)[i];
  }
  // every lane of a group accumulated the same loss terms; count them once (lane 0)
  float lsum = block_sum(lane == 0 ? loss_acc : 0.f, red);
  float rsum = block_sum(reg_acc, red);
  if (threadIdx.x == 0) { A.loss_part[blockIdx.x] = lsum * invBN; A.reg_part[blockIdx.x] = rsum; }
}

// dz1[b,:] = (sum_slabs dh) * h (1-h)      one workgroup per batch row, groups stride over slabs
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_hidden_bwd(int ld, int B, int n_slabs, const float *__restrict__ slab,
                                                       const float *__restrict__ h, float *__restrict__ dz1) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int s = r; s < n_slabs; s += R) {
    float4 v[J];
    load_row<G, J>(slab, (size_t)s * B + b, ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J], hv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
    }
    load_row<G, J>(h, (size_t)b, ld, lane, hv);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      t[j].x *= hv[j].x * (1.0f - hv[j].x); t[j].y *= hv[j].y * (1.0f - hv[j].y);
      t[j].z *= hv[j].z * (1.0f - hv[j].z); t[j].w *= hv[j].w * (1.0f - hv[j].w);
    }
    store_row<G, J>(dz1, (size_t)b, ld, lane, t);
  }
}

// Dense sweep over W rows [0,N) and V rows [N, N+U): gradient from the per-row batch bitmasks (bits
// ascending => deterministic sum order) + L2, Adam.  The last workgroup updates the hidden bias b.
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_in_sweep(DrxCdaeParams P, DrxOptim opt, int B, float scale, DenseAux aux,
                                                     const float *__restrict__ dz1, float *reg_part) {
  __shared__ float red[kBlock / 64];
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  const int ld = P.ld;
  float reg_acc = 0.f;
  if (blockIdx.x == gridDim.x - 1) {   // hidden bias: g = sum_b dz1[b,:]   (no L2 on biases, cdae.py:82)
    if (threadIdx.x < G) {
      float4 g[J], w[J];
#pragma unroll
      for (int j = 0; j < J; ++j) g[j] = f4_zero();
      for (int b = 0; b < B; ++b) {
        float4 v[J];
        load_row<G, J>(dz1, (size_t)b, ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
      }
      load_row<G, J>(P.b, 0, ld, lane, w);
      OptScalars o = opt_for(opt, 3, B);
      o.rb = 0.f;
      row_update<G, J>(o, P.b, opt.s1[3], opt.s2[3], 0, ld, lane, w, g);
    }
    if (threadIdx.x == 0) reg_part[blockIdx.x] = 0.f;
    return;
  }
  const int total = P.n_items + P.n_users;
  const OptScalars oW = opt_for(opt, 0, B), oV = opt_for(opt, 2, B);
  for (int row = blockIdx.x * gpb + threadIdx.x / G; row < total; row += (gridDim.x - 1) * gpb) {
    const bool isW = row < P.n_items;
    const size_t rr = isW ? row : row - P.n_items;
    const uint32_t *mask = isW ? aux.km + rr * aux.Bw : aux.vm + rr * aux.Bw;
    float4 g[J], w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = f4_zero();
    for (int wd = 0; wd < aux.Bw; ++wd) {
      uint32_t m = mask[wd];
      while (m) {
        const int b = wd * 32 + __builtin_ctz(m);
        m &= m - 1;
        float4 v[J];
        load_row<G, J>(dz1, (size_t)b, ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
      }
    }
    if (isW) {
#pragma unroll
      for (int j = 0; j < J; ++j) { g[j].x *= scale; g[j].y *= scale; g[j].z *= scale; g[j].w *= scale; }
      load_row<G, J>(P.W, rr, ld, lane, w);
      reg_acc += row_update<G, J>(oW, P.W, opt.s1[0], opt.s2[0], rr, ld, lane, w, g);
    } else {
      load_row<G, J>(P.V, rr, ld, lane, w);
      reg_acc += row_update<G, J>(oV, P.V, opt.s1[2], opt.s2[2], rr, ld, lane, w, g);
    }
  }
  float rsum = block_sum(reg_acc, red);
  if (threadIdx.x == 0) reg_part[blockIdx.x] = rsum;
}

__global__ void k_loss_final(const float *lp, int nl, const float *rp1, int n1, const float *rp2, int n2, float reg_half_rb,
                             float *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float l = 0.f, r = 0.f;
    for (int i = 0; i < nl; ++i) l += lp[i];
    for (int i = 0; i < n1; ++i) r += rp1[i];
    for (int i = 0; i < n2; ++i) r += rp2[i];
    out[0] = l;
    out[1] = r * reg_half_rb;
  }
}

// ------------------------------------------------------------------------------------------------
// sampled-output mode: one group per (u, i, y) triple — gather, hidden, one output unit, loss,
// backward to dz1 / g2 rows, and the (row key, sample) touch list for the inverted index.
// key space: [0,N) W rows, [N,2N) W2T rows, [2N, 2N+U) V rows.
// ------------------------------------------------------------------------------------------------
struct SparseBufs {
  float *dz1;       // [B, ld]
  float *g2;        // [B, ld]  dz2_b * h_b
  float *dz2;       // [B]
  float *lossb;     // [B]
  uint32_t *keys, *vals, *keys_s, *vals_s;    // [T]
  float *phead, *ptail;                       // [n_chunks, ld]
  float *phs, *pts;                           // [n_chunks] scalar (b2) partials
  uint32_t *span_list, *long_list;            // [n_chunks] each
  uint32_t *n_span;                           // [0] crossing segments, [1] long ones
  float *bpart;                               // [n_bpart, ld]
  int T, n_chunks, n_bpart;
};

constexpr int kChunk = 32;      // touches per group in the segmented reduction

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_sampled_fwd_bwd(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale,
                                                            uint32_t qthr, int loss_kind, SparseBufs S) {
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  const int u = bt.uid[b], i = bt.iid[b];
  const float y = bt.y[b];
  const int base = bt.keep_off[b] + 2 * b;
  const int deg = bt.keep_off[b + 1] - bt.keep_off[b];
  float4 acc[J], h[J], w2[J];
  DenseAux none{};
  gather_bag<G, J, 2>(P, H, bt, qthr, b, lane, acc, none, S.keys, S.vals, base);
  if (lane == 0) {
    S.keys[base + deg] = (uint32_t)(P.n_items + i);       S.vals[base + deg] = (uint32_t)b;
    S.keys[base + deg + 1] = (uint32_t)(2 * P.n_items + u); S.vals[base + deg + 1] = (uint32_t)b;
  }
  hidden_act<G, J>(P, u, scale, lane, acc, h);
  load_row<G, J>(P.W2T, (size_t)i, P.ld, lane, w2);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) d += f4_dot(w2[j], h[j]);
  d = group_sum<G>(d);
  const float p = sigmoidf_(d + P.b2[i]);
  const float invB = 1.0f / (float)bt.B;
  float lval, dp;
  if (loss_kind == DRX_LOSS_BCE) { lval = bce_elem(y, p); dp = bce_grad(y, p) * invB; }
  else { lval = (p - y) * (p - y); dp = 2.0f * (p - y) * invB; }
  const float dz2 = dp * p * (1.0f - p);
  float4 dz1[J], g2[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    dz1[j].x = dz2 * w2[j].x * h[j].x * (1.0f - h[j].x); dz1[j].y = dz2 * w2[j].y * h[j].y * (1.0f - h[j].y);
    dz1[j].z = dz2 * w2[j].z * h[j].z * (1.0f - h[j].z); dz1[j].w = dz2 * w2[j].w * h[j].w * (1.0f - h[j].w);
    g2[j].x = dz2 * h[j].x; g2[j].y = dz2 * h[j].y; g2[j].z = dz2 * h[j].z; g2[j].w = dz2 * h[j].w;
  }
  store_row<G, J>(S.dz1, (size_t)b, P.ld, lane, dz1);
  store_row<G, J>(S.g2, (size_t)b, P.ld, lane, g2);
  if (lane == 0) { S.dz2[b] = dz2; S.lossb[b] = lval; }
}

template <int G, int J>
__device__ __forceinline__ void sparse_apply(const DrxCdaeParams &P, const DrxOptim &opt, int B, uint32_t key, int lane,
                                             const float4 (&g)[J], float gs) {
  const uint32_t N = (uint32_t)P.n_items;
  float *tab, *s1, *s2;
  size_t row;
  int var;
  if (key < N) { tab = P.W; var = 0; row = key; }
  else if (key < 2 * N) { tab = P.W2T; var = 1; row = key - N; }
  else { tab = P.V; var = 2; row = key - 2 * N; }
  s1 = opt.s1[var]; s2 = opt.s2[var];
  OptScalars o = opt_for(opt, 0, B);
  float4 w[J];
  load_row<G, J>(tab, row, P.ld, lane, w);
  row_update<G, J>(o, tab, s1, s2, row, P.ld, lane, w, g);
  if (var == 1 && lane == 0) {
    float pb = P.b2[row], m = opt.s1[4][row], v = o.kind == DRX_OPT_ADAM ? opt.s2[4][row] : 0.f;
    o.rb = 0.f;
    opt_update1(o, gs, pb, m, v);
    P.b2[row] = pb; opt.s1[4][row] = m;
    if (o.kind == DRX_OPT_ADAM) opt.s2[4][row] = v;
  }
}

// Segmented reduction over the sorted touch list in fixed chunks of kChunk touches per group.
// Segments that lie inside one chunk are updated here; segments crossing chunk borders leave
// partial rows that k_span_fixup combines in chunk order (deterministic).
// The chunk's (key, sample) pairs are fetched with one coalesced load per lane and broadcast by shuffles; the
// contribution rows are then loaded LB at a time (independent loads in flight) before they are folded in order.
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_seg_reduce(DrxCdaeParams P, DrxOptim opt, int B, float scale, SparseBufs S) {
  const int lane = threadIdx.x % G;
  const int g = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (g >= S.n_chunks) return;
  const int start = g * kChunk, end = min(S.T, start + kChunk);
  const int n = end - start;
  const uint32_t N = (uint32_t)P.n_items;
  const uint32_t prev_key = start > 0 ? S.keys_s[start - 1] : DRX_KEY_NONE;
  const uint32_t next_key = end < S.T ? S.keys_s[end] : DRX_KEY_NONE;
  constexpr int KPL = (kChunk + G - 1) / G;          // (key, val) registers per lane
  constexpr int LB = J == 1 ? 8 : (J == 2 ? 4 : 2);  // rows in flight per group
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
  uint32_t cur = DRX_KEY_NONE, last_key = DRX_KEY_NONE;
  bool cur_from_start = false;
  auto flush = [&](bool at_end) {
    if (cur == DRX_KEY_NONE) return;
    const bool cont_left = cur_from_start && prev_key == cur;
    const bool cont_right = at_end && next_key == cur;
    if (!cont_left && !cont_right) {
      sparse_apply<G, J>(P, opt, B, cur, lane, acc, accs);
    } else if (cont_left) {
      store_row<G, J>(S.phead, (size_t)g, P.ld, lane, acc);
      if (lane == 0) S.phs[g] = accs;
    } else {
      store_row<G, J>(S.ptail, (size_t)g, P.ld, lane, acc);
      if (lane == 0) {
        S.pts[g] = accs;
        const uint32_t slot = atomicAdd(S.n_span, 1u);
        S.span_list[slot] = (uint32_t)g;
      }
    }
  };
  for (int t0 = 0; t0 < n; t0 += LB) {
    uint32_t k8[LB];
    float s8[LB];
    float4 rows[LB][J];
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int t = t0 + u;
      k8[u] = t < n ? bcast(kreg, t) : DRX_KEY_NONE;    // padding (dropped inputs) sorts last
      const uint32_t b = bcast(vreg, t < n ? t : 0);
      s8[u] = 0.f;
#pragma unroll
      for (int jx = 0; jx < J; ++jx) rows[u][jx] = f4_zero();
      if (k8[u] != DRX_KEY_NONE) {
        const bool is_out = k8[u] >= N && k8[u] < 2 * N;
        load_row<G, J>(is_out ? S.g2 : S.dz1, (size_t)b, P.ld, lane, rows[u]);
        if (is_out) s8[u] = S.dz2[b];
      }
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
        const float c = key < N ? scale : 1.0f;
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_fma(acc[jx], c, rows[u][jx]);
        accs += s8[u];
        last_key = key;
      }
    }
  }
  // the last segment ends at the chunk border iff the final touch of the chunk is a real key
  const bool ran_to_end = n > 0 && bcast(kreg, n - 1) != DRX_KEY_NONE;
  (void)last_key;
  flush(ran_to_end);
}

// Fix-up of chunk-crossing segments, two tiers.
//   k_span_short : one GROUP per crossing segment: tail partial of its first chunk + head partials of the next chunks
//                  whose first key equals the segment key, in chunk order.  Segments that cross more than
//                  kShortSpan chunk borders (hot items) are queued for
//   k_span_long  : one 1024-thread workgroup per such segment; its R = 1024/G groups stride over the chunks and the
//                  R partial sums are combined in a fixed order.  Both tiers are deterministic.
constexpr int kShortSpan = 6;
constexpr int kFixBlock = 1024;

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_span_short(DrxCdaeParams P, DrxOptim opt, int B, SparseBufs S) {
  const int lane = threadIdx.x % G;
  const uint32_t n_span = S.n_span[0];
  const int gpb = kBlock / G;
  for (uint32_t si = blockIdx.x * gpb + threadIdx.x / G; si < n_span; si += gridDim.x * gpb) {
    const int g0 = (int)S.span_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    // number of following chunks that continue this segment (bounded look-ahead)
    int m = 0;
    while (m <= kShortSpan && g0 + 1 + m < S.n_chunks && S.keys_s[(g0 + 1 + m) * kChunk] == key) ++m;
    if (m > kShortSpan) {
      if (lane == 0) S.long_list[atomicAdd(&S.n_span[1], 1u)] = (uint32_t)g0;
      continue;
    }
    float4 t[J];
    load_row<G, J>(S.ptail, (size_t)g0, P.ld, lane, t);
    float ts = S.pts[g0];
    for (int c = g0 + 1; c <= g0 + m; ++c) {
      float4 v[J];
      load_row<G, J>(S.phead, (size_t)c, P.ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
      ts += S.phs[c];
    }
    sparse_apply<G, J>(P, opt, B, key, lane, t, ts);
  }
}

template <int G, int J>
__global__ __launch_bounds__(kFixBlock) void k_span_long(DrxCdaeParams P, DrxOptim opt, int B, SparseBufs S) {
  extern __shared__ __align__(16) float lds[];   // [R, ld] + [R]
  constexpr int R = kFixBlock / G;
  float *sc = lds + (size_t)R * P.ld;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const uint32_t n_long = S.n_span[1];
  for (uint32_t si = blockIdx.x; si < n_long; si += gridDim.x) {
    const int g0 = (int)S.long_list[si];
    const uint32_t key = S.keys_s[min(S.T, (g0 + 1) * kChunk) - 1];
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
    float accs = 0.f;
    for (int c = g0 + 1 + r; c < S.n_chunks; c += 2 * R) {
      if (S.keys_s[c * kChunk] != key) break;
      const int c2 = c + R;
      const bool two = c2 < S.n_chunks && S.keys_s[c2 * kChunk] == key;
      float4 v[J], v2[J];
      load_row<G, J>(S.phead, (size_t)c, P.ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) v2[j] = f4_zero();
      float s2 = 0.f;
      if (two) { load_row<G, J>(S.phead, (size_t)c2, P.ld, lane, v2); s2 = S.phs[c2]; }
#pragma unroll
      for (int j = 0; j < J; ++j) { f4_add(acc[j], v[j]); f4_add(acc[j], v2[j]); }
      accs += S.phs[c];
      accs += s2;
      if (!two) break;
    }
    __syncthreads();
    store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
    if (lane == 0) sc[r] = accs;
    __syncthreads();
    if (r == 0) {
      float4 t[J];
      load_row<G, J>(S.ptail, (size_t)g0, P.ld, lane, t);
      float ts = S.pts[g0];
      for (int rr = 0; rr < R; ++rr) {
        float4 v[J];
        load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
        ts += sc[rr];
      }
      sparse_apply<G, J>(P, opt, B, key, lane, t, ts);
    }
  }
}

// hidden bias b: column sums of dz1 in two deterministic stages, then a dense optimizer update.
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_bias_partial(int ld, int B, const float *__restrict__ dz1, float *__restrict__ part,
                                                         int rows_per_block) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b0 = blockIdx.x * rows_per_block, b1 = min(B, b0 + rows_per_block);
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int b = b0 + r; b < b1; b += R) {
    float4 v[J];
    load_row<G, J>(dz1, (size_t)b, ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], v[j]);
    }
    store_row<G, J>(part, (size_t)blockIdx.x, ld, lane, t);
  }
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_bias_final(DrxCdaeParams P, DrxOptim opt, int B, const float *__restrict__ part,
                                                       int n_part, const float *__restrict__ lossb, float *loss_out) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  __shared__ float red[kBlock / 64];
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int i = r; i < n_part; i += R) {
    float4 v[J];
    load_row<G, J>(part, (size_t)i, P.ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 g[J], w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = f4_zero();
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
    }
    load_row<G, J>(P.b, 0, P.ld, lane, w);
    OptScalars o = opt_for(opt, 0, B);
    o.rb = 0.f;
    row_update<G, J>(o, P.b, opt.s1[3], opt.s2[3], 0, P.ld, lane, w, g);
  }
  if (loss_out) {   // mean of the per-sample losses, fixed order
    float a = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) a += lossb[b];
    float t = block_sum(a, red);
    if (threadIdx.x == 0) { loss_out[0] = t / (float)B; loss_out[1] = 0.f; }
  }
}

// ------------------------------------------------------------------------------------------------
// scratch layouts (shared by the sizing entry point and the step functions)
// ------------------------------------------------------------------------------------------------
constexpr int kOutGrid = 256;        // persistent workgroups of k_out_dense (one per CU)
constexpr int kSweepGrid = 1024;
constexpr size_t kLdsBudget = 144 * 1024;

struct DenseLayout {
  float *h, *dz1, *dh_slab, *loss_part, *reg_part1, *reg_part2, *gbuf, *gb2buf;
  int32_t *cnt;
  uint32_t *km, *vm, *tb;
  size_t zero_begin, zero_end;
  int Bw, Nw, Bs, n_sub, out_grid;
  size_t lds_bytes;
};

static DenseLayout dense_layout(Carver &cv, const DrxCdaeParams &P, int B, bool per_row) {
  DenseLayout L{};
  const Geom gm = pick_geom(P.ld);
  const int R = kBlock / gm.G;
  const int n_tiles = (P.n_items + R - 1) / R;
  L.out_grid = n_tiles < kOutGrid ? n_tiles : kOutGrid;
  // LDS: 2*Bs*ld + R*ld + Bs*R floats
  size_t fixed = (size_t)R * P.ld * 4;
  size_t per_b = ((size_t)2 * P.ld + R) * 4;
  int Bs = (int)((kLdsBudget - fixed) / per_b);
  if (Bs > B) Bs = B;
  if (Bs < 1) Bs = 1;
  L.Bs = Bs;
  L.n_sub = (B + Bs - 1) / Bs;
  L.lds_bytes = fixed + per_b * Bs;
  L.Bw = (B + 31) / 32;
  L.Nw = (P.n_items + 31) / 32;
  L.h = cv.take<float>((size_t)B * P.ld);
  L.dz1 = cv.take<float>((size_t)B * P.ld);
  L.dh_slab = cv.take<float>((size_t)L.out_grid * B * P.ld);
  L.loss_part = cv.take<float>(L.out_grid);
  L.reg_part1 = cv.take<float>(L.out_grid);
  L.reg_part2 = cv.take<float>(kSweepGrid + 1);
  L.gbuf = L.n_sub > 1 ? cv.take<float>((size_t)P.n_items * P.ld) : nullptr;
  L.gb2buf = L.n_sub > 1 ? cv.take<float>(P.n_items) : nullptr;
  cv.off = align_up(cv.off, 256);
  L.zero_begin = cv.off;
  L.cnt = cv.take<int32_t>(P.n_items);
  L.km = cv.take<uint32_t>((size_t)P.n_items * L.Bw);
  L.vm = cv.take<uint32_t>((size_t)P.n_users * L.Bw);
  L.tb = cv.take<uint32_t>((size_t)B * L.Nw);      // always reserved so the size does not depend on the mode
  L.zero_end = cv.off;
  (void)per_row;
  return L;
}

static SparseBufs sparse_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots, size_t *sort_bytes,
                                void **sort_temp) {
  SparseBufs S{};
  S.T = n_touch_slots + 2 * B;
  S.n_chunks = (S.T + kChunk - 1) / kChunk;
  S.n_bpart = 256;
  S.dz1 = cv.take<float>((size_t)B * P.ld);
  S.g2 = cv.take<float>((size_t)B * P.ld);
  S.dz2 = cv.take<float>(B);
  S.lossb = cv.take<float>(B);
  S.keys = cv.take<uint32_t>(S.T);
  S.vals = cv.take<uint32_t>(S.T);
  S.keys_s = cv.take<uint32_t>(S.T);
  S.vals_s = cv.take<uint32_t>(S.T);
  S.phead = cv.take<float>((size_t)S.n_chunks * P.ld);
  S.ptail = cv.take<float>((size_t)S.n_chunks * P.ld);
  S.phs = cv.take<float>(S.n_chunks);
  S.pts = cv.take<float>(S.n_chunks);
  S.span_list = cv.take<uint32_t>(S.n_chunks);
  S.long_list = cv.take<uint32_t>(S.n_chunks);
  S.n_span = cv.take<uint32_t>(64);
  S.bpart = cv.take<float>((size_t)S.n_bpart * P.ld);
  const int bits = bits_for((uint64_t)2 * P.n_items + P.n_users + 1);
  *sort_bytes = sort_pairs_temp_bytes(S.T, bits);
  *sort_temp = cv.take<char>(*sort_bytes);
  return S;
}

static int check_params(const DrxCdaeParams *p) {
  if (!p || !p->W || !p->W2T || !p->V || !p->b || !p->b2) return DRX_EINVAL;
  if (p->k < 1 || p->k > DRX_MAX_K || p->ld < p->k || (p->ld & 3) || p->ld > DRX_MAX_K) return DRX_EINVAL;
  if (p->n_users < 1 || p->n_items < 1) return DRX_EINVAL;
  return DRX_OK;
}

static int check_batch(const DrxHistory *h, const DrxBatch *bt) {
  if (!h || !h->indptr || !h->indices || !bt || !bt->uid) return DRX_EINVAL;
  if (bt->keep && !bt->keep_off) return DRX_EINVAL;
  if (bt->B < 1 || bt->q < 0.f || bt->q >= 1.f) return DRX_EINVAL;
  return DRX_OK;
}

}  // namespace drx

using namespace drx;

extern "C" {

int drx_version(void) { return DRX_VERSION; }

const char *drx_strerror(int code) {
  switch (code) {
    case DRX_OK: return "ok";
    case DRX_EINVAL: return "invalid argument";
    case DRX_ESCRATCH: return "scratch buffer too small";
    case DRX_ENOTIMPL: return "not implemented";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown drx error";
  }
}

uint32_t drx_hash_u32(uint64_t seed, uint32_t a, uint32_t b) { return hash_u32(seed, a, b); }

int drx_cdae_forward(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, float *h, float *pred,
                     void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !h) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  DenseAux none{};
#define CALL(G, J)                                                                                        \
  {                                                                                                       \
    const int gpb = kBlock / G;                                                                           \
    hipLaunchKernelGGL((k_hidden_fwd<G, J, 0>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt, \
                       scale, qthr, h, none);                                                             \
    if (pred) {                                                                                           \
      int blocks = (p->n_items + gpb - 1) / gpb;                                                          \
      if (blocks > 2048) blocks = 2048;                                                                   \
      hipLaunchKernelGGL((k_out_fwd<G, J>), dim3(blocks), dim3(kBlock), 0, st, *p, h, bt->B, pred);        \
    }                                                                                                     \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_cdae_scratch_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots) {
  if (!p || B < 1 || n_touch_slots < 0) return 0;
  Carver c1(nullptr, 0);
  (void)dense_layout(c1, *p, B, true);
  Carver c2(nullptr, 0);
  size_t sb; void *stmp;
  (void)sparse_layout(c2, *p, B, n_touch_slots, &sb, &stmp);
  size_t m = c1.off > c2.off ? c1.off : c2.off;
  return align_up(m, 256) + 256;
}

int drx_cdae_step_dense(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                        int32_t loss_kind, int32_t targets_kind, void *scratch, size_t scratch_bytes, float *loss_out,
                        void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !opt || !scratch) return DRX_EINVAL;
  if (opt->kind != DRX_OPT_ADAM && opt->kind != DRX_OPT_ADAGRAD) return DRX_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!opt->s1[i] || (opt->kind == DRX_OPT_ADAM && !opt->s2[i])) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  const bool per_row = targets_kind == DRX_TARGETS_PER_ROW;
  DenseLayout L = dense_layout(cv, *p, bt->B, per_row);
  if (!cv.ok()) return DRX_ESCRATCH;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  DRX_HIP(hipMemsetAsync((char *)scratch + L.zero_begin, 0, L.zero_end - L.zero_begin, st));
  DenseAux aux{L.cnt, L.km, L.vm, per_row ? L.tb : nullptr, L.Bw, L.Nw};
  OutDenseArgs A{};
  A.h = L.h; A.cnt = L.cnt; A.tb = aux.tb; A.Nw = L.Nw; A.B = bt->B; A.Bs = L.Bs; A.n_sub = L.n_sub;
  A.gbuf = L.gbuf; A.gb2buf = L.gb2buf; A.dh_slab = L.dh_slab; A.loss_part = L.loss_part; A.reg_part = L.reg_part1;
  A.loss_kind = loss_kind;
  const int total_rows = p->n_items + p->n_users;
#define CALL(G, J)                                                                                                   \
  {                                                                                                                  \
    const int gpb = kBlock / G;                                                                                      \
    hipLaunchKernelGGL((k_hidden_fwd<G, J, 1>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt,  \
                       scale, qthr, L.h, aux);                                                                       \
    DRX_HIP(hipFuncSetAttribute((const void *)k_out_dense<G, J>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                (int)L.lds_bytes));                                                                  \
    hipLaunchKernelGGL((k_out_dense<G, J>), dim3(L.out_grid), dim3(kBlock), L.lds_bytes, st, *p, *opt, A);           \
    hipLaunchKernelGGL((k_hidden_bwd<G, J>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, p->ld, bt->B,   \
                       L.out_grid, L.dh_slab, L.h, L.dz1);                                                           \
    int sweep = (total_rows + gpb - 1) / gpb;                                                                        \
    if (sweep > kSweepGrid) sweep = kSweepGrid;                                                                      \
    hipLaunchKernelGGL((k_in_sweep<G, J>), dim3(sweep + 1), dim3(kBlock), 0, st, *p, *opt, bt->B, scale, aux, L.dz1, \
                       L.reg_part2);                                                                                 \
    if (loss_out)                                                                                                    \
      hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, st, L.loss_part, L.out_grid, L.reg_part1, L.out_grid,   \
                         L.reg_part2, sweep + 1, 0.5f * opt->reg_rate / (float)bt->B, loss_out);                     \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

static int step_sparse_impl(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                            int32_t loss_kind, void *scratch, size_t scratch_bytes, float *loss_out, void *const *events,
                            void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !opt || !scratch || !bt->iid || !bt->y || !bt->keep_off) return DRX_EINVAL;
  if (opt->kind != DRX_OPT_ADAM && opt->kind != DRX_OPT_ADAGRAD) return DRX_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!opt->s1[i] || (opt->kind == DRX_OPT_ADAM && !opt->s2[i])) return DRX_EINVAL;
  if ((uint64_t)2 * p->n_items + p->n_users + 1 >= 0xFFFFFFFFull) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  size_t sort_bytes; void *sort_temp;
  SparseBufs S = sparse_layout(cv, *p, bt->B, bt->n_touch_slots, &sort_bytes, &sort_temp);
  if (!cv.ok()) return DRX_ESCRATCH;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  const int bits = bits_for((uint64_t)2 * p->n_items + p->n_users + 1);
  const int rows_per_block = (bt->B + S.n_bpart - 1) / S.n_bpart;
  const int n_bpart = (bt->B + rows_per_block - 1) / rows_per_block;
#define EV(i) do { if (events) DRX_HIP(hipEventRecord((hipEvent_t)events[i], st)); } while (0)
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    DRX_HIP(hipMemsetAsync(S.n_span, 0, 2 * sizeof(uint32_t), st));                                                        \
    /* slots beyond keep_off[B] (n_touch_slots may be an upper bound) must read as padding */                          \
    DRX_HIP(hipMemsetAsync(S.keys, 0xFF, (size_t)S.T * sizeof(uint32_t), st));                                         \
    EV(0);                                                                                                             \
    hipLaunchKernelGGL((k_sampled_fwd_bwd<G, J>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt,  \
                       scale, qthr, loss_kind, S);                                                                     \
    EV(1);                                                                                                             \
    rc = sort_pairs(sort_temp, sort_bytes, S.keys, S.keys_s, S.vals, S.vals_s, (size_t)S.T, bits, st);                 \
    if (rc) return rc;                                                                                                 \
    EV(2);                                                                                                             \
    hipLaunchKernelGGL((k_seg_reduce<G, J>), dim3((S.n_chunks + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, bt->B, \
                       scale, S);                                                                                      \
    EV(3);                                                                                                             \
    hipLaunchKernelGGL((k_span_short<G, J>), dim3(1024), dim3(kBlock), 0, st, *p, *opt, bt->B, S);                     \
    if (((size_t)(kFixBlock / G) * (p->ld + 1)) * 4 > 48 * 1024)                                                       \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_long<G, J>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                  (int)(((size_t)(kFixBlock / G) * (p->ld + 1)) * 4)));                                \
    hipLaunchKernelGGL((k_span_long<G, J>), dim3(256), dim3(kFixBlock), ((size_t)(kFixBlock / G) * (p->ld + 1)) * 4,   \
                       st, *p, *opt, bt->B, S);                                                                        \
    EV(4);                                                                                                             \
    hipLaunchKernelGGL((k_bias_partial<G, J>), dim3(n_bpart), dim3(kBlock), (size_t)gpb * p->ld * 4, st, p->ld, bt->B, \
                       S.dz1, S.bpart, rows_per_block);                                                                \
    hipLaunchKernelGGL((k_bias_final<G, J>), dim3(1), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *opt, bt->B,      \
                       S.bpart, n_bpart, S.lossb, loss_out);                                                           \
    EV(5);                                                                                                             \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
#undef EV
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_cdae_step_sparse(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                         int32_t loss_kind, void *scratch, size_t scratch_bytes, float *loss_out, void *stream) {
  return step_sparse_impl(p, opt, hist, bt, loss_kind, scratch, scratch_bytes, loss_out, nullptr, stream);
}

int drx_cdae_step_sparse_timed(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                               int32_t loss_kind, void *scratch, size_t scratch_bytes, float *loss_out, void *const *events,
                               void *stream) {
  if (!events) return DRX_EINVAL;
  return step_sparse_impl(p, opt, hist, bt, loss_kind, scratch, scratch_bytes, loss_out, events, stream);
}

}  // extern "C"
