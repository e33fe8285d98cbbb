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
#include "drx_rows.hpp"
#include "drx_segreduce.hpp"
#include <cstring>
#include "drx_scan.hpp"

#ifndef DRX_GATHER_ROWS
#define DRX_GATHER_ROWS 8
#endif

#ifdef DRX_STAMPS
static unsigned long long *h_stamps = nullptr;       // device buffer [110000 x 16], handed to the sparse step's kernels in their arguments
extern "C" int drx_debug_set_stamps(unsigned long long *buf, unsigned int) { h_stamps = buf; return 0; }
#endif

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

// Gathers scale * sum_{kept} W[n] for one batch row.  MODE 0: plain; 1: also builds DenseAux;
// 2: also emits the (key,val) touch list of the sampled mode.
template <int G, int J, int MODE>
__device__ __forceinline__ void gather_bag(const DrxCdaeParams &P, const DrxHistory &H, const DrxBatch &bt,
                                           uint32_t qthr, int b, int lane, float4 (&acc)[J],
                                           const DenseAux &aux, uint32_t *tkeys, uint32_t *tvals,
                                           int touch_base, int part = 0, int nparts = 1, unsigned long long *stamps = nullptr) {
  (void)stamps;
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
#ifdef DRX_STAMPS
  if (s >= 0) DRX_STAMP(stamps, b, 2, lane);          // (uses s: the stamp waits for the row pointers)
  bool first_rows = true;
#endif
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  // a group fetches CH history entries per round: its G lanes hold IPL each, so that narrow groups (rows of <= 32 floats:
  // G = 4 or 8) do not walk the history in rounds of 4 or 8 dependent index loads
  constexpr int IPL = G >= 16 ? 1 : 16 / G;
  constexpr int CH = G * IPL;
  for (int64_t c = s + (int64_t)part * CH; c < e; c += (int64_t)nparts * CH) {
    int idx[IPL], kf[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int64_t j = c + r * G + lane;
      idx[r] = -1; kf[r] = 0;
      if (j < e) {
        idx[r] = H.indices[j];
        const uint32_t jj = (uint32_t)(j - s);
        kf[r] = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
        if (MODE == 1) {
          atomicAdd(&aux.cnt[idx[r]], 1);
          if (aux.tb) atomicOr(&aux.tb[(size_t)b * aux.Nw + (idx[r] >> 5)], 1u << (idx[r] & 31));
          if (kf[r]) atomicOr(&aux.km[(size_t)idx[r] * aux.Bw + (b >> 5)], 1u << (b & 31));
        }
        if (MODE == 2) {
          tkeys[touch_base + jj] = kf[r] ? (uint32_t)idx[r] : DRX_KEY_NONE;
          tvals[touch_base + jj] = (uint32_t)b;
        }
      }
    }
    const int n_here = (int)((e - c) < (int64_t)CH ? (e - c) : (int64_t)CH);
    constexpr int NF = J == 1 ? DRX_GATHER_ROWS : 4;       // rows in flight per group
#ifdef DRX_STAMPS
    if (first_rows && idx[0] >= -1) DRX_STAMP(stamps, b, 3, lane);      // (uses idx: the stamp waits for the indices)
#endif
    for (int t = 0; t < n_here; t += NF) {
      float4 r[NF][J];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int tt = t + q;
        int si = idx[0], sk = kf[0];
#pragma unroll
        for (int rr = 1; rr < IPL; ++rr) { si = (tt / G == rr) ? idx[rr] : si; sk = (tt / G == rr) ? kf[rr] : sk; }
        const int iq = __shfl(si, tt % G, G);
        const int kq = (tt < n_here) ? __shfl(sk, tt % G, G) : 0;
#pragma unroll
        for (int jx = 0; jx < J; ++jx) r[q][jx] = f4_zero();
        if (kq) load_row<G, J>(P.W, (size_t)iq, P.ld, lane, r[q]);
      }
#pragma unroll
      for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int jx = 0; jx < J; ++jx) f4_add(acc[jx], r[q][jx]);
#ifdef DRX_STAMPS
      if (first_rows && acc[0].x == acc[0].x) { DRX_STAMP(stamps, b, 4, lane); first_rows = false; }      // (uses acc: after the first rows landed)
#endif
    }
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

// Small batches (the reference's own B = 32..64): one WORKGROUP per batch row — its 256/G groups split the user's history,
// each keeps 4 row loads in flight, and the partial bags are combined in LDS in group order.  With one group per row a
// 155-item history is a chain of ~40 dependent load batches on 8 workgroups of the whole chip (measured 94 us at ml-1m).
template <int G, int J, int MODE, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS) void k_hidden_fwd_wg(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale,
                                                          uint32_t qthr, float *__restrict__ hout, DenseAux aux) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = THREADS / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x;
  float4 acc[J], h[J];
  gather_bag<G, J, MODE>(P, H, bt, qthr, b, lane, acc, aux, nullptr, nullptr, 0, r, R);
  store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
  __syncthreads();
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = f4_zero();
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
    }
    const int u = bt.uid[b];
    if (MODE == 1 && lane == 0) atomicOr(&aux.vm[(size_t)u * aux.Bw + (b >> 5)], 1u << (b & 31));
    hidden_act<G, J>(P, u, scale, lane, acc, h);
    store_row<G, J>(hout, (size_t)b, P.ld, lane, h);
  }
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
// reference mode, output layer: forward over ALL units, loss vs batch-mean / per-row target, dz2,
// dW2T/db2 (+L2) with fused Adam, and the per-workgroup partial of dh = dz2 . W_^T.
// One GROUP owns one output unit (its W2T row stays in registers), the sub-batch's hidden rows live
// in LDS; workgroups are persistent over tiles of R = 256/G units.
// ------------------------------------------------------------------------------------------------
struct OutDenseArgs {
  const float *h;         // [B, ld]
  int32_t *cnt;           // [N]  (zeroed again by the tile that read it: the next step finds it clean)
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

template <int G, int J, bool WANT_LOSS>
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
        if (lane == 0 && sb + 1 == A.n_sub) A.cnt[n] = 0;       // (every lane of the group has read it: same instruction)
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
            if (WANT_LOSS) loss_acc += bce_elem(t, p);
            dp = bce_grad(t, p) * invBN;
          } else {
            const float df = p - t;
            // (B,B,N) broadcast of squared error: (p - tbar)^2 + var(t) for binary targets
            if (WANT_LOSS) loss_acc += df * df + (A.tb ? 0.f : t * (1.0f - t));
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

// The same output-layer step for batches that fit LDS whole (the reference's own B = 64): k_out_dense walks the batch rows one
// after the other with a cross-lane reduction per (row, unit) — a chain of B dependent shuffles that leaves the chip idle when
// there are only N/R tiles to spread.  Here a tile of kTileR units is three small register-tiled products out of LDS:
//   A  every thread owns (row, unit) pairs and forms their dot products serially over the columns -> p, loss, dz
//   B  every thread owns one float4 of one unit's gradient row: sum_b dz[b] h[b,:] in batch order, then the optimizer update
//   C  every thread owns (row, float4) cells of the tile's contribution to dh
// h rows and W2T rows are padded by 4 floats in LDS so that 8 different rows read by a wave fall on different banks.
constexpr int kTileR = 8;

__host__ __device__ inline size_t out_tile_lds_floats(int B, int ld) {
  return (size_t)B * (ld + 4) + (size_t)B * ld + (size_t)kTileR * (ld + 4) + (size_t)B * kTileR + 2 * kTileR;
}

template <bool WANT_LOSS>
__global__ __launch_bounds__(kBlock) void k_out_dense_tile(DrxCdaeParams P, DrxOptim opt, OutDenseArgs A) {
  constexpr int R = kTileR;
  extern __shared__ __align__(16) float lds[];
  const int ld = P.ld, ldp = ld + 4, B = A.B, c4n = ld / 4;
  float *h_s = lds;                              // [B, ldp]
  float *dh_s = h_s + (size_t)B * ldp;           // [B, ld]
  float *w_s = dh_s + (size_t)B * ld;            // [R, ldp]
  float *dz_s = w_s + (size_t)R * ldp;           // [B, R]
  float *bias_s = dz_s + (size_t)B * R;          // [R]
  float *tbar_s = bias_s + R;                    // [R]
  __shared__ float red[kBlock / 64];
  const int n_tiles = (P.n_items + R - 1) / R;
  const OptScalars oW = opt_for(opt, 1, B), oB = opt_for(opt, 4, B);
  const float invBN = 1.0f / ((float)B * (float)P.n_items);
  const float invB = 1.0f / (float)B;
  float loss_acc = 0.f, reg_acc = 0.f;

  for (int i = threadIdx.x; i < B * c4n; i += kBlock) {
    const int b = i / c4n, c = i % c4n;
    reinterpret_cast<float4 *>(h_s + (size_t)b * ldp)[c] = reinterpret_cast<const float4 *>(A.h + (size_t)b * ld)[c];
    reinterpret_cast<float4 *>(dh_s + (size_t)b * ld)[c] = f4_zero();
  }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();                             // previous tile's readers of w_s / dz_s are done (and h_s is loaded)
    for (int i = threadIdx.x; i < R * c4n; i += kBlock) {
      const int r = i / c4n, c = i % c4n, n = tile * R + r;
      reinterpret_cast<float4 *>(w_s + (size_t)r * ldp)[c] =
          n < P.n_items ? reinterpret_cast<const float4 *>(P.W2T + (size_t)n * ld)[c] : f4_zero();
    }
    if (threadIdx.x < R) {
      const int n = tile * R + threadIdx.x;
      bias_s[threadIdx.x] = n < P.n_items ? P.b2[n] : 0.f;
      tbar_s[threadIdx.x] = n < P.n_items ? (float)A.cnt[n] * invB : 0.f;
      if (n < P.n_items) A.cnt[n] = 0;
    }
    __syncthreads();
    // A: dot products, predictions, dz
    for (int pair = threadIdx.x; pair < B * R; pair += kBlock) {
      const int r = pair % R, b = pair / R, n = tile * R + r;
      const float4 *wr = reinterpret_cast<const float4 *>(w_s + (size_t)r * ldp);
      const float4 *hr = reinterpret_cast<const float4 *>(h_s + (size_t)b * ldp);
      float d0 = 0.f, d1 = 0.f;
      int c = 0;
#pragma unroll 4
      for (; c + 1 < c4n; c += 2) { d0 += f4_dot(wr[c], hr[c]); d1 += f4_dot(wr[c + 1], hr[c + 1]); }
      if (c < c4n) d0 += f4_dot(wr[c], hr[c]);
      float dz = 0.f;
      if (n < P.n_items) {
        const float p = sigmoidf_((d0 + d1) + bias_s[r]);
        float t = tbar_s[r];
        if (A.tb) t = (A.tb[(size_t)b * A.Nw + (n >> 5)] >> (n & 31)) & 1u ? 1.0f : 0.0f;
        float dp;
        if (A.loss_kind == DRX_LOSS_BCE) {
          if (WANT_LOSS) loss_acc += bce_elem(t, p);
          dp = bce_grad(t, p) * invBN;
        } else {
          const float df = p - t;
          if (WANT_LOSS) loss_acc += df * df + (A.tb ? 0.f : t * (1.0f - t));
          dp = 2.0f * df * invBN;
        }
        dz = dp * p * (1.0f - p);
      }
      dz_s[b * R + r] = dz;
    }
    __syncthreads();
    // C: dh_s[b,:] += sum_r dz[b,r] * w[r,:]
    for (int i = threadIdx.x; i < B * c4n; i += kBlock) {
      const int b = i / c4n, c = i % c4n;
      float4 a = reinterpret_cast<float4 *>(dh_s + (size_t)b * ld)[c];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) f4_fma(a, dz_s[b * R + rr], reinterpret_cast<const float4 *>(w_s + (size_t)rr * ldp)[c]);
      reinterpret_cast<float4 *>(dh_s + (size_t)b * ld)[c] = a;
    }
    // B: gradient row of each unit (batch order) and its update; one thread per (unit, float4)
    for (int i = threadIdx.x; i < R * c4n; i += kBlock) {
      const int r = i / c4n, c = i % c4n, n = tile * R + r;
      if (n >= P.n_items) continue;
      float4 g = f4_zero();
#pragma unroll 8
      for (int b = 0; b < B; ++b) f4_fma(g, dz_s[b * R + r], reinterpret_cast<const float4 *>(h_s + (size_t)b * ldp)[c]);
      float4 p = reinterpret_cast<const float4 *>(w_s + (size_t)r * ldp)[c];
      float4 *pw = reinterpret_cast<float4 *>(P.W2T + (size_t)n * ld) + c;
      float4 *p1 = reinterpret_cast<float4 *>(opt.s1[1] + (size_t)n * ld) + c;
      float4 *p2 = oW.kind == DRX_OPT_ADAM ? reinterpret_cast<float4 *>(opt.s2[1] + (size_t)n * ld) + c : nullptr;
      float4 m = *p1, v = p2 ? *p2 : f4_zero();
      reg_acc += f4_dot(p, p);
      opt_update1(oW, fmaf(oW.rb, p.x, g.x), p.x, m.x, v.x);
      opt_update1(oW, fmaf(oW.rb, p.y, g.y), p.y, m.y, v.y);
      opt_update1(oW, fmaf(oW.rb, p.z, g.z), p.z, m.z, v.z);
      opt_update1(oW, fmaf(oW.rb, p.w, g.w), p.w, m.w, v.w);
      *pw = p; *p1 = m;
      if (p2) *p2 = v;
    }
    if (threadIdx.x >= kBlock - R) {             // the last R threads (idle in B for every supported width): the unit's bias
      const int r = threadIdx.x - (kBlock - R), n = tile * R + r;
      if (n < P.n_items) {
        float gb2 = 0.f;
        for (int b = 0; b < B; ++b) gb2 += dz_s[b * R + r];
        float pb = bias_s[r], m = opt.s1[4][n], v = oB.kind == DRX_OPT_ADAM ? opt.s2[4][n] : 0.f;
        OptScalars ob = oB; ob.rb = 0.f;
        opt_update1(ob, gb2, pb, m, v);
        P.b2[n] = pb; opt.s1[4][n] = m;
        if (oB.kind == DRX_OPT_ADAM) opt.s2[4][n] = v;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B * c4n; i += kBlock)
    reinterpret_cast<float4 *>(A.dh_slab + (size_t)blockIdx.x * B * ld)[i] = reinterpret_cast<float4 *>(dh_s)[i];
  float lsum = block_sum(loss_acc, red);
  float rsum = block_sum(reg_acc, red);
  if (threadIdx.x == 0) { A.loss_part[blockIdx.x] = lsum * invBN; A.reg_part[blockIdx.x] = rsum; }
}

// dz1[b,:] = (sum_slabs dh) * h (1-h)      one workgroup per batch row, groups stride over slabs
template <int G, int J, int THREADS = kBlock>
__global__ __launch_bounds__(THREADS) void k_hidden_bwd(int ld, int B, int n_slabs, const float *__restrict__ slab,
                                                       const float *__restrict__ h, float *__restrict__ dz1) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = THREADS / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  for (int s = r; s < n_slabs; s += 4 * R) {            // 4 independent slab rows in flight, folded in slab order
    float4 v[4][J];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < J; ++j) v[u][j] = f4_zero();
      if (s + u * R < n_slabs) load_row<G, J>(slab, (size_t)(s + u * R) * B + b, ld, lane, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(acc[j], v[u][j]);
  }
  store_row<G, J>(lds, (size_t)r, ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 t[J], hv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
#pragma unroll 8
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
struct DensePrefetch {          // n16 16-byte words from src (pinned host memory) to dst (device), or n16 == 0
  const uint4 *src;
  uint4 *dst;
  size_t n16;
};

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_in_sweep(DrxCdaeParams P, DrxOptim opt, int B, float scale, DenseAux aux,
                                                     const float *__restrict__ dz1, float *reg_part, DensePrefetch pf) {
  __shared__ float red[kBlock / 64];
  const int lane = threadIdx.x % G;
  const int gpb = kBlock / G;
  const int ld = P.ld;
  float reg_acc = 0.f;
  if (blockIdx.x == gridDim.x - 1) {   // hidden bias: g = sum_b dz1[b,:]   (no L2 on biases, cdae.py:82)
    // the next batch, if the caller has it: from its pinned staging slot into device memory while the sweep runs (16-byte words)
    for (size_t i = threadIdx.x; i < pf.n16; i += kBlock) pf.dst[i] = pf.src[i];
    // every group sums the rows b = group, group + gpb, ...; the first group adds the partial sums up in group order
    __shared__ float4 part[kBlock / G][G * J];
    const int grp = threadIdx.x / G;
    float4 g[J], w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = f4_zero();
    for (int b = grp; b < B; b += gpb) {
      float4 v[J];
      load_row<G, J>(dz1, (size_t)b, ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) part[grp][j * G + lane] = g[j];
    __syncthreads();
    if (threadIdx.x < G) {
      for (int q = 1; q < gpb; ++q) {
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(g[j], part[q][j * G + lane]);
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
    uint32_t *mask = isW ? aux.km + rr * aux.Bw : aux.vm + rr * aux.Bw;
    float4 g[J], w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = f4_zero();
    for (int wd = 0; wd < aux.Bw; ++wd) {
      uint32_t m = mask[wd];
      if (m && lane == 0) mask[wd] = 0;            // consumed (all lanes of the group loaded it with the same instruction): clean for the next step
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
  float *phead, *ptail;                       // [n_chunks, ld]
  float *phs, *pts;                           // [n_chunks] scalar (b2) partials
  float *pblock, *pbs;                        // [n_blocks, ld], [n_blocks]: partials of all-inner workgroups (k_seg_reduce_planned)
  float *hot_part, *hot_ps;                   // [kMaxHot * kHotTiles, ld], [kMaxHot * kHotTiles]: (hot segment, sample tile) partials
  float *bpart;                               // [n_bpart, ld]
  const uint8_t *solo_v, *solo_o;             // [B] each or nullptr: sample b is the ONLY toucher of its V / W2T row; solo_v + 2B:
                                              //   [B] sample b holds at least one W row that only it touches
  const uint32_t *solo_w;                     // [ceil(N/32)] or nullptr: bit n set = W row n is touched by ONE sample of the batch
  unsigned long long *stamps;                 // diagnostic builds (DRX_STAMPS) only
  const int32_t *order;                       // [B] or nullptr: launch order of the forward kernel's triples (k_order_by_degree)
  int T, n_chunks, n_bpart;
};

// The V keys of a touch list are 2N + user id: with ten million users that is a 24-bit key space and a THIRD pass of the radix sort
// for the sake of 65 536 of 1.4 M touches.  A batch holds at most B distinct users, so they are numbered by their SLOT in an
// open-addressing table of 4B entries (atomicCAS; cleared per batch): V key = 2N + slot, 22 bits at N = 10^6 — two 11-bit passes.
// The table travels with the prepared list (the reduction turns a slot back into a user with one load per V segment).  Used when
// n_users exceeds the table; smaller user sets keep their ids.
__device__ __forceinline__ uint32_t user_hash(uint32_t u) { u *= 0x9E3779B1u; return u ^ (u >> 15); }

__global__ __launch_bounds__(kBlock) void k_user_slots(const int32_t *__restrict__ uid, int B, uint32_t *__restrict__ vtab, uint32_t vt_mask,
                                                       uint32_t *__restrict__ vslot) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const uint32_t u = (uint32_t)uid[b];
  uint32_t h = user_hash(u) & vt_mask;
  for (;;) {
    const uint32_t old = atomicCAS(&vtab[h], 0xFFFFFFFFu, u);
    if (old == 0xFFFFFFFFu || old == u) break;
    h = (h + 1) & vt_mask;
  }
  vslot[b] = h;
}

// Touch list of one batch (row key, sample): depends only on the batch, never on the parameters, so it can be built
// and sorted for batch t+1 while batch t trains (drx_cdae_sparse_prepare on a second stream).
// Also clears the sole-toucher marks of the batch (solo: [2B] bytes, solo_w: one bit per item; or nullptr) and pads the slots beyond
// the last sample's up to T with DRX_KEY_NONE (n_touch_slots may be an upper bound) — memsets the preparation would otherwise launch.
__global__ __launch_bounds__(kBlock) void k_sparse_touches(int n_items, DrxHistory H, DrxBatch bt, uint32_t qthr, uint32_t *keys,
                                                           uint32_t *vals, int T, uint8_t *solo, uint32_t *solo_w, uint32_t *zero_a,
                                                           int n_zero_a, uint32_t *zero_b, int n_zero_b, const uint32_t *__restrict__ vslot,
                                                           uint32_t *zero_c, int n_zero_c) {
  constexpr int G = 16;
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (solo_w)
    for (int w = blockIdx.x * kBlock + threadIdx.x; w < (n_items + 31) / 32; w += gridDim.x * kBlock) solo_w[w] = 0u;
  // (two more ranges of words the preparation wants zeroed: the span plan's counters + window bytes, the degree-order work area)
  // (word 2 of the first range = "the V keys are table slots")
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_a; w += gridDim.x * kBlock) zero_a[w] = (w == 2 && vslot) ? 1u : 0u;
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_b; w += gridDim.x * kBlock) zero_b[w] = 0u;
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < n_zero_c; w += gridDim.x * kBlock) zero_c[w] = 0u;      // (the sort's counters and tile words)
  if (b >= bt.B) return;
  if (solo && lane == 0) { solo[b] = 0; solo[bt.B + b] = 0; solo[2 * (size_t)bt.B + b] = 0; }
  if (b == bt.B - 1)
    for (int j = bt.keep_off[bt.B] + 2 * bt.B + lane; j < T; j += G) { keys[j] = DRX_KEY_NONE; vals[j] = 0; }
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const int base = bt.keep_off[b] + 2 * b;
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  for (int64_t j = s + lane; j < e; j += G) {
    const uint32_t jj = (uint32_t)(j - s);
    const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, jj) >= qthr);
    keys[base + jj] = kf ? (uint32_t)H.indices[j] : DRX_KEY_NONE;
    vals[base + jj] = (uint32_t)b;
  }
  if (lane == 0) {
    const int deg = (int)(e - s);
    keys[base + deg] = (uint32_t)(n_items + bt.iid[b]);       vals[base + deg] = (uint32_t)b;
    keys[base + deg + 1] = (uint32_t)(2 * n_items) + (vslot ? vslot[b] : (uint32_t)u);       vals[base + deg + 1] = (uint32_t)b;
  }
}

template <int G, int J, int KIND = -1>
__device__ __forceinline__ void sparse_apply(const DrxCdaeParams &P, const DrxOptim &opt, int B, uint32_t key, int lane,
                                             const float4 (&g)[J], float gs) {
  const uint32_t N = (uint32_t)P.n_items;
  // read every candidate pointer as a scalar first, then select VALUES (a dynamic index into the kernarg pointer
  // arrays would become a vector load + s_waitcnt vmcnt(0))
  float *const tW = P.W, *const tO = P.W2T, *const tV = P.V;
  float *const a0 = opt.s1[0], *const a1 = opt.s1[1], *const a2 = opt.s1[2];
  float *const c0 = opt.s2[0], *const c1 = opt.s2[1], *const c2 = opt.s2[2];
  const int var = key < N ? 0 : (key < 2 * N ? 1 : 2);
  const size_t row = key - (uint32_t)var * N;
  float *const tab = var == 0 ? tW : (var == 1 ? tO : tV);
  float *const s1 = var == 0 ? a0 : (var == 1 ? a1 : a2);
  float *const s2 = var == 0 ? c0 : (var == 1 ? c1 : c2);
  OptScalars o = opt_for(opt, 0, B);
  o.inv_k = 1.0f / (float)P.k;
  float4 w[J];
  load_row<G, J>(tab, row, P.ld, lane, w);
  row_update<G, J, KIND>(o, tab, s1, s2, row, P.ld, lane, w, g);
  if (var == 1 && lane == 0) {
    const int kind = KIND >= 0 ? KIND : o.kind;
    float pb = P.b2[row], m = opt.s1[4][row], v = kind == DRX_OPT_ADAM ? opt.s2[4][row] : 0.f;
    o.rb = 0.f;
    opt_update1<KIND>(o, gs, pb, m, v);
    P.b2[row] = pb; opt.s1[4][row] = m;
    if (kind == DRX_OPT_ADAM) opt.s2[4][row] = v;
  }
}

// Launch order of the forward kernel's triples: longest histories first, triples of similar length side by side (r03 phase stamps:
// a triple lives 19 us on average but 25 us at the 90th percentile and far longer for the few users with hundreds of items; a
// workgroup waits for its slowest triple and the launch for its last workgroups — 47 % of the chip's group slots were occupied on
// average).  A counting sort by history length in 256 buckets of 4 items, two small launches on the preparation's stream (counts;
// scatter), each workgroup over 1024 triples; `work` = 512 zeroed ints (k_sparse_touches clears them).  A first version did it all
// in ONE workgroup: 152 us of side-stream time per step, which made the preparation — not the training — the pipeline's bound.
// The order inside a bucket comes from atomics and differs from run to run: it decides only WHERE a triple is computed, never a result.
// (degree_bucket / same_bucket_lanes / degree_counts_body: drx_common.hpp — the counts can ride in the sort's first launch)
__global__ __launch_bounds__(1024) void k_degree_counts(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work) {
  __shared__ unsigned int cnt[256];
  degree_counts_body<1024>(keep_off, B, work, (int)blockIdx.x, cnt);
}

template <int NT>
__device__ __forceinline__ void degree_scatter_body(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work,
                                                    int32_t *__restrict__ order, int block_id, unsigned int *lds /* [768] */) {
  unsigned int *start = lds, *cnt = lds + 256, *base = lds + 512;
  for (int i = threadIdx.x; i < 256; i += NT) cnt[i] = 0;
  if (threadIdx.x < 64) {                            // exclusive scan of the 256 global counts by one wave: 4 bins per lane
    unsigned int c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c[q] = work[threadIdx.x * 4 + q]; sum += c[q]; }
    unsigned int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
    unsigned int run = inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) { start[threadIdx.x * 4 + q] = run; run += c[q]; }
  }
  __syncthreads();
  const int b = block_id * NT + (int)threadIdx.x, lane = threadIdx.x & 63;
  const bool valid = b < B;
  const int d = valid ? degree_bucket(keep_off, b) : 0;
  const unsigned long long m = same_bucket_lanes(valid, d);
  const int leader = valid ? __ffsll((long long)m) - 1 : lane;
  unsigned int at = 0;
  if (valid && lane == leader) at = atomicAdd(&cnt[d], (unsigned int)__popcll(m));        // place inside this workgroup's share
  at = __shfl(at, leader);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += NT) base[i] = cnt[i] ? atomicAdd(&work[256 + i], cnt[i]) : 0u;     // the share's place in the bucket
  __syncthreads();
  if (valid) order[start[d] + base[d] + at + __popcll(m & ((1ull << lane) - 1ull))] = b;
}

__global__ __launch_bounds__(1024) void k_degree_scatter(const int32_t *__restrict__ keep_off, int B, unsigned int *__restrict__ work,
                                                         int32_t *__restrict__ order) {
  __shared__ unsigned int lds[768];
  degree_scatter_body<1024>(keep_off, B, work, order, (int)blockIdx.x, lds);
}

// V and W2T rows are mostly touched by ONE sample of the batch (a user is drawn once, output items are uniform), and so are the W
// rows of the long tail of unpopular items (10M x 1M set, B = 65 536: 2/3 of the distinct W rows of a batch, 1/8 of the W touches).
// When the touch list is prepared ahead of the step, such rows are marked here: the forward/backward kernel, which holds the
// sample's gradient rows in registers, then applies their update itself (no g2 row written, no re-read of the gradient
// and of the parameter row later), and the touch is blanked (DRX_KEY_NONE) so that the segmented reduction passes over
// it.  A sole toucher cannot race: no other sample of the batch reads or writes that row.  V / W2T marks are a byte per sample;
// a W mark is a bit per ITEM (solo_w; nullptr = W rows are not marked) plus a byte per sample "holds a marked item" (solo_v + 2B):
// such a sample walks its history a second time and finds the item by its bit (solo_w_pass).
__global__ void k_mark_solo(uint32_t *keys_s, const uint32_t *__restrict__ vals_s, int T, uint32_t n_items, int B, uint8_t *solo_v,
                            uint8_t *solo_o, uint32_t *solo_w) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    const uint32_t k = keys_s[j];
    if (k == DRX_KEY_NONE || (k < n_items && !solo_w)) continue;
    const uint32_t prev = j > 0 ? keys_s[j - 1] : DRX_KEY_NONE, next = j + 1 < T ? keys_s[j + 1] : DRX_KEY_NONE;
    if (k == prev || k == next) continue;            // (a neighbour blanked concurrently was a different key anyway)
    const uint32_t b = vals_s[j];
    if (k < n_items) { atomicOr(&solo_w[k >> 5], 1u << (k & 31)); solo_v[2 * (size_t)B + b] = 1; }
    else if (k < 2 * n_items) solo_o[b] = 1;
    else solo_v[b] = 1;
    keys_s[j] = DRX_KEY_NONE;
  }
}

// The span plan and the sole-toucher marks in ONE launch over the freshly sorted list (two launches cost the preparation — the
// pipeline's bound once the training kernels got faster — a launch gap and 15 us): the first n_chunks threads plan their chunk, then
// every thread marks its share of the touches.  The two do not disturb each other: a key that is blanked has one touch, a key whose
// run the plan measures crosses a chunk border (>= 2 touches), and the plan's searches only test keys for equality with such a key.
// order_blocks workgroups behind the plan's: the scatter half of the launch order (k_degree_scatter's work, 256 triples each; the
// counts were taken before the sort) — one launch less for the preparation's stream to wait for.
__global__ __launch_bounds__(256) void k_plan_and_mark(uint32_t *keys_s, const uint32_t *__restrict__ vals_s, int T, int n_chunks, int cpb,
                                                       SpanPlan P, uint32_t n_items, int B, uint8_t *solo_v, uint8_t *solo_o,
                                                       uint32_t *solo_w, int order_blocks, const int32_t *keep_off, unsigned int *order_work,
                                                       int32_t *order) {
  const int plan_blocks = (int)gridDim.x - order_blocks;
  if ((int)blockIdx.x >= plan_blocks) {
    __shared__ unsigned int lds[768];
    degree_scatter_body<256>(keep_off, B, order_work, order, (int)blockIdx.x - plan_blocks, lds);
    return;
  }
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < n_chunks) plan_chunk(keys_s, T, n_chunks, cpb, P, tid);
  for (int j = tid; j < T; j += plan_blocks * blockDim.x) {
    const uint32_t k = keys_s[j];
    if (k == DRX_KEY_NONE || (k < n_items && !solo_w)) continue;
    const uint32_t prev = j > 0 ? keys_s[j - 1] : DRX_KEY_NONE, next = j + 1 < T ? keys_s[j + 1] : DRX_KEY_NONE;
    if (k == prev || k == next) continue;
    const uint32_t b = vals_s[j];
    if (k < n_items) { atomicOr(&solo_w[k >> 5], 1u << (k & 31)); solo_v[2 * (size_t)B + b] = 1; }
    else if (k < 2 * n_items) solo_o[b] = 1;
    else solo_v[b] = 1;
    keys_s[j] = DRX_KEY_NONE;
  }
}

// Second walk over a triple's history, after its gradient row dz1 is known: the kept items whose W row carries a sole-toucher
// mark get their sparse update here, from registers (gradient scale * dz1[b], exactly the one-touch segment the reduction would have
// folded: same arithmetic, same bits).  The indices come from L1/L2 (the gather read them a moment ago), the mark words from the
// 125 KB bitmap; per marked row: parameter row (just gathered: L2) + slot row in, both out.
template <int G, int J, int KIND = -1>
__device__ __forceinline__ void solo_w_pass(const DrxCdaeParams &P, const DrxOptim &opt, const DrxHistory &H, const DrxBatch &bt,
                                            uint32_t qthr, float scale, const uint32_t *__restrict__ solo_w, int b, int lane,
                                            const float4 (&dz1)[J]) {
  const int gshift = (threadIdx.x & 63) / G * G;
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  float4 g[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { g[j] = f4_zero(); f4_fma(g[j], scale, dz1[j]); }
  // (b laundered: the walk re-reads uid / indptr instead of keeping the gather's copies alive in registers through the whole kernel —
  // 65 instead of 62 VGPRs would cost the forward kernel its eighth wave per SIMD)
  int bq = b;
  asm volatile("" : "+v"(bq));
  const int u = bt.uid[bq];
  const int64_t s = H.indptr[u];
  const int n = (int)(H.indptr[u + 1] - s);
  const int32_t *const ind = H.indices + s;
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  constexpr int IPL = G >= 16 ? 1 : 16 / G;
  constexpr int CH = G * IPL;
  for (int c = 0; c < n; c += CH) {
    int idx[IPL], sw[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int jj = c + r * G + lane;
      idx[r] = 0; sw[r] = 0;
      if (jj < n) {
        idx[r] = ind[jj];
        const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, (uint32_t)jj) >= qthr);
        if (kf) sw[r] = (int)((solo_w[idx[r] >> 5] >> (idx[r] & 31)) & 1u);
      }
    }
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      unsigned long long m = (__ballot(sw[r] != 0) >> gshift) & gmask;
      while (m) {
        const int t = __builtin_ctzll(m);
        m &= m - 1;
        sparse_apply<G, J, KIND>(P, opt, bt.B, (uint32_t)__shfl(idx[r], t, G), lane, g, 0.f);
      }
    }
  }
}

// One triple after its input bag is known, in two halves.  sampled_hidden: hidden layer and the dot product with the
// sampled output row (on the columns this table holds).  sampled_rest: loss, backward, gradient rows (and the in-place update
// of rows only this triple touches) from the COMPLETE dot product — the same value in the single-GPU step, the sum over ranks
// of the partial dots in the column-sharded one.
template <int G, int J>
__device__ __forceinline__ float sampled_hidden(const DrxCdaeParams &P, const DrxBatch &bt, float scale, int b, int lane,
                                                const float4 (&acc)[J], float4 (&h)[J], float4 (&w2)[J]) {
  hidden_act<G, J>(P, bt.uid[b], scale, lane, acc, h);
  load_row<G, J>(P.W2T, (size_t)bt.iid[b], P.ld, lane, w2);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) d += f4_dot(w2[j], h[j]);
  return group_sum<G>(d);
}

template <int G, int J, int KIND = -1>
__device__ __forceinline__ void sampled_rest(const DrxCdaeParams &P, const DrxOptim &opt, const DrxHistory &H, const DrxBatch &bt,
                                             float scale, uint32_t qthr, int loss_kind, const SparseBufs &S, int b, int lane,
                                             float d, const float4 (&h)[J], const float4 (&w2)[J],
                                             float4 (*dz1_out)[J] = nullptr) {
  const int u = bt.uid[b], i = bt.iid[b];
  const float y = bt.y[b];
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
  if (lane == 0) S.lossb[b] = lval;
  const uint8_t *const pv = S.solo_v, *const po = S.solo_o;
  const bool solo_v = pv && pv[b], solo_o = po && po[b];
  if (solo_o) {      // this sample alone touches W2T[i] and b2[i]: update them here (same arithmetic as the segment path)
    sparse_apply<G, J, KIND>(P, opt, bt.B, (uint32_t)P.n_items + (uint32_t)i, lane, g2, dz2);
  } else {
    store_row<G, J>(S.g2, (size_t)b, P.ld, lane, g2);
    if (lane == 0) S.dz2[b] = dz2;
  }
  if (solo_v) sparse_apply<G, J, KIND>(P, opt, bt.B, 2u * (uint32_t)P.n_items + (uint32_t)u, lane, dz1, 0.f);
  const uint32_t *const pw = S.solo_w;
  if (pw && pv[2 * (size_t)bt.B + b]) solo_w_pass<G, J, KIND>(P, opt, H, bt, qthr, scale, pw, b, lane, dz1);
  if (dz1_out) {
#pragma unroll
    for (int j = 0; j < J; ++j) (*dz1_out)[j] = dz1[j];
  }
}

template <int G, int J, int KIND = -1>
__device__ __forceinline__ void sampled_finish(const DrxCdaeParams &P, const DrxOptim &opt, const DrxHistory &H, const DrxBatch &bt,
                                               float scale, uint32_t qthr, int loss_kind, const SparseBufs &S, int b, int lane,
                                               const float4 (&acc)[J]) {
  float4 h[J], w2[J];
  const float d = sampled_hidden<G, J>(P, bt, scale, b, lane, acc, h, w2);
  sampled_rest<G, J, KIND>(P, opt, H, bt, scale, qthr, loss_kind, S, b, lane, d, h, w2);
}

// ---- column-sharded ("K-sharded") step: the two halves as kernels of their own, the all-reduce of dot[] between them ----------
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_kshard_fwd(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale, uint32_t qthr,
                                                       float *__restrict__ h_out, float *__restrict__ dot_out,
                                                       const int32_t *__restrict__ order) {
  const int lane = threadIdx.x % G;
  const int slot = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (slot >= bt.B) return;
  const int b = order ? order[slot] : slot;        // longest histories first, similar lengths side by side (k_degree_counts)
  float4 acc[J], h[J], w2[J];
  DenseAux none{};
  gather_bag<G, J, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0);
  const float d = sampled_hidden<G, J>(P, bt, scale, b, lane, acc, h, w2);
  store_row<G, J>(h_out, (size_t)b, P.ld, lane, h);
  if (lane == 0) dot_out[b] = d;
}

// (long histories / small batches: one workgroup per triple, like k_sampled_fwd_bwd_wg)
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_kshard_fwd_wg(DrxCdaeParams P, DrxHistory H, DrxBatch bt, float scale, uint32_t qthr,
                                                          float *__restrict__ h_out, float *__restrict__ dot_out) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x;
  float4 acc[J], h[J], w2[J];
  DenseAux none{};
  gather_bag<G, J, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0, r, R);
  store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
  __syncthreads();
  if (r != 0) return;
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
#pragma unroll 8
  for (int rr = 0; rr < R; ++rr) {
    float4 v[J];
    load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  const float d = sampled_hidden<G, J>(P, bt, scale, b, lane, acc, h, w2);
  store_row<G, J>(h_out, (size_t)b, P.ld, lane, h);
  if (lane == 0) dot_out[b] = d;
}

template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_kshard_rest(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                        uint32_t qthr, int loss_kind, SparseBufs S,
                                                        const float *__restrict__ h_in, const float *__restrict__ dot_total) {
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  float4 h[J], w2[J];
  load_row<G, J>(h_in, (size_t)b, P.ld, lane, h);
  load_row<G, J>(P.W2T, (size_t)bt.iid[b], P.ld, lane, w2);
  sampled_rest<G, J>(P, opt, H, bt, scale, qthr, loss_kind, S, b, lane, dot_total[b], h, w2);
}

template <int G, int J, int KIND = -1>
__global__ __launch_bounds__(kBlock) void k_sampled_fwd_bwd(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                            uint32_t qthr, int loss_kind, SparseBufs S) {
  const int lane = threadIdx.x % G;
  const int slot = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (slot >= bt.B) return;
  const int32_t *const ord = S.order;
  const int b = ord ? ord[slot] : slot;            // longest histories first, similar lengths side by side (k_degree_counts)
  float4 acc[J];
  DenseAux none{};
  gather_bag<G, J, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0);
  sampled_finish<G, J, KIND>(P, opt, H, bt, scale, qthr, loss_kind, S, b, lane, acc);
}

// The forward/backward kernel with sole-toucher W rows (DRX_BATCH_MARK_W; Adagrad).  A triple that holds marked rows (byte mark)
// fetches, beside the rows of its gather, the items' bits of the mark bitmap and notes the marked items in LDS.  Right after the
// gather — when the 8 row buffers of the loop are free — the parameter rows (again: they are in L2) and accumulator rows of those
// items are requested; they travel in the shadow of the hidden layer's own loads.  At the end the update is arithmetic on registers
// and two stores per row: no second walk, no extra round trip in the triple's dependent chain (r03a measured the second walk at +120 us for the 10M x 1M batch; the reduction it relieves runs 50 us shorter).  A triple with
// more than kStash marked rows (rare) takes the walk (solo_w_pass) for all of them.
constexpr int kStash = 4;

template <int G, int J>
__device__ __forceinline__ void adagrad_commit(const OptScalars &o, float *tab, float *s1, size_t row, int ld, int lane,
                                               const float4 (&w)[J], const float4 (&a)[J], const float4 (&g)[J]) {
  float4 *pr = reinterpret_cast<float4 *>(tab + row * (size_t)ld), *ar = reinterpret_cast<float4 *>(s1 + row * (size_t)ld);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = lane + j * G;
    if (4 * c < ld) {
      float4 p = w[j], m = a[j];
      float unused = 0.f;
      opt_update1<DRX_OPT_ADAGRAD>(o, fmaf(o.rb, p.x, g[j].x), p.x, m.x, unused);
      opt_update1<DRX_OPT_ADAGRAD>(o, fmaf(o.rb, p.y, g[j].y), p.y, m.y, unused);
      opt_update1<DRX_OPT_ADAGRAD>(o, fmaf(o.rb, p.z, g[j].z), p.z, m.z, unused);
      opt_update1<DRX_OPT_ADAGRAD>(o, fmaf(o.rb, p.w, g[j].w), p.w, m.w, unused);
      pr[c] = p;
      ar[c] = m;
    }
  }
}

#ifdef DRX_STASH_W8
#define DRX_STASH_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define DRX_STASH_ATTR
#endif
template <int G, int J>      // J == 1 only (rows of <= 256 floats: a row is one 16-byte piece per lane)
__global__ __launch_bounds__(kBlock) DRX_STASH_ATTR void k_sampled_fwd_bwd_stash(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                                  uint32_t qthr, int loss_kind, SparseBufs S) {
  __shared__ int ids_lds[(kBlock / G) * (kStash + 1)];   // per group: how many W rows this triple alone touches, and which items
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x * (kBlock / G) + r;
  if (b >= bt.B) return;
  const int gshift = (threadIdx.x & 63) / G * G;
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const uint32_t *const pw = S.solo_w;
  const uint8_t *const pv = S.solo_v;
  float4 acc[J];
  DenseAux none{};
  gather_bag<G, J, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0);
  // A triple that holds marked rows (byte mark) walks its history once more right away — indices from L1/L2, mark bits from the
  // 125 KB bitmap — and notes the marked items in LDS, in history order.  (Done inside the gather loop this cost the kernel its
  // eighth wave per SIMD: 69 VGPRs; b is laundered so that the gather's copies of uid / indptr are not kept alive for it.)
  int cnt = 0;
  if (pv[2 * (size_t)bt.B + b]) {
    int bq = b;
    asm volatile("" : "+v"(bq));
    const int u = bt.uid[bq];
    const int64_t s = H.indptr[u];
    const int n = (int)(H.indptr[u + 1] - s);
    const int32_t *const ind = H.indices + s;
    const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[bq] : nullptr;
    for (int c = 0; c < n; c += G) {
      const int jj = c + lane;
      int item = 0;
      bool mk = false;
      if (jj < n) {
        item = ind[jj];
        const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)bq, (uint32_t)jj) >= qthr);
        mk = kf && ((pw[item >> 5] >> (item & 31)) & 1u);
      }
      const unsigned long long m = (__ballot(mk) >> gshift) & gmask;
      if (m) {
        const int at = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (mk && at < kStash) ids_lds[r * (kStash + 1) + 1 + at] = item;
        cnt += __popcll(m);
      }
    }
  }
  // accumulator rows of the marked items (HBM misses: the slow half of their update), requested NOW and straight into LDS
  // (global_load_lds: no register is held while they travel in the shadow of the hidden layer's own loads).  One wave-instruction
  // writes 64 lanes x 16 B = the rows of the wave's 64 / G groups side by side; slot q of the wave lives at q KiB of its region.
  extern __shared__ __align__(16) float slot_lds[];          // [4 waves][kStash][64 lanes x 4 floats]
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
  float *const wave_slots = slot_lds + (size_t)wave * kStash * 256;
  if (lane == 0) ids_lds[r * (kStash + 1)] = cnt;          // (kept in LDS, not in a register, across the rest of the kernel: 64 VGPRs)
  if (cnt > 0 && cnt <= kStash) {
    wave_lds_sync();                            // (ids written by some lanes of the group are read by all of them)
    const float *const a0 = opt.s1[0];
#pragma unroll
    for (int q = 0; q < kStash; ++q) {
      if (q < cnt && 4 * lane < P.ld) {
        const float *src = a0 + (size_t)ids_lds[r * (kStash + 1) + 1 + q] * P.ld + 4 * lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(wave_slots + q * 256), 16, 0, 0);
      }
    }
  }
  float4 h[J], w2[J];
  const float d = sampled_hidden<G, J>(P, bt, scale, b, lane, acc, h, w2);
  SparseBufs S2 = S;                            // (sampled_rest with the W walk switched off: the marks are served below)
  S2.solo_w = nullptr;
  float4 dz1[J];
  sampled_rest<G, J, DRX_OPT_ADAGRAD>(P, opt, H, bt, scale, qthr, loss_kind, S2, b, lane, d, h, w2, &dz1);
  wave_lds_sync();
  const int cnt2 = ids_lds[(threadIdx.x / G) * (kStash + 1)];
  const int *const my_ids = ids_lds + (threadIdx.x / G) * (kStash + 1) + 1;
  if (cnt2 > 0 && cnt2 <= kStash) {
    float4 g[J];
#pragma unroll
    for (int jx = 0; jx < J; ++jx) { g[jx] = f4_zero(); f4_fma(g[jx], scale, dz1[jx]); }
    const OptScalars o = opt_for(opt, 0, bt.B);
    float *const tW = P.W, *const a0 = opt.s1[0];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's LDS-DMA has landed (nothing else orders a ds_read behind it)
#pragma unroll
    for (int q0 = 0; q0 < kStash; q0 += 2) {
      float4 prow[2][J];                         // the parameter rows again: the gather had them a moment ago, L2 still does
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        prow[q][0] = f4_zero();
        if (q0 + q < cnt2) load_row<G, J>(tW, (size_t)my_ids[q0 + q], P.ld, lane, prow[q]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q0 + q < cnt2) {
          float4 slot[J];
          slot[0] = *reinterpret_cast<const float4 *>(wave_slots + (q0 + q) * 256 + 4 * (threadIdx.x % 64));
          adagrad_commit<G, J>(o, tW, a0, (size_t)my_ids[q0 + q], P.ld, lane, prow[q], slot, g);
        }
      }
    }
  } else if (cnt2 > kStash) {
    solo_w_pass<G, J, DRX_OPT_ADAGRAD>(P, opt, H, bt, qthr, scale, pw, b, lane, dz1);
  }
}

// The forward/backward kernel with its late rows PREFETCHED INTO LDS (Adagrad, one float4 per lane; r03).  One triple is a chain of
// dependent round trips: uid -> indptr -> indices -> rows (two rounds) -> V[u], b -> W2T[i] -> b2[i], y -> marks -> slot rows of the
// sole-toucher updates, and with 16 such chains per SIMD the kernel is bound by their LENGTH (r02: 69 % of the wave-cycles in
// s_waitcnt at 3.6 TB/s).  Everything behind the gather depends on (u, i) alone, but fetching it early into registers cost two waves
// per SIMD (r02e: 90 VGPRs, no gain).  global_load_lds needs no register: V[u], W2T[i] and — where the marks say this triple is their
// only toucher — the two accumulator rows are requested as soon as uid / iid are known, land in LDS while the gather runs, and are
// read back with ds_read: the chain behind the gather shrinks to one round trip (b, b2[i], y: L2 hits) and the stores.
// One wave-instruction writes 64 lanes x 16 B: the rows of the wave's 64 / G groups side by side; slot s of a wave is at s KiB.
template <int G>
__device__ __forceinline__ void glds_row(const float *tab, size_t row, int ld, int lane, float *lds_slot) {
  if (4 * lane < ld)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tab + row * (size_t)ld + 4 * lane),
                                     (__attribute__((address_space(3))) void *)lds_slot, 16, 0, 0);
}

template <int G>
__global__ __launch_bounds__(kBlock) void k_sampled_fwd_bwd_pf(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                               uint32_t qthr, int loss_kind, SparseBufs S) {
  extern __shared__ __align__(16) float pf_lds[];            // [kBlock / 64 waves][4 slots][64 lanes x 4 floats]
  const int lane = threadIdx.x % G;
  const int slot = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (slot >= bt.B) return;
  const int32_t *const ord = S.order;
  const int b = ord ? ord[slot] : slot;            // longest histories first, similar lengths side by side
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
  float *const ws = pf_lds + (size_t)wave * 4 * 256;
  const int wl4 = 4 * (int)(threadIdx.x % 64);
  const bool act = 4 * lane < P.ld;
  DRX_STAMP(S.stamps, b, 0, lane);
  {
    // (the two accumulator rows are requested whatever the marks say — nine triples in ten are the only toucher of their V and
    // W2T rows — so that nothing here waits for a mark byte before the gather's own chain starts)
    const int u = bt.uid[b], i = bt.iid[b];
    glds_row<G>(P.V, (size_t)u, P.ld, lane, ws);
    glds_row<G>(P.W2T, (size_t)i, P.ld, lane, ws + 256);
    if (S.solo_v) {
      glds_row<G>(opt.s1[2], (size_t)u, P.ld, lane, ws + 512);
      glds_row<G>(opt.s1[1], (size_t)i, P.ld, lane, ws + 768);
    }
  }
  DRX_STAMP(S.stamps, b, 1, lane);
  float4 acc[1];
  DenseAux none{};
  gather_bag<G, 1, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0, 0, 1, S.stamps);
  DRX_STAMP(S.stamps, b, 5, lane);
  // (b laundered: ids and marks are re-read — L1 hits — instead of living in registers through the gather)
  int bq = b;
  asm volatile("" : "+v"(bq));
  const int u = bt.uid[bq], i = bt.iid[bq];
  const uint8_t *const pv = S.solo_v, *const po = S.solo_o;
  const bool solo_v = pv && pv[bq], solo_o = po && po[bq];
  float4 bb[1];
  load_row<G, 1>(P.b, 0, P.ld, lane, bb);
  const float y = bt.y[bq], pb2 = P.b2[i];
  const float mb2 = solo_o ? opt.s1[4][i] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's LDS-DMA has landed (nothing else orders a ds_read behind it)
  DRX_STAMP(S.stamps, b, 6, lane);
  const float4 zero = f4_zero();
  const float4 v = act ? *reinterpret_cast<const float4 *>(ws + wl4) : zero;
  const float4 w2 = act ? *reinterpret_cast<const float4 *>(ws + 256 + wl4) : zero;
  float4 h;
  {
    const int col = 4 * lane;
    h.x = colmask(col + 0, P.k, sigmoidf_(fmaf(scale, acc[0].x, v.x + bb[0].x)));
    h.y = colmask(col + 1, P.k, sigmoidf_(fmaf(scale, acc[0].y, v.y + bb[0].y)));
    h.z = colmask(col + 2, P.k, sigmoidf_(fmaf(scale, acc[0].z, v.z + bb[0].z)));
    h.w = colmask(col + 3, P.k, sigmoidf_(fmaf(scale, acc[0].w, v.w + bb[0].w)));
  }
  const float d = group_sum<G>(f4_dot(w2, h));
  const float p = sigmoidf_(d + pb2);
  const float invB = 1.0f / (float)bt.B;
  float lval, dp;
  if (loss_kind == DRX_LOSS_BCE) { lval = bce_elem(y, p); dp = bce_grad(y, p) * invB; }
  else { lval = (p - y) * (p - y); dp = 2.0f * (p - y) * invB; }
  const float dz2 = dp * p * (1.0f - p);
  float4 dz1[1], g2[1];
  dz1[0].x = dz2 * w2.x * h.x * (1.0f - h.x); dz1[0].y = dz2 * w2.y * h.y * (1.0f - h.y);
  dz1[0].z = dz2 * w2.z * h.z * (1.0f - h.z); dz1[0].w = dz2 * w2.w * h.w * (1.0f - h.w);
  g2[0].x = dz2 * h.x; g2[0].y = dz2 * h.y; g2[0].z = dz2 * h.z; g2[0].w = dz2 * h.w;
  store_row<G, 1>(S.dz1, (size_t)b, P.ld, lane, dz1);
  if (lane == 0) S.lossb[b] = lval;
  DRX_STAMP(S.stamps, b, 7, lane);
  OptScalars o = opt_for(opt, 0, bt.B);
  if (solo_o) {      // this triple alone touches W2T[i] and b2[i] (same arithmetic as sparse_apply / the segment path)
    float4 w[1] = {w2}, a[1];
    a[0] = act ? *reinterpret_cast<const float4 *>(ws + 768 + wl4) : zero;
    adagrad_commit<G, 1>(o, P.W2T, opt.s1[1], (size_t)i, P.ld, lane, w, a, g2);
    if (lane == 0) {
      OptScalars ob = o;
      ob.rb = 0.f;
      float pb = pb2, m = mb2, unused = 0.f;
      opt_update1<DRX_OPT_ADAGRAD>(ob, dz2, pb, m, unused);
      P.b2[i] = pb; opt.s1[4][i] = m;
    }
  } else {
    store_row<G, 1>(S.g2, (size_t)b, P.ld, lane, g2);
    if (lane == 0) S.dz2[b] = dz2;
  }
  if (solo_v) {
    float4 w[1] = {v}, a[1];
    a[0] = act ? *reinterpret_cast<const float4 *>(ws + 512 + wl4) : zero;
    adagrad_commit<G, 1>(o, P.V, opt.s1[2], (size_t)u, P.ld, lane, w, a, dz1);
  }
  DRX_STAMP(S.stamps, b, 8, lane);
}

// Small batches of long histories (ml-1m: 155 items per user, B of a few thousand): with one group per triple the gather is
// a chain of ~20 dependent load rounds on a chip that is mostly idle (measured 143 us at B = 4096).  Here one WORKGROUP
// takes a triple: its 256/G groups split the history, the partial bags are summed in LDS in group order.
template <int G, int J>
__global__ __launch_bounds__(kBlock) void k_sampled_fwd_bwd_wg(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                               uint32_t qthr, int loss_kind, SparseBufs S) {
  extern __shared__ __align__(16) float lds[];   // [R, ld]
  constexpr int R = kBlock / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b = blockIdx.x;
  float4 acc[J];
  DenseAux none{};
  gather_bag<G, J, 0>(P, H, bt, qthr, b, lane, acc, none, nullptr, nullptr, 0, r, R);
  store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
  __syncthreads();
  if (r != 0) return;
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
#pragma unroll 8
  for (int rr = 0; rr < R; ++rr) {
    float4 v[J];
    load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
    for (int j = 0; j < J; ++j) f4_add(acc[j], v[j]);
  }
  sampled_finish<G, J>(P, opt, H, bt, scale, qthr, loss_kind, S, b, lane, acc);
}

// Policy of the single-GPU sparse step for the generic segmented reduction (drx_segreduce.hpp):
// key space [0,N) W rows (contribution dz1[b] * 1/(1-q)), [N,2N) W2T rows (g2[b], scalar dz2[b] for b2), [2N,2N+U) V rows.
template <int KIND>
struct DirectPolicyT {
  DrxCdaeParams P;
  DrxOptim opt;
  int B;
  float scale;
  // contribution rows: dz1 [B,ld] and g2 [B,ld]; g2 is addressed as dz1 + g2_off so that the choice between them is
  // a VALUE select (a select between the two kernarg pointer FIELDS makes hipcc fetch the pointer with a vector load
  // and an s_waitcnt vmcnt(0) in front of every row load, which serialises the loads: measured 0.26 -> 0.33 ms)
  const float *dz1;
  long long g2_off;
  const float *dz2;
  // V keys of this list: 2N + user id, or — cnt_flags[2] != 0 — 2N + the user's slot in the batch's user table vtab (see k_user_slots)
  const uint32_t *vtab, *cnt_flags;
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t key, uint32_t b, int lane, float4 (&row)[J], float &sc, float &coef) const {
    const uint32_t N = (uint32_t)P.n_items;
    const bool is_out = key >= N && key < 2 * N;
    load_row<G, J>(dz1 + (is_out ? g2_off : 0ll), (size_t)b, P.ld, lane, row);
    if (is_out) sc = dz2[b];
    coef = key < N ? scale : 1.0f;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int, int lane, const float4 (&g)[J], float gs) const {
    const uint32_t N2 = 2u * (uint32_t)P.n_items;
    if (key >= N2 && cnt_flags[2]) key = N2 + vtab[key - N2];          // slot -> user
    sparse_apply<G, J, KIND>(P, opt, B, key, lane, g, gs);
  }
};
using DirectPolicy = DirectPolicyT<-1>;                      // optimizer chosen at run time
using DirectPolicyAdagrad = DirectPolicyT<DRX_OPT_ADAGRAD>;  // the throughput configuration's optimizer, known at compile time

// hidden bias b: column sums of dz1 in two deterministic stages, then a dense optimizer update.  Both stages are ROLES of
// the two tail launches of the sparse step (k_sparse_tail_a / _b below): they share a launch with the span fix-ups, which
// they do not depend on.
struct BiasArgs {
  const float *dz1;       // [B, ld]
  float *part;            // [n_part, ld] column-sum partials, then [n_part] loss partials
  const float *lossb;     // [B]
  float *loss_out;        // nullptr: no loss wanted
  int B, n_part, rows_per_block;
};

template <int G, int J, int NT>
__device__ __forceinline__ void bias_partial_body(int ld, const BiasArgs &A, int block_id, float *lds /* [NT/G, ld] */, float *red) {
  constexpr int R = NT / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  const int b0 = block_id * A.rows_per_block, b1 = min(A.B, b0 + A.rows_per_block);
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  constexpr int NB = J == 1 ? 8 : 2;            // rows in flight per group (one at a time made these sums chains of dependent loads)
  for (int b = b0 + r; b < b1; b += NB * R) {
    float4 v[NB][J];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
#pragma unroll
      for (int j = 0; j < J; ++j) v[q][j] = f4_zero();
      if (b + q * R < b1) load_row<G, J>(A.dz1, (size_t)(b + q * R), ld, lane, v[q]);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(acc[j], v[q][j]);
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
    store_row<G, J>(A.part, (size_t)block_id, ld, lane, t);
  }
  if (A.loss_out) {                                // this block's slice of the per-sample losses
    float a = 0.f;
    for (int b = b0 + (int)threadIdx.x; b < b1; b += NT) a += A.lossb[b];
    const float tl = block_sum(a, red);
    if (threadIdx.x == 0) A.part[(size_t)A.n_part * ld + block_id] = tl;
  }
}

template <int G, int J, int NT>
__device__ __forceinline__ void bias_final_body(const DrxCdaeParams &P, const DrxOptim &opt, const BiasArgs &A,
                                                float *lds /* [NT/G, ld] */, float *red) {
  constexpr int R = NT / G;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = f4_zero();
  constexpr int NB = J == 1 ? 8 : 2;
  for (int i = r; i < A.n_part; i += NB * R) {
    float4 v[NB][J];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
#pragma unroll
      for (int j = 0; j < J; ++j) v[q][j] = f4_zero();
      if (i + q * R < A.n_part) load_row<G, J>(A.part, (size_t)(i + q * R), P.ld, lane, v[q]);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(acc[j], v[q][j]);
  }
  store_row<G, J>(lds, (size_t)r, P.ld, lane, acc);
  __syncthreads();
  if (r == 0) {
    float4 g[J], w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = f4_zero();
#pragma unroll 8
    for (int rr = 0; rr < R; ++rr) {
      float4 v[J];
      load_row<G, J>(lds, (size_t)rr, P.ld, lane, v);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(g[j], v[j]);
    }
    load_row<G, J>(P.b, 0, P.ld, lane, w);
    OptScalars o = opt_for(opt, 0, A.B);
    if (o.kind == DRX_OPT_ROWWISE_ADAGRAD) o.kind = DRX_OPT_ADAGRAD;      // the bias vectors keep one accumulator per element
    o.rb = 0.f;
    row_update<G, J>(o, P.b, opt.s1[3], opt.s2[3], 0, P.ld, lane, w, g);
  }
  if (A.loss_out) {   // mean of the per-sample losses from the per-block partials, fixed order
    const float *lp = A.part + (size_t)A.n_part * P.ld;
    float a = 0.f;
    for (int b = threadIdx.x; b < A.n_part; b += NT) a += lp[b];
    float t = block_sum(a, red);
    if (threadIdx.x == 0) { A.loss_out[0] = t / (float)A.B; A.loss_out[1] = 0.f; }
  }
}

// The two stages of the hidden-bias gradient ride along with the planned segmented reduction (drx_segreduce.hpp): the column-sum
// partials of dz1 as extra workgroups of k_seg_reduce_planned (they depend on the forward kernel only), the final sum + update of b
// (+ the mean loss) as one extra workgroup of k_span_planned.  The sparse step is three launches: forward/backward, reduction, spans.
template <int G, int J>
struct BiasPartialExtra {
  int ld;
  BiasArgs A;
  __device__ __forceinline__ void operator()(float *lds) const {
    __shared__ float red[kSegBlock / 64 > 0 ? kSegBlock / 64 : 1];
    bias_partial_body<G, J, kSegBlock>(ld, A, (int)blockIdx.x, lds, red);
  }
};

template <int G, int J>
struct BiasFinalExtra {
  DrxCdaeParams P;
  DrxOptim opt;
  BiasArgs A;
  __device__ __forceinline__ void operator()(float *lds) const {
    __shared__ float red[kFixBlock / 64];
    bias_final_body<G, J, kFixBlock>(P, opt, A, lds, red);
  }
};

// ------------------------------------------------------------------------------------------------
// scratch layouts (shared by the sizing entry point and the step functions)
// ------------------------------------------------------------------------------------------------
constexpr int kSmallBatch = 1024;    // at or below: one workgroup per batch row in the hidden-layer gather
constexpr int kFewRows = 128;        // at or below: that workgroup has 1024 threads (fewer rows than CUs: spread each row wider)
constexpr int kOutGrid = 512;        // persistent workgroups of k_out_dense (two per CU when LDS allows)
constexpr int kSweepGrid = 1024;
constexpr int kLongBlocks = 256, kShortBlocks = 1024, kHotFinBlocks = 256;
constexpr int kHotBlocks = 2048;     // workgroups of the hot-tiles role inside the reduction's launch (a multiple of 8: one share per XCD)   // k_span_planned: workgroups striding over the long / the short spans
constexpr size_t kLdsBudget = 144 * 1024;

struct DenseLayout {
  float *h, *dz1, *dh_slab, *loss_part, *reg_part1, *reg_part2, *gbuf, *gb2buf;
  int32_t *cnt;
  uint32_t *km, *vm, *tb;
  size_t zero_begin, zero_end;
  int Bw, Nw, Bs, n_sub, out_grid;
  size_t lds_bytes;
  bool tile_path;               // k_out_dense_tile: the whole batch in LDS
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
  L.tile_path = out_tile_lds_floats(B, P.ld) * 4 <= kLdsBudget / 2 && B <= 256;      // two workgroups per CU
  if (L.tile_path) {
    const int nt = (P.n_items + kTileR - 1) / kTileR;
    L.out_grid = nt < kOutGrid ? nt : kOutGrid;
    L.Bs = B; L.n_sub = 1;
    L.lds_bytes = out_tile_lds_floats(B, P.ld) * 4;
  }
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

struct PrepBufs {
  uint32_t *keys_s, *vals_s, *keys, *vals;
  void *sort_temp;
  size_t sort_bytes;
  uint8_t *solo_v, *solo_o;     // [B] each (see k_mark_solo); then [B] "holds a marked W row"
  uint32_t *solo_w;             // [ceil(N/32)] one bit per item
  SpanPlan plan;                // chunk-crossing segments of the list (k_plan_spans)
  uint32_t *vtab;               // [vt] the batch's user table (k_user_slots), part of the result: slot -> user id
  uint32_t *vslot;              // [B] slot of every sample's user (work area)
  int vt, bits_hashed;          // table entries (power of two >= 4B); key bits when the V keys are slots
  int32_t *order;               // [B] launch order of the forward kernel (k_degree_counts / k_degree_scatter)
  unsigned int *order_work;     // [512] bucket counts | running places (zeroed by k_sparse_touches)
  int n_chunks;
  size_t result_bytes;
  int T, bits;
};

static PrepBufs prep_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots) {
  PrepBufs R{};
  R.T = n_touch_slots + 2 * B;
  R.bits = bits_for((uint64_t)2 * P.n_items + P.n_users + 1);
  // the RESULT first and contiguous (drx_cdae_prep_result_bytes: what a step reads, and all that has to travel when one rank
  // prepares a list for the others), then what only the preparation itself needs
  R.keys_s = cv.take<uint32_t>(R.T);
  R.vals_s = cv.take<uint32_t>(R.T);
  R.solo_v = cv.take<uint8_t>((size_t)3 * B);
  R.solo_o = R.solo_v ? R.solo_v + B : nullptr;
  R.solo_w = cv.take<uint32_t>(((size_t)P.n_items + 31) / 32);
  R.n_chunks = (R.T + kChunk - 1) / kChunk;
  R.plan.desc = cv.take<uint2>(R.n_chunks);
  R.plan.cnt = cv.take<uint32_t>(64);
  R.plan.ext = cv.take<uint8_t>(R.n_chunks);
  R.plan.hflag = cv.take<uint8_t>(R.n_chunks);
  R.plan.hot = cv.take<uint4>(kMaxHot);
  R.plan.hot_bounds = cv.take<uint32_t>((size_t)kMaxHot * (kHotTiles + 1));
  R.plan.hot_pref = cv.take<uint32_t>(kMaxHot + 1);
  R.order = cv.take<int32_t>(B);
  R.vt = 1;
  while (R.vt < 4 * B) R.vt <<= 1;
  R.vtab = cv.take<uint32_t>(R.vt);
  R.bits_hashed = bits_for((uint64_t)2 * P.n_items + (uint64_t)R.vt + 1);
  R.result_bytes = align_up(cv.off, 256);
  R.keys = cv.take<uint32_t>(R.T);
  R.vals = cv.take<uint32_t>(R.T);
  R.sort_bytes = std::max(sort_pairs_temp_bytes(R.T, R.bits), sort_pairs_temp_bytes(R.T, R.bits_hashed));     // (11-bit digits need more)
  R.sort_temp = cv.take<char>(R.sort_bytes);
  R.order_work = cv.take<unsigned int>(512);
  R.vslot = cv.take<uint32_t>(B);
  return R;
}

static SparseBufs sparse_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots) {
  SparseBufs S{};
  S.T = n_touch_slots + 2 * B;
  S.n_chunks = (S.T + kChunk - 1) / kChunk;
  S.n_bpart = 1024;      // (256: each row group of a bias block summed 32 rows one load at a time; tail_a 25.0 -> 23.5 us)
  S.dz1 = cv.take<float>((size_t)B * P.ld);
  S.g2 = cv.take<float>((size_t)B * P.ld);
  S.dz2 = cv.take<float>(B);
  S.lossb = cv.take<float>(B);
  S.phead = cv.take<float>((size_t)S.n_chunks * P.ld);
  S.ptail = cv.take<float>((size_t)S.n_chunks * P.ld);
  S.phs = cv.take<float>(S.n_chunks);
  S.pts = cv.take<float>(S.n_chunks);
  {
    const int cpb = kSegBlock / pick_geom(P.ld).G;
    const int n_blocks = (S.n_chunks + cpb - 1) / cpb;
    S.pblock = cv.take<float>((size_t)n_blocks * P.ld);
    S.pbs = cv.take<float>(n_blocks);
    // (a hot segment has at least max(32, T / kMaxHot) touches — plan_hot_min —; a tile's share of it is cut into hot_slices(len) slices:
    // one partial row per (segment, tile, slice))
    const size_t n_hp = ((size_t)S.T / (kHotTiles * kHotSlice) + 1 + (size_t)std::min<long long>(kMaxHot, (long long)S.T / 32 + 1)) * kHotTiles;
    S.hot_part = cv.take<float>(n_hp * P.ld);
    S.hot_ps = cv.take<float>(n_hp);
  }
  S.bpart = cv.take<float>((size_t)S.n_bpart * (P.ld + 1));     // partial rows + per-block loss partials
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

// ---- touch list prepared in PARTS (column-sharded multi-GPU: every rank needs the same list of the same global batch) ---------
// Sorting it on every rank is the one cost of that layout that does not shrink with N (10 M pairs at 8 GPUs: 0.75 ms per step).
// Any order that keeps equal keys adjacent (and their touches in sample order) serves the segmented reduction, so rank r sorts
// only the touches whose row it "owns" — row id modulo the number of parts, which spreads rows evenly however ids were assigned — and
// the global list is the concatenation of the parts in rank order.
// Two passes over the batch (count, then write at the scanned offsets) take the owned touches in sample order straight from the
// histories: nothing of the size of the whole list is ever written.  WRITE = false: cnt[b] = owned touches of sample b;
// WRITE = true: cnt[] holds the inclusive scan of those counts.
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_owned_touches(int n_items, DrxHistory H, DrxBatch bt, uint32_t qthr, int part, int parts,
                                                          int *__restrict__ cnt, int cap, uint32_t *__restrict__ ck,
                                                          uint32_t *__restrict__ cv, int32_t *__restrict__ header) {
  constexpr int G = 16;
  const int lane = threadIdx.x % G;
  const int b = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (b >= bt.B) return;
  const int gshift = ((threadIdx.x % 64) / G) * G;            // this group's 16 bits of the wave's ballot
  const int u = bt.uid[b];
  const int64_t s = H.indptr[u], e = H.indptr[u + 1];
  const int deg = (int)(e - s);
  const uint8_t *kp = bt.keep ? bt.keep + bt.keep_off[b] : nullptr;
  int run = WRITE ? (b > 0 ? cnt[b - 1] : 0) : 0;
  for (int j0 = 0; j0 < deg; j0 += G) {
    const int jj = j0 + lane;
    bool mine = false;
    uint32_t key = 0;
    if (jj < deg) {
      const bool kf = kp ? (kp[jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)b, (uint32_t)jj) >= qthr);
      key = (uint32_t)H.indices[s + jj];
      mine = kf && (int)(key % (uint32_t)parts) == part;
    }
    const uint32_t m = (uint32_t)(__ballot(mine) >> gshift) & 0xFFFFu;
    if (WRITE && mine) {
      const int pos = run + __popc(m & ((1u << lane) - 1u));
      if (pos < cap) { ck[pos] = key; cv[pos] = (uint32_t)b; }
    }
    run += __popc(m);
  }
  if (lane == 0) {
    const uint32_t i = (uint32_t)bt.iid[b];
    const bool own_o = (int)(i % (uint32_t)parts) == part, own_v = (int)((uint32_t)u % (uint32_t)parts) == part;
    if (WRITE) {
      if (own_o) { if (run < cap) { ck[run] = (uint32_t)n_items + i; cv[run] = (uint32_t)b; } ++run; }
      if (own_v) { if (run < cap) { ck[run] = 2u * (uint32_t)n_items + (uint32_t)u; cv[run] = (uint32_t)b; } ++run; }
      if (b == bt.B - 1) { header[0] = run < cap ? run : cap; header[2] = run > cap ? 1 : 0; }
    } else {
      cnt[b] = run + (own_o ? 1 : 0) + (own_v ? 1 : 0);
    }
  }
}

// A part travels as [header: 4 int32 = touches, runs, overflow, 0 | runs: (key << 32 | first position) per distinct key | the samples
// of the touches, grouped by key]: 4 bytes per touch instead of 8, which is what the exchange costs.
__global__ void k_run_flags(const uint32_t *__restrict__ ks, int n, int *__restrict__ flag) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t k = ks[j];
    flag[j] = (k != DRX_KEY_NONE && (j == 0 || ks[j - 1] != k)) ? 1 : 0;
  }
}

__global__ void k_take_runs(const uint32_t *__restrict__ ks, const int *__restrict__ scan, int n, int rcap,
                            unsigned long long *__restrict__ runs, int32_t *__restrict__ header) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int incl = scan[j], prev = j > 0 ? scan[j - 1] : 0;
    if (incl != prev && incl <= rcap) runs[incl - 1] = ((unsigned long long)ks[j] << 32) | (uint32_t)j;
    if (j == n - 1) {
      header[1] = incl < rcap ? incl : rcap;
      if (incl > rcap) header[2] = 1;               // cannot happen (rcap bounds the distinct keys of a part); checked all the same
      header[3] = 0;
    }
  }
}

struct PartView {               // one part inside the exchanged buffer
  const int32_t *header;
  const unsigned long long *runs;
  const uint32_t *vals;
};

__device__ __forceinline__ PartView part_view(const char *all, size_t part_bytes, size_t runs_off, size_t vals_off, int r) {
  const char *b = all + (size_t)r * part_bytes;
  return PartView{(const int32_t *)b, (const unsigned long long *)(b + runs_off), (const uint32_t *)(b + vals_off)};
}

// parts in rank order -> keys_s / vals_s of the whole batch, padded with DRX_KEY_NONE.  One position per thread: neighbouring
// positions walk the same path through a part's runs, so the binary search costs a cache line or two per step and wave.
__global__ void k_assemble_parts(const char *__restrict__ all, size_t part_bytes, size_t runs_off, size_t vals_off, int parts, int T,
                                 uint32_t *__restrict__ keys_s, uint32_t *__restrict__ vals_s, int32_t *__restrict__ flags_out) {
  __shared__ int off[DRX_MAX_WORLD + 1];
  if (threadIdx.x == 0) {
    int run = 0, bad = 0;
    for (int r = 0; r < parts; ++r) {
      const int32_t *h = part_view(all, part_bytes, runs_off, vals_off, r).header;
      off[r] = run; run += h[0]; bad |= h[2];
    }
    off[parts] = run;
    if (blockIdx.x == 0) flags_out[0] = (bad || run > T) ? 1 : 0;
  }
  __syncthreads();
  const int total = off[parts] < T ? off[parts] : T;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < T; j += gridDim.x * blockDim.x) {
    if (j >= total) { keys_s[j] = DRX_KEY_NONE; vals_s[j] = 0; continue; }
    int r = 0;
    while (r + 1 < parts && j >= off[r + 1]) ++r;
    const PartView pv = part_view(all, part_bytes, runs_off, vals_off, r);
    const uint32_t local = (uint32_t)(j - off[r]);
    int lo = 0, hi = pv.header[1] - 1;               // last run starting at or before `local`
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((uint32_t)pv.runs[mid] <= local) lo = mid; else hi = mid - 1;
    }
    keys_s[j] = (uint32_t)(pv.runs[lo] >> 32);
    vals_s[j] = pv.vals[local];
  }
}

struct PartOut {                 // layout of one exchanged part
  size_t runs_off, vals_off, bytes;
  int cap, rcap;
};

static PartOut part_out_layout(const DrxCdaeParams &P, int B, int n_touch_slots, int parts) {
  PartOut o{};
  const long long T = (long long)n_touch_slots + 2ll * B;
  o.cap = (int)(T / parts + T / (4 * parts) + 16384);         // 1.25 x the even share + slack (ids are spread evenly)
  if (parts == 1 || o.cap > T) o.cap = (int)T;
  o.rcap = 2 * ((P.n_items + parts - 1) / parts + 1) + (P.n_users + parts - 1) / parts + 1;   // distinct rows a part can own
  o.runs_off = 256;
  o.vals_off = align_up(o.runs_off + (size_t)o.rcap * 8, 256);
  o.bytes = align_up(o.vals_off + (size_t)o.cap * 4, 256);
  return o;
}

struct PartBufs {
  PrepBufs R;                    // full touch arrays (keys / vals) + sort temp of the full size
  int *flag;
  void *scan_temp;
  size_t scan_bytes;
  uint32_t *ck, *cv, *ck_s;
  PartOut out;
};

static PartBufs part_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots, int parts) {
  PartBufs L{};
  L.R = prep_layout(cv, P, B, n_touch_slots);
  L.out = part_out_layout(P, B, n_touch_slots, parts);
  L.flag = cv.take<int>(L.R.T);                 // [B] counts of the samples, later [cap] run flags
  L.scan_bytes = scan_i32_temp_bytes((size_t)(L.R.T > 0 ? L.R.T : 1));
  L.scan_temp = cv.take<char>(L.scan_bytes);
  L.ck = cv.take<uint32_t>(L.out.cap); L.cv = cv.take<uint32_t>(L.out.cap);
  L.ck_s = cv.take<uint32_t>(L.out.cap);
  return L;
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
    case DRX_ERETRY: return "sampler gave up after its maximum number of consecutive failed attempts";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown drx error";
  }
}

uint32_t drx_hash_u32(uint64_t seed, uint32_t a, uint32_t b) { return hash_u32(seed, a, b); }

// ---- light events for the run-ahead pipelines: ordering between two streams of ONE device.  hipEventDisableSystemFence: the record
// releases at agent scope instead of writing the L2 back for the host and peers — all a same-device hipStreamWaitEvent needs.
void *drx_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
  return (void *)e;
}
void drx_event_destroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int drx_event_record(void *ev, void *stream) { return ev ? (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) : DRX_EINVAL; }
int drx_stream_wait_event(void *stream, void *ev) { return ev ? (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) : DRX_EINVAL; }
int drx_event_synchronize(void *ev) { return ev ? (int)hipEventSynchronize((hipEvent_t)ev) : DRX_EINVAL; }

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
    if (bt->B <= kSmallBatch)                                                                             \
      hipLaunchKernelGGL((k_hidden_fwd_wg<G, J, 0>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *hist, *bt, \
                         scale, qthr, h, none);                                                           \
    else                                                                                                  \
      hipLaunchKernelGGL((k_hidden_fwd<G, J, 0>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt, \
                         scale, qthr, h, none);                                                           \
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

size_t drx_cdae_scratch_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t dense_mode) {
  if (!p || B < 1 || n_touch_slots < 0) return 0;
  Carver c(nullptr, 0);
  if (dense_mode) {
    (void)dense_layout(c, *p, B, true);
  } else {
    (void)sparse_layout(c, *p, B, n_touch_slots);
    (void)prep_layout(c, *p, B, n_touch_slots);
  }
  return align_up(c.off, 256) + 256;
}

static int step_dense_impl(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                           int32_t loss_kind, int32_t targets_kind, void *scratch, size_t scratch_bytes, float *loss_out,
                           void *stream, const DensePrefetch &pf) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !opt || !scratch) return DRX_EINVAL;
  if (opt->kind != DRX_OPT_ADAM && opt->kind != DRX_OPT_ADAGRAD) return DRX_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!opt->s1[i] || (opt->kind == DRX_OPT_ADAM && !opt->s2[i])) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  // DRX_DENSE_AUX_CLEAN: the caller vouches that the batch-membership arrays in `scratch` are all zero — true for zero-initialised
  // scratch and after every completed dense step of the same batch size, because the kernels that consume an entry clear it
  const bool aux_clean = (targets_kind & DRX_DENSE_AUX_CLEAN) != 0;
  targets_kind &= 0xFF;
  const bool per_row = targets_kind == DRX_TARGETS_PER_ROW;
  DenseLayout L = dense_layout(cv, *p, bt->B, per_row);
  if (!cv.ok()) return DRX_ESCRATCH;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  if (!aux_clean) DRX_HIP(hipMemsetAsync((char *)scratch + L.zero_begin, 0, L.zero_end - L.zero_begin, st));
  else if (per_row) DRX_HIP(hipMemsetAsync(L.tb, 0, (size_t)bt->B * L.Nw * 4, st));      // (target bits are shared by tiles: no single consumer)
  DenseAux aux{L.cnt, L.km, L.vm, per_row ? L.tb : nullptr, L.Bw, L.Nw};
  OutDenseArgs A{};
  A.h = L.h; A.cnt = L.cnt; A.tb = aux.tb; A.Nw = L.Nw; A.B = bt->B; A.Bs = L.Bs; A.n_sub = L.n_sub;
  A.gbuf = L.gbuf; A.gb2buf = L.gb2buf; A.dh_slab = L.dh_slab; A.loss_part = L.loss_part; A.reg_part = L.reg_part1;
  A.loss_kind = loss_kind;
  const int total_rows = p->n_items + p->n_users;
#define CALL(G, J)                                                                                                   \
  {                                                                                                                  \
    const int gpb = kBlock / G;                                                                                      \
    /* a 256-thread workgroup keeps 4 * gpb history rows in flight per round: go wide when a row needs more than two rounds */ \
    const bool wide_fwd = bt->B <= kFewRows && (int64_t)bt->n_touch_slots > (int64_t)bt->B * 8 * gpb;                \
    if (wide_fwd)               /* few rows with long histories: 1024 threads per batch row (32 groups split the history) */ \
      hipLaunchKernelGGL((k_hidden_fwd_wg<G, J, 1, 1024>), dim3(bt->B), dim3(1024), (size_t)(1024 / G) * p->ld * 4, st, *p, *hist, \
                         *bt, scale, qthr, L.h, aux);                                                                \
    else if (bt->B <= kSmallBatch)                                                                                   \
      hipLaunchKernelGGL((k_hidden_fwd_wg<G, J, 1>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *hist, *bt, \
                         scale, qthr, L.h, aux);                                                                     \
    else                                                                                                             \
      hipLaunchKernelGGL((k_hidden_fwd<G, J, 1>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt, \
                         scale, qthr, L.h, aux);                                                                     \
    if (L.tile_path) {                                                                                               \
      if (loss_out) {                                                                                                \
        DRX_HIP(hipFuncSetAttribute((const void *)k_out_dense_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)L.lds_bytes));                                                              \
        hipLaunchKernelGGL((k_out_dense_tile<true>), dim3(L.out_grid), dim3(kBlock), L.lds_bytes, st, *p, *opt, A);  \
      } else {                                                                                                       \
        DRX_HIP(hipFuncSetAttribute((const void *)k_out_dense_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)L.lds_bytes));                                                              \
        hipLaunchKernelGGL((k_out_dense_tile<false>), dim3(L.out_grid), dim3(kBlock), L.lds_bytes, st, *p, *opt, A); \
      }                                                                                                              \
    } else if (loss_out) {                                                                                           \
      DRX_HIP(hipFuncSetAttribute((const void *)k_out_dense<G, J, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)L.lds_bytes));                                                                \
      hipLaunchKernelGGL((k_out_dense<G, J, true>), dim3(L.out_grid), dim3(kBlock), L.lds_bytes, st, *p, *opt, A);   \
    } else {                                                                                                         \
      DRX_HIP(hipFuncSetAttribute((const void *)k_out_dense<G, J, false>, hipFuncAttributeMaxDynamicSharedMemorySize,\
                                  (int)L.lds_bytes));                                                                \
      hipLaunchKernelGGL((k_out_dense<G, J, false>), dim3(L.out_grid), dim3(kBlock), L.lds_bytes, st, *p, *opt, A);  \
    }                                                                                                                \
    if (bt->B <= kFewRows && L.out_grid > 16 * gpb)          /* many partial slabs per row: 1024 threads fold them */      \
      hipLaunchKernelGGL((k_hidden_bwd<G, J, 1024>), dim3(bt->B), dim3(1024), (size_t)(1024 / G) * p->ld * 4, st, p->ld, bt->B, \
                         L.out_grid, L.dh_slab, L.h, L.dz1);                                                         \
    else                                                                                                             \
      hipLaunchKernelGGL((k_hidden_bwd<G, J>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, p->ld, bt->B, \
                         L.out_grid, L.dh_slab, L.h, L.dz1);                                                         \
    int sweep = (total_rows + gpb - 1) / gpb;                                                                        \
    if (sweep > kSweepGrid) sweep = kSweepGrid;                                                                      \
    hipLaunchKernelGGL((k_in_sweep<G, J>), dim3(sweep + 1), dim3(kBlock), 0, st, *p, *opt, bt->B, scale, aux, L.dz1, \
                       L.reg_part2, pf);                                                                             \
    if (loss_out)                                                                                                    \
      hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, st, L.loss_part, L.out_grid, L.reg_part1, L.out_grid,   \
                         L.reg_part2, sweep + 1, 0.5f * opt->reg_rate / (float)bt->B, loss_out);                     \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_cdae_step_dense(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                        int32_t loss_kind, int32_t targets_kind, void *scratch, size_t scratch_bytes, float *loss_out,
                        void *stream) {
  return step_dense_impl(p, opt, hist, bt, loss_kind, targets_kind, scratch, scratch_bytes, loss_out, stream, DensePrefetch{});
}

// ---- the quiet fit() loop of reference mode in one call -------------------------------------------------------------------------
// What RecommenderABC.fit() does per epoch when nobody watches single steps (recommender_abc.py:189-205 with verbose off and no
// early-stopping rule): take the next drawn batch, queue its training step, keep the draw-ahead workers fed.  A staging slot
// (pinned, device-addressable) is laid out
//   [uid int32 B | keep_off int32 B+1 | keep u8 keep_capacity | iid int32 B | value f64 B | is_negative u8 B], every array 16-byte
// aligned: what a step reads comes first.  The step of batch s copies that prefix of batch s+1 into device memory from the last
// workgroup of its parameter sweep (k_in_sweep: one PCIe round trip hidden behind the sweep), so that the gather kernel of step
// s+1 reads its batch from HBM: 7.8 instead of 12.2 us at the ml-100k shape; only the first batch of a call is read in place.
struct FitSlot {
  size_t uid, keep_off, keep, iid, val, neg, total;
};
static FitSlot fit_slot_layout(int32_t B, int64_t keep_capacity) {
  FitSlot s{};
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off = align_up(off + bytes, 16); return at; };
  s.uid = take((size_t)B * 4);
  s.keep_off = take((size_t)(B + 1) * 4);
  s.keep = take((size_t)(keep_capacity > 0 ? keep_capacity : 1));
  s.iid = take((size_t)B * 4);
  s.val = take((size_t)B * 8);
  s.neg = take((size_t)B);
  s.total = align_up(off, 256);
  return s;
}

size_t drx_cdae_fit_slot_bytes(int32_t B, int64_t keep_capacity) {
  if (B < 1 || keep_capacity < 0) return 0;
  return fit_slot_layout(B, keep_capacity).total;
}

int drx_cdae_fit_dense(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, DrxDrawAhead *draws, int64_t *cursor,
                       int32_t B, float q, int64_t keep_capacity, int32_t loss_kind, int32_t targets_kind, int64_t n_steps,
                       const float *h_alphas, void *h_slots, size_t slot_bytes, int32_t n_slots, void *d_stage,
                       size_t d_stage_bytes, void *scratch, size_t scratch_bytes, void *stream) {
  constexpr int kAhead = 4;             // draws in flight beyond the two batches the loop holds: two per worker
  constexpr int kMaxSlots = 64;
  if (!p || !opt || !hist || !draws || !cursor || !h_alphas || !h_slots || !scratch || B < 1 || n_steps < 0 || keep_capacity < 0)
    return DRX_EINVAL;
  if (n_slots < kAhead + 4 || n_slots > kMaxSlots) return DRX_EINVAL;
  const FitSlot S = fit_slot_layout(B, keep_capacity);
  if (slot_bytes < S.total) return DRX_ESCRATCH;
  if (d_stage && d_stage_bytes < 2 * S.iid) return DRX_ESCRATCH;       // two device copies of the prefix a step reads
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev[kMaxSlots] = {};
  bool busy[kMaxSlots] = {};
  for (int k = 0; k < n_slots; ++k)
    if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
      for (int j = 0; j < k; ++j) (void)hipEventDestroy(ev[j]);
      return DRX_EINVAL;
    }
  int64_t ticket = cursor[0];
  uint64_t mask_pos = (uint64_t)cursor[1], mask_at[2] = {(uint64_t)cursor[2], (uint64_t)cursor[3]};
  const uint64_t words = (uint64_t)2 * (uint64_t)p->n_items * (uint64_t)B;      // cdae.py:63: N uniform draws per batch row
  struct Pending { int gen; int64_t job; int slot; } ring[kAhead + 1];
  int head = 0, count = 0;
  int64_t next_slot = 0, submitted = 0;
  int rc = DRX_OK;
  hipError_t herr = hipSuccess;
  auto slot_at = [&](int k) { return (char *)h_slots + (size_t)k * slot_bytes; };
  auto submit = [&]() -> int {
    const int k = (int)(next_slot % n_slots);
    if (busy[k]) {                      // the step that last read this slot must have finished before a worker refills it
      if ((herr = hipEventSynchronize(ev[k])) != hipSuccess) return (int)herr;
      busy[k] = false;
    }
    char *b = slot_at(k);
    const int gen = (int)(ticket & 1);
    const int64_t job = drx_drawahead_submit(draws, gen, ticket, mask_pos - mask_at[gen], B, (double)q, (int32_t *)(b + S.uid),
                                             (int32_t *)(b + S.iid), (double *)(b + S.val), (uint8_t *)(b + S.neg),
                                             (int32_t *)(b + S.keep_off), (uint8_t *)(b + S.keep), keep_capacity > 0 ? keep_capacity : 1);
    if (job < 0) return (int)job;
    ++next_slot; ++ticket; ++submitted;
    mask_pos += words;
    mask_at[gen] = mask_pos;
    ring[(head + count) % (kAhead + 1)] = Pending{gen, job, k};
    ++count;
    return DRX_OK;
  };
  // never beyond the last step: the sampler and corruption streams end where the reference's do
  auto refill = [&]() { while (rc == DRX_OK && count < kAhead && submitted < n_steps) rc = submit(); };
  auto take = [&](Pending &out) {       // the oldest draw in flight, completed
    out = ring[head];
    head = (head + 1) % (kAhead + 1);
    --count;
    const int wrc = drx_drawahead_wait(draws, out.gen, out.job);
    if (rc == DRX_OK) rc = wrc;
  };
  DrxOptim o = *opt;
  Pending cur{}, nxt{};
  bool have_nxt = false;
  int64_t s = 0;
  if (n_steps > 0) {
    refill();
    if (rc == DRX_OK) take(cur);
  }
  for (; s < n_steps && rc == DRX_OK; ++s) {
    refill();
    if (rc) break;
    have_nxt = false;
    if (s + 1 < n_steps) {
      take(nxt);
      have_nxt = true;
      refill();
      if (rc) break;
    }
    const char *hb = slot_at(cur.slot);
    const bool on_device = d_stage && s > 0;          // (copied there by the previous step's sweep)
    const char *b = on_device ? (const char *)d_stage + (size_t)(s & 1) * S.iid : hb;
    DrxBatch bt{};
    bt.B = B;
    bt.uid = (const int32_t *)(b + S.uid);
    bt.keep_off = (const int32_t *)(b + S.keep_off);
    bt.keep = (const uint8_t *)(b + S.keep);
    bt.q = q;
    bt.n_touch_slots = ((const int32_t *)(hb + S.keep_off))[B];
    DensePrefetch pf{};
    if (have_nxt && d_stage) {
      const char *nb = slot_at(nxt.slot);
      const int32_t n_keep = ((const int32_t *)(nb + S.keep_off))[B];
      pf.src = (const uint4 *)nb;
      pf.dst = (uint4 *)((char *)d_stage + (size_t)((s + 1) & 1) * S.iid);
      pf.n16 = (S.keep + (size_t)(n_keep > 0 ? n_keep : 1) + 15) / 16;
    }
    for (int j = 0; j < 5; ++j) o.alpha[j] = h_alphas[s * 5 + j];
    rc = step_dense_impl(p, &o, hist, &bt, loss_kind, targets_kind | DRX_DENSE_AUX_CLEAN, scratch, scratch_bytes, nullptr, stream, pf);
    if (rc) break;
    // this step is the last reader of its own slot when it read it in place, and of the next batch's slot when it copied it
    const int released = pf.n16 ? nxt.slot : (on_device ? -1 : cur.slot);
    if (!on_device && pf.n16) {         // (first step of a call: both)
      if ((herr = hipEventRecord(ev[cur.slot], st)) != hipSuccess) { rc = (int)herr; break; }
      busy[cur.slot] = true;
    }
    if (released >= 0) {
      if ((herr = hipEventRecord(ev[released], st)) != hipSuccess) { rc = (int)herr; break; }
      busy[released] = true;
    }
    cur = nxt;
  }
  // draws still in flight write into the caller's slots: wait for them whatever happened (their place in the streams is consumed)
  while (count > 0) { Pending drop; take(drop); }
  cursor[0] = ticket;
  cursor[1] = (int64_t)mask_pos;
  cursor[2] = (int64_t)mask_at[0];
  cursor[3] = (int64_t)mask_at[1];
  // the slots are free for the caller's next run once the steps queued here have read them (the events do not outlive the call)
  herr = hipStreamSynchronize(st);
  if (rc == DRX_OK && herr != hipSuccess) rc = (int)herr;
  for (int k = 0; k < n_slots; ++k) (void)hipEventDestroy(ev[k]);
  return rc;
}

// plan.cnt (64 words), plan.ext and plan.hflag (n_chunks bytes each) are consecutive 256-byte-aligned allocations: one range of words to zero
static int plan_zero_words(const PrepBufs &R) { return (int)(((const char *)R.plan.hflag + R.n_chunks - (const char *)R.plan.cnt + 3) / 4); }

// The chunk-crossing segments of a sorted list, short ones and long ones (drx_segreduce.hpp, planned variant).  On the pristine list:
// BEFORE the sole-toucher marks blank any key.
static int plan_spans(const DrxCdaeParams *p, const DrxBatch *bt, const PrepBufs &R0, hipStream_t st, bool cleared) {
  PrepBufs R = R0;
  R.plan.hot_min = plan_hot_min(R.T, bt->flags);
  if (!cleared) DRX_HIP(hipMemsetAsync(R.plan.cnt, 0, (size_t)plan_zero_words(R) * 4, st));     // (prepare_impl's touch kernel clears them)
  hipLaunchKernelGGL(k_plan_spans<0>, dim3((R.n_chunks + 255) / 256), dim3(256), 0, st, R.keys_s, R.T, R.n_chunks,
                     kSegBlock / pick_geom(p->ld).G, R.plan);
  if (R.plan.hot_min != 0x7FFFFFFF)
    hipLaunchKernelGGL(k_hot_bounds, dim3((kMaxHot * (kHotTiles + 1) + 255) / 256), dim3(256), 0, st, R.vals_s, R.plan,
                       (bt->B + kHotTiles - 1) / kHotTiles);
  return DRX_OK;
}

// (see k_degree_counts)
static void order_by_degree(const DrxBatch *bt, const PrepBufs &R, hipStream_t st, bool cleared) {
  if (!cleared) (void)hipMemsetAsync(R.order_work, 0, 512 * sizeof(unsigned int), st);
  const int blocks = (bt->B + 1023) / 1024;
  hipLaunchKernelGGL(k_degree_counts, dim3(blocks), dim3(1024), 0, st, bt->keep_off, bt->B, R.order_work);
  hipLaunchKernelGGL(k_degree_scatter, dim3(blocks), dim3(1024), 0, st, bt->keep_off, bt->B, R.order_work, R.order);
}

// W rows get sole-toucher marks when the caller asks for them (DRX_BATCH_MARK_W: worth it where a batch leaves most of its distinct
// W rows with one touch — large catalogues; at MovieLens shapes every item collects hundreds of touches and nothing would be marked).
// Rows of <= 16 floats are never marked (see prepare_impl).
static bool mark_w_rows(const DrxCdaeParams *p, const DrxBatch *bt) { return (bt->flags & DRX_BATCH_MARK_W) != 0 && p->ld > 16; }

static int prepare_impl(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const PrepBufs &R, hipStream_t st,
                        bool with_marks = false) {
  const int gpb = kBlock / 16;
  // V keys = slots of the batch's user table: on request (DRX_BATCH_V_SLOTS).  Measured at 10 M users (r03l): the narrower key trades
  // three 8-bit passes of the sort (3 x 60 us beside the training kernels) for two 11-bit ones (2 x 109 us) — a loss with this sort,
  // whose ranking step costs one ballot per digit bit; kept for sorts / shapes where the pass count decides.
  const bool hashed = (bt->flags & DRX_BATCH_V_SLOTS) != 0 && p->n_users > R.vt && R.bits_hashed < R.bits;
  if (hashed) {
    DRX_HIP(hipMemsetAsync(R.vtab, 0xFF, (size_t)R.vt * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_user_slots, dim3((bt->B + kBlock - 1) / kBlock), dim3(kBlock), 0, st, bt->uid, bt->B, R.vtab, (uint32_t)(R.vt - 1),
                       R.vslot);
  }
  uint32_t *sort_zero = nullptr;
  size_t sort_zero_words = 0;
  sort_pairs_zero_region(R.sort_temp, (size_t)R.T, hashed ? R.bits_hashed : R.bits, &sort_zero, &sort_zero_words);
  hipLaunchKernelGGL(k_sparse_touches, dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, p->n_items, *hist, *bt,
                     q_threshold(bt->q), R.keys, R.vals, R.T, R.solo_v, R.solo_w, R.plan.cnt, plan_zero_words(R), R.order_work, 512,
                     hashed ? R.vslot : nullptr, sort_zero, (int)sort_zero_words);
  // the launch order (see k_degree_counts): its counts ride in the sort's first launch, its scatter in the plan + marks launch below
  const bool fused_order = with_marks && p->ld > 16;
  const SortRider rider{fused_order ? bt->keep_off : nullptr, bt->B, R.order_work};
  // dropped inputs (DRX_KEY_NONE) take no part in the sort: its last pass writes them back behind the sorted touches
  const int rc = sort_pairs_ex(R.sort_temp, R.sort_bytes, R.keys, R.keys_s, R.vals, R.vals_s, (size_t)R.T, hashed ? R.bits_hashed : R.bits,
                               true, st, true, rider);
  if (rc) return rc;
  // rows of <= 16 floats (K = 128 sharded over 8 GPUs): a 64-byte random read-modify-write in the forward kernel costs more than
  // the segmented reduction saves (measured 1.018 vs 0.995 ms per step); no marks = no fusion
  if (with_marks && p->ld > 16) {
    const int blocks = std::max(2048, (R.n_chunks + 255) / 256);
    SpanPlan plan = R.plan;
    plan.hot_min = plan_hot_min(R.T, bt->flags);
    const int order_blocks = (bt->B + 255) / 256;
    hipLaunchKernelGGL(k_plan_and_mark, dim3(blocks + order_blocks), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, R.n_chunks,
                       kSegBlock / pick_geom(p->ld).G, plan, (uint32_t)p->n_items, bt->B, R.solo_v, R.solo_o,
                       mark_w_rows(p, bt) ? R.solo_w : nullptr, order_blocks, bt->keep_off, R.order_work, R.order);
    if (plan.hot_min != 0x7FFFFFFF)
      hipLaunchKernelGGL(k_hot_bounds, dim3((kMaxHot * (kHotTiles + 1) + 255) / 256), dim3(256), 0, st, R.vals_s, plan,
                         (bt->B + kHotTiles - 1) / kHotTiles);
    return DRX_OK;
  }
  return plan_spans(p, bt, R, st, true);
}

// Only for touch lists prepared AHEAD of the step (the forward kernel must see the marks): see k_mark_solo.
static int mark_solo(const DrxCdaeParams *p, const DrxBatch *bt, const PrepBufs &R, hipStream_t st, bool cleared) {
  if (!cleared) {                                                                  // (prepare_impl's touch kernel clears them)
    DRX_HIP(hipMemsetAsync(R.solo_v, 0, (size_t)bt->B * 3, st));
    DRX_HIP(hipMemsetAsync(R.solo_w, 0, (((size_t)p->n_items + 31) / 32) * sizeof(uint32_t), st));
  }
  // rows of <= 16 floats (K = 128 sharded over 8 GPUs): a 64-byte random read-modify-write in the forward kernel costs more than
  // the segmented reduction saves (measured 1.018 vs 0.995 ms per step); no marks = no fusion
  if (p->ld <= 16) return DRX_OK;       // solo_v and solo_o are adjacent
  hipLaunchKernelGGL(k_mark_solo, dim3(2048), dim3(256), 0, st, R.keys_s, R.vals_s, R.T, (uint32_t)p->n_items, bt->B, R.solo_v,
                     R.solo_o, mark_w_rows(p, bt) ? R.solo_w : nullptr);
  return DRX_OK;
}

static int step_sparse_impl(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                            int32_t loss_kind, const void *prepared, size_t prepared_bytes, void *scratch, size_t scratch_bytes,
                            float *loss_out, void *const *events, void *stream, const float *ks_h = nullptr,
                            const float *ks_dot = nullptr) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !opt || !scratch || !bt->iid || !bt->y || !bt->keep_off) return DRX_EINVAL;
  if (opt->kind != DRX_OPT_ADAM && opt->kind != DRX_OPT_ADAGRAD && opt->kind != DRX_OPT_ROWWISE_ADAGRAD) return DRX_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!opt->s1[i] || (opt->kind == DRX_OPT_ADAM && !opt->s2[i])) return DRX_EINVAL;
  if ((uint64_t)2 * p->n_items + p->n_users + 1 >= 0xFFFFFFFFull) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  SparseBufs S = sparse_layout(cv, *p, bt->B, bt->n_touch_slots);
  PrepBufs R{};
  if (prepared) {
    Carver cp(const_cast<void *>(prepared), prepared_bytes);
    R = prep_layout(cp, *p, bt->B, bt->n_touch_slots);
    if (!cp.ok()) return DRX_ESCRATCH;
  } else {
    R = prep_layout(cv, *p, bt->B, bt->n_touch_slots);
  }
  if (!cv.ok()) return DRX_ESCRATCH;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  const int rows_per_block = (bt->B + S.n_bpart - 1) / S.n_bpart;
  const int n_bpart = (bt->B + rows_per_block - 1) / rows_per_block;
  S.solo_v = prepared ? R.solo_v : nullptr;
  S.solo_o = prepared ? R.solo_o : nullptr;
  S.solo_w = (prepared && mark_w_rows(p, bt)) ? R.solo_w : nullptr;       // (the rule mark_solo followed)
  static const bool use_order = [] { const char *e = getenv("DRX_FWD_ORDER"); return !e || atoi(e) != 0; }();      // (A/B switch)
  S.order = (prepared && use_order) ? R.order : nullptr;
  SegBufs SB{R.keys_s, R.vals_s, S.phead, S.ptail, S.phs, S.pts, nullptr, nullptr, nullptr, nullptr, S.T, S.n_chunks, p->ld, nullptr};
#ifdef DRX_STAMPS
  S.stamps = SB.stamps = h_stamps;
#endif
  PlanBufs PB{S.pblock, S.pbs, S.hot_part, S.hot_ps, (bt->B + kHotTiles - 1) / kHotTiles};
  const int hot_blocks = (bt->flags >> 16) ? kHotBlocks : 0, hot_fin_blocks = (bt->flags >> 16) ? kHotFinBlocks : 0;
  // more than 8 touches per table row on average: rows collect long runs of touches (MovieLens shapes), k_seg_reduce's LB1 = 8
  const bool long_segments = (int64_t)S.T > 8 * ((int64_t)2 * p->n_items + p->n_users);
#define EV(i) do { if (events) DRX_HIP(hipEventRecord((hipEvent_t)events[i], st)); } while (0)
  // One workgroup per triple (its groups split the history) instead of one group per triple: when a group would walk many
  // dependent load rounds.  Short histories (mean <= 64 items): only while the batch cannot fill the chip anyway.
  constexpr int wg_long = 64;
  const long long mean_hist = bt->n_touch_slots / (long long)bt->B;
  const bool per_wg = mean_hist > wg_long || (bt->B <= 8192 && mean_hist > 16);
  BiasArgs BA{S.dz1, S.bpart, S.lossb, loss_out, bt->B, n_bpart, rows_per_block};
  // the LDS-prefetch forward kernel (k_sampled_fwd_bwd_pf) is OPT-IN (DRX_FWD_PF=1): with the launch order in both, the plain kernel's
  // phase is the shorter one (r03u, 3 runs each: 132 against 137 us; step 0.367 against 0.365 ms — inside the run-to-run spread)
  static const bool use_pf = [] { const char *e = getenv("DRX_FWD_PF"); return e && atoi(e) != 0; }();
  // the segmented reduction (+ the bias column sums as extra workgroups) and the ONE launch that combines the chunk-crossing segments
  // (+ the bias update), with the policy type POLT (optimizer at run time, or Adagrad compiled in)
#define REDUCE_AND_SPANS(G, J, POLT)                                                                                   \
  {                                                                                                                    \
    POLT polk{*p, *opt, bt->B, scale, S.dz1, (long long)(S.g2 - S.dz1), S.dz2, R.vtab, R.plan.cnt};                    \
    BiasPartialExtra<G, J> bpx{p->ld, BA};                                                                             \
    BiasFinalExtra<G, J> bfx{*p, *opt, BA};                                                                            \
    const int cpb = kSegBlock / G;                                                                                     \
    const dim3 rgrid(n_bpart + hot_blocks + (S.n_chunks + cpb - 1) / cpb);                                             \
    const size_t lds_r = (size_t)cpb * (p->ld + 1) * 4;                                                                \
    if (long_segments)                                                                                                 \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, 8, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB,    \
                         R.plan, polk, n_bpart, hot_blocks, bpx);                                                      \
    else                                                                                                               \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, 2, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB,    \
                         R.plan, polk, n_bpart, hot_blocks, bpx);                                                      \
    EV(3);                                                                                                             \
    if (lds_b > 48 * 1024)                                                                                             \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_planned<G, J, POLT, BiasFinalExtra<G, J>>,                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                            \
    hipLaunchKernelGGL((k_span_planned<G, J, POLT, BiasFinalExtra<G, J>>), dim3(kLongBlocks + kShortBlocks + hot_fin_blocks + 1), \
                       dim3(kFixBlock), lds_b, st, SB, PB, R.plan, polk, kLongBlocks, kShortBlocks, hot_fin_blocks, bfx); \
    EV(4);                                                                                                             \
    EV(5);                                                                                                             \
  }
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    const size_t lds_b = ((size_t)(kFixBlock / G) * (p->ld + 1)) * 4;                                                  \
    EV(0);                                                                                                             \
    if (ks_h)                                                                                                          \
      hipLaunchKernelGGL((k_kshard_rest<G, J>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, *hist, *bt, scale, \
                         qthr, loss_kind, S, ks_h, ks_dot);                                                            \
    else if (per_wg)                                                                                                   \
      hipLaunchKernelGGL((k_sampled_fwd_bwd_wg<G, J>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *opt, *hist, \
                         *bt, scale, qthr, loss_kind, S);                                                              \
    else if (opt->kind == DRX_OPT_ADAGRAD && S.solo_w && J == 1)                                                       \
      hipLaunchKernelGGL((k_sampled_fwd_bwd_stash<G, 1>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock),                 \
                         (size_t)(kBlock / 64) * kStash * 1024, st, *p, *opt, *hist, *bt, scale, qthr, loss_kind, S);  \
    else if (opt->kind == DRX_OPT_ADAGRAD && J == 1 && use_pf)                                                         \
      hipLaunchKernelGGL((k_sampled_fwd_bwd_pf<G>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), (size_t)(kBlock / 64) * 4 * 1024, \
                         st, *p, *opt, *hist, *bt, scale, qthr, loss_kind, S);                                         \
    else if (opt->kind == DRX_OPT_ADAGRAD)                                                                             \
      hipLaunchKernelGGL((k_sampled_fwd_bwd<G, J, DRX_OPT_ADAGRAD>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, \
                         *hist, *bt, scale, qthr, loss_kind, S);                                                       \
    else                                                                                                               \
      hipLaunchKernelGGL((k_sampled_fwd_bwd<G, J>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *opt, *hist, *bt,  \
                         scale, qthr, loss_kind, S);                                                                   \
    EV(1);                                                                                                             \
    if (!prepared) {                                                                                                   \
      rc = prepare_impl(p, hist, bt, R, st);                                                                           \
      if (rc) return rc;                                                                                               \
    }                                                                                                                  \
    EV(2);                                                                                                             \
    if (opt->kind == DRX_OPT_ADAGRAD) { REDUCE_AND_SPANS(G, J, DirectPolicyAdagrad); }                                 \
    else { REDUCE_AND_SPANS(G, J, DirectPolicy); }                                                                     \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
#undef REDUCE_AND_SPANS
#undef EV
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_cdae_prep_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots) {
  if (!p || B < 1 || n_touch_slots < 0) return 0;
  Carver c(nullptr, 0);
  (void)prep_layout(c, *p, B, n_touch_slots);
  return align_up(c.off, 256) + 256;
}

int drx_cdae_sparse_prepare(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, void *prepared,
                            size_t prepared_bytes, void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !prepared || !bt->iid || !bt->keep_off) return DRX_EINVAL;
  Carver cp(prepared, prepared_bytes);
  PrepBufs R = prep_layout(cp, *p, bt->B, bt->n_touch_slots);
  if (!cp.ok()) return DRX_ESCRATCH;
  rc = prepare_impl(p, hist, bt, R, (hipStream_t)stream, true);      // touches, sort, span plan + sole-toucher marks (+ launch order)
  if (rc) return rc;
  if (p->ld <= 16) order_by_degree(bt, R, (hipStream_t)stream, true);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

size_t drx_cdae_prep_result_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots) {
  if (!p || B < 1 || n_touch_slots < 0) return 0;
  Carver c(nullptr, 0);
  return prep_layout(c, *p, B, n_touch_slots).result_bytes;
}

size_t drx_cdae_prep_part_out_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts) {
  if (!p || B < 1 || n_touch_slots < 0 || parts < 1 || parts > DRX_MAX_WORLD) return 0;
  return part_out_layout(*p, B, n_touch_slots, parts).bytes;
}

int drx_cdae_prep_part_layout(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts, size_t *out4) {
  if (!p || !out4 || B < 1 || n_touch_slots < 0 || parts < 1 || parts > DRX_MAX_WORLD) return DRX_EINVAL;
  const PartOut o = part_out_layout(*p, B, n_touch_slots, parts);
  out4[0] = o.runs_off; out4[1] = o.vals_off; out4[2] = (size_t)o.rcap; out4[3] = (size_t)o.cap;
  return DRX_OK;
}

size_t drx_cdae_prep_part_bytes(const DrxCdaeParams *p, int32_t B, int32_t n_touch_slots, int32_t parts) {
  if (!p || B < 1 || n_touch_slots < 0 || parts < 1 || parts > DRX_MAX_WORLD) return 0;
  Carver c(nullptr, 0);
  (void)part_layout(c, *p, B, n_touch_slots, parts);
  return align_up(c.off, 256) + 256;
}

int drx_cdae_sparse_prepare_part(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, int32_t part, int32_t parts,
                                 void *part_out, size_t part_out_bytes, void *scratch, size_t scratch_bytes, void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !bt->iid || !bt->keep_off || !part_out || !scratch || parts < 1 || parts > DRX_MAX_WORLD || part < 0 || part >= parts)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cv(scratch, scratch_bytes);
  PartBufs L = part_layout(cv, *p, bt->B, bt->n_touch_slots, parts);
  if (!cv.ok() || part_out_bytes < L.out.bytes) return DRX_ESCRATCH;
  const int T = L.R.T, cap = L.out.cap;
  int32_t *header = (int32_t *)part_out;
  unsigned long long *runs = (unsigned long long *)((char *)part_out + L.out.runs_off);
  uint32_t *vals_out = (uint32_t *)((char *)part_out + L.out.vals_off);
  const int gpb = kBlock / 16;
  const dim3 grid((bt->B + gpb - 1) / gpb);
  const uint32_t qthr = q_threshold(bt->q);
  hipLaunchKernelGGL(k_owned_touches<false>, grid, dim3(kBlock), 0, st, p->n_items, *hist, *bt, qthr, part, parts, L.flag, cap, L.ck, L.cv,
                     header);
  rc = scan_i32(L.scan_temp, L.scan_bytes, L.flag, L.flag, (size_t)bt->B, true, st);
  if (rc) return rc;
  DRX_HIP(hipMemsetAsync(L.ck, 0xFF, (size_t)cap * sizeof(uint32_t), st));
  DRX_HIP(hipMemsetAsync(L.cv, 0, (size_t)cap * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_owned_touches<true>, grid, dim3(kBlock), 0, st, p->n_items, *hist, *bt, qthr, part, parts, L.flag, cap, L.ck, L.cv,
                     header);
  rc = sort_pairs(L.R.sort_temp, L.R.sort_bytes, L.ck, L.ck_s, L.cv, vals_out, (size_t)cap, L.R.bits, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_run_flags, dim3(1024), dim3(256), 0, st, L.ck_s, cap, L.flag);
  rc = scan_i32(L.scan_temp, L.scan_bytes, L.flag, L.flag, (size_t)cap, true, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_take_runs, dim3(1024), dim3(256), 0, st, L.ck_s, L.flag, cap, L.out.rcap, runs, header);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_cdae_sparse_prepare_assemble(const DrxCdaeParams *p, const DrxBatch *bt, const void *all_parts, int32_t parts, void *prepared,
                                     size_t prepared_bytes, int32_t *overflow_out, void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  if (!bt || bt->B < 1 || !all_parts || !prepared || !overflow_out || parts < 1 || parts > DRX_MAX_WORLD) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Carver cp(prepared, prepared_bytes);
  PrepBufs R = prep_layout(cp, *p, bt->B, bt->n_touch_slots);
  if (!cp.ok()) return DRX_ESCRATCH;
  const PartOut o = part_out_layout(*p, bt->B, bt->n_touch_slots, parts);
  hipLaunchKernelGGL(k_assemble_parts, dim3(2048), dim3(256), 0, st, (const char *)all_parts, o.bytes, o.runs_off, o.vals_off, parts, R.T,
                     R.keys_s, R.vals_s, overflow_out);
  rc = plan_spans(p, bt, R, st, false);
  if (rc) return rc;
  rc = mark_solo(p, bt, R, st, false);
  if (rc) return rc;
  order_by_degree(bt, R, st, false);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_cdae_step_sparse_prepared(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                                  int32_t loss_kind, const void *prepared, size_t prepared_bytes, void *scratch,
                                  size_t scratch_bytes, float *loss_out, void *const *events, void *stream) {
  if (!prepared) return DRX_EINVAL;
  return step_sparse_impl(p, opt, hist, bt, loss_kind, prepared, prepared_bytes, scratch, scratch_bytes, loss_out, events, stream);
}

int drx_cdae_kshard_forward(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, float *h_out, float *dot_partial,
                            void *stream) {
  return drx_cdae_kshard_forward_prepared(p, hist, bt, nullptr, 0, h_out, dot_partial, stream);
}

int drx_cdae_kshard_forward_prepared(const DrxCdaeParams *p, const DrxHistory *hist, const DrxBatch *bt, const void *prepared,
                                     size_t prepared_bytes, float *h_out, float *dot_partial, void *stream) {
  int rc = check_params(p);
  if (rc) return rc;
  rc = check_batch(hist, bt);
  if (rc || !bt->iid || !bt->keep_off || !h_out || !dot_partial) return DRX_EINVAL;
  const int32_t *order = nullptr;                  // the launch order built with the prepared list (drx_cdae_sparse_prepare*)
  if (prepared) {
    Carver cp(const_cast<void *>(prepared), prepared_bytes);
    const PrepBufs R = prep_layout(cp, *p, bt->B, bt->n_touch_slots);
    if (!cp.ok()) return DRX_ESCRATCH;
    static const bool use_order = [] { const char *e = getenv("DRX_FWD_ORDER"); return !e || atoi(e) != 0; }();      // (A/B switch)
    if (use_order) order = R.order;
  }
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / (1.0f - bt->q);
  const uint32_t qthr = q_threshold(bt->q);
  const long long mean_hist = bt->n_touch_slots / (long long)bt->B;
  const bool per_wg = mean_hist > 64 || (bt->B <= 8192 && mean_hist > 16);      // as in step_sparse_impl
#define CALL(G, J)                                                                                                     \
  {                                                                                                                    \
    const int gpb = kBlock / G;                                                                                        \
    if (per_wg)                                                                                                        \
      hipLaunchKernelGGL((k_kshard_fwd_wg<G, J>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *hist, *bt, scale, \
                         qthr, h_out, dot_partial);                                                                    \
    else                                                                                                               \
      hipLaunchKernelGGL((k_kshard_fwd<G, J>), dim3((bt->B + gpb - 1) / gpb), dim3(kBlock), 0, st, *p, *hist, *bt, scale, qthr, \
                         h_out, dot_partial, order);                                                                   \
  }
  DRX_DISPATCH_GEOM(p->ld, CALL);
#undef CALL
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_cdae_kshard_step(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt, int32_t loss_kind,
                         const float *h, const float *dot_total, const void *prepared, size_t prepared_bytes, void *scratch,
                         size_t scratch_bytes, float *loss_out, void *const *events, void *stream) {
  if (!h || !dot_total) return DRX_EINVAL;
  if (opt && opt->kind == DRX_OPT_ROWWISE_ADAGRAD) return DRX_EINVAL;      // its row statistic would cover the local columns only
  return step_sparse_impl(p, opt, hist, bt, loss_kind, prepared, prepared_bytes, scratch, scratch_bytes, loss_out, events, stream,
                          h, dot_total);
}

int drx_cdae_step_sparse(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                         int32_t loss_kind, void *scratch, size_t scratch_bytes, float *loss_out, void *stream) {
  return step_sparse_impl(p, opt, hist, bt, loss_kind, nullptr, 0, scratch, scratch_bytes, loss_out, nullptr, stream);
}

int drx_cdae_step_sparse_timed(const DrxCdaeParams *p, const DrxOptim *opt, const DrxHistory *hist, const DrxBatch *bt,
                               int32_t loss_kind, void *scratch, size_t scratch_bytes, float *loss_out, void *const *events,
                               void *stream) {
  if (!events) return DRX_EINVAL;
  return step_sparse_impl(p, opt, hist, bt, loss_kind, nullptr, 0, scratch, scratch_bytes, loss_out, events, stream);
}

}  // extern "C"
