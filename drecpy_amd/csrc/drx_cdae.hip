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
#include "drx_prep.hpp"
#include "drx_segstream.hpp"
#include <type_traits>

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
  float *bpart;                               // [n_bpart, ld]
  const uint8_t *solo_v, *solo_o;             // [B] each or nullptr: sample b is the ONLY toucher of its V / W2T row
  unsigned long long *stamps;                 // diagnostic builds (DRX_STAMPS) only
  const int32_t *order;                       // [B] or nullptr: launch order of the forward kernel's triples (k_order_by_degree)
  // DRX_BATCH_SHARE_USERS (k_items_fwd_bwd): the batch's work items (drx_prep.hpp PrepBufs); dz1 then holds B more rows behind the
  // samples': the items' summed gradients.  Else nullptr.
  const int32_t *usamp, *worder, *n_items;
  const WorkItem *witem;
  int T, n_chunks, n_bpart;
};

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

// the lanes of this thread's row group for which f holds (bit i: lane i of the group)
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool f) {
  const unsigned long long m = __ballot(f);
  if (G >= 64) return m;
  const int sh = ((int)(threadIdx.x & 63) / G) * G;
  return (m >> sh) & ((1ull << (G & 63)) - 1ull);
}

// DRX_BATCH_SHARE_USERS: one WORKGROUP per work item — up to kShareTriples (16) triples of ONE user (drx_prep.hpp k_tp_item_*).
// Every row of the user's history is loaded ONCE for all the item's triples (the plain kernel: once per triple):
//   A. the bags of the item's triples are a small masked matrix product [16 triples x history] x [history x K]: the workgroup's waves
//      split the history, 16 rows per wave and round, and v_mfma_f32_16x16x4_f32 adds them into the 16 bags under the 0 / 1 keep
//      coefficients — fp32 products with 0 or 1 are exact; the sums run in the instruction's fixed order.  The coefficient a lane
//      feeds the instruction is the one (triple, row) pair whose mask bit it evaluates: no ballot, no broadcast;
//   B. the waves' partial bags meet in LDS; row group r (the G x J geometry of every other kernel) sums those of triples r, r + R, ...
//      in wave order, then forward / loss / backward as in k_sampled_fwd_bwd;
//   C. the item's summed gradient row dz1[B + item] = sum of its triples' dz1 (per group, then in group order): what the touch
//      list's shared entries (sample field B + item) name.
// At the ml-1m shape (6 040 users, 65 536 triples: 11 per user, 165 rows per history) the plain kernel gathers 8.6 M rows in 350 us;
// summing every user's rows once and subtracting each triple's dropped rows (4.1 M rows): 222 us; the masked product on the vector
// ALUs (a v_readlane + a packed FMA per row and triple: the kernel was bound by instruction issue): 200 us.
template <int G, int J, int KIND = -1>
__global__ __launch_bounds__(kItemThreads, (4 * G * J <= 128 ? 5 : 1)) void k_items_fwd_bwd(DrxCdaeParams P, DrxOptim opt, DrxHistory H, DrxBatch bt, float scale,
                                                          uint32_t qthr, int loss_kind, SparseBufs S) {
  extern __shared__ __align__(16) float lds[];   // [NWV * RT, ld] floats, then RT sample ids
  constexpr int R = kItemThreads / G, NWV = kItemThreads / 64, RT = kShareTriples;
  constexpr int NH = 4 * G * J / 64;                               // blocks of 64 columns of a row
  static_assert(NH >= 1 && NH <= 8 && RT == 16, "rows of 64 .. 512 floats; the 16 x 16 x 4 matrix instruction");
  constexpr int NF = NH <= 2 ? 16 : (NH == 4 ? 8 : 4);             // rows in flight per wave
  constexpr int TPG = (RT + R - 1) / R;                            // triples per row group in B
  if ((int)blockIdx.x >= S.n_items[0]) return;
  const int w = S.worder[blockIdx.x];              // longest histories first
  const WorkItem wi = S.witem[w];
  const int p0 = wi.p0, n = wi.n, u = wi.user;
  const int64_t s = wi.hist_start, e = s + wi.deg;
  const int lane = threadIdx.x % G, r = threadIdx.x / G;
  int *lb = (int *)(lds + (size_t)NWV * RT * P.ld);
  if ((int)threadIdx.x < RT) lb[threadIdx.x] = (int)threadIdx.x < n ? S.usamp[p0 + threadIdx.x] : 0;
  // the user's row of V (B needs it, not the bags): under way before A's rounds
  float4 vrow[J];
  load_row<G, J>(P.V, (size_t)u, P.ld, lane, vrow);
  const int n_waves = (int)min((int64_t)NWV, (e - s + NF - 1) / NF);         // waves that have rows at all
  {  // A
    // v_mfma_f32_16x16x4_f32 per (4 rows, 16 columns): A[m][k] = the keep coefficient of (triple m, row k) — lane l supplies
    // A[l % 16][l / 16], the very pair whose mask bit it evaluates; B[k][n] = lane l's column of row k = l / 16.  A lane loads float4s
    // (columns 64 h + 4 (l % 16) ...), so the tile of (h, j) holds columns 64 h + 4 n + j: C[4 (l / 16) + i][l % 16] in register i.
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int wl = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kr = wl >> 4, nc = wl & 15;            // this lane's row of a group of 4, its triple (A) / its column quad (B)
    f4v acc[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][j] = f4v{0.f, 0.f, 0.f, 0.f};
    const int64_t c0 = s + (int64_t)wv * NF;
    int ridx[NF / 4];
#pragma unroll
    for (int g = 0; g < NF / 4; ++g) ridx[g] = c0 + 4 * g + kr < e ? H.indices[c0 + 4 * g + kr] : 0;
    __syncthreads();                                 // (lb)
    const int bmine = lb[nc];
    for (int64_t c = c0; c < e; c += (int64_t)NWV * NF) {
      const int64_t cn = c + (int64_t)NWV * NF;
      int ridx_n[NF / 4];
#pragma unroll
      for (int g = 0; g < NF / 4; ++g) ridx_n[g] = cn + 4 * g + kr < e ? H.indices[cn + 4 * g + kr] : 0;      // (under way while this round's rows are)
      float4 bv[NF / 4][NH];
#pragma unroll
      for (int g = 0; g < NF / 4; ++g)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          bv[g][h] = f4_zero();
          if (c + 4 * g + kr < e && 64 * h + 4 * nc < P.ld)
            bv[g][h] = *reinterpret_cast<const float4 *>(P.W + (size_t)ridx[g] * P.ld + 64 * h + 4 * nc);
        }
#pragma unroll
      for (int g = 0; g < NF / 4; ++g) {
        const int64_t row = c + 4 * g + kr;
        bool kf = false;
        if (nc < n && row < e) {
          const uint32_t jj = (uint32_t)(row - s);
          kf = bt.keep ? (bt.keep[bt.keep_off[bmine] + jj] != 0) : (hash_u32(bt.mask_seed, (uint32_t)bmine, jj) >= qthr);
        }
        const float a = kf ? 1.0f : 0.0f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[g][h].x, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[g][h].y, acc[h][1], 0, 0, 0);
          acc[h][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[g][h].z, acc[h][2], 0, 0, 0);
          acc[h][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[g][h].w, acc[h][3], 0, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < NF / 4; ++g) ridx[g] = ridx_n[g];
    }
    if (wv < n_waves) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = 4 * kr + i;
        if (t < n) {
#pragma unroll
          for (int h = 0; h < NH; ++h)
            if (64 * h + 4 * nc < P.ld)
              *reinterpret_cast<float4 *>(lds + (size_t)(wv * RT + t) * P.ld + 64 * h + 4 * nc) =
                  make_float4(acc[h][0][i], acc[h][1][i], acc[h][2][i], acc[h][3][i]);
        }
      }
    }
  }
  __syncthreads();
  // B
  float4 dsum[J];
#pragma unroll
  for (int j = 0; j < J; ++j) dsum[j] = f4_zero();
#pragma unroll
  for (int kt = 0; kt < TPG; ++kt) {
    const int t = r + kt * R;
    if (t < n) {
      const int b = lb[t];
      float4 acc[J], h[J], w2[J], dz1[J];
#pragma unroll
      for (int j = 0; j < J; ++j) acc[j] = f4_zero();
      for (int wv = 0; wv < n_waves; ++wv) {
        float4 x[J];
        load_row<G, J>(lds, (size_t)(wv * RT + t), P.ld, lane, x);
#pragma unroll
        for (int j = 0; j < J; ++j) f4_add(acc[j], x[j]);
      }
      float4 brow[J];
      load_row<G, J>(P.b, 0, P.ld, lane, brow);
      hidden_act_rows<G, J>(P, scale, lane, acc, vrow, brow, h);
      load_row<G, J>(P.W2T, (size_t)bt.iid[b], P.ld, lane, w2);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) d += f4_dot(w2[j], h[j]);
      d = group_sum<G>(d);
      sampled_rest<G, J, KIND>(P, opt, H, bt, scale, qthr, loss_kind, S, b, lane, d, h, w2, &dz1);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(dsum[j], dz1[j]);
    }
  }
  // C
  __syncthreads();                                   // (every group has read its partial bags)
  if (r < n) store_row<G, J>(lds, (size_t)r, P.ld, lane, dsum);
  __syncthreads();
  if (r == 0) {
    float4 t[J];
#pragma unroll
    for (int j = 0; j < J; ++j) t[j] = f4_zero();
    for (int rr = 0; rr < min(n, R); ++rr) {
      float4 x[J];
      load_row<G, J>(lds, (size_t)rr, P.ld, lane, x);
#pragma unroll
      for (int j = 0; j < J; ++j) f4_add(t[j], x[j]);
    }
    store_row<G, J>(S.dz1, (size_t)bt.B + w, P.ld, lane, t);
  }
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
  template <int G, int J>
  __device__ __forceinline__ void load(uint32_t key, uint32_t bv, int lane, float4 (&row)[J], float &sc, float &coef) const {
    const uint32_t N = (uint32_t)P.n_items;
    const bool is_out = key >= N && key < 2 * N;
    // (DRX_BATCH_SHARE_USERS lists: the top bit of a W touch's sample field says "subtract"; fields >= B name a user's summed row,
    // stored behind the samples' rows)
    const uint32_t b = bv & 0x7FFFFFFFu;
    load_row<G, J>(dz1 + (is_out ? g2_off : 0ll), (size_t)b, P.ld, lane, row);
    if (is_out) sc = dz2[b];
    coef = key < N ? ((bv >> 31) ? -scale : scale) : 1.0f;
  }
  template <int G, int J>
  __device__ __forceinline__ void finish(uint32_t key, int, int lane, const float4 (&g)[J], float gs) const {
    sparse_apply<G, J, KIND>(P, opt, B, key, lane, g, gs);
  }
  // what the streamed reduction (drx_segstream.hpp) needs to know — the key space as three row arrays: W rows (touches carry the sample
  // whose dz1 row they add, with the coefficient 1/(1-q)), W2T rows (g2 rows, coefficient 1, and the scalar dz2 for the output bias),
  // V rows (dz1 rows, coefficient 1); the optimizer is element-wise with one slot (Adagrad: KIND says so at compile time); a row's
  // own value enters its gradient with reg/B (sparse_apply -> row_update), the output bias's does not
  static constexpr bool kStreamParks = false;                  // every finished row is applied to its table
  __device__ __forceinline__ StreamArrays stream_arrays() const {
    const uint32_t N = (uint32_t)P.n_items;
    return StreamArrays{{0u, N, 2u * N}, {dz1, dz1 + g2_off, dz1}, {P.W, P.W2T, P.V}, {opt.s1[0], opt.s1[1], opt.s1[2]},
                        {scale, 1.0f, 1.0f}, {nullptr, dz2, nullptr}, {nullptr, P.b2, nullptr}, {nullptr, opt.s1[4], nullptr}};
  }
  __device__ __forceinline__ float stream_decay() const { return opt.reg_rate / (float)B; }
  __device__ __forceinline__ void stream_update(float g, float &p, float &a) const {
    static_assert(KIND == DRX_OPT_ADAGRAD || KIND < 0, "one slot per element");
    OptScalars o = opt_for(opt, 0, B);
    float unused = 0.f;
    opt_update1<DRX_OPT_ADAGRAD>(o, g, p, a, unused);
  }
  __device__ __forceinline__ void stream_update_scalar(float g, float &p, float &a) const { stream_update(g, p, a); }      // sparse_apply's output-bias half
};
using DirectPolicy = DirectPolicyT<-1>;                      // optimizer chosen at run time
using DirectPolicyAdagrad = DirectPolicyT<DRX_OPT_ADAGRAD>;  // the throughput configuration's optimizer, known at compile time

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

static SparseBufs sparse_layout(Carver &cv, const DrxCdaeParams &P, int B, int n_touch_slots) {
  SparseBufs S{};
  S.T = n_touch_slots + 2 * B;
  const int chunk = seg_chunk(long_segments(S.T, P));           // (as prep_layout: the list's chunks)
  S.n_chunks = (S.T + chunk - 1) / chunk;
  S.n_bpart = 1024;      // (256: each row group of a bias block summed 32 rows one load at a time; tail_a 25.0 -> 23.5 us)
  S.dz1 = cv.take<float>((size_t)2 * B * P.ld);       // (the second half: the work items' summed rows of DRX_BATCH_SHARE_USERS)
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
    case DRX_ECOMM: return "RCCL transport error (drx_comm_last_error() has the text)";
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

void *drx_stream_create_cu_slice(int32_t cus_per_xcd) {
  constexpr int kXcds = 8, kCusPerXcd = 32;          // gfx950
  if (cus_per_xcd < 1 || cus_per_xcd > kCusPerXcd) return nullptr;
  uint32_t mask[kXcds * kCusPerXcd / 32] = {};
  for (int c = kCusPerXcd - cus_per_xcd; c < kCusPerXcd; ++c)
    for (int x = 0; x < kXcds; ++x) {
      const int bit = c * kXcds + x;
      mask[bit >> 5] |= 1u << (bit & 31);
    }
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)(sizeof(mask) / sizeof(mask[0])), mask) != hipSuccess) return nullptr;
  return (void *)st;
}
void drx_stream_destroy(void *stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

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
  S.order = prepared ? R.order : nullptr;
  SegBufs SB{R.keys_s, R.vals_s, S.phead, S.ptail, S.phs, S.pts, nullptr, nullptr, nullptr, nullptr, S.T, S.n_chunks, p->ld, nullptr};
#ifdef DRX_STAMPS
  S.stamps = SB.stamps = h_stamps;
#endif
  PlanBufs PB{S.pblock, S.pbs};
  // rows collect long runs of touches (MovieLens shapes): k_seg_reduce_planned's LONG form (chunks of 64, 8 rows in flight), XCD placement of its workgroups
  const bool long_segments = drx::long_segments(S.T, *p);
#define EV(i) do { if (events) DRX_HIP(hipEventRecord((hipEvent_t)events[i], st)); } while (0)
  // One workgroup per triple (its groups split the history) instead of one group per triple: when a group would walk many
  // dependent load rounds.  Short histories (mean <= 64 items): only while the batch cannot fill the chip anyway.
  constexpr int wg_long = 64;
  // DRX_BATCH_SHARE_USERS (lists prepared ahead through the history's transpose only): one forward workgroup per work item
  // (k_items_fwd_bwd); the reduction reads the items' summed gradient rows
  // (a list in the shared form has no gradient row per triple: the column-sharded step's forward kernels cannot read it)
  if (prepared && ks_h && share_users(p, hist, bt, R)) return DRX_EINVAL;
  const bool share = prepared && !ks_h && share_users(p, hist, bt, R);
  S.usamp = share ? R.usamp : nullptr;
  S.witem = share ? R.witem : nullptr;
  S.worder = share ? R.worder : nullptr;
  S.n_items = share ? R.n_du : nullptr;
  const long long mean_hist = bt->n_touch_slots / (long long)bt->B;
  const bool per_wg = mean_hist > wg_long || (bt->B <= 8192 && mean_hist > 16);
  BiasArgs BA{S.dz1, S.bpart, S.lossb, loss_out, bt->B, n_bpart, rows_per_block};
  // the segmented reduction (+ the bias column sums as extra workgroups) and the ONE launch that combines the chunk-crossing segments
  // (+ the bias update), with the policy type POLT (optimizer at run time, or Adagrad compiled in)
#define REDUCE_AND_SPANS(G, J, POLT)                                                                                   \
  {                                                                                                                    \
    POLT polk{*p, *opt, bt->B, scale, S.dz1, (long long)(S.g2 - S.dz1), S.dz2};                    \
    BiasPartialExtra<G, J> bpx{p->ld, BA};                                                                             \
    BiasFinalExtra<G, J> bfx{*p, *opt, BA};                                                                            \
    const int cpb = kSegBlock / G;                                                                                     \
    const dim3 rgrid(n_bpart + (S.n_chunks + cpb - 1) / cpb);                                                        \
    const size_t lds_r = seg_reduce_lds_bytes(cpb, p->ld, long_segments);                                              \
    bool streamed = false;                                                                                             \
    if constexpr (kStreamDepth > 0 && J == 1 && G >= 16 && std::is_same<POLT, DirectPolicyAdagrad>::value) {           \
      /* lists of short segments over rows of exactly 64 / 128 / 256 floats: the streamed form (drx_segstream.hpp) */  \
      if (!long_segments && p->ld == 4 * G && bt->B < (1 << kStreamIndexBits) && p->n_items < (1 << kStreamIndexBits) &&           \
          p->n_users < (1 << kStreamIndexBits)) { \
        BiasPartialExtra<G, J, cpb * 64> bpxs{p->ld, BA};                                                              \
        const dim3 sgrid(n_bpart + (S.n_chunks + cpb - 1) / cpb);                                        \
        hipLaunchKernelGGL((k_seg_reduce_stream<4 * G, kStreamDepth, POLT, BiasPartialExtra<G, J, cpb * 64>>), sgrid, dim3(cpb * 64), \
                           seg_stream_lds_bytes(p->ld, kStreamDepth), st, SB, PB, R.plan, polk, n_bpart, bpxs);        \
        streamed = true;                                                                                               \
      }                                                                                                                \
    }                                                                                                                  \
    if (streamed) { }                                                                                                  \
    else if (long_segments)                                                                                            \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, true, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB,    \
                         R.plan, polk, n_bpart, bpx);                                                                  \
    else                                                                                                               \
      hipLaunchKernelGGL((k_seg_reduce_planned<G, J, POLT, false, BiasPartialExtra<G, J>>), rgrid, dim3(kSegBlock), lds_r, st, SB, PB,    \
                         R.plan, polk, n_bpart, bpx);                                                                  \
    EV(3);                                                                                                             \
    if (lds_b > 48 * 1024)                                                                                             \
      DRX_HIP(hipFuncSetAttribute((const void *)k_span_planned<G, J, POLT, BiasFinalExtra<G, J>>,                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                            \
    hipLaunchKernelGGL((k_span_planned<G, J, POLT, BiasFinalExtra<G, J>>), dim3(kLongBlocks + kShortBlocks + 1),                  \
                       dim3(kFixBlock), lds_b, st, SB, PB, R.plan, polk, kLongBlocks, kShortBlocks, bfx);                 \
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
    else if (share) {                                                                                                  \
      /* work items: at most one per distinct user + one per full item, never more than triples */                   \
      const long long wmax = (long long)(p->n_users < bt->B ? p->n_users : bt->B) + bt->B / share_item_triples(p->ld) + 1;                   \
      const dim3 igrid((unsigned)(wmax < bt->B ? wmax : bt->B));                                                       \
      if constexpr (G >= 16 && J <= 2) {                        /* (share_users(): rows of 64 .. 512 floats) */        \
        if (opt->kind == DRX_OPT_ADAGRAD)                                                                              \
          hipLaunchKernelGGL((k_items_fwd_bwd<G, J, DRX_OPT_ADAGRAD>), igrid, dim3(kItemThreads), share_item_lds_bytes(p->ld), st, *p, *opt, \
                             *hist, *bt, scale, qthr, loss_kind, S);                                                   \
        else                                                                                                           \
          hipLaunchKernelGGL((k_items_fwd_bwd<G, J>), igrid, dim3(kItemThreads), share_item_lds_bytes(p->ld), st, *p, *opt, *hist, *bt, \
                             scale, qthr, loss_kind, S);                                                               \
      }                                                                                                                \
    } else if (per_wg)                                                                                                 \
      hipLaunchKernelGGL((k_sampled_fwd_bwd_wg<G, J>), dim3(bt->B), dim3(kBlock), (size_t)gpb * p->ld * 4, st, *p, *opt, *hist, \
                         *bt, scale, qthr, loss_kind, S);                                                              \
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
    order = R.order;
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
