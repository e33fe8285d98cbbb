// Caser training step (caser.py:86-120 under the tape of recommender_abc.py:191-203) on the matrix cores.
//
// A workgroup takes TILES OF 16 SAMPLES.  Every layer of the model is then a product whose M (or K) dimension is the 16 samples of the
// tile, computed with v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) out of LDS:
//   forward   vertical conv      [16 x (t, c)] . [(t, c) x n_v]
//             horizontal convs   per (height i, position t): [16 x (s, c)] . [(s, c) x n_h]      -> max over t, act_h, dropout
//             dense_0            [16 x nx] . [nx x d]
//   targets   per sample, lane = channel (gathered rows of dense_1: not a product over the tile)
//   backward  dense_0            [16 x d] . [d x nx]                                             -> dropout, act_h', scatter to arg-max
//             convs -> item rows per position t': [16 x f] . [f x c] summed over the (i, s) taps that reach t'
//   weights   horizontal kernels per (i, s): [f x (t, 16)] . [(t, 16) x c];  vertical kernels, dense_0 likewise with K = the 16 samples
// The wave-per-sample kernel this replaces (r02 - r04) spent 14 k instructions per sample on channel-parallel FMAs and wave
// reductions and was bound by instruction issue at 110 us for 4096 samples; here a tile costs about 2 000 MFMAs over 8 waves.
//
// MFMA operand layout (lane l): A[l % 16][l / 16], B[l / 16][l % 16], C register r = C[4 (l / 16) + r][l % 16].
// LDS: the convolution weights (a copy of sw[0 .. off_wd), when they fit), the tile's item rows E[b][t][c] with a per-sample stride
// whose quarter is odd (the 16 samples of an A fragment fall into 16 different bank quads), 16 x 16 tiles stored [column][17],
// per-sample vectors x / pre / dx / arg-max with the same kind of stride.  dense_0's kernel is read from global memory (17 KB: L1 / L2).
// Sums over the batch: each tile's 16 samples inside one MFMA (hardware order, fixed), tiles in the order a workgroup takes them,
// workgroups by k_sum_partials in block order — the same bits run after run.
#pragma once
#include "drx_common.hpp"
#include "drx_rows.hpp"

#ifdef DRX_STAMPS
static unsigned long long *h_caser_stamps = nullptr;       // device buffer [tiles x 16] (diagnostic builds: scripts/stamps_caser.py)
extern "C" int drx_debug_set_caser_stamps(unsigned long long *buf) { h_caser_stamps = buf; return 0; }
#define CASER_STAMP_ARG , unsigned long long *stamps
#define CASER_STAMP_PASS , h_caser_stamps
#define TSTAMP(i) DRX_STAMP(stamps, tile_id, i, threadIdx.x)
#else
#define CASER_STAMP_ARG
#define CASER_STAMP_PASS
#define TSTAMP(i) do { } while (0)
#endif

namespace drx {

constexpr int kCaserMaxL = 8;

// Sums 8 per-lane values across the 64 lanes with 10 shuffles instead of 8 full butterflies (48): each exchange halves the number of
// values a lane carries.  Afterwards lane l holds the wave total of v[slot8(l)] (eight lanes per value); lane8(j) is the first lane that
// holds value j.
__device__ __forceinline__ int slot8(int lane) { return ((lane >> 5) & 1) << 2 | ((lane >> 4) & 1) << 1 | ((lane >> 3) & 1); }
__device__ __forceinline__ constexpr int lane8(int j) { return ((j >> 2) & 1) << 5 | ((j >> 1) & 1) << 4 | (j & 1) << 3; }
__device__ __forceinline__ float reduce8(const float (&v)[8], int lane) {
  float a[4], b[2];
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = (h5 ? v[4 + k] : v[k]) + __shfl_xor(h5 ? v[k] : v[4 + k], 32);
#pragma unroll
  for (int k = 0; k < 2; ++k) b[k] = (h4 ? a[2 + k] : a[k]) + __shfl_xor(h4 ? a[k] : a[2 + k], 16);
  float d = (h3 ? b[1] : b[0]) + __shfl_xor(h3 ? b[0] : b[1], 8);
  d += __shfl_xor(d, 4);
  d += __shfl_xor(d, 2);
  d += __shfl_xor(d, 1);
  return d;
}

__device__ __forceinline__ float lane_f(float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

// act_h / act_mlp of caser.py:29-30 (Keras activation names): value and derivative at pre-activation v (a = act(v)).
__device__ __forceinline__ float act_f(int kind, float v) {
  switch (kind) {
    case DRX_ACT_RELU: return fmaxf(v, 0.f);
    case DRX_ACT_TANH: return tanhf(v);
    case DRX_ACT_SIGMOID: return sigmoidf_(v);
    default: return v;
  }
}
__device__ __forceinline__ float act_df(int kind, float v) {
  switch (kind) {
    case DRX_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case DRX_ACT_TANH: { const float a = tanhf(v); return 1.f - a * a; }
    case DRX_ACT_SIGMOID: { const float a = sigmoidf_(v); return a * (1.f - a); }
    default: return 1.f;
  }
}


constexpr int kTileSamples = 16;
constexpr int kCT = 16 * 17;            // floats of one 16 x 16 tile in LDS: [column][17]
constexpr int kSP = 68;                 // per-sample stride of the 64-wide vectors (quarter odd)
#ifndef DRX_CASER_TILE_WAVES
#define DRX_CASER_TILE_WAVES 8
#endif
constexpr int kTileWaves = DRX_CASER_TILE_WAVES;       // 16: one sample per wave in the per-sample phases, 8: two

typedef float f4v __attribute__((ext_vector_type(4)));

struct CaserTileGeom {
  int nwl;          // floats of weights kept in LDS
  int SB;           // per-sample stride of E
  int SX;           // per-sample stride of x / pre / dx / arg
  int KC;           // k-steps over the channels (ceil(d / 4))
  int NC;           // 16-column tiles over the channels (ceil(ld / 16))
  int NTh, NTv;     // 16-filter tiles of a horizontal / the vertical convolution
  int NJ;           // 16-unit tiles over nx
  int TV0;          // first vertical tile in the tile array
  int n_ct;         // tiles in the array
  int o_E, o_PU, o_Z0, o_Z1, o_DZ0, o_XD, o_PRE, o_DX, o_ARG, o_CT;   // float offsets in dynamic LDS
  int floats;
};

__host__ __device__ inline int odd_quarter(int n) {          // n rounded up to a multiple of 4 whose quarter is odd
  n = (n + 3) & ~3;
  return ((n >> 2) & 1) ? n : n + 4;
}

// wl: 2 = every small weight in LDS, 1 = the convolution weights (sw[0 .. off_wd)), 0 = none (read from global memory)
__host__ __device__ inline CaserTileGeom caser_tile_geom(const DrxCaserDims &D, int wl) {
  CaserTileGeom g;
  const int nx = D.n_v + D.L * D.n_h;
  g.nwl = wl == 2 ? D.n_small + 64 : wl == 1 ? D.off_wd + 64 : 0;
  g.SB = odd_quarter(D.L * D.ld);
  g.SX = odd_quarter(nx);
  g.KC = (D.d + 3) / 4;
  g.NC = (D.ld + 15) / 16;
  g.NTh = (D.n_h + 15) / 16;
  g.NTv = (D.n_v + 15) / 16;
  g.NJ = (nx + 15) / 16;
  g.TV0 = (D.L * (D.L + 1) / 2) * g.NTh;
  g.n_ct = g.TV0 + g.NTv;
  int o = g.nwl;
  g.o_E = o; o += kTileSamples * g.SB + 64;
  g.o_PU = o; o += kTileSamples * kSP;
  g.o_Z0 = o; o += kTileSamples * kSP;
  g.o_Z1 = o; o += kTileSamples * kSP;
  g.o_DZ0 = o; o += kTileSamples * kSP;
  g.o_XD = o; o += kTileSamples * g.SX;
  g.o_PRE = o; o += kTileSamples * g.SX;
  g.o_DX = o; o += kTileSamples * g.SX;
  g.o_ARG = o; o += kTileSamples * g.SX;
  g.o_CT = o; o += g.n_ct * kCT;
  g.floats = o;
  return g;
}

__device__ __forceinline__ f4v mfma4(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// N k-steps of one 16 x 16 product: the 2 N fragment loads first (fa(k), fb(k): unconditional loads — a branch per load would put a
// wait in front of every one of them), then the N MFMAs, alternating between two accumulators
template <int N, class FA, class FB>
__device__ __forceinline__ void mfma_chunk(int k0, FA fa, FB fb, f4v &a0, f4v &a1) {
  float av[N], bv[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { av[n] = fa(k0 + n); bv[n] = fb(k0 + n); }
  __builtin_amdgcn_sched_barrier(0);            // (left alone the scheduler pairs every two loads with a wait and two MFMAs)
#pragma unroll
  for (int n = 0; n < N; ++n) {
    if (n & 1) a1 = mfma4(av[n], bv[n], a1);
    else a0 = mfma4(av[n], bv[n], a0);
  }
}
// k-steps [k, K): chunks of 8, one of 4, single steps — no step is padded
template <class FA, class FB>
__device__ __forceinline__ void mfma_k(int k, int K, FA fa, FB fb, f4v &a0, f4v &a1) {
  for (; k + 8 <= K; k += 8) mfma_chunk<8>(k, fa, fb, a0, a1);
  if (k + 4 <= K) { mfma_chunk<4>(k, fa, fb, a0, a1); k += 4; }
  for (; k < K; ++k) mfma_chunk<1>(k, fa, fb, a0, a1);
}

// Workgroup barrier that orders LDS traffic only: the waves of a tile exchange everything through LDS, and a __syncthreads() would also
// wait for every global store in flight (the gradient rows of the lookups, the partial sums) at each of the tile's seven barriers.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// first tile of (height i, position 0) in the tile array, NTh tiles per (i, t)
__device__ __forceinline__ int ct_first(int i, int L, int NTh) { return (i * L - (i * (i - 1)) / 2) * NTh; }

// FX = 1: the dimensions of examples/caser.py:13 (BASELINE configuration 5: L = 5, d = 50, n_v = 4, n_h = 16, T = 3, 9 negatives) as
// compile-time constants — trip counts, strides and tile counts fold, the loops over heights / positions / taps unroll; FX = 0: any.
template <int WL, int FX>
__global__ __launch_bounds__(64 * kTileWaves) void k_caser_tile(DrxCaserDims D, DrxCaserArgs A CASER_STAMP_ARG) {
  // (D stays the kernel argument — its offset arrays are indexed with run-time heights, which would send a modified copy to scratch
  //  memory, every field read a scratch load; the dimensions the instantiation fixes are local constants instead)
  const int L = FX ? 5 : D.L, d = FX ? 50 : D.d, ld = FX ? 52 : D.ld, ld2 = FX ? 100 : D.ld2, n_v = FX ? 4 : D.n_v, n_h = FX ? 16 : D.n_h;
  const int T = FX ? 3 : D.T, Tp = FX ? 12 : D.Tp, nx = n_v + L * n_h;
  DrxCaserDims Dg = {};                                // (what the LDS geometry depends on)
  Dg.L = L; Dg.d = d; Dg.ld = ld; Dg.n_v = n_v; Dg.n_h = n_h; Dg.n_small = D.n_small; Dg.off_wd = D.off_wd;
  constexpr int NW = kTileWaves, SPW = kTileSamples / NW;          // samples per wave in the per-sample phases
  extern __shared__ __align__(16) float lds[];
  const CaserTileGeom G = caser_tile_geom(Dg, WL);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int SB = G.SB, SX = G.SX, KC = G.KC, NC = G.NC, NTh = G.NTh, NTv = G.NTv, NJ = G.NJ, TV0 = G.TV0;
  float *const wl = lds;
  float *const E = lds + G.o_E, *const PU = lds + G.o_PU, *const Z0 = lds + G.o_Z0, *const Z1 = lds + G.o_Z1, *const DZ0 = lds + G.o_DZ0;
  float *const XD = lds + G.o_XD, *const PRE = lds + G.o_PRE, *const DX = lds + G.o_DX, *const CT = lds + G.o_CT;
  int *const ARG = reinterpret_cast<int *>(lds + G.o_ARG);
  __shared__ float wloss[NW];
  // a convolution weight / a dense_0 weight
  auto cw = [&](int off) __attribute__((always_inline)) -> float { return WL >= 1 ? wl[off] : A.sw[off]; };
  auto gw = [&](int off) __attribute__((always_inline)) -> float { return WL == 2 ? wl[off] : A.sw[off]; };

  float loss_acc = 0.f;
  const float inv_bt = 1.0f / ((float)A.B * (float)Tp);
  const float inv_keep = 1.0f / (1.0f - A.rate);
  const bool hashed = !A.keep && A.rate > 0.f;
  const uint32_t rthr = hashed ? q_threshold(A.rate) : 0u;
  auto kept = [&](int gb, int j) __attribute__((always_inline)) -> bool {
    if (gb >= A.B) return false;                       // (a sample of the tile's padding: nothing of it is used)
    if (A.keep) return A.keep[(size_t)gb * nx + j] != 0;
    if (hashed) return hash_u32(A.mask_seed, (uint32_t)gb, (uint32_t)j) >= rthr;
    return true;
  };
  // the rows of dense_1 of eight targets of sample gb (targets j0 .. j0 + 7) and, in the lanes that will hold target jm's score, its bias
  auto load_targets = [&](int gb, int j0, float (&wa)[8], float (&wb)[8], float &bm) __attribute__((always_inline)) {
    // (clamped indices, unconditional loads, the value selected afterwards: a load under a predicate becomes a branch, and the counter
    //  waits behind branches drain every load in flight)
    const bool live = lane < d;
    const int cl = min(lane, d - 1), gbc = min(gb, A.B - 1);
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const bool on = live && gb < A.B && j0 + qq < Tp;
      const int n = A.after[(size_t)gbc * Tp + min(j0 + qq, Tp - 1)];
      const float va = A.W1[(size_t)n * ld2 + cl], vb = A.W1[(size_t)n * ld2 + d + cl];
      wa[qq] = on ? va : 0.f;
      wb[qq] = on ? vb : 0.f;
    }
    const int jm = j0 + slot8(lane);
    const float vbm = A.b1[A.after[(size_t)gbc * Tp + min(jm, Tp - 1)]];
    bm = (gb < A.B && jm < Tp) ? vbm : 0.f;
  };
  float *const gp = A.gsw_part + (size_t)blockIdx.x * D.n_small;
  const int n_tiles = (A.B + kTileSamples - 1) / kTileSamples;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b0 = tile * kTileSamples;
    const bool first = tile == (int)blockIdx.x;
    [[maybe_unused]] const int tile_id = tile;
    TSTAMP(0);
#ifdef DRX_CASER_EXIT_EARLY                                  // (diagnostic builds: what an empty launch of this shape costs)
    if (A.B > 0) break;
#endif
    // (the lane's coordinates in an MFMA fragment, opaque per tile: what is derived from them is computed where it is used instead of
    //  being hoisted — by the dozen — in front of the tile loop)
    int m16 = lane & 15, q4 = lane >> 4;
    asm volatile("" : "+v"(m16), "+v"(q4));
    // ---- 0. the tile's item rows and user rows: every wave its samples, all row reads in flight together, with the weights' copy
    //         into LDS (first tile) ------------------------------------------------------------------------------------------------
    // Loads in the order of their dependence chains: the samples' indices, the weights (first tile), the item / user rows (they need
    // the indices).  Loads return in order, so the stores into LDS below wait for no more than they need.  (The dense_1 rows of the
    // targets — 64 more loads per wave, used in step 4 — are issued behind the barrier: in front of it they held up these stores.)
    constexpr int WCH = 10;                               // float4s of weights a thread has in flight: 80 KB per workgroup and pass
    const int nw = WL == 2 ? D.n_small : WL == 1 ? D.off_wd : 0;
    int ib[SPW], ia[SPW], iu[SPW];
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int gb = min(b0 + w + NW * k, A.B - 1);       // (a sample of the tile's padding reads the last sample's indices; its rows are zeroed below)
      ib[k] = A.before[(size_t)gb * L + min(lane, L - 1)];
      ia[k] = A.after[(size_t)gb * Tp + min(lane, Tp - 1)];                     // (lane j: target j; the first 16 are used here)
      iu[k] = A.uid[gb];
    }
    float e[SPW][kCaserMaxL], pu[SPW];
    auto fetch_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const bool on = b0 + w + NW * k < A.B && lane < d;
        const int cl = min(lane, d - 1);
#pragma unroll
        for (int t = 0; t < kCaserMaxL; ++t) {
          const int n = __shfl(ib[k], min(t, L - 1));
          const float v = A.item_emb[(size_t)n * ld + cl];
          e[k][t] = (t < L && on) ? v : 0.f;
        }
        const float vu = A.user_emb[(size_t)iu[k] * ld + cl];
        pu[k] = on ? vu : 0.f;
      }
    };
    if (first && WL) {
      // (one block from the weights' loads to their stores: registers; split around the rows' loads the buffer went to scratch memory.
      //  Clamped indices: the last float4 may be read and written more than once — no branch per load or store, so the wait in front of
      //  the stores counts loads instead of draining them all)
      float4 wbuf[WCH];
#pragma unroll
      for (int c = 0; c < WCH; ++c)                       // (every segment of sw is a multiple of 4 floats long)
        wbuf[c] = *reinterpret_cast<const float4 *>(A.sw + min((int)(threadIdx.x + c * blockDim.x) * 4, nw - 4));
      fetch_rows();
#pragma unroll
      for (int c = 0; c < WCH; ++c)
        *reinterpret_cast<float4 *>(wl + min((int)(threadIdx.x + c * blockDim.x) * 4, nw - 4)) = wbuf[c];
      for (int i = (int)(threadIdx.x + WCH * blockDim.x) * 4; i < nw; i += blockDim.x * 4)      // (more than 80 KB of weights)
        *reinterpret_cast<float4 *>(wl + i) = *reinterpret_cast<const float4 *>(A.sw + i);
      if (threadIdx.x < 64) wl[nw + threadIdx.x] = 0.f;
    } else {
      fetch_rows();
    }
    if (first) {
      // E's padding (the gap behind every sample's rows, the tail a 16-column read runs into), x's k-padding and the tiles' unused
      // columns stay zero throughout (none of it is a place a row is written to: no barrier between this and the rows' stores)
      const int gapE = SB - L * ld, gapX = SX - nx;
      for (int i = threadIdx.x; i < kTileSamples * gapE; i += blockDim.x) E[(i / gapE) * SB + L * ld + i % gapE] = 0.f;
      if (threadIdx.x < 64) E[kTileSamples * SB + threadIdx.x] = 0.f;
      for (int i = threadIdx.x; i < kTileSamples * gapX; i += blockDim.x) XD[(i / gapX) * SX + nx + i % gapX] = 0.f;
      for (int i = threadIdx.x; i < G.n_ct * kCT; i += blockDim.x) CT[i] = 0.f;
    }
    TSTAMP(1);
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int b = w + NW * k;
#pragma unroll
      for (int t = 0; t < kCaserMaxL; ++t)
        if (t < L && lane < ld) E[b * SB + t * ld + lane] = e[k][t];
      PU[b * kSP + lane] = pu[k];
    }
    lds_barrier();
    TSTAMP(2);
    // the dense_1 rows (and biases) of the first 8 targets of every sample: issued now, used in step 4 (the next 8 behind step 1: a wave
    // cannot have more than 63 loads in flight, and would wait here for the first ones to return)
    float pwa[SPW][2][8], pwb[SPW][2][8], b1v[SPW];
    auto fetch_targets = [&](int r) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const bool on = b0 + w + NW * k < A.B && lane < d;
        const int cl = min(lane, d - 1);
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
          const int n = __shfl(ia[k], 8 * r + qq);                      // (lanes beyond Tp hold target Tp - 1: a valid row)
          const bool t_on = on && 8 * r + qq < Tp;
          const float va = A.W1[(size_t)n * ld2 + cl], vb = A.W1[(size_t)n * ld2 + d + cl];
          pwa[k][r][qq] = t_on ? va : 0.f;
          pwb[k][r][qq] = t_on ? vb : 0.f;
        }
      }
    };
#pragma unroll
    for (int k = 0; k < SPW; ++k) b1v[k] = A.b1[ia[k]];                  // (lane j: the bias of target j; lanes beyond Tp are not read)
    fetch_targets(0);
    // ---- 1. convolutions forward.  One unit = a 16-filter tile of the vertical conv, or of the horizontal conv of height i with its
    //         positions t in turn: 16 samples x 16 filters of pre-activations per position in the MFMA's accumulators, the lane keeps
    //         the running maximum of its four (sample, filter) pairs — the first maximum wins, like the max-pool gradient — and writes
    //         x after dropout, the pre-activation at the arg-max and the arg-max (caser.py:103-114) ------------------------------------
    {
      int u = 0;
      for (int nt = 0; nt < NTv; ++nt, ++u) {
        if (u % NW != w) continue;
        const int f = 16 * nt + m16;
        const bool fv = f < n_v;
        f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int t = 0; t < L; ++t) {
          const float *const ep = E + m16 * SB + t * ld + q4;
          const int wo = D.off_kv + (t * n_v + (fv ? f : 0)) * ld + q4;
          mfma_k(0, KC, [&](int k) __attribute__((always_inline)) { return ep[4 * k]; },
                 [&](int k) __attribute__((always_inline)) { const float v = cw(wo + 4 * k); return fv ? v : 0.f; }, a0, a1);
        }
        if (fv) {
          const float bias = cw(D.off_bv + f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int b = 4 * q4 + r;
            XD[b * SX + f] = kept(b0 + b, f) ? (a0[r] + a1[r] + bias) * inv_keep : 0.f;
          }
        }
      }
      for (int i = L - 1; i >= 0; --i)
        for (int nt = 0; nt < NTh; ++nt, ++u) {
          if (u % NW != w) continue;
          const int f = 16 * nt + m16;
          const bool fv = f < n_h;
          const float bias = fv ? cw(D.off_bh[i] + f) : 0.f;
          float best[4], bpre[4];
          int barg[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { best[r] = -3.0e38f; bpre[r] = 0.f; barg[r] = 0; }
          for (int t = 0; t + i < L; ++t) {
            f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            for (int s = 0; s <= i; ++s) {
              const float *const ep = E + m16 * SB + (t + s) * ld + q4;
              const int wo = D.off_kh[i] + (s * n_h + (fv ? f : 0)) * ld + q4;
              mfma_k(0, KC, [&](int k) __attribute__((always_inline)) { return ep[4 * k]; },
                     [&](int k) __attribute__((always_inline)) { const float v = cw(wo + 4 * k); return fv ? v : 0.f; }, a0, a1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = a0[r] + a1[r] + bias;
              const float rr = act_f(D.act_h, v);
              if (rr > best[r]) { best[r] = rr; barg[r] = t; bpre[r] = v; }
            }
          }
          if (fv) {
            const int j = n_v + i * n_h + f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int b = 4 * q4 + r;
              XD[b * SX + j] = kept(b0 + b, j) ? best[r] * inv_keep : 0.f;
              PRE[b * SX + j] = bpre[r];
              ARG[b * SX + j] = barg[r];
            }
          }
        }
    }
    lds_barrier();
    TSTAMP(3);
    fetch_targets(1);
    // ---- 3. dense_0: [16 x nx] . [nx x 16 channels] per channel tile, the k range in two halves (Z0 = first half + bias, Z1 = second) --
    {
      const int KX = (nx + 3) >> 2, KXh = (KX + 1) >> 1;
      for (int u = w; u < 2 * NC; u += NW) {
        const int nc = u >> 1, half = u & 1;
        const int cc = 16 * nc + m16;
        const bool cv = cc < ld;
        const int lo = half * KXh, hi = half ? KX : KXh;
        f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const float *const xa = XD + m16 * SX + q4;
        const int ccs = cv ? cc : 0;
        mfma_k(lo, hi, [&](int k) __attribute__((always_inline)) { return xa[4 * k]; },
               [&](int k) __attribute__((always_inline)) {
                 const int j = 4 * k + q4;
                 const float v = gw(D.off_wd + min(j, nx - 1) * ld + ccs);
                 return (cv && j < nx) ? v : 0.f;
               }, a0, a1);
        const float bias = (cc < d && !half) ? gw(D.off_bd + cc) : 0.f;
        float *const zo = half ? Z1 : Z0;
#pragma unroll
        for (int r = 0; r < 4; ++r) zo[(4 * q4 + r) * kSP + cc] = cc < d ? a0[r] + a1[r] + bias : 0.f;
      }
    }
    lds_barrier();
    TSTAMP(5);
    // ---- 4. targets: score, sigmoid, Keras BCE, backward to the lookups — per sample, lane = channel (caser.py:115-120) ------------
    // eight targets at a time: the eight dot products leave through ONE reduce8, the lanes that hold target j's score do its sigmoid /
    // loss / gradient, every lane then fetches the eight gradients with v_readlane.  (The rows of the first sixteen came with step 0.)
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int b = w + NW * k, gb = b0 + b;
      float dz0 = 0.f;
      if (gb < A.B) {
        const int c = lane;
        const bool live = c < d;
        const float z0 = live ? Z0[b * kSP + c] + Z1[b * kSP + c] : 0.f, pu1 = PU[b * kSP + c];
        const float z = live ? act_f(D.act_mlp, z0) : 0.f;
        float dz = 0.f, dpu = 0.f;
        auto round8 = [&](int j0, const float (&wa)[8], const float (&wb)[8], float bm) __attribute__((always_inline)) {
          const int jm = j0 + slot8(c);                           // the target whose score this lane receives
          float prod[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) prod[qq] = fmaf(z, wa[qq], pu1 * wb[qq]);
          const float sc = reduce8(prod, c) + bm;
          const float p = sigmoidf_(sc);
          const float y = jm < T ? 1.f : 0.f;
          const float dsm = jm < Tp ? bce_grad(y, p) * inv_bt * p * (1.f - p) : 0.f;
          if ((c & 7) == 0 && jm < Tp) {
            loss_acc += bce_elem(y, p);
            A.db1[(size_t)gb * Tp + jm] = dsm;
          }
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            if (j0 + qq < Tp) {
              const float ds = lane_f(dsm, lane8(qq));
              const size_t row = (size_t)gb * Tp + j0 + qq;
              if (live && A.dW1) { A.dW1[row * ld2 + c] = ds * z; A.dW1[row * ld2 + d + c] = ds * pu1; }
              dz = fmaf(ds, wa[qq], dz);
              dpu = fmaf(ds, wb[qq], dpu);
            }
          }
        };
        round8(0, pwa[k][0], pwb[k][0], __shfl(b1v[k], slot8(c)));
        if (Tp > 8) round8(8, pwa[k][1], pwb[k][1], __shfl(b1v[k], 8 + slot8(c)));
#pragma unroll 1
        for (int j0 = 16; j0 < Tp; j0 += 8) {
          float wa[8], wb[8], bm;
          load_targets(gb, j0, wa, wb, bm);
          round8(j0, wa, wb, bm);
        }
        if (live) A.dPu[(size_t)gb * ld + c] = dpu;
        if (live && A.cat_out) { A.cat_out[(size_t)gb * ld2 + c] = z; A.cat_out[(size_t)gb * ld2 + d + c] = pu1; }
        dz0 = live ? dz * act_df(D.act_mlp, z0) : 0.f;
      }
      DZ0[b * kSP + lane] = dz0;
      if (k == 0) TSTAMP(4);
    }
    lds_barrier();
    TSTAMP(6);
    // ---- 5. dense_0 backward: dx[16 x 16 units] = dz0[16 x c] . Wd^T, through the dropout mask; a horizontal unit goes on through
    //         act_h at its arg-max step: the tile array receives dC[(i, t)][f][b] (zero away from the arg-max) and dV[f][b], DX the
    //         pre-activation gradients (the bias sums of step 8 read them) ------------------------------------------------------------
    for (int nt = w; nt < NJ; nt += NW) {
      const int jj = 16 * nt + m16;
      const bool jv = jj < nx;
      f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float *const za = DZ0 + m16 * kSP + q4;
      const int wo = D.off_wd + (jv ? jj : 0) * ld + q4;
      mfma_k(0, KC, [&](int k) __attribute__((always_inline)) { return za[4 * k]; },
             [&](int k) __attribute__((always_inline)) { const float v = gw(wo + 4 * k); return jv ? v : 0.f; }, a0, a1);
      if (jv) {
        const int pq = jj - n_v, i = pq >= 0 ? pq / n_h : 0, f = pq - i * n_h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = 4 * q4 + r;
          const float g = kept(b0 + b, jj) ? (a0[r] + a1[r]) * inv_keep : 0.f;
          if (pq < 0) {
            DX[b * SX + jj] = g;
            CT[(TV0 + (jj >> 4)) * kCT + (jj & 15) * 17 + b] = g;
          } else {
            const float dc = g * act_df(D.act_h, PRE[b * SX + jj]);
            const int arg = ARG[b * SX + jj];
            DX[b * SX + jj] = dc;
            float *const c0 = CT + (ct_first(i, L, NTh) + (f >> 4)) * kCT + (f & 15) * 17 + b;
            for (int t = 0; t + i < L; ++t) c0[t * NTh * kCT] = t == arg ? dc : 0.f;
          }
        }
      }
    }
    lds_barrier();
    TSTAMP(7);
    // ---- 7 + 8. one list of units dealt out over the waves:
    //   gradient rows of the item lookups, dE[16 x 16 channels] at position t' = dV . Kv[t'] + sum over the taps (i, s) that reach t' of
    //     dC[(i, t' - s)] . Kh_i[s];
    //   small-weight gradients of the tile (K = the 16 samples; the channel tiles of a unit share its A fragments), added to the
    //     workgroup's partial sums -------------------------------------------------------------------------------------------------
    {
      int u = 0;
      for (int tp = 0; tp < L; ++tp)
        for (int nc = 0; nc < NC; ++nc, ++u) {
          if (u % NW != w) continue;
          const int cc = 16 * nc + m16;
          const bool cv = cc < ld;
          const int cs = cv ? cc : 0;
          f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
          {
            const float *const c0 = CT + TV0 * kCT + m16;
            const int wo = D.off_kv + tp * n_v * ld + cs;
            mfma_k(0, (n_v + 3) >> 2, [&](int k) __attribute__((always_inline)) { const int f = 4 * k + q4; return c0[(f >> 4) * kCT + (f & 15) * 17]; },
                   [&](int k) __attribute__((always_inline)) {
                     const int f = 4 * k + q4;
                     const float v = cw(wo + min(f, n_v - 1) * ld);
                     return (cv && f < n_v) ? v : 0.f;
                   }, a0, a1);
          }
          for (int i = 0; i < L; ++i)
            for (int s = 0; s <= i; ++s) {
              const int t = tp - s;
              if (t < 0 || t + i >= L) continue;
              const float *const c0 = CT + (ct_first(i, L, NTh) + t * NTh) * kCT + m16;
              const int wo = D.off_kh[i] + s * n_h * ld + cs;
              mfma_k(0, (n_h + 3) >> 2, [&](int k) __attribute__((always_inline)) { const int f = 4 * k + q4; return c0[(f >> 4) * kCT + (f & 15) * 17]; },
                     [&](int k) __attribute__((always_inline)) {
                       const int f = 4 * k + q4;
                       const float v = cw(wo + min(f, n_h - 1) * ld);
                       return (cv && f < n_h) ? v : 0.f;
                     }, a0, a1);
            }
          if (cc < d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int gb = b0 + 4 * q4 + r;
              if (gb < A.B) A.dE[((size_t)gb * L + tp) * ld + cc] = a0[r] + a1[r];
            }
          }
        }
      TSTAMP(8);
      // three kinds of units, one loop: A = 16 weights x 16 samples (a tile of dC / dV, or 16 units of x), B = 16 samples x 16 channels
      // (item rows at one position, or dz0), summed over `nit` positions
      //   horizontal kernels  g[(i, s)][f][c] = sum over t, b of dC[(i, t)][f][b] * E[b][t + s][c]
      //   vertical kernel     g[t][f][c]      = sum over b of dV[f][b] * E[b][t][c]
      //   dense_0 kernel      g[j][c]         = sum over b of xd[b][j] * dz0[b][c]
      const int nH = (L * (L + 1) / 2) * NTh, nV = L * NTv;
#pragma unroll 1
      for (int v = 0; v < nH + nV + NJ; ++v, ++u) {
        if (u % NW != w) continue;
        const float *ap, *bp;
        int ak, at, bk, bt, nit, off, row0, n_rows;
        if (v < nH) {
          int r = v / NTh, i = 0;
          const int nt = v - r * NTh;
          while (r > i) { r -= i + 1; ++i; }
          ap = CT + (ct_first(i, L, NTh) + nt) * kCT + m16 * 17 + q4; ak = 4; at = NTh * kCT;
          bp = E + q4 * SB + r * ld + m16; bk = 4 * SB; bt = ld;
          nit = L - i; off = D.off_kh[i] + r * n_h * ld; row0 = 16 * nt; n_rows = n_h;
        } else if (v < nH + nV) {
          const int t = (v - nH) / NTv, nt = (v - nH) - t * NTv;
          ap = CT + (TV0 + nt) * kCT + m16 * 17 + q4; ak = 4; at = 0;
          bp = E + q4 * SB + t * ld + m16; bk = 4 * SB; bt = 0;
          nit = 1; off = D.off_kv + t * n_v * ld; row0 = 16 * nt; n_rows = n_v;
        } else {
          const int nt = v - nH - nV, ja = min(16 * nt + m16, nx - 1);        // (rows beyond nx are computed and not stored)
          ap = XD + q4 * SX + ja; ak = 4 * SX; at = 0;
          bp = DZ0 + q4 * kSP + m16; bk = 4 * kSP; bt = 0;
          nit = 1; off = D.off_wd; row0 = 16 * nt; n_rows = nx;
        }
        f4v acc[4];
#pragma unroll
        for (int nc = 0; nc < 4; ++nc) acc[nc] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < nit; ++it, ap += at, bp += bt) {
          float av[4], bv[4][4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            av[kb] = ap[kb * ak];
#pragma unroll
            for (int nc = 0; nc < 4; ++nc) bv[kb][nc] = bp[kb * bk + 16 * nc];     // (tiles beyond NC: finite values, never stored)
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int nc = 0; nc < 4; ++nc) acc[nc] = mfma4(av[kb], bv[kb][nc], acc[nc]);
        }
#pragma unroll
        for (int nc = 0; nc < 4; ++nc)
          if (nc < NC) {
            const int cc = 16 * nc + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = row0 + 4 * q4 + r;
              if (row < n_rows && cc < ld) {
                float *const o = gp + off + row * ld + cc;
                *o = (first ? 0.f : *o) + acc[nc][r];
              }
            }
          }
      }
      TSTAMP(9);
      // biases (and the zero padding of their segments): one thread per entry, the tile's samples in order
      const int padv = (n_v + 3) & ~3, padh = (n_h + 3) & ~3;
      for (int en = threadIdx.x; en < ld + padv + L * padh; en += blockDim.x) {
        float acc = 0.f;
        float *o;
        if (en < ld) {
          o = gp + D.off_bd + en;
          for (int b = 0; b < kTileSamples; ++b) acc += DZ0[b * kSP + en];
        } else if (en < ld + padv) {
          const int f = en - ld;
          o = gp + D.off_bv + f;
          if (f < n_v)
            for (int b = 0; b < kTileSamples; ++b) acc += DX[b * SX + f];
        } else {
          const int q = en - ld - padv, i = q / padh, f = q - i * padh;
          o = gp + D.off_bh[i] + f;
          if (f < n_h)
            for (int b = 0; b < kTileSamples; ++b) acc += DX[b * SX + n_v + i * n_h + f];
        }
        *o = (first ? 0.f : *o) + acc;
      }
    }
    lds_barrier();                                    // (the next tile rewrites E and the vectors)
    TSTAMP(10);
  }
  loss_acc = group_sum<64>(loss_acc);                 // (the lanes that held a target's score carry its loss term)
  if (lane == 0) wloss[w] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int ww = 0; ww < NW; ++ww) t += wloss[ww];
    A.loss_part[blockIdx.x] = t * inv_bt;
  }
}

}  // namespace drx
