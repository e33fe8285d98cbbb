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
static unsigned long long *h_caser_stamps = nullptr;       // device buffer [B x 16] (diagnostic builds: scripts/stamps_caser.py)
extern "C" int drx_debug_set_caser_stamps(unsigned long long *buf) { h_caser_stamps = buf; return 0; }
#define CASER_STAMP_ARG , unsigned long long *stamps
#define CASER_STAMP_PASS , h_caser_stamps
#define CSTAMP(i) DRX_STAMP(stamps, b, i, c)
#define TSTAMP(i) DRX_STAMP(stamps, blockIdx.x, i, threadIdx.x)
#else
#define CASER_STAMP_ARG
#define CASER_STAMP_PASS
#define CSTAMP(i) do { } while (0)
#define TSTAMP(i) do { } while (0)
#endif

namespace drx {

constexpr int kCaserMaxL = 8;

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// Sums 16 per-lane values across the 64 lanes with 17 shuffles instead of 16 full butterflies (96): each exchange halves
// the number of values a lane carries.  Afterwards lane l holds the wave total of v[slot16(l)] (four lanes per value).
__device__ __forceinline__ int slot16(int lane) { return ((lane >> 5) & 1) << 3 | ((lane >> 4) & 1) << 2 | ((lane >> 3) & 1) << 1 | ((lane >> 2) & 1); }
__device__ __forceinline__ float reduce16(const float (&v)[16], int lane) {
  float a[8], b[4], c[2];
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (h5 ? v[8 + k] : v[k]) + __shfl_xor(h5 ? v[k] : v[8 + k], 32);
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = (h4 ? a[4 + k] : a[k]) + __shfl_xor(h4 ? a[k] : a[4 + k], 16);
#pragma unroll
  for (int k = 0; k < 2; ++k) c[k] = (h3 ? b[2 + k] : b[k]) + __shfl_xor(h3 ? b[k] : b[2 + k], 8);
  float d = (h2 ? c[1] : c[0]) + __shfl_xor(h2 ? c[0] : c[1], 4);
  d += __shfl_xor(d, 2);
  d += __shfl_xor(d, 1);
  return d;
}

// The same for 8 values (10 shuffles): afterwards lane l holds the wave total of v[slot8(l)] (eight lanes per value); lane8(j) is
// the first lane that holds value j.
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

__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

constexpr int kTileWaves = 8;
constexpr int kTileSamples = 16;
constexpr int kCT = 16 * 17;            // floats of one 16 x 16 tile in LDS: [column][17]
constexpr int kSP = 68;                 // per-sample stride of the 64-wide vectors (quarter odd)

typedef float f4v __attribute__((ext_vector_type(4)));

struct CaserTileGeom {
  int nwl;          // floats of weights kept in LDS (0: read from global)
  int SB;           // per-sample stride of E
  int SX;           // per-sample stride of x / pre / dx / arg
  int KC;           // k-steps over the channels (ceil(d / 4))
  int NC;           // 16-column tiles over the channels (ceil(ld / 16))
  int NTh, NTv;     // 16-filter tiles of a horizontal / the vertical convolution
  int NJ;           // 16-unit tiles over nx
  int TV0;          // first vertical tile in the tile array
  int n_ct;         // tiles in the array
  int o_E, o_PU, o_Z0, o_Z, o_DZ0, o_XD, o_PRE, o_DX, o_ARG, o_CT;   // float offsets in dynamic LDS
  int floats;
};

__host__ __device__ inline int odd_quarter(int n) {          // n rounded up to a multiple of 4 whose quarter is odd
  n = (n + 3) & ~3;
  return ((n >> 2) & 1) ? n : n + 4;
}

__host__ __device__ inline CaserTileGeom caser_tile_geom(const DrxCaserDims &D, bool weights_in_lds) {
  CaserTileGeom g;
  const int nx = D.n_v + D.L * D.n_h;
  g.nwl = weights_in_lds ? D.off_wd + 64 : 0;
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
  g.o_Z = o; o += kTileSamples * kSP;
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

// first tile of (height i, position 0) in the tile array, NTh tiles per (i, t)
__device__ __forceinline__ int ct_first(int i, int L, int NTh) { return (i * L - (i * (i - 1)) / 2) * NTh; }

template <bool WL>
__global__ __launch_bounds__(64 * kTileWaves) void k_caser_tile(DrxCaserDims D, DrxCaserArgs A CASER_STAMP_ARG) {
  extern __shared__ __align__(16) float lds[];
  const CaserTileGeom G = caser_tile_geom(D, WL);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m16 = lane & 15, q4 = lane >> 4;
  const int L = D.L, d = D.d, ld = D.ld, n_v = D.n_v, n_h = D.n_h, nx = n_v + L * n_h;
  const int SB = G.SB, SX = G.SX, KC = G.KC, NC = G.NC, NTh = G.NTh, NTv = G.NTv, NJ = G.NJ, TV0 = G.TV0;
  float *const wl = lds;
  float *const E = lds + G.o_E, *const PU = lds + G.o_PU, *const Z0 = lds + G.o_Z0, *const Z = lds + G.o_Z, *const DZ0 = lds + G.o_DZ0;
  float *const XD = lds + G.o_XD, *const PRE = lds + G.o_PRE, *const DX = lds + G.o_DX, *const CT = lds + G.o_CT;
  int *const ARG = reinterpret_cast<int *>(lds + G.o_ARG);
  __shared__ float wloss[kTileWaves];
  // a convolution weight / a dense_0 weight
  auto cw = [&](int off) __attribute__((always_inline)) -> float { return WL ? wl[off] : A.sw[off]; };
  auto gw = [&](int off) __attribute__((always_inline)) -> float { return A.sw[off]; };

  TSTAMP(0);
  if (WL) {
    for (int i = threadIdx.x * 4; i < D.off_wd; i += blockDim.x * 4)   // (every segment of sw is a multiple of 4 floats long)
      *reinterpret_cast<float4 *>(wl + i) = *reinterpret_cast<const float4 *>(A.sw + i);
    if (threadIdx.x < 64) wl[D.off_wd + threadIdx.x] = 0.f;
  }
  // E's padding (gaps between samples, the tail a 16-column read runs into) and x's k-padding stay zero throughout
  for (int i = threadIdx.x; i < kTileSamples * SB + 64; i += blockDim.x) E[i] = 0.f;
  for (int i = threadIdx.x; i < kTileSamples * SX; i += blockDim.x) XD[i] = 0.f;
  __syncthreads();
  TSTAMP(1);

  float loss_acc = 0.f;
  const float inv_bt = 1.0f / ((float)A.B * (float)D.Tp);
  const float inv_keep = 1.0f / (1.0f - A.rate);
  const bool hashed = !A.keep && A.rate > 0.f;
  const uint32_t rthr = hashed ? q_threshold(A.rate) : 0u;
  auto kept = [&](int gb, int j) __attribute__((always_inline)) -> bool {
    if (gb >= A.B) return false;                       // (a sample of the tile's padding: nothing of it is used)
    if (A.keep) return A.keep[(size_t)gb * nx + j] != 0;
    if (hashed) return hash_u32(A.mask_seed, (uint32_t)gb, (uint32_t)j) >= rthr;
    return true;
  };
  float *const gp = A.gsw_part + (size_t)blockIdx.x * D.n_small;
  const int n_tiles = (A.B + kTileSamples - 1) / kTileSamples;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b0 = tile * kTileSamples;
    const bool first = tile == (int)blockIdx.x;
    // ---- 0. the tile's item rows and user rows: every wave its two samples, all row reads in flight together ------------------
    {
      float e[2][kCaserMaxL], pu[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gb = b0 + w + kTileWaves * k;
        const bool has = gb < A.B;
        const int mine = (has && lane < L) ? A.before[(size_t)gb * L + lane] : 0;
#pragma unroll
        for (int t = 0; t < kCaserMaxL; ++t) {
          const int n = __shfl(mine, t);
          e[k][t] = (t < L && has && lane < d) ? A.item_emb[(size_t)n * ld + lane] : 0.f;
        }
        const int u = has ? A.uid[gb] : 0;
        pu[k] = (has && lane < d) ? A.user_emb[(size_t)u * ld + lane] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int b = w + kTileWaves * k;
#pragma unroll
        for (int t = 0; t < kCaserMaxL; ++t)
          if (t < L && lane < ld) E[b * SB + t * ld + lane] = e[k][t];
        PU[b * kSP + lane] = pu[k];
      }
    }
    __syncthreads();
    TSTAMP(2);
    // ---- 1. convolutions forward: one 16 x 16 tile of pre-activations per (vertical filter tile) / (height, position, filter tile)
    {
      int u = 0;
      for (int nt = 0; nt < NTv; ++nt, ++u) {
        if (u % kTileWaves != w) continue;
        const int f = 16 * nt + m16;
        const bool fv = f < n_v;
        f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int t = 0; t < L; ++t) {
          const float *const e = E + m16 * SB + t * ld + q4;
          const int wo = D.off_kv + (t * n_v + (fv ? f : 0)) * ld + q4;
          int kc = 0;
          for (; kc + 1 < KC; kc += 2) {
            a0 = mfma4(e[4 * kc], fv ? cw(wo + 4 * kc) : 0.f, a0);
            a1 = mfma4(e[4 * kc + 4], fv ? cw(wo + 4 * kc + 4) : 0.f, a1);
          }
          if (kc < KC) a0 = mfma4(e[4 * kc], fv ? cw(wo + 4 * kc) : 0.f, a0);
        }
        float *const ct = CT + (TV0 + nt) * kCT + m16 * 17 + 4 * q4;
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[r] = a0[r] + a1[r];
      }
      for (int i = L - 1; i >= 0; --i)                      // (tallest filters first: the units are dealt out in order of cost)
        for (int t = 0; t + i < L; ++t)
          for (int nt = 0; nt < NTh; ++nt, ++u) {
            if (u % kTileWaves != w) continue;
            const int f = 16 * nt + m16;
            const bool fv = f < n_h;
            f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            for (int s = 0; s <= i; ++s) {
              const float *const e = E + m16 * SB + (t + s) * ld + q4;
              const int wo = D.off_kh[i] + (s * n_h + (fv ? f : 0)) * ld + q4;
              int kc = 0;
              for (; kc + 1 < KC; kc += 2) {
                a0 = mfma4(e[4 * kc], fv ? cw(wo + 4 * kc) : 0.f, a0);
                a1 = mfma4(e[4 * kc + 4], fv ? cw(wo + 4 * kc + 4) : 0.f, a1);
              }
              if (kc < KC) a0 = mfma4(e[4 * kc], fv ? cw(wo + 4 * kc) : 0.f, a0);
            }
            float *const ct = CT + ((ct_first(i, L, NTh) + t * NTh) + nt) * kCT + m16 * 17 + 4 * q4;
#pragma unroll
            for (int r = 0; r < 4; ++r) ct[r] = a0[r] + a1[r];
          }
    }
    __syncthreads();
    TSTAMP(3);
    // ---- 2. bias, act_h, max over time (the first maximum wins, like the max-pool gradient), dropout (caser.py:103-114) -----------
    for (int idx = threadIdx.x; idx < kTileSamples * nx; idx += blockDim.x) {
      const int b = idx & 15, j = idx >> 4;
      float x, pre;
      int arg = 0;
      if (j < n_v) {
        x = pre = CT[(TV0 + (j >> 4)) * kCT + (j & 15) * 17 + b] + cw(D.off_bv + j);
      } else {
        const int pq = j - n_v, i = pq / n_h, f = pq - i * n_h;
        const float bias = cw(D.off_bh[i] + f);
        const float *const c0 = CT + (ct_first(i, L, NTh) + (f >> 4)) * kCT + (f & 15) * 17 + b;
        float best = -3.0e38f;
        pre = 0.f;
        for (int t = 0; t + i < L; ++t) {
          const float v = c0[t * NTh * kCT] + bias;
          const float r = act_f(D.act_h, v);
          if (r > best) { best = r; arg = t; pre = v; }
        }
        x = best;
      }
      XD[b * SX + j] = kept(b0 + b, j) ? x * ((A.keep || hashed) ? inv_keep : 1.f) : 0.f;
      PRE[b * SX + j] = pre;
      ARG[b * SX + j] = arg;
    }
    __syncthreads();
    TSTAMP(4);
    // ---- 3. dense_0: [16 x nx] . [nx x 16 channels] per channel tile (kernel rows from global memory) ------------------------------
    for (int nc = w; nc < NC; nc += kTileWaves) {
      const int cc = 16 * nc + m16;
      const bool cv = cc < ld;
      f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float *const xa = XD + m16 * SX + q4;
      const int KX = (nx + 3) >> 2;
      int kj = 0;
      for (; kj + 1 < KX; kj += 2) {
        const int j0 = 4 * kj + q4, j1 = j0 + 4;
        a0 = mfma4(xa[4 * kj], (cv && j0 < nx) ? gw(D.off_wd + j0 * ld + cc) : 0.f, a0);
        a1 = mfma4(xa[4 * kj + 4], (cv && j1 < nx) ? gw(D.off_wd + j1 * ld + cc) : 0.f, a1);
      }
      if (kj < KX) {
        const int j0 = 4 * kj + q4;
        a0 = mfma4(xa[4 * kj], (cv && j0 < nx) ? gw(D.off_wd + j0 * ld + cc) : 0.f, a0);
      }
      const float bias = cc < d ? gw(D.off_bd + cc) : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = 4 * q4 + r;
        const float z0 = cc < d ? a0[r] + a1[r] + bias : 0.f;
        Z0[b * kSP + cc] = z0;
        Z[b * kSP + cc] = cc < d ? act_f(D.act_mlp, z0) : 0.f;
      }
    }
    __syncthreads();
    TSTAMP(5);
    // ---- 4. targets: score, sigmoid, Keras BCE, backward to the lookups — per sample, lane = channel (caser.py:115-120) ------------
    // eight targets at a time: their sixteen row reads are in flight together, the eight dot products leave through ONE reduce8, the
    // lanes that hold target j's score do its sigmoid / loss / gradient, every lane then fetches the eight gradients with v_readlane
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const int b = w + kTileWaves * k, gb = b0 + b;
      float dz0 = 0.f;
      if (gb < A.B) {
        const int c = lane;
        const bool live = c < d;
        const float z = live ? Z[b * kSP + c] : 0.f, z0 = live ? Z0[b * kSP + c] : 0.f, pu = PU[b * kSP + c];
        float dz = 0.f, dpu = 0.f;
        for (int j0 = 0; j0 < D.Tp; j0 += 8) {
          int n[8];
          float wa[8], wb[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) n[qq] = j0 + qq < D.Tp ? A.after[(size_t)gb * D.Tp + j0 + qq] : 0;
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            const bool on = live && j0 + qq < D.Tp;
            wa[qq] = on ? A.W1[(size_t)n[qq] * D.ld2 + c] : 0.f;
            wb[qq] = on ? A.W1[(size_t)n[qq] * D.ld2 + d + c] : 0.f;
          }
          const int jm = j0 + slot8(c);                           // the target whose score this lane receives
          const float bm = jm < D.Tp ? A.b1[A.after[(size_t)gb * D.Tp + jm]] : 0.f;
          float prod[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) prod[qq] = fmaf(z, wa[qq], pu * wb[qq]);
          const float sc = reduce8(prod, c) + bm;
          const float p = sigmoidf_(sc);
          const float y = jm < D.T ? 1.f : 0.f;
          const float dsm = jm < D.Tp ? bce_grad(y, p) * inv_bt * p * (1.f - p) : 0.f;
          if ((c & 7) == 0 && jm < D.Tp) {
            loss_acc += bce_elem(y, p);
            A.db1[(size_t)gb * D.Tp + jm] = dsm;
          }
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            if (j0 + qq < D.Tp) {
              const float ds = lane_f(dsm, lane8(qq));
              const size_t row = (size_t)gb * D.Tp + j0 + qq;
              if (live) { A.dW1[row * D.ld2 + c] = ds * z; A.dW1[row * D.ld2 + d + c] = ds * pu; }
              dz = fmaf(ds, wa[qq], dz);
              dpu = fmaf(ds, wb[qq], dpu);
            }
          }
        }
        if (live) A.dPu[(size_t)gb * ld + c] = dpu;
        dz0 = live ? dz * act_df(D.act_mlp, z0) : 0.f;
      }
      DZ0[b * kSP + lane] = dz0;
    }
    __syncthreads();
    TSTAMP(6);
    // ---- 5. dense_0 backward: dx[16 x 16 units] = dz0[16 x c] . Wd^T, through the dropout mask ----------------------------------
    for (int nt = w; nt < NJ; nt += kTileWaves) {
      const int jj = 16 * nt + m16;
      const bool jv = jj < nx;
      f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float *const za = DZ0 + m16 * kSP + q4;
      const int wo = D.off_wd + (jv ? jj : 0) * ld + q4;
      int kc = 0;
      for (; kc + 1 < KC; kc += 2) {
        a0 = mfma4(za[4 * kc], jv ? gw(wo + 4 * kc) : 0.f, a0);
        a1 = mfma4(za[4 * kc + 4], jv ? gw(wo + 4 * kc + 4) : 0.f, a1);
      }
      if (kc < KC) a0 = mfma4(za[4 * kc], jv ? gw(wo + 4 * kc) : 0.f, a0);
      if (jv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = 4 * q4 + r;
          const float g = a0[r] + a1[r];
          DX[b * SX + jj] = kept(b0 + b, jj) ? g * ((A.keep || hashed) ? inv_keep : 1.f) : 0.f;
        }
      }
    }
    __syncthreads();
    TSTAMP(7);
    // ---- 6. through act_h at the arg-max step: the tile array now holds dC[(i, t)][f][b] (zero away from the arg-max) and dV[f][b];
    //         dx of a horizontal unit becomes its pre-activation gradient in place (the bias sums of step 8 read it) -------------------
    for (int idx = threadIdx.x; idx < kTileSamples * 16 * NTv; idx += blockDim.x) {
      const int b = idx & 15, f = idx >> 4;
      CT[(TV0 + (f >> 4)) * kCT + (f & 15) * 17 + b] = f < n_v ? DX[b * SX + f] : 0.f;
    }
    for (int idx = threadIdx.x; idx < kTileSamples * 16 * NTh * L; idx += blockDim.x) {
      const int b = idx & 15, fi = idx >> 4, i = fi / (16 * NTh), f = fi - i * 16 * NTh;
      float dc = 0.f;
      int arg = -1;
      if (f < n_h) {
        const int j = n_v + i * n_h + f;
        dc = DX[b * SX + j] * act_df(D.act_h, PRE[b * SX + j]);
        arg = ARG[b * SX + j];
        DX[b * SX + j] = dc;
      }
      float *const c0 = CT + (ct_first(i, L, NTh) + (f >> 4)) * kCT + (f & 15) * 17 + b;
      for (int t = 0; t + i < L; ++t) c0[t * NTh * kCT] = t == arg ? dc : 0.f;
    }
    __syncthreads();
    TSTAMP(8);
    // ---- 7. gradient rows of the item lookups: dE[16 x 16 channels] at position t' = dV . Kv[t'] + sum over taps (i, s) of
    //         dC[(i, t' - s)] . Kh_i[s] -----------------------------------------------------------------------------------------------
    {
      int u = 0;
      for (int tp = 0; tp < L; ++tp)
        for (int nc = 0; nc < NC; ++nc, ++u) {
          if (u % kTileWaves != w) continue;
          const int cc = 16 * nc + m16;
          const bool cv = cc < ld;
          const int cs = cv ? cc : 0;
          f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
          for (int kf = 0; 4 * kf < n_v; ++kf) {
            const int f = 4 * kf + q4;
            const float a = CT[(TV0 + (f >> 4)) * kCT + (f & 15) * 17 + m16];
            a0 = mfma4(a, (cv && f < n_v) ? cw(D.off_kv + (tp * n_v + f) * ld + cs) : 0.f, a0);
          }
          for (int i = 0; i < L; ++i)
            for (int s = 0; s <= i; ++s) {
              const int t = tp - s;
              if (t < 0 || t + i >= L) continue;
              const float *const c0 = CT + (ct_first(i, L, NTh) + t * NTh) * kCT + m16;
              for (int kf = 0; 4 * kf < n_h; kf += 2) {
                const int f0 = 4 * kf + q4, f1 = f0 + 4;
                a0 = mfma4(c0[(f0 >> 4) * kCT + (f0 & 15) * 17], (cv && f0 < n_h) ? cw(D.off_kh[i] + (s * n_h + f0) * ld + cs) : 0.f, a0);
                if (4 * (kf + 1) < n_h)
                  a1 = mfma4(c0[(f1 >> 4) * kCT + (f1 & 15) * 17], (cv && f1 < n_h) ? cw(D.off_kh[i] + (s * n_h + f1) * ld + cs) : 0.f, a1);
              }
            }
          if (cc < d) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int gb = b0 + 4 * q4 + r;
              if (gb < A.B) A.dE[((size_t)gb * L + tp) * ld + cc] = a0[r] + a1[r];
            }
          }
        }
    }
#ifdef DRX_STAMPS
    __syncthreads();
    TSTAMP(9);
#endif
    // ---- 8. small-weight gradients of the tile (K = the 16 samples), added to the workgroup's partial sums -------------------------
    {
      int u = 0;
      // horizontal kernels: g[(i, s)][f][c] = sum over t, b of dC[(i, t)][f][b] * E[b][t + s][c]
      for (int i = 0; i < L; ++i)
        for (int s = 0; s <= i; ++s)
          for (int nt = 0; nt < NTh; ++nt)
            for (int nc = 0; nc < NC; ++nc, ++u) {
              if (u % kTileWaves != w) continue;
              f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
              for (int t = 0; t + i < L; ++t) {
                const float *const ca = CT + (ct_first(i, L, NTh) + t * NTh + nt) * kCT + m16 * 17 + q4;
                const float *const eb = E + q4 * SB + (t + s) * ld + 16 * nc + m16;
                a0 = mfma4(ca[0], eb[0], a0);
                a1 = mfma4(ca[4], eb[4 * SB], a1);
                a0 = mfma4(ca[8], eb[8 * SB], a0);
                a1 = mfma4(ca[12], eb[12 * SB], a1);
              }
              const int cc = 16 * nc + m16;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int f = 16 * nt + 4 * q4 + r;
                if (f < n_h && cc < ld) {
                  float *const o = gp + D.off_kh[i] + (s * n_h + f) * ld + cc;
                  *o = (first ? 0.f : *o) + (a0[r] + a1[r]);
                }
              }
            }
      // vertical kernel: g[t][f][c] = sum over b of dV[f][b] * E[b][t][c]
      for (int t = 0; t < L; ++t)
        for (int nt = 0; nt < NTv; ++nt)
          for (int nc = 0; nc < NC; ++nc, ++u) {
            if (u % kTileWaves != w) continue;
            f4v a0 = {0.f, 0.f, 0.f, 0.f};
            const float *const ca = CT + (TV0 + nt) * kCT + m16 * 17 + q4;
            const float *const eb = E + q4 * SB + t * ld + 16 * nc + m16;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) a0 = mfma4(ca[4 * kb], eb[4 * kb * SB], a0);
            const int cc = 16 * nc + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int f = 16 * nt + 4 * q4 + r;
              if (f < n_v && cc < ld) {
                float *const o = gp + D.off_kv + (t * n_v + f) * ld + cc;
                *o = (first ? 0.f : *o) + a0[r];
              }
            }
          }
      // dense_0 kernel: g[j][c] = sum over b of xd[b][j] * dz0[b][c]
      for (int nt = 0; nt < NJ; ++nt)
        for (int nc = 0; nc < NC; ++nc, ++u) {
          if (u % kTileWaves != w) continue;
          f4v a0 = {0.f, 0.f, 0.f, 0.f};
          const int ja = 16 * nt + m16;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            const int b = 4 * kb + q4;
            a0 = mfma4(ja < nx ? XD[b * SX + ja] : 0.f, DZ0[b * kSP + 16 * nc + m16], a0);
          }
          const int cc = 16 * nc + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * nt + 4 * q4 + r;
            if (j < nx && cc < ld) {
              float *const o = gp + D.off_wd + j * ld + cc;
              *o = (first ? 0.f : *o) + a0[r];
            }
          }
        }
      // biases (and the zero padding of their segments): one thread per entry, the tile's samples in order
      const int padv = (n_v + 3) & ~3, padh = (n_h + 3) & ~3;
      for (int e = threadIdx.x; e < ld + padv + L * padh; e += blockDim.x) {
        float acc = 0.f;
        float *o;
        if (e < ld) {
          o = gp + D.off_bd + e;
          for (int b = 0; b < kTileSamples; ++b) acc += DZ0[b * kSP + e];
        } else if (e < ld + padv) {
          const int f = e - ld;
          o = gp + D.off_bv + f;
          if (f < n_v)
            for (int b = 0; b < kTileSamples; ++b) acc += DX[b * SX + f];
        } else {
          const int q = e - ld - padv, i = q / padh, f = q - i * padh;
          o = gp + D.off_bh[i] + f;
          if (f < n_h)
            for (int b = 0; b < kTileSamples; ++b) acc += DX[b * SX + n_v + i * n_h + f];
        }
        *o = (first ? 0.f : *o) + acc;
      }
    }
    __syncthreads();                                  // (the next tile rewrites E and the vectors)
    TSTAMP(10);
  }
  loss_acc = group_sum<64>(loss_acc);                 // (the lanes that held a target's score carry its loss term)
  if (lane == 0) wloss[w] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int ww = 0; ww < kTileWaves; ++ww) t += wloss[ww];
    A.loss_part[blockIdx.x] = t * inv_bt;
  }
}

}  // namespace drx
