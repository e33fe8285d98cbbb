// Caser (DRecPy/Recommender/caser.py) on gfx950.
//
// Training (forward, Keras BCE, backward): k_caser_tile (drx_caser_tile.hpp) — tiles of 16 samples, every layer a product on the matrix
// cores.  Inference (the hidden state [dense_0 output | user embedding] that drx_rows_dot scores against dense_1): k_caser_hidden below —
// one wavefront per sample, lane c = embedding channel (d <= 64): the L item rows and the user row are single coalesced row reads; the
// vertical conv (caser.py:53,103 — kernel [L,d,n_v], it sums over d), the L horizontal convs with act_h + max over time
// (caser.py:55-58,106-108) and dense_0 (caser.py:63,114) are channel-parallel FMAs followed by wave reductions out of a copy of the small
// weights in LDS (channel-fastest, [s][f][c]: every lane reads its own column).  Through r04 the same wave-per-sample program also did
// the backward pass (HISTORY.md).
#include "drx_caser_tile.hpp"


namespace drx {

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

struct CaserLds {
  float *E;       // [L][64] item rows of the current sample
  float *x;       // [nx] concat(out_v, out_h)
};

__global__ __launch_bounds__(1024) void k_caser_hidden(DrxCaserDims D, DrxCaserArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int c = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), W = blockDim.x >> 6;
  const int L = D.L, d = D.d, ld = D.ld, nx = D.n_v + D.L * D.n_h;
  const int per_wave = L * 64 + nx;
  float *const wl = lds;                                              // [n_small] small weights, then 64 zeros (a row read runs 64 wide)
  float *const scratch0 = lds + D.n_small + 64;
  CaserLds S;
  S.E = scratch0 + (size_t)w * per_wave;
  S.x = S.E + L * 64;
  for (int i = threadIdx.x * 4; i < D.n_small; i += blockDim.x * 4)  // (every segment of sw is a multiple of 4 floats long)
    *reinterpret_cast<float4 *>(wl + i) = *reinterpret_cast<const float4 *>(A.sw + i);
  if (threadIdx.x < 64) wl[D.n_small + threadIdx.x] = 0.f;
  __syncthreads();
  const bool live = c < d;

  for (int b0 = blockIdx.x * W; b0 < A.B; b0 += gridDim.x * W) {      // uniform over the workgroup: its waves take b0 .. b0 + W - 1
    const int b = b0 + w;
    if (b >= A.B) continue;
    // ---- 1. embeddings: the L row reads in flight together ---------------------------------------------------------------------
    {
      const int mine = c < L ? A.before[b * L + c] : 0;
      float e[kCaserMaxL];
#pragma unroll
      for (int t = 0; t < kCaserMaxL; ++t) {
        const int n = __shfl(mine, t);
        e[t] = (t < L && live) ? A.item_emb[(size_t)n * ld + c] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < kCaserMaxL; ++t)
        if (t < L) S.E[t * 64 + c] = e[t];
    }
    const int u = A.uid[b];
    const float pu = live ? A.user_emb[(size_t)u * ld + c] : 0.f;
    wave_lds_sync();
    // ---- 2. vertical conv (dead lanes hold E = 0: whatever they read beside a weight row adds nothing) -------------------------
    for (int f = 0; f < D.n_v; ++f) {
      float part = 0.f;
      for (int t = 0; t < L; ++t) part = fmaf(S.E[t * 64 + c], wl[D.off_kv + (t * D.n_v + f) * ld + c], part);
      const float v = wave_sum(part) + wl[D.off_bv + f];
      if (c == 0) S.x[f] = v;
    }
    // ---- 3. horizontal convs + act_h + max over time -------------------------------------------------------------------
    // 16 filters at a time: the 16 channel sums leave through one reduce16, and the lanes that end up holding filter f track its
    // maximum over the window positions.
    for (int i = 0; i < L; ++i) {
      const float *const kh = wl + D.off_kh[i] + c;
      for (int f0 = 0; f0 < D.n_h; f0 += 16) {
        const int fm = slot16(c);                               // the filter (of this block) whose total this lane receives
        float best = -3.0e38f;
        const float bias = (f0 + fm < D.n_h) ? wl[D.off_bh[i] + f0 + fm] : 0.f;
        for (int t = 0; t + i < L; ++t) {
          float part[16];
#pragma unroll
          for (int ff = 0; ff < 16; ++ff) part[ff] = 0.f;
          if (f0 + 16 <= D.n_h) {
            for (int s2 = 0; s2 <= i; ++s2) {
              const float e = S.E[(t + s2) * 64 + c];
              const float *const row = kh + (s2 * D.n_h + f0) * ld;
#pragma unroll
              for (int ff = 0; ff < 16; ++ff) part[ff] = fmaf(e, row[ff * ld], part[ff]);
            }
          } else {
            for (int s2 = 0; s2 <= i; ++s2) {
              const float e = S.E[(t + s2) * 64 + c];
              const float *const row = kh + (s2 * D.n_h + f0) * ld;
#pragma unroll
              for (int ff = 0; ff < 16; ++ff) part[ff] = fmaf(e, (f0 + ff < D.n_h) ? row[ff * ld] : 0.f, part[ff]);
            }
          }
          best = fmaxf(best, act_f(D.act_h, reduce16(part, c) + bias));
        }
        if ((c & 3) == 0 && f0 + fm < D.n_h) S.x[D.n_v + i * D.n_h + f0 + fm] = best;      // one of the four lanes that hold filter fm
      }
    }
    wave_lds_sync();
    // ---- 4. dense_0 (act_mlp; no dropout at inference: caser.py:61,114 with training=False) -------------------------------------
    float z0 = wl[D.off_bd + c];
    {
      const float *const wd = wl + D.off_wd + c;
#pragma unroll 4
      for (int j = 0; j < nx; ++j) z0 = fmaf(S.x[j], wd[j * ld], z0);
    }
    if (live) {
      A.cat_out[(size_t)b * D.ld2 + c] = act_f(D.act_mlp, z0);
      A.cat_out[(size_t)b * D.ld2 + d + c] = pu;
    }
    wave_lds_sync();                                    // (the next sample rewrites E and x)
  }
}

// out[j] = sum_r part[r][j] for j < n, out[n] = sum_r tail[r]: 64 columns per workgroup, its 16 waves take every 16th row,
// their partial sums are combined in wave order (fixed order of additions).  (One thread per column walking all rows took
// 144 us of a 0.68 ms Caser step.)
__global__ __launch_bounds__(1024) void k_sum_partials(const float *__restrict__ part, int n_rows, int n, const float *__restrict__ tail,
                                                       float *__restrict__ out) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float a = 0.f;
  if (j < n) for (int r = q; r < n_rows; r += 16) a += part[(size_t)r * n + j];
  else if (j == n) for (int r = q; r < n_rows; r += 16) a += tail[r];
  red[q][c] = a;
  __syncthreads();
  if (q == 0 && j <= n) {
    float t = 0.f;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) t += red[qq][c];
    out[j] = t;
  }
}

// k_sum_partials with the small weights' Keras Adam behind it: the thread that has column j's sum applies l2 + Adam of j's segment
// (one launch instead of two; same operations in the same order as k_sum_partials + k_adam_segments).
__global__ __launch_bounds__(1024) void k_sum_partials_adam(const float *__restrict__ part, int n_rows, int n, const float *__restrict__ tail,
                                                            float *__restrict__ out, float *p, float *m, float *v, DrxAdamSegments sg,
                                                            float b1, float b2, float eps) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float a = 0.f;
  if (j < n) for (int r = q; r < n_rows; r += 16) a += part[(size_t)r * n + j];
  else if (j == n) for (int r = q; r < n_rows; r += 16) a += tail[r];
  red[q][c] = a;
  __syncthreads();
  if (q == 0 && j <= n) {
    float t = 0.f;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) t += red[qq][c];
    out[j] = t;
    if (j < n) {
      int s = -1;
      for (int k = 0; k < sg.n; ++k)
        if (j >= sg.start[k] && j < sg.start[k] + sg.len[k]) s = k;
      if (s >= 0) {                                  // (padding entries of a bias segment belong to no segment: left alone)
        const OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, sg.alpha[s]};
        float pp = p[j], mm = m[j], vv = v[j];
        opt_update1(o, fmaf(sg.l2_coef[s], pp, t), pp, mm, vv);
        p[j] = pp; m[j] = mm; v[j] = vv;
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_adam_segments(float *p, float *m, float *v, const float *g, DrxAdamSegments sg, float b1,
                                                          float b2, float eps) {
  for (int s = 0; s < sg.n; ++s) {
    OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, sg.alpha[s]};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < sg.len[s]; i += gridDim.x * kBlock) {
      const int k = sg.start[s] + i;
      float pp = p[k], mm = m[k], vv = v[k];
      opt_update1(o, fmaf(sg.l2_coef[s], pp, g[k]), pp, mm, vv);
      p[k] = pp; m[k] = mm; v[k] = vv;
    }
  }
}

constexpr size_t kCaserLdsLimit = 160 * 1024 - 256;    // dynamic LDS a workgroup may ask for (beside the kernels' static words)

// k_caser_hidden: the small weights (+ 64 zeros) and a scratch per wave, E | x
static size_t hidden_lds_bytes(const DrxCaserDims &D, int waves) {
  const int nx = D.n_v + D.L * D.n_h;
  return ((size_t)D.n_small + 64 + (size_t)waves * ((size_t)D.L * 64 + (size_t)nx)) * 4 + 64;
}

// its waves per workgroup: as many (16, 8, 4, 2, 1) as fit the CU's LDS beside the weights, and no more than leave 128 workgroups
static int hidden_waves(const DrxCaserDims &D, int B) {
  int w = 16;
  while (w > 1 && (hidden_lds_bytes(D, w) > kCaserLdsLimit || (B + w - 1) / w < 128)) w >>= 1;
  return w;
}

// the training kernel's variant for these dimensions: 2 = every small weight in LDS, 1 = the convolution weights, 0 = all read from global
// memory, -1 = the tile's own scratch does not fit a workgroup's LDS (nx too large)
static int caser_tile_variant(const DrxCaserDims &D) {
#ifdef DRX_CASER_FORCE_WL                                   // (diagnostic builds)
  return DRX_CASER_FORCE_WL;
#endif
  for (int wl = 2; wl >= 0; --wl)
    if ((size_t)caser_tile_geom(D, wl).floats * 4 <= kCaserLdsLimit) return wl;
  return -1;
}

static int check_dims(const DrxCaserDims *D) {
  if (!D || D->L < 1 || D->L > kCaserMaxL || D->d < 1 || D->d > 64 || D->ld < D->d || (D->ld & 3) || D->ld2 < 2 * D->d ||
      (D->ld2 & 3) || D->n_v < 1 || D->n_h < 1 || D->T < 1 || D->Tp < D->T || D->n_small < 1 || (D->n_small & 3) || D->act_h < 0 || D->act_h > 3 ||
      D->act_mlp < 0 || D->act_mlp > 3)
    return DRX_EINVAL;
  // both kernels must fit: the inference kernel keeps every small weight in LDS, the training kernel at least its tile's scratch
  return (hidden_lds_bytes(*D, 1) <= kCaserLdsLimit && caser_tile_variant(*D) >= 0) ? DRX_OK : DRX_EINVAL;
}

}  // namespace drx

using namespace drx;

extern "C" {

int drx_caser_grid(const DrxCaserDims *D, int32_t B) {
  if (!D || B < 1 || check_dims(D) != DRX_OK) return 0;
  const int g = (B + kTileSamples - 1) / kTileSamples;       // one workgroup per CU (its LDS), tiles of 16 samples in turn
#ifdef DRX_CASER_GRID                                          // (diagnostic builds: several tiles per workgroup at a small batch)
  return g < DRX_CASER_GRID ? g : DRX_CASER_GRID;
#endif
  return g < 256 ? g : 256;
}

// the training kernel of drx_caser_fwd_bwd / drx_caser_step_small (the partial sums are reduced by the caller)
static int launch_caser_tile(const DrxCaserDims *D, const DrxCaserArgs *A, const float *gsw_out, hipStream_t st, int *grid_out) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->item_emb || !A->user_emb || !A->W1 || !A->b1 || !A->sw || !A->uid || !A->before || !A->after || !A->dE ||
      (!A->dW1 && !A->cat_out) || !A->db1 || !A->dPu || !A->gsw_part || !A->loss_part || !gsw_out || A->B < 1 || A->rate < 0.f || A->rate >= 1.f)
    return DRX_EINVAL;
  const int grid = drx_caser_grid(D, A->B);
  const int var = caser_tile_variant(*D);
  const size_t lds = (size_t)caser_tile_geom(*D, var).floats * 4;
  auto launch = [&](auto kernel) -> int {
    DRX_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * kTileWaves), lds, st, *D, *A CASER_STAMP_PASS);
    return DRX_OK;
  };
  // (examples/caser.py:13 with fit(neg_ratio=3): its dimensions as compile-time constants)
  const bool fx = var == 2 && D->L == 5 && D->d == 50 && D->ld == 52 && D->ld2 == 100 && D->n_v == 4 && D->n_h == 16 && D->T == 3 && D->Tp == 12;
  rc = fx ? launch(k_caser_tile<2, 1>) : var == 2 ? launch(k_caser_tile<2, 0>) : var == 1 ? launch(k_caser_tile<1, 0>) : launch(k_caser_tile<0, 0>);
  *grid_out = grid;
  return rc;
}

int drx_caser_fwd_bwd(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int grid = 0;
  const int rc = launch_caser_tile(D, A, gsw_out, st, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum_partials, dim3((D->n_small + 64) / 64), dim3(1024), 0, st, A->gsw_part, grid, D->n_small,
                     A->loss_part, gsw_out);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_caser_step_small(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                         const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream) {
  if (!sw || !sw_m || !sw_v || !sg || sg->n < 1 || sg->n > DRX_MAX_SEGMENTS || (A && sw != A->sw)) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = 0;
  const int rc = launch_caser_tile(D, A, gsw_out, st, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum_partials_adam, dim3((D->n_small + 64) / 64), dim3(1024), 0, st, A->gsw_part, grid, D->n_small,
                     A->loss_part, gsw_out, sw, sw_m, sw_v, *sg, beta1, beta2, eps);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_caser_hidden(const DrxCaserDims *D, const DrxCaserArgs *A, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->item_emb || !A->user_emb || !A->sw || !A->uid || !A->before || !A->cat_out || A->B < 1) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int waves = hidden_waves(*D, A->B);             // (every workgroup copies the small weights into its LDS: few, large workgroups)
  const size_t lds = hidden_lds_bytes(*D, waves);
  const int g = (A->B + waves - 1) / waves;
  DRX_HIP(hipFuncSetAttribute((const void *)k_caser_hidden, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_caser_hidden, dim3(g < 512 ? g : 512), dim3(64 * waves), lds, st, *D, *A);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_adam_segments(float *p, float *m, float *v, const float *g, const DrxAdamSegments *sg, float beta1, float beta2, float eps,
                      void *stream) {
  if (!p || !m || !v || !g || !sg || sg->n < 1 || sg->n > DRX_MAX_SEGMENTS) return DRX_EINVAL;
  hipLaunchKernelGGL(k_adam_segments, dim3(64), dim3(kBlock), 0, (hipStream_t)stream, p, m, v, g, *sg, beta1, beta2, eps);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // extern "C"
