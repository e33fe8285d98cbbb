// Caser (DRecPy/Recommender/caser.py) forward / backward on gfx950.
//
// One wavefront per sample, lane c = embedding channel (d <= 64): the L item rows and the user row are single
// coalesced row reads; the vertical conv (caser.py:53,103 — kernel [L,d,n_v], it sums over d), the L horizontal convs
// with relu + max over time (caser.py:55-58,106-108), dense_0 (caser.py:63,114) and the T' target dots
// (caser.py:115-120) are channel-parallel FMAs followed by wave reductions.  The gradients of the small weights are
// accumulated per workgroup in LDS by their owning lane (no atomics, fixed sample order) and reduced over workgroups
// in a second, ordered pass; the gradients of the embedding lookups leave as one row per lookup for
// drx_scatter_rows.  Small weights are stored channel-fastest ([s][f][c]) so that every lane reads its own column.
#include "drx_common.hpp"
#include "drx_rows.hpp"

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

struct CaserLds {
  float *gsw;     // [n_small] gradient accumulators of the small weights
  float *E;       // [L][64] item rows of the current sample
  float *dE;      // [L][64]
  float *x;       // [nx] concat(out_v, out_h) before dropout
  float *xd;      // [nx] after dropout
  float *pre;     // [nx] pre-activation at the arg-max step (horizontal convs)
  float *dx;      // [nx]
  int *arg;       // [nx]
  float *dz0s;    // [64] dense_0's pre-activation gradient of the sample (read by the accumulator owners)
};

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

// A workgroup is W waves, one sample per wave and round, sharing ONE set of small-weight gradient accumulators in LDS (72 KB at the
// reference configuration: with a private copy per wave only two waves fit a CU and the chip idles — 0.74 ms at B = 4096).  Forward
// and backward of the W samples run side by side in per-wave scratch; afterwards the waves add their weight-gradient contributions
// to the shared accumulators ONE AFTER THE OTHER, in sample order (a barrier between turns): no atomics, and the order of every
// sum is fixed by the batch alone.
template <bool TRAIN>
__global__ __launch_bounds__(1024) void k_caser(DrxCaserDims D, DrxCaserArgs A) {
  extern __shared__ __align__(16) float lds[];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int L = D.L, d = D.d, nx = D.n_v + D.L * D.n_h;
  const int per_wave = 2 * L * 64 + 5 * nx + 64;
  CaserLds S;
  S.gsw = lds;
  float *q = lds + (TRAIN ? D.n_small : 0) + (size_t)w * per_wave;
  S.E = q; q += L * 64;
  S.dE = q; q += L * 64;
  S.x = q; q += nx;
  S.xd = q; q += nx;
  S.pre = q; q += nx;
  S.dx = q; q += nx;
  S.arg = reinterpret_cast<int *>(q); q += nx;
  S.dz0s = q;
  __shared__ float wloss[16];
  const float *sw = A.sw;
  if (TRAIN) {
    for (int i = threadIdx.x; i < D.n_small; i += blockDim.x) S.gsw[i] = 0.f;
    __syncthreads();
  }
  float loss_acc = 0.f;
  const bool live = c < d;
  const float inv_bt = 1.0f / ((float)A.B * (float)D.Tp);
  const float inv_keep = 1.0f / (1.0f - A.rate);
  const bool hashed = TRAIN && !A.keep && A.rate > 0.f;                // counter-based dropout mask (DrxCaserArgs.mask_seed)
  const uint32_t rthr = hashed ? q_threshold(A.rate) : 0u;

  for (int b0 = blockIdx.x * W; b0 < A.B; b0 += gridDim.x * W) {      // uniform over the workgroup: its waves take b0 .. b0 + W - 1
    const int b = b0 + w;
    const bool has = b < A.B;
    float dz0 = 0.f;
    if (has) {
    // ---- 1. embeddings ---------------------------------------------------------------------------------------------
    for (int t = 0; t < L; ++t) {
      const int n = A.before[b * L + t];
      S.E[t * 64 + c] = live ? A.item_emb[(size_t)n * D.ld + c] : 0.f;
      S.dE[t * 64 + c] = 0.f;
    }
    const int u = A.uid[b];
    const float pu = live ? A.user_emb[(size_t)u * D.ld + c] : 0.f;
    wave_lds_sync();
    // ---- 2. vertical conv ------------------------------------------------------------------------------------------
    for (int f = 0; f < D.n_v; ++f) {
      float part = 0.f;
      for (int t = 0; t < L; ++t) part = fmaf(S.E[t * 64 + c], live ? sw[D.off_kv + (t * D.n_v + f) * D.ld + c] : 0.f, part);
      const float v = wave_sum(part) + sw[D.off_bv + f];
      if (c == 0) { S.x[f] = v; S.pre[f] = v; S.arg[f] = 0; }
    }
    // ---- 3. horizontal convs + act_h + max over time -------------------------------------------------------------------
    // 16 filters at a time: their (i+1) x 16 kernel values of this channel are requested together, the 16 channel sums
    // leave through one reduce16, and the lanes that end up holding filter f track its maximum over the window positions.
    for (int i = 0; i < L; ++i) {
      for (int f0 = 0; f0 < D.n_h; f0 += 16) {
        const int fm = slot16(c);                               // the filter (of this block) whose total this lane receives
        float best = -3.0e38f, bpre = 0.f;
        int bt = 0;
        for (int t = 0; t + i < L; ++t) {
          float part[16];
#pragma unroll
          for (int ff = 0; ff < 16; ++ff) part[ff] = 0.f;
          for (int s2 = 0; s2 <= i; ++s2) {
            const float e = S.E[(t + s2) * 64 + c];
            float w16[16];
#pragma unroll
            for (int ff = 0; ff < 16; ++ff)
              w16[ff] = (live && f0 + ff < D.n_h) ? sw[D.off_kh[i] + (s2 * D.n_h + f0 + ff) * D.ld + c] : 0.f;
#pragma unroll
            for (int ff = 0; ff < 16; ++ff) part[ff] = fmaf(e, w16[ff], part[ff]);
          }
          const float tot = reduce16(part, c);
          const float v = tot + ((f0 + fm < D.n_h) ? sw[D.off_bh[i] + f0 + fm] : 0.f);
          const float r = act_f(D.act_h, v);
          if (r > best) { best = r; bt = t; bpre = v; }        // first maximum wins, like the max-pool gradient
        }
        if ((c & 3) == 0 && f0 + fm < D.n_h) {                  // one of the four lanes that hold filter fm
          const int j = D.n_v + i * D.n_h + f0 + fm;
          S.x[j] = best; S.pre[j] = bpre; S.arg[j] = bt;
        }
      }
    }
    wave_lds_sync();
    // ---- 4. dropout (mask injected by the host; caser.py:61,114) -------------------------------------------------------
    for (int j = c; j < nx; j += 64) {
      float v = S.x[j];
      if (TRAIN && A.keep) v = A.keep[(size_t)b * nx + j] ? v * inv_keep : 0.f;
      else if (hashed) v = hash_u32(A.mask_seed, (uint32_t)b, (uint32_t)j) >= rthr ? v * inv_keep : 0.f;
      S.xd[j] = v;
    }
    wave_lds_sync();
    // ---- 5. dense_0 (act_mlp) ---------------------------------------------------------------------------------------------
    float z0 = live ? sw[D.off_bd + c] : 0.f;
    if (live)
      for (int j = 0; j < nx; ++j) z0 = fmaf(S.xd[j], sw[D.off_wd + j * D.ld + c], z0);
    const float z = act_f(D.act_mlp, z0);
    if (!TRAIN) {
      if (live) { A.cat_out[(size_t)b * D.ld2 + c] = z; A.cat_out[(size_t)b * D.ld2 + d + c] = pu; }
    } else {
    // ---- 6. targets: score, sigmoid, Keras BCE, backward to the lookups -------------------------------------------------
    float dz = 0.f, dpu = 0.f;
    for (int j = 0; j < D.Tp; ++j) {
      const int n = A.after[b * D.Tp + j];
      const float wa = live ? A.W1[(size_t)n * D.ld2 + c] : 0.f;
      const float wb = live ? A.W1[(size_t)n * D.ld2 + d + c] : 0.f;
      const float sc = wave_sum(fmaf(z, wa, pu * wb)) + A.b1[n];
      const float p = sigmoidf_(sc);
      const float y = j < D.T ? 1.f : 0.f;
      loss_acc += bce_elem(y, p);
      const float ds = bce_grad(y, p) * inv_bt * p * (1.f - p);
      const size_t row = (size_t)b * D.Tp + j;
      if (live) { A.dW1[row * D.ld2 + c] = ds * z; A.dW1[row * D.ld2 + d + c] = ds * pu; }
      if (c == 0) A.db1[row] = ds;
      dz = fmaf(ds, wa, dz);
      dpu = fmaf(ds, wb, dpu);
    }
    if (live) A.dPu[(size_t)b * D.ld + c] = dpu;
    dz0 = dz * act_df(D.act_mlp, z0);
    // ---- 7. dense_0 backward: dx[j] = sum_c dz0[c] * Wd[j][c], 16 rows of Wd per reduce16 ---------------------------------------
    for (int j0 = 0; j0 < nx; j0 += 16) {
      float prod[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        const float wv = (live && j < nx) ? sw[D.off_wd + j * D.ld + c] : 0.f;
        prod[jj] = dz0 * wv;
      }
      float g = reduce16(prod, c);
      const int j = j0 + slot16(c);
      if ((c & 3) == 0 && j < nx) {
        if (A.keep) g = A.keep[(size_t)b * nx + j] ? g * inv_keep : 0.f;
        else if (hashed) g = hash_u32(A.mask_seed, (uint32_t)b, (uint32_t)j) >= rthr ? g * inv_keep : 0.f;
        S.dx[j] = g;
      }
    }
    wave_lds_sync();
    // ---- 8. vertical conv backward (to the item rows) ------------------------------------------------------------------------
    for (int f = 0; f < D.n_v; ++f) {
      const float dv = S.dx[f];
      if (live)
        for (int t = 0; t < L; ++t) S.dE[t * 64 + c] = fmaf(dv, sw[D.off_kv + (t * D.n_v + f) * D.ld + c], S.dE[t * 64 + c]);
    }
    // ---- 9. horizontal convs backward (through relu at the arg-max step) -------------------------------------------------
    for (int i = 0; i < L; ++i)
      for (int f = 0; f < D.n_h; ++f) {
        const int j = D.n_v + i * D.n_h + f;
        const float dc = S.dx[j] * act_df(D.act_h, S.pre[j]);
        if (dc == 0.f) continue;
        const int t = S.arg[j];
        if (live)
          for (int s = 0; s <= i; ++s)
            S.dE[(t + s) * 64 + c] = fmaf(dc, sw[D.off_kh[i] + (s * D.n_h + f) * D.ld + c], S.dE[(t + s) * 64 + c]);
      }
    // ---- 10. gradient rows of the item lookups ------------------------------------------------------------------------------
    if (live)
      for (int t = 0; t < L; ++t) A.dE[((size_t)b * L + t) * D.ld + c] = S.dE[t * 64 + c];
    }   // TRAIN
    }   // has
    // ---- 11. small-weight gradients into the shared accumulators ---------------------------------------------------------------
    // Every accumulator row has ONE owner wave (dense_0 row j: wave j % W; conv_v filter f: f % W; horizontal filter (i, f):
    // (i * n_h + f) % W; the dense_0 bias: wave 0).  An owner walks the W samples of the round in sample order and adds their
    // contributions to its rows out of the samples' scratch — no two waves ever touch the same accumulator, and every sum is taken
    // in the order of the batch.
    if (TRAIN) {
      if (has) S.dz0s[c] = dz0;
      __syncthreads();
      for (int t = 0; t < W && b0 + t < A.B; ++t) {
        const float *T0 = lds + D.n_small + (size_t)t * per_wave;
        const float *tE = T0, *txd = T0 + 2 * L * 64 + nx, *tpre = txd + nx, *tdx = tpre + nx;
        const int *targ = reinterpret_cast<const int *>(tdx + nx);
        const float tdz0 = reinterpret_cast<const float *>(targ + nx)[c];
        if (w == 0 && live) S.gsw[D.off_bd + c] += tdz0;
        if (live)
          for (int j = w; j < nx; j += W) S.gsw[D.off_wd + j * D.ld + c] = fmaf(txd[j], tdz0, S.gsw[D.off_wd + j * D.ld + c]);
        for (int f = w; f < D.n_v; f += W) {
          const float dv = tdx[f];
          if (c == 0) S.gsw[D.off_bv + f] += dv;
          if (live)
            for (int tt = 0; tt < L; ++tt) {
              const int k = D.off_kv + (tt * D.n_v + f) * D.ld + c;
              S.gsw[k] = fmaf(tE[tt * 64 + c], dv, S.gsw[k]);
            }
        }
        for (int pq = w; pq < L * D.n_h; pq += W) {
          const int i = pq / D.n_h, f = pq - i * D.n_h;
          const int j = D.n_v + pq;
          const float dc = tdx[j] * act_df(D.act_h, tpre[j]);
          if (dc == 0.f) continue;
          const int ta = targ[j];
          if (c == 0) S.gsw[D.off_bh[i] + f] += dc;
          if (live)
            for (int s2 = 0; s2 <= i; ++s2) {
              const int k = D.off_kh[i] + (s2 * D.n_h + f) * D.ld + c;
              S.gsw[k] = fmaf(tE[(ta + s2) * 64 + c], dc, S.gsw[k]);
            }
        }
      }
      __syncthreads();
    }
  }
  if (TRAIN) {
    if (c == 0) wloss[w] = loss_acc;
    __syncthreads();
    for (int i = threadIdx.x; i < D.n_small; i += blockDim.x) A.gsw_part[(size_t)blockIdx.x * D.n_small + i] = S.gsw[i];
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int ww = 0; ww < W; ++ww) t += wloss[ww];
      A.loss_part[blockIdx.x] = t * inv_bt;
    }
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

static size_t caser_lds_bytes(const DrxCaserDims &D, bool train, int waves) {
  const int nx = D.n_v + D.L * D.n_h;
  return ((size_t)(train ? D.n_small : 0) + (size_t)waves * (2 * (size_t)D.L * 64 + 5 * (size_t)nx + 64)) * 4 + 64;
}

// waves per workgroup: as many (16, 8, 4, 2, 1) as fit the CU's LDS beside the shared accumulators, and no more than the batch needs
static int caser_waves(const DrxCaserDims &D, bool train, int B) {
  int w = 16;
  while (w > 1 && (caser_lds_bytes(D, train, w) > 150 * 1024 || (B + w - 1) / w < 128)) w >>= 1;
  return w;
}

static int check_dims(const DrxCaserDims *D) {
  if (!D || D->L < 1 || D->L > kCaserMaxL || D->d < 1 || D->d > 64 || D->ld < D->d || (D->ld & 3) || D->ld2 < 2 * D->d ||
      (D->ld2 & 3) || D->n_v < 1 || D->n_h < 1 || D->T < 1 || D->Tp < D->T || D->n_small < 1 || D->act_h < 0 || D->act_h > 3 ||
      D->act_mlp < 0 || D->act_mlp > 3)
    return DRX_EINVAL;
  return caser_lds_bytes(*D, true, 1) <= 150 * 1024 ? DRX_OK : DRX_EINVAL;
}

}  // namespace drx

using namespace drx;

extern "C" {

int drx_caser_grid(const DrxCaserDims *D, int32_t B) {
  if (!D || B < 1) return 0;
  const int w = caser_waves(*D, true, B);
  const int g = (B + w - 1) / w;
  return g < 512 ? g : 512;
}

int drx_caser_fwd_bwd(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->item_emb || !A->user_emb || !A->W1 || !A->b1 || !A->sw || !A->uid || !A->before || !A->after || !A->dE ||
      !A->dW1 || !A->db1 || !A->dPu || !A->gsw_part || !A->loss_part || !gsw_out || A->B < 1 || A->rate < 0.f || A->rate >= 1.f)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = drx_caser_grid(D, A->B);
  const int waves = caser_waves(*D, true, A->B);
  const size_t lds = caser_lds_bytes(*D, true, waves);
  DRX_HIP(hipFuncSetAttribute((const void *)k_caser<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_caser<true>, dim3(grid), dim3(64 * waves), lds, st, *D, *A);
  hipLaunchKernelGGL(k_sum_partials, dim3((D->n_small + 64) / 64), dim3(1024), 0, st, A->gsw_part, grid, D->n_small,
                     A->loss_part, gsw_out);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_caser_hidden(const DrxCaserDims *D, const DrxCaserArgs *A, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->item_emb || !A->user_emb || !A->sw || !A->uid || !A->before || !A->cat_out || A->B < 1) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int waves = 4;
  const size_t lds = caser_lds_bytes(*D, false, waves);
  const int g = (A->B + waves - 1) / waves;
  hipLaunchKernelGGL(k_caser<false>, dim3(g < 2048 ? g : 2048), dim3(64 * waves), lds, st, *D, *A);
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
