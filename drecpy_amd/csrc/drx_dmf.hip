// DMF (DRecPy/Recommender/dmf.py) forward / backward on gfx950, plus the bf16-MFMA all-pairs cosine scorer.
//
// One wavefront per (user, item, target) triple, lane k = hidden units k, k + 64, ... (NU = 1: layer widths <= 64 — the default towers
// [64, 32]; NU = 2: widths <= 128 — examples/consistency_eval/dmf.py:20 builds [128, 64]).  The first Dense layer of
// each tower acts on an l2-normalised sparse interaction row / column (dmf.py:75-86): it is an embedding bag over the
// first-layer kernel rows (one coalesced row read per non-zero), not a [1,N]x[N,f] product.  Deeper layers are tiny
// (64x32): activations are broadcast with wave shuffles, kernels read straight from L2.  Small-weight gradients are
// accumulated per workgroup in LDS by their owning lane (fixed order, no atomics); the first-layer kernel gradient
// leaves as (row key, sample, coefficient) touches + one dz0 row per sample for drx_scatter_rows.
#include "drx_common.hpp"
#include "drx_rows.hpp"

namespace drx {

constexpr float kL2NEps = 1e-12f;
constexpr int kDmfMaxLayers = 4;

// value of lane j (the same j for the whole wave: a loop counter) — v_readlane_b32 instead of a ds_bpermute through the LDS pipe
__device__ __forceinline__ float lane_value(float x, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
}

struct TowerIO {
  const float *K0;            // [rows, ld0] first-layer kernel
  int ld0;
  const int64_t *indptr;      // CSR (user tower) / CSC (item tower) of the interaction matrix, raw values
  const int32_t *indices;
  const float *values;
  const int32_t *ids;         // [B] uid / iid
  const int32_t *off;         // [B+1] touch offsets
  float *dz0;                 // [B, ld0]
  uint32_t *tkeys, *tsrc;     // [T] touches: first-layer row, sample
  float *tcoef;               // [T] normalised input value
  const float *rho = nullptr; // [ids' range] l2 normaliser of every row / column, precomputed (drx_dmf_norms), or null
};

// 1 / |row|_2 of a sparse row as every kernel here forms it (lane partial sums, then a butterfly): one value per id of the dataset
__device__ __forceinline__ float row_rho(const DrxDmfDims &D, const float *values, int64_t s, int64_t e, int k) {
  float q = 0.f;
  for (int64_t j = s + k; j < e; j += 64) { const float v = values[j]; q = fmaf(v, v, q); }
  q = group_sum<64>(q);
  return D.l2_norm_vectors ? rsqrtf(fmaxf(q, kL2NEps)) : 1.0f;
}

__global__ __launch_bounds__(256) void k_dmf_norms(DrxDmfDims D, const int64_t *indptr, const float *values, int n, float *out) {
  const int k = threadIdx.x & 63;
  for (int id = blockIdx.x * 4 + (threadIdx.x >> 6); id < n; id += gridDim.x * 4) {
    const float rho = row_rho(D, values, indptr[id], indptr[id + 1], k);
    if (k == 0) out[id] = rho;
  }
}

// waves of a workgroup that share the sparse first layer of one sample (WV below): 16 for batches that cannot fill the chip
// otherwise (one workgroup per sample, the gather of ~150-250 rows is the sample's critical path: 0.37 -> 0.29 ms per step at
// B = 256), 4 for large ones (more samples in flight per CU: 1.17 vs 1.48 ms at B = 4096)
inline int dmf_waves(int B) { return B <= 512 ? 16 : (B <= 2048 ? 8 : 4); }

// First (sparse) layer of one tower for sample b: the l2-normalised rating row / column against the kernel rows it names —
// an embedding bag.  The workgroup's WV waves take every WV-th block of 64 non-zeros, 8 kernel rows in flight each;
// returns this wave's partial pre-activation of lane k (popular items have thousands of non-zeros: with one wave and
// 4 loads in flight this loop was 60 % of a DMF step).
template <int WV, int NU>
__device__ __forceinline__ void tower_gather(const DrxDmfDims &D, int tw, const TowerIO &T, int b, int k, int w, float (&acc)[NU]) {
  const int id = T.ids[b];
  const int64_t s = T.indptr[id], e = T.indptr[id + 1];
  float q = 0.f;
  for (int64_t j = s + k; j < e; j += 64) { const float v = T.values[j]; q = fmaf(v, v, q); }
  q = group_sum<64>(q);
  const float rho_in = D.l2_norm_vectors ? rsqrtf(fmaxf(q, kL2NEps)) : 1.0f;
  const int f0 = D.f[tw][0];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = 0.f;
  for (int64_t c = s + 64 * (int64_t)w; c < e; c += 64 * WV) {   // lanes fetch 64 (index, value) pairs, then broadcast them
    const int64_t j = c + k;
    int idx = 0;
    float v = 0.f;
    if (j < e) {
      idx = T.indices[j];
      v = T.values[j] * rho_in;
    }
    const int n_here = (int)((e - c) < 64 ? (e - c) : 64);
    constexpr int UF = 8 / NU;                        // kernel rows in flight
    for (int t = 0; t < n_here; t += UF) {
      float r[UF][NU], vv[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int iu = __shfl(idx, t + u);
        vv[u] = (t + u < n_here) ? __shfl(v, t + u) : 0.f;
#pragma unroll
        for (int h = 0; h < NU; ++h) r[u][h] = (k + 64 * h < f0 && t + u < n_here) ? T.K0[(size_t)iu * T.ld0 + k + 64 * h] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UF; ++u)
#pragma unroll
        for (int h = 0; h < NU; ++h) acc[h] = fmaf(vv[u], r[u][h], acc[h]);
    }
  }
}

// The same embedding bag with 16 NU lanes per kernel row (a float4 each): one load instruction of the wave fetches 4 / NU rows, one per
// part of the wave, so a non-zero costs a fraction of the loads and broadcasts of tower_gather.  Part r takes the non-zeros t with
// t % (4 / NU) == r; the partial sums are combined by exchanges.  Writes this wave's partial pre-activations (column k at out[k]) —
// the training path (k_dmf_gather).
template <int WV, int NU>
__device__ __forceinline__ void tower_gather_q(const DrxDmfDims &D, int tw, const TowerIO &T, int b, int k, int w, float *out, int seg = 0,
                                               int seg_len = 0) {
  constexpr int LPR = 16 * NU, RPW = 64 / LPR;      // lanes per kernel row, rows per load instruction
  const int id = T.ids[b];
  const int64_t s0 = T.indptr[id], e0 = T.indptr[id + 1];
  // (a long row / column cut into segments of seg_len non-zeros: DrxDmfArgs::work_order — this workgroup's part of it)
  const int64_t s = seg_len > 0 ? s0 + (int64_t)seg * seg_len : s0;
  const int64_t e = seg_len > 0 ? (s + seg_len < e0 ? s + seg_len : e0) : e0;
  // (a popular item has thousands of non-zeros: forming the norm here, one load in flight per wave, was the kernel's tail)
  const float rho_in = T.rho ? T.rho[id] : row_rho(D, T.values, s0, e0, k);
  const int r = k / LPR, c = k % LPR;
  const bool ok = 4 * c < T.ld0;
  float4 acc = f4_zero();
  const bool touches = T.tkeys != nullptr;          // (the scan update of the first layer needs none: k_dmf_k0_update)
  const int base = touches ? T.off[b] : 0;
  for (int64_t c0 = s + 64 * (int64_t)w; c0 < e; c0 += 64 * WV) {
    const int64_t j = c0 + k;
    int idx = 0;
    float v = 0.f;
    if (j < e) {
      idx = T.indices[j];
      v = T.values[j] * rho_in;
      if (touches) { T.tkeys[base + (j - s0)] = (uint32_t)idx; T.tsrc[base + (j - s0)] = (uint32_t)b; T.tcoef[base + (j - s0)] = v; }
    }
    const int n_here = (int)((e - c0) < 64 ? (e - c0) : 64);
    for (int t = 0; t < n_here; t += 8 * RPW) {
      float4 rowv[8];
      float vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tt = t + RPW * u + r;
        const int iu = __shfl(idx, tt & 63);
        const float vs = __shfl(v, tt & 63);
        const bool on = tt < n_here;
        vv[u] = on ? vs : 0.f;
        rowv[u] = (on && ok) ? *reinterpret_cast<const float4 *>(T.K0 + (size_t)iu * T.ld0 + 4 * c) : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) f4_fma(acc, vv[u], rowv[u]);
    }
  }
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
  }
  if (r == 0) *reinterpret_cast<float4 *>(out + 4 * c) = acc;
}

// Dense layers of one tower from the summed first-layer pre-activation; returns the final activation of lane k's units.  zs[l], as[l]
// keep pre/post activations.  Unit j of a layer lives on lane j % 64, slot j / 64.
template <int NU>
__device__ __forceinline__ void tower_dense(const DrxDmfDims &D, int tw, const float *sw, int k, const float (&acc)[NU],
                                            float (&zs)[kDmfMaxLayers][NU], float (&as)[kDmfMaxLayers][NU], float (&out)[NU]) {
  const int nl = D.n_layers[tw];
  const int f0 = D.f[tw][0];
  float a[NU];
#pragma unroll
  for (int h = 0; h < NU; ++h) {
    const float z = k + 64 * h < f0 ? acc[h] + sw[D.off_b[tw][0] + k + 64 * h] : 0.f;
    a[h] = fmaxf(z, 0.f);
    zs[0][h] = z; as[0][h] = a[h];
  }
#pragma unroll
  for (int l = 1; l < kDmfMaxLayers; ++l) {
    if (l >= nl) break;
    const int fin = D.f[tw][l - 1], fo = D.f[tw][l];
    float zz[NU];
#pragma unroll
    for (int h = 0; h < NU; ++h) zz[h] = k + 64 * h < fo ? sw[D.off_b[tw][l] + k + 64 * h] : 0.f;
#pragma unroll
    for (int hj = 0; hj < NU; ++hj) {
      const int jn = min(64, fin - 64 * hj);
      for (int j = 0; j < jn; ++j) {
        const float aj = lane_value(a[hj], j);
#pragma unroll
        for (int h = 0; h < NU; ++h)
          if (k + 64 * h < fo) zz[h] = fmaf(aj, sw[D.off_k[tw][l] + (64 * hj + j) * fo + k + 64 * h], zz[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < NU; ++h) { a[h] = fmaxf(zz[h], 0.f); zs[l][h] = zz[h]; as[l][h] = a[h]; }
  }
#pragma unroll
  for (int h = 0; h < NU; ++h) out[h] = a[h];
}

// the sample's cosine from its towers' outputs (l2_normalize each, dot): shared by the predict kernel and the training step
template <int NU>
__device__ __forceinline__ float wave_sum_units(const float (&x)[NU], const float (&y)[NU]) {
  float t = 0.f;
#pragma unroll
  for (int h = 0; h < NU; ++h) t = fmaf(x[h], y[h], t);
  return group_sum<64>(t);
}

// drx_dmf_predict: one workgroup per pair — its WV waves gather the two sparse first layers together, one of them runs the dense
// layers and the cosine.  (r01's training step had this shape too; r02 split it into the three kernels below.)
template <int WV, int NU>
__global__ __launch_bounds__(64 * WV) void k_dmf_predict(DrxDmfDims D, DrxDmfArgs A) {
  extern __shared__ __align__(16) float lds[];       // [2][WV][64 NU] first-layer partials
  constexpr int W = 64 * NU;
  float *part = lds;
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6;
  TowerIO Tu{A.K0u, D.ld0[0], A.u_indptr, A.u_indices, A.u_values, A.uid, A.off_u, A.dz0u, A.tkeys_u, A.tsrc_u, A.tcoef_u};
  TowerIO Ti{A.K0i, D.ld0[1], A.i_indptr, A.i_indices, A.i_values, A.iid, A.off_i, A.dz0i, A.tkeys_i, A.tsrc_i, A.tcoef_i};
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    float gu[NU], gi[NU];
    tower_gather<WV, NU>(D, 0, Tu, b, k, w, gu);
    tower_gather<WV, NU>(D, 1, Ti, b, k, w, gi);
#pragma unroll
    for (int h = 0; h < NU; ++h) { part[(0 * WV + w) * W + k + 64 * h] = gu[h]; part[(1 * WV + w) * W + k + 64 * h] = gi[h]; }
    __syncthreads();
    if (w == 0) {                                    // one wave finishes the sample: dense layers, cosine
      float pu[NU], pi[NU];
#pragma unroll
      for (int h = 0; h < NU; ++h) {
        pu[h] = 0.f; pi[h] = 0.f;
#pragma unroll
        for (int ww = 0; ww < WV; ++ww) { pu[h] += part[(0 * WV + ww) * W + k + 64 * h]; pi[h] += part[(1 * WV + ww) * W + k + 64 * h]; }
      }
      float zu[kDmfMaxLayers][NU], au[kDmfMaxLayers][NU], zi[kDmfMaxLayers][NU], ai[kDmfMaxLayers][NU], ru[NU], ri[NU];
      tower_dense<NU>(D, 0, A.sw, k, pu, zu, au, ru);
      tower_dense<NU>(D, 1, A.sw, k, pi, zi, ai, ri);
      const float qu = wave_sum_units<NU>(ru, ru), qi = wave_sum_units<NU>(ri, ri);
      const float rhou = rsqrtf(fmaxf(qu, kL2NEps)), rhoi = rsqrtf(fmaxf(qi, kL2NEps));
      float nu[NU], ni[NU];
#pragma unroll
      for (int h = 0; h < NU; ++h) { nu[h] = ru[h] * rhou; ni[h] = ri[h] * rhoi; }
      const float s = wave_sum_units<NU>(nu, ni);
      const float cosv = fmaxf(1e-6f, s);
      // optional registered scalar multiplying every prediction (examples/extending_recommender_dmf.py:9-18)
      const float wsc = D.off_scale >= 0 ? A.sw[D.off_scale] : 1.0f;
      if (k == 0 && A.pred_out) A.pred_out[b] = wsc * cosv;
#pragma unroll
      for (int h = 0; h < NU; ++h) {                 // l2-normalised representations (zero beyond f_last), rows of 64 NU floats
        if (A.rep_u_out) A.rep_u_out[(size_t)b * W + k + 64 * h] = nu[h];
        if (A.rep_i_out) A.rep_i_out[(size_t)b * W + k + 64 * h] = ni[h];
      }
    }
    __syncthreads();                                 // the partials are free for the next sample
  }
}

// ---- training step in three kernels (r02) ------------------------------------------------------------------------------------
// k_dmf<true> did everything for a sample inside one workgroup: its waves gathered the two sparse first layers together, then ONE of
// them ran the dense layers, the loss and the whole backward while the others waited at the barrier, adding every small-weight
// gradient into LDS accumulators on the way (0.47 ms at B = 4096; 84 % of the wave-cycles waiting).  The three phases want different
// shapes, so they are kernels of their own:
//   k_dmf_gather   memory-bound: the embedding bags of both towers, WV waves per sample -> z0[tw][b][64] (+ the scatter's touches)
//   k_dmf_dense    ALU-bound: one wave per sample, small weights staged in LDS: dense layers, cosine, loss, backward; stores every
//                  layer's activation and pre-activation gradient rows — no shared accumulators, no barrier between samples
//   k_dmf_wgrad    the small-weight gradients as what they are, tiny products dW_l = A_{l-1}^T DZ_l (and column sums for the
//                  biases), over fixed chunks of the batch in sample order -> partial rows for k_sum_partials2
// work area (caller's, drx_dmf_work_bytes): z0 [2][B][W] | act [2][B][4][W] | dz [2][B][4][W] | samp [B][2] (loss, d scale), W = 64 unit slots
struct DmfWork {
  float *z0, *act, *dz, *samp;
  int W;                       // floats per row of z0 / act / dz: 64 per unit slot of a lane
};
__host__ __device__ inline int dmf_units(const DrxDmfDims &D) {      // unit slots per lane: 1 (widths <= 64) or 2 (<= 128)
  int m = 0;
  for (int tw = 0; tw < 2; ++tw) {
    for (int l = 0; l < D.n_layers[tw]; ++l) m = D.f[tw][l] > m ? D.f[tw][l] : m;
    m = D.ld0[tw] > m ? D.ld0[tw] : m;
  }
  return m > 64 ? 2 : 1;
}
__host__ __device__ inline DmfWork dmf_work(float *base, int B, int W) {
  DmfWork Wk;
  Wk.W = W;
  Wk.z0 = base;
  Wk.act = Wk.z0 + (size_t)2 * B * W;
  Wk.dz = Wk.act + (size_t)2 * B * kDmfMaxLayers * W;
  Wk.samp = Wk.dz + (size_t)2 * B * kDmfMaxLayers * W;
  return Wk;
}

template <int WV, int NU>
__global__ __launch_bounds__(64 * WV) void k_dmf_gather(DrxDmfDims D, DrxDmfArgs A) {
  // one workgroup per DISTINCT user / item of the batch: the first layer of a tower depends on the id alone, and a batch of 4096
  // pairs over 6040 users / 3706 items repeats ids a lot (popular items above all, whose columns are the long ones)
  constexpr int W = 64 * NU;
  __shared__ __align__(16) float part[WV * W];
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6;
  const DmfWork Wk = dmf_work(A.work, A.B, W);
  TowerIO Tu{A.K0u, D.ld0[0], A.u_indptr, A.u_indices, A.u_values, A.uid, A.off_u, A.dz0u, A.tkeys_u, A.tsrc_u, A.tcoef_u};
  TowerIO Ti{A.K0i, D.ld0[1], A.i_indptr, A.i_indices, A.i_values, A.iid, A.off_i, A.dz0i, A.tkeys_i, A.tsrc_i, A.tcoef_i};
  Tu.rho = A.rho_u; Ti.rho = A.rho_i;
  const int n_du = A.nd_dev ? A.nd_dev[0] : A.n_du;
  // (longest rows first, long ones in segments: include/drx.h DrxDmfArgs::work_order; a device-prepared batch brings a list built there)
  const bool listed = A.work_order && (!A.nd_dev || A.n_work_dev);
  const int total = listed ? (A.n_work_dev ? A.n_work_dev[0] : A.n_work) : n_du + (A.nd_dev ? A.nd_dev[1] : A.n_di);
  for (int it0 = blockIdx.x; it0 < total; it0 += gridDim.x) {
    const int enc = listed ? A.work_order[it0] : it0;
    const int it = enc & 0xFFFFFF, seg = listed ? (int)((unsigned)enc >> 24) : 0, seg_len = listed ? A.seg_len : 0;
    const int tw = it < n_du ? 0 : 1, d = tw ? it - n_du : it;
    if (tw) tower_gather_q<WV, NU>(D, 1, Ti, d, k, w, part + w * W, seg, seg_len);
    else tower_gather_q<WV, NU>(D, 0, Tu, d, k, w, part + w * W, seg, seg_len);
    if (A.map_u && threadIdx.x == 0 && seg == 0) {   // for k_dmf_k0_update: which distinct index this id has in THIS step
      const int id = tw ? A.iid[d] : A.uid[d];
      (tw ? A.map_i : A.map_u)[id] = ((unsigned long long)A.stamp << 32) | (unsigned long long)(uint32_t)d;
    }
    __syncthreads();
    if (w == 0) {                                    // partials summed in wave order
#pragma unroll
      for (int h = 0; h < NU; ++h) {
        float p = 0.f;
#pragma unroll
        for (int ww = 0; ww < WV; ++ww) p += part[ww * W + k + 64 * h];
        // segment 0: the id's first-layer sum; a later segment: its partial row (the dense kernel adds them in segment order)
        if (seg == 0) Wk.z0[((size_t)tw * A.B + d) * W + k + 64 * h] = p;
        else A.zpart[((size_t)(A.zseg[it] >> 8) + seg - 1) * W + k + 64 * h] = p;
      }
    }
    __syncthreads();
  }
}

// dz0 of a distinct id = sum of the first-layer pre-activation gradients of the samples that carry it, in sample order (gptr / grows =
// CSR of samples per distinct id): the scatter then runs over the touches of DISTINCT ids.
__global__ __launch_bounds__(256) void k_dmf_dzsum(DrxDmfDims D, DrxDmfArgs A, int W) {
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6;
  const DmfWork Wk = dmf_work(A.work, A.B, W);
  const int n_du = A.nd_dev ? A.nd_dev[0] : A.n_du, total = n_du + (A.nd_dev ? A.nd_dev[1] : A.n_di);
  for (int it = blockIdx.x * 4 + w; it < total; it += gridDim.x * 4) {
    const int tw = it < n_du ? 0 : 1, d = tw ? it - n_du : it;
    const int32_t *gp = tw ? A.gptr_i : A.gptr_u, *gr = tw ? A.grows_i : A.grows_u;
    const int ld0 = D.ld0[tw];
    float *out = tw ? A.dz0i : A.dz0u;
    for (int kk = k; kk < ld0; kk += 64) {
      float acc = 0.f;
      for (int q = gp[d]; q < gp[d + 1]; ++q) acc += Wk.dz[((size_t)tw * A.B + gr[q]) * kDmfMaxLayers * W + kk];
      out[(size_t)d * ld0 + kk] = acc;
    }
  }
}

// First-layer kernels K0u [N, ld0] / K0i [U, ld0]: gradient + dense Keras Adam in ONE pass over the table, no touch list and no sort.
//   dK0u[n] = sum over the batch's distinct users d that have item n in their row of  (value(d, n) * rho_d) * dz0u[d]
// Row n's candidates are column n of the interaction matrix — the CSC the item tower reads anyway (for K0i: the CSR) — and a
// candidate is in the batch iff map[id] carries this step's stamp (written by k_dmf_gather together with rho_d; the map is never
// cleared).  Every step walks all nnz entries once (8 bytes each + an L2-resident map probe: 16 MB at ml-1m) instead of sorting the
// ~40 k - 500 k touches of the batch twice and reducing them through a dense gradient arena; Adam's moments decay on every row of
// these tables anyway (dense tf.Variable), so the update of a row rides in the same workgroup.
// One workgroup per row: its 4 waves take every 4th block of 64 candidates; a wave compacts its hits in LDS and its quarter-waves
// (16 lanes x float4 = one dz0 row) take every 4th hit; sums are combined quarter by quarter, wave by wave: a fixed order.
struct K0Tables {
  float *K0[2], *m[2], *v[2];
  int rows[2];
  float alpha[2];
  float l2c, b1, b2, eps;
  const int32_t *order;       // workgroup -> row (longest first), or nullptr
  int n_long;                 // with `order`: its first n_long rows take a workgroup each, the others a WAVE each (four per workgroup)
};

// One WAVE per kernel row (r06; rows whose column / row of the interaction matrix holds at most kK0WaveMax entries: all but a few hundred).
// A workgroup per row left four fifths of its lanes idle on the average row (193 entries against 1024 per round) and kept a CU at 8 rows
// in flight; a wave per row keeps 32.  Same walk, the wave's own 256 entries per round, no cross-wave combine.
constexpr int kK0WaveMax = 1024;
__device__ __forceinline__ void k0_update_wave_row(const DrxDmfDims &D, const DrxDmfArgs &A, const K0Tables &U, int row, int (*hd)[64],
                                                   float (*hc)[64]) {
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6, r = k >> 4, c = k & 15;
  const int tw = row < U.rows[0] ? 0 : 1;
  const int n = tw ? row - U.rows[0] : row;
  const int ld0 = D.ld0[tw];
  const int64_t *ip = tw ? A.u_indptr : A.i_indptr;
  const int32_t *ix = tw ? A.u_indices : A.i_indices;
  const float *vals = tw ? A.u_values : A.i_values;
  const unsigned long long *map = tw ? A.map_i : A.map_u;
  const float *rho = tw ? A.rho_i : A.rho_u;
  const float *dz0 = tw ? A.dz0i : A.dz0u;
  const bool col = 4 * c < ld0;
  const bool upd = r == 0 && col;
  float4 p = f4_zero(), m = f4_zero(), v = f4_zero();
  const size_t at = (size_t)n * ld0 + 4 * c;
  if (upd) {
    p = *reinterpret_cast<const float4 *>(U.K0[tw] + at);
    m = *reinterpret_cast<const float4 *>(U.m[tw] + at);
    v = *reinterpret_cast<const float4 *>(U.v[tw] + at);
  }
  const int64_t s = ip[n], e = ip[n + 1];
  float4 acc = f4_zero();
  constexpr int UN = 4;
  for (int64_t c0 = s; c0 < e; c0 += 64 * UN) {
    int id[UN];
    float val[UN];
    unsigned long long ent[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t j = c0 + 64 * u + k;
      id[u] = j < e ? ix[j] : -1;
      val[u] = j < e ? vals[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      ent[u] = id[u] >= 0 ? map[id[u]] : 0ull;
      val[u] *= id[u] >= 0 ? rho[id[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (c0 + 64 * u >= e) break;                   // (wave-uniform)
      const bool hit = (uint32_t)(ent[u] >> 32) == A.stamp;
      const unsigned long long mask = __ballot(hit);
      if (!mask) continue;
      if (hit) {
        const int rank = __popcll(mask & ((1ull << k) - 1ull));
        hd[w][rank] = (int)(uint32_t)ent[u];
        hc[w][rank] = val[u];
      }
      wave_lds_sync();
      const int nh = __popcll(mask);
      int t = r;
      for (; t + 4 < nh; t += 8) {                   // two dz0 rows in flight per quarter-wave
        const int d0 = hd[w][t], d1 = hd[w][t + 4];
        const float c0f = hc[w][t], c1f = hc[w][t + 4];
        float4 x0 = f4_zero(), x1 = f4_zero();
        if (col) {
          x0 = *reinterpret_cast<const float4 *>(dz0 + (size_t)d0 * ld0 + 4 * c);
          x1 = *reinterpret_cast<const float4 *>(dz0 + (size_t)d1 * ld0 + 4 * c);
        }
        f4_fma(acc, c0f, x0);
        f4_fma(acc, c1f, x1);
      }
      if (t < nh) {
        const int dd = hd[w][t];
        const float cc = hc[w][t];
        if (col) f4_fma(acc, cc, *reinterpret_cast<const float4 *>(dz0 + (size_t)dd * ld0 + 4 * c));
      }
      wave_lds_sync();
    }
  }
  acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
  if (upd) {
    const OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, U.b1, U.b2, U.eps, U.alpha[tw]};
    opt_update1(o, fmaf(U.l2c, p.x, acc.x), p.x, m.x, v.x);
    opt_update1(o, fmaf(U.l2c, p.y, acc.y), p.y, m.y, v.y);
    opt_update1(o, fmaf(U.l2c, p.z, acc.z), p.z, m.z, v.z);
    opt_update1(o, fmaf(U.l2c, p.w, acc.w), p.w, m.w, v.w);
    *reinterpret_cast<float4 *>(U.K0[tw] + at) = p;
    *reinterpret_cast<float4 *>(U.m[tw] + at) = m;
    *reinterpret_cast<float4 *>(U.v[tw] + at) = v;
  }
}

__global__ __launch_bounds__(256) void k_dmf_k0_update(DrxDmfDims D, DrxDmfArgs A, K0Tables U) {
  __shared__ int hd[4][64];
  __shared__ float hc[4][64];
  __shared__ float4 part[4][16];
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6, r = k >> 4, c = k & 15;
  if (U.order && (int)blockIdx.x >= U.n_long) {      // a workgroup of four short rows, a wave each (whole waves leave: no barrier in there)
    const int i = U.n_long + ((int)blockIdx.x - U.n_long) * 4 + w;
    if (i < U.rows[0] + U.rows[1]) k0_update_wave_row(D, A, U, U.order[i], hd, hc);
    return;
  }
  const int row = U.order ? U.order[blockIdx.x] : (int)blockIdx.x;
  const int tw = row < U.rows[0] ? 0 : 1;
  const int n = tw ? row - U.rows[0] : row;
  const int ld0 = D.ld0[tw];
  const int64_t *ip = tw ? A.u_indptr : A.i_indptr;
  const int32_t *ix = tw ? A.u_indices : A.i_indices;
  const float *vals = tw ? A.u_values : A.i_values;
  const unsigned long long *map = tw ? A.map_i : A.map_u;
  const float *rho = tw ? A.rho_i : A.rho_u;
  const float *dz0 = tw ? A.dz0i : A.dz0u;
  const bool col = 4 * c < ld0;
  const bool upd = w == 0 && r == 0 && col;
  float4 p = f4_zero(), m = f4_zero(), v = f4_zero();
  const size_t at = (size_t)n * ld0 + 4 * c;
  if (upd) {                                         // (issued before the scan: the row is not written by anyone else)
    p = *reinterpret_cast<const float4 *>(U.K0[tw] + at);
    m = *reinterpret_cast<const float4 *>(U.m[tw] + at);
    v = *reinterpret_cast<const float4 *>(U.v[tw] + at);
  }
  const int64_t s = ip[n], e = ip[n + 1];
  float4 acc = f4_zero();
  constexpr int UN = 4;                              // blocks of 64 candidates a wave has in flight (popular items: 3 400 candidates)
  for (int64_t c0 = s + 64 * (int64_t)w; c0 < e; c0 += 256 * UN) {
    int id[UN];
    float val[UN];
    unsigned long long ent[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t j = c0 + 256 * u + k;
      id[u] = j < e ? ix[j] : -1;
      val[u] = j < e ? vals[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      ent[u] = id[u] >= 0 ? map[id[u]] : 0ull;
      val[u] *= id[u] >= 0 ? rho[id[u]] : 0.f;       // the normalised input value (what the forward multiplied the kernel row by)
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (c0 + 256 * u >= e) break;                  // (wave-uniform)
      const bool hit = (uint32_t)(ent[u] >> 32) == A.stamp;
      const unsigned long long mask = __ballot(hit);
      if (!mask) continue;
      if (hit) {
        const int d = (int)(uint32_t)ent[u];
        const int rank = __popcll(mask & ((1ull << k) - 1ull));
        hd[w][rank] = d;
        hc[w][rank] = val[u];
      }
      wave_lds_sync();
      const int nh = __popcll(mask);
      int t = r;
      for (; t + 4 < nh; t += 8) {                   // two dz0 rows in flight per quarter-wave
        const int d0 = hd[w][t], d1 = hd[w][t + 4];
        const float c0f = hc[w][t], c1f = hc[w][t + 4];
        float4 x0 = f4_zero(), x1 = f4_zero();
        if (col) {
          x0 = *reinterpret_cast<const float4 *>(dz0 + (size_t)d0 * ld0 + 4 * c);
          x1 = *reinterpret_cast<const float4 *>(dz0 + (size_t)d1 * ld0 + 4 * c);
        }
        f4_fma(acc, c0f, x0);
        f4_fma(acc, c1f, x1);
      }
      if (t < nh) {
        const int dd = hd[w][t];
        const float cc = hc[w][t];
        if (col) f4_fma(acc, cc, *reinterpret_cast<const float4 *>(dz0 + (size_t)dd * ld0 + 4 * c));
      }
      wave_lds_sync();
    }
  }
  acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
  acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
  if (r == 0) part[w][c] = acc;
  __syncthreads();
  if (upd) {
    float4 g = part[0][c];
    f4_add(g, part[1][c]); f4_add(g, part[2][c]); f4_add(g, part[3][c]);
    const OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, U.b1, U.b2, U.eps, U.alpha[tw]};
    opt_update1(o, fmaf(U.l2c, p.x, g.x), p.x, m.x, v.x);
    opt_update1(o, fmaf(U.l2c, p.y, g.y), p.y, m.y, v.y);
    opt_update1(o, fmaf(U.l2c, p.z, g.z), p.z, m.z, v.z);
    opt_update1(o, fmaf(U.l2c, p.w, g.w), p.w, m.w, v.w);
    *reinterpret_cast<float4 *>(U.K0[tw] + at) = p;
    *reinterpret_cast<float4 *>(U.m[tw] + at) = m;
    *reinterpret_cast<float4 *>(U.v[tw] + at) = v;
  }
}

// Backward of one tower given da = dL/d(final activation) on lane k's units: stores dz_l rows (and leaves the activations stored by the
// forward) for k_dmf_wgrad, the first layer's dz0 row for the scatter.
template <int NU>
__device__ __forceinline__ void tower_bwd_store(const DrxDmfDims &D, int tw, const float *sw, float *dzrow /* [4][64 NU] */, int k,
                                                const float (&zs)[kDmfMaxLayers][NU], const float (&da_in)[NU]) {
  constexpr int W = 64 * NU;
  const int nl = D.n_layers[tw];
  float da[NU];
#pragma unroll
  for (int h = 0; h < NU; ++h) da[h] = da_in[h];
#pragma unroll
  for (int l = kDmfMaxLayers - 1; l >= 1; --l) {
    if (l >= nl) continue;
    const int fin = D.f[tw][l - 1], fo = D.f[tw][l];
    float dz[NU], dprev[NU];
#pragma unroll
    for (int h = 0; h < NU; ++h) {
      dz[h] = (k + 64 * h < fo && zs[l][h] > 0.f) ? da[h] : 0.f;
      dzrow[l * W + k + 64 * h] = dz[h];
      dprev[h] = 0.f;
    }
#pragma unroll
    for (int hk = 0; hk < NU; ++hk) {                // da_{l-1}[j] = sum_k dz[k] * K[j][k] : lane j walks its kernel rows
      const int kn = min(64, fo - 64 * hk);
      for (int kk = 0; kk < kn; ++kk) {
        const float dzk = lane_value(dz[hk], kk);
#pragma unroll
        for (int h = 0; h < NU; ++h)
          if (k + 64 * h < fin) dprev[h] = fmaf(dzk, sw[D.off_k[tw][l] + (k + 64 * h) * fo + 64 * hk + kk], dprev[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < NU; ++h) da[h] = dprev[h];
  }
  const int f0 = D.f[tw][0];
#pragma unroll
  for (int h = 0; h < NU; ++h)                       // (zero beyond f0; k_dmf_dzsum folds these rows per distinct id)
    dzrow[k + 64 * h] = (k + 64 * h < f0 && zs[0][h] > 0.f) ? da[h] : 0.f;
}

template <int NU>
__global__ __launch_bounds__(256) void k_dmf_dense(DrxDmfDims D, DrxDmfArgs A) {
  extern __shared__ __align__(16) float swl[];       // [n_small] the small weights
  constexpr int W = 64 * NU;
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < D.n_small; i += 256) swl[i] = A.sw[i];
  __syncthreads();
  const DmfWork Wk = dmf_work(A.work, A.B, W);
  const float inv_b = 1.0f / (float)A.B;
  for (int b = blockIdx.x * 4 + w; b < A.B; b += gridDim.x * 4) {
    float pu[NU], pi[NU];
#pragma unroll
    for (int h = 0; h < NU; ++h) {
      pu[h] = Wk.z0[((size_t)0 * A.B + A.inv_u[b]) * W + k + 64 * h];
      pi[h] = Wk.z0[((size_t)1 * A.B + A.inv_i[b]) * W + k + 64 * h];
    }
    if (A.zseg && A.work_order && (!A.nd_dev || A.n_work_dev)) {       // long rows / columns gathered in segments: their partial rows, in segment order
      const int zu_ = A.zseg[A.inv_u[b]], zi_ = A.zseg[(A.nd_dev ? A.nd_dev[0] : A.n_du) + A.inv_i[b]];
      for (int g = 0; g < (zu_ & 255); ++g)
#pragma unroll
        for (int h = 0; h < NU; ++h) pu[h] += A.zpart[((size_t)(zu_ >> 8) + g) * W + k + 64 * h];
      for (int g = 0; g < (zi_ & 255); ++g)
#pragma unroll
        for (int h = 0; h < NU; ++h) pi[h] += A.zpart[((size_t)(zi_ >> 8) + g) * W + k + 64 * h];
    }
    float zu[kDmfMaxLayers][NU], au[kDmfMaxLayers][NU], zi[kDmfMaxLayers][NU], ai[kDmfMaxLayers][NU], ru[NU], ri[NU];
    tower_dense<NU>(D, 0, swl, k, pu, zu, au, ru);
    tower_dense<NU>(D, 1, swl, k, pi, zi, ai, ri);
    float *actu = Wk.act + ((size_t)0 * A.B + b) * kDmfMaxLayers * W, *acti = Wk.act + ((size_t)1 * A.B + b) * kDmfMaxLayers * W;
#pragma unroll
    for (int l = 0; l < kDmfMaxLayers; ++l) {
#pragma unroll
      for (int h = 0; h < NU; ++h) {
        if (l < D.n_layers[0]) actu[l * W + k + 64 * h] = au[l][h];
        if (l < D.n_layers[1]) acti[l * W + k + 64 * h] = ai[l][h];
      }
    }
    const float qu = wave_sum_units<NU>(ru, ru), qi = wave_sum_units<NU>(ri, ri);
    const float rhou = rsqrtf(fmaxf(qu, kL2NEps)), rhoi = rsqrtf(fmaxf(qi, kL2NEps));
    float nu[NU], ni[NU];
#pragma unroll
    for (int h = 0; h < NU; ++h) { nu[h] = ru[h] * rhou; ni[h] = ri[h] * rhoi; }
    const float s = wave_sum_units<NU>(nu, ni);
    const float cosv = fmaxf(1e-6f, s);
    const float wsc = D.off_scale >= 0 ? swl[D.off_scale] : 1.0f;
    const float pred = wsc * cosv;
    // target_mode 1: Keras BCE of (B,) targets against (B,1) predictions broadcasts to (B,B); its mean equals the BCE against the
    // batch-mean target because the element is affine in t
    const float y = A.target_mode == 1 ? (A.y_mean_dev ? A.y_mean_dev[0] : A.y_mean) : A.y[b];
    const float gp = bce_grad(y, pred) * inv_b;
    if (k == 0) { Wk.samp[(size_t)b * 2] = bce_elem(y, pred); Wk.samp[(size_t)b * 2 + 1] = gp * cosv; }
    const float ds = s > 1e-6f ? gp * wsc : 0.f;
    // l2_normalize backward (tf.nn.l2_normalize: x * rsqrt(max(sum x^2, eps)))
    float dnu[NU], dni[NU], dru[NU], dri[NU];
#pragma unroll
    for (int h = 0; h < NU; ++h) { dnu[h] = ds * ni[h]; dni[h] = ds * nu[h]; }
    const float du = wave_sum_units<NU>(nu, dnu), di = wave_sum_units<NU>(ni, dni);
#pragma unroll
    for (int h = 0; h < NU; ++h) {
      dru[h] = qu > kL2NEps ? rhou * (dnu[h] - nu[h] * du) : rhou * dnu[h];
      dri[h] = qi > kL2NEps ? rhoi * (dni[h] - ni[h] * di) : rhoi * dni[h];
    }
    tower_bwd_store<NU>(D, 0, swl, Wk.dz + ((size_t)0 * A.B + b) * kDmfMaxLayers * W, k, zu, dru);
    tower_bwd_store<NU>(D, 1, swl, Wk.dz + ((size_t)1 * A.B + b) * kDmfMaxLayers * W, k, zi, dri);
  }
}

// ---- k_dmf_dense on the matrix cores (r06; towers whose layers are all <= 64 wide — the defaults [64, 32], every BASELINE shape) -------
// The wave-per-sample kernel above walks every dense layer as a chain of 64 v_readlane + FMA pairs per sample: 42 us for 4096 samples
// and bound by that dependent chain, not by any pipe.  Here a workgroup of FOUR WAVES (two per tower) takes a TILE OF 16 SAMPLES and every layer is a product whose M
// dimension is the tile, on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) out of LDS:
//   forward   Z_l [16 x fo]   = A_{l-1} [16 x fin] . K_l [fin x fo] + b_l,  A_l = relu(Z_l)          (dmf.py:47-58 Dense layers)
//   cosine    per sample: l2-normalise both towers' outputs, dot, clip, BCE                           (dmf.py:92-99)
//   backward  dA_{l-1} [16 x fin] = dZ_l [16 x fo] . K_l^T,  dZ_l = dA_l where A_l > 0 (relu: A_l > 0 <=> Z_l > 0)
// Operand layout (lane l): A[l % 16][l / 16], B[l / 16][l % 16], C register r = C[4 (l / 16) + r][l % 16].  The tile's activations of
// every layer of both towers stay in LDS (rows of 68 floats: the 16 samples of an A fragment fall in 16 different bank quads) beside the
// small weights; the same rows leave for k_dmf_wgrad / k_dmf_dzsum exactly as the wave kernel wrote them (act, dz, samp): the rest of
// the step does not change.  Sums run in the MFMA's order (k ascending in steps of 4): another association than the wave kernel's —
// same oracle, same tolerance, bit-reproducible.
typedef float dmf_f4v __attribute__((ext_vector_type(4)));
constexpr int kDmfTS = 68;                       // floats per sample row of a tile in LDS
constexpr int kDmfTile = 16 * kDmfTS;

__global__ __launch_bounds__(256) void k_dmf_dense_tile(DrxDmfDims D, DrxDmfArgs A) {
  extern __shared__ __align__(16) float swl[];    // [n_small] weights | X[2][4][16][68] activations | G[2][16][68] gradient tiles
  constexpr int W = 64;
  // FOUR waves per tile: waves 2 tw and 2 tw + 1 run tower tw, splitting its 16-column tiles (and the samples of the row-wise passes)
  // between them — a tile's life is a chain of dependent layers (27 us with one wave, the whole kernel's time: every CU holds one tile)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tw = wv >> 1, half = wv & 1;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int nsw = (D.n_small + 3) & ~3;
  float *const X = swl + nsw;                                    // X[(tw * 4 + l) * kDmfTile + s * 68 + c]
  float *const G = X + 2 * kDmfMaxLayers * kDmfTile;             // one tile per tower
  for (int i = threadIdx.x; i < D.n_small; i += 256) swl[i] = A.sw[i];
  const DmfWork Wk = dmf_work(A.work, A.B, W);
  const float inv_b = 1.0f / (float)A.B;
  const int n_tiles = (A.B + 15) / 16;
  const int nl = D.n_layers[tw], f0 = D.f[tw][0];
  const int32_t *const inv = tw ? A.inv_i : A.inv_u;
  float *const X0 = X + (tw * kDmfMaxLayers) * kDmfTile;
  float *const Gt = G + tw * kDmfTile;
  const int nl_max = D.n_layers[0] > D.n_layers[1] ? D.n_layers[0] : D.n_layers[1];
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b0 = tile * 16;
    __syncthreads();                                            // (the weights' copy; the previous tile's rows are free)
    // ---- forward.  Layer 0: the gathered pre-activations of the samples' ids + bias, relu -> A_0 (LDS + the act rows); a wave: 8 samples
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int s_ = half * 8 + it * 4 + q4, c4 = m16 * 4, b = b0 + s_;
      float4 z = f4_zero();
      if (b < A.B) {
        z = *reinterpret_cast<const float4 *>(Wk.z0 + ((size_t)tw * A.B + inv[b]) * W + c4);
        if (A.zseg && A.work_order && (!A.nd_dev || A.n_work_dev)) {     // a long row / column gathered in segments: its partial rows, in segment order
          const int zs = A.zseg[(tw ? (A.nd_dev ? A.nd_dev[0] : A.n_du) : 0) + inv[b]];
          for (int g = 0; g < (zs & 255); ++g) f4_add(z, *reinterpret_cast<const float4 *>(A.zpart + ((size_t)(zs >> 8) + g) * W + c4));
        }
      }
      float a[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = (c4 + j < f0) ? fmaxf(a[j] + swl[D.off_b[tw][0] + c4 + j], 0.f) : 0.f;
      *reinterpret_cast<float4 *>(X0 + s_ * kDmfTS + c4) = make_float4(a[0], a[1], a[2], a[3]);
      if (b < A.B) *reinterpret_cast<float4 *>(Wk.act + ((size_t)tw * A.B + b) * kDmfMaxLayers * W + c4) = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < nl_max; ++l) {                           // (both towers walk the same number of barriers)
      if (l < nl) {
        const int fin = D.f[tw][l - 1], fo = D.f[tw][l];
        const float *const Kl = swl + D.off_k[tw][l], *const bl = swl + D.off_b[tw][l];
        const float *const Xi = X0 + (l - 1) * kDmfTile;
        float *const Xo = X0 + l * kDmfTile;
        float av[16];                                            // this lane's A operands of every k-step: one batch of LDS reads
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { const int kk = 4 * ks + q4; av[ks] = kk < fin ? Xi[m16 * kDmfTS + kk] : 0.f; }
        for (int nt = half; nt < 4; nt += 2) {                   // all four 16-column tiles of the 64-float row: beyond fo they are zero
          const int col = nt * 16 + m16;
          dmf_f4v acc = {0.f, 0.f, 0.f, 0.f};
          if (nt * 16 < fo) {
            float bv[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) { const int kk = 4 * ks + q4; bv[ks] = (kk < fin && col < fo) ? Kl[kk * fo + col] : 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
              if (4 * ks < fin) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s_ = 4 * q4 + r, b = b0 + s_;
            const float a = col < fo ? fmaxf(acc[r] + bl[col], 0.f) : 0.f;
            Xo[s_ * kDmfTS + col] = a;
            if (b < A.B) Wk.act[((size_t)tw * A.B + b) * kDmfMaxLayers * W + l * W + col] = a;
          }
        }
      }
      __syncthreads();
    }
    // ---- per sample: cosine of the two outputs, loss, gradient wrt the outputs (lane = sample m16, quarter q4 of the columns); every
    //      wave computes the scalars, the leading wave of a tower writes that tower's gradient tile
    {
      const float *const Ru = X + (0 * kDmfMaxLayers + D.n_layers[0] - 1) * kDmfTile + m16 * kDmfTS;
      const float *const Ri = X + (1 * kDmfMaxLayers + D.n_layers[1] - 1) * kDmfTile + m16 * kDmfTS;
      float qu = 0.f, qi = 0.f, ui = 0.f;
      for (int c = q4 * 16; c < q4 * 16 + 16; ++c) { const float u = Ru[c], v = Ri[c]; qu = fmaf(u, u, qu); qi = fmaf(v, v, qi); ui = fmaf(u, v, ui); }
      qu += __shfl_xor(qu, 16); qu += __shfl_xor(qu, 32);
      qi += __shfl_xor(qi, 16); qi += __shfl_xor(qi, 32);
      ui += __shfl_xor(ui, 16); ui += __shfl_xor(ui, 32);
      const float rhou = rsqrtf(fmaxf(qu, kL2NEps)), rhoi = rsqrtf(fmaxf(qi, kL2NEps));
      const float s = ui * rhou * rhoi;
      const float cosv = fmaxf(1e-6f, s);
      const float wsc = D.off_scale >= 0 ? swl[D.off_scale] : 1.0f;
      const float pred = wsc * cosv;
      const int b = b0 + m16;
      const float y = A.target_mode == 1 ? (A.y_mean_dev ? A.y_mean_dev[0] : A.y_mean) : (b < A.B ? A.y[b] : 0.f);
      const float gp = bce_grad(y, pred) * inv_b;
      if (wv == 0 && q4 == 0 && b < A.B) { Wk.samp[(size_t)b * 2] = bce_elem(y, pred); Wk.samp[(size_t)b * 2 + 1] = gp * cosv; }
      const float ds = (s > 1e-6f && b < A.B) ? gp * wsc : 0.f;
      // l2_normalize backward of both towers: dr = rho * (dn - n * (n . dn)), dn_u = ds * n_i: dr_u = rho_u * ds * (n_i - n_u * s)
      if (half == 0) {
        float *const Gs = Gt + m16 * kDmfTS;
        for (int c = q4 * 16; c < q4 * 16 + 16; ++c) {
          const float nu = Ru[c] * rhou, ni = Ri[c] * rhoi;
          Gs[c] = tw == 0 ? (qu > kL2NEps ? rhou * ds * (ni - nu * s) : rhou * ds * ni) : (qi > kL2NEps ? rhoi * ds * (nu - ni * s) : rhoi * ds * nu);
        }
      }
    }
    __syncthreads();
    // ---- backward: Gt holds dA of the tower's last layer; dZ_l overwrites it (rows leave for k_dmf_wgrad / k_dmf_dzsum), then dA_{l-1}
#pragma unroll 1
    for (int l = nl_max - 1; l >= 0; --l) {
      const bool on = l < nl;
      const int fo = on ? D.f[tw][l] : 0;
      if (on) {
        const float *const Al = X0 + l * kDmfTile;
#pragma unroll
        for (int it = 0; it < 2; ++it) {                         // dZ_l = dA_l where A_l > 0; a wave: 8 samples
          const int s_ = half * 8 + it * 4 + q4, c4 = m16 * 4, b = b0 + s_;
          const float4 a = *reinterpret_cast<const float4 *>(Al + s_ * kDmfTS + c4);
          float4 g = *reinterpret_cast<const float4 *>(Gt + s_ * kDmfTS + c4);
          g.x = (c4 + 0 < fo && a.x > 0.f) ? g.x : 0.f; g.y = (c4 + 1 < fo && a.y > 0.f) ? g.y : 0.f;
          g.z = (c4 + 2 < fo && a.z > 0.f) ? g.z : 0.f; g.w = (c4 + 3 < fo && a.w > 0.f) ? g.w : 0.f;
          *reinterpret_cast<float4 *>(Gt + s_ * kDmfTS + c4) = g;
          if (b < A.B) *reinterpret_cast<float4 *>(Wk.dz + ((size_t)tw * A.B + b) * kDmfMaxLayers * W + l * W + c4) = g;
        }
      }
      __syncthreads();
      if (l == 0) break;
      dmf_f4v acc[2] = {dmf_f4v{0.f, 0.f, 0.f, 0.f}, dmf_f4v{0.f, 0.f, 0.f, 0.f}};
      const int fin = on ? D.f[tw][l - 1] : 0;
      if (on) {
        const float *const Kl = swl + D.off_k[tw][l];
        float av[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { const int kk = 4 * ks + q4; av[ks] = kk < fo ? Gt[m16 * kDmfTS + kk] : 0.f; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                            // this wave's two 16-unit tiles of layer l - 1
          const int nt = half + 2 * j, row = nt * 16 + m16;
          if (nt * 16 < fin) {
            float bv[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) { const int kk = 4 * ks + q4; bv[ks] = (kk < fo && row < fin) ? Kl[row * fo + kk] : 0.f; }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
              if (4 * ks < fo) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc[j], 0, 0, 0);
          }
        }
      }
      __syncthreads();                                          // (every wave has read dZ_l: the tile may be overwritten)
      if (on) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = (half + 2 * j) * 16 + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r) Gt[(4 * q4 + r) * kDmfTS + col] = col < fin ? acc[j][r] : 0.f;
        }
      }
      __syncthreads();
    }
  }
}

// gsw_part[chunk][i] = sum over the chunk's samples (ascending b) of the gradient of small weight i:
//   kernel element (tw, l >= 1, j, k): act[tw][b][l-1][j] * dz[tw][b][l][k];   bias (tw, l, k): dz[tw][b][l][k];   scale: samp[b][1]
// loss_part[chunk] = sum of samp[b][0] / B.  One thread per small weight, blockIdx.y = chunk.
__global__ __launch_bounds__(256) void k_dmf_wgrad(DrxDmfDims D, DrxDmfArgs A, int chunk, int W) {
  const DmfWork Wk = dmf_work(A.work, A.B, W);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * chunk, b1 = min(A.B, b0 + chunk);
  if (i == D.n_small) {                              // the loss rides in the same launch
    float t = 0.f;
    for (int b = b0; b < b1; ++b) t += Wk.samp[(size_t)b * 2];
    A.loss_part[blockIdx.y] = t / (float)A.B;
    return;
  }
  if (i >= D.n_small) return;
  const float *pa = nullptr, *pd = nullptr;          // per-sample strides of kDmfMaxLayers * W floats
  bool is_scale = i == D.off_scale;
  for (int tw = 0; tw < 2 && !pd && !is_scale; ++tw)
    for (int l = 0; l < D.n_layers[tw]; ++l) {
      const int fo = D.f[tw][l];
      if (l >= 1) {
        const int r = i - D.off_k[tw][l];
        if (r >= 0 && r < D.f[tw][l - 1] * fo) {
          pa = Wk.act + ((size_t)tw * A.B * kDmfMaxLayers + (l - 1)) * W + r / fo;
          pd = Wk.dz + ((size_t)tw * A.B * kDmfMaxLayers + l) * W + r % fo;
          break;
        }
      }
      const int rb = i - D.off_b[tw][l];
      if (rb >= 0 && rb < fo) { pd = Wk.dz + ((size_t)tw * A.B * kDmfMaxLayers + l) * W + rb; break; }
    }
  float g = 0.f;
  const size_t stride = (size_t)kDmfMaxLayers * W;
  if (is_scale) {
    for (int b = b0; b < b1; ++b) g += Wk.samp[(size_t)b * 2 + 1];
  } else if (pd && pa) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) g = fmaf(pa[b * stride], pd[b * stride], g);
  } else if (pd) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) g += pd[b * stride];
  }
  A.gsw_part[(size_t)blockIdx.y * D.n_small + i] = g;
}

// out[j] = sum_r part[r][j] for j < n, out[n] = sum_r tail[r]: 64 columns per workgroup, its 16 waves take every 16th row,
// their partial sums are combined in wave order (fixed order of additions).
__global__ __launch_bounds__(1024) void k_sum_partials2(const float *__restrict__ part, int n_rows, int n, const float *__restrict__ tail,
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

// k_sum_partials2 with the small weights' Keras Adam behind it (as drx_caser.hip's k_sum_partials_adam): the thread that has column j's
// sum applies l2 + Adam of j's segment — one launch instead of two, the same operations in the same order.
__global__ __launch_bounds__(1024) void k_sum_partials2_adam(const float *__restrict__ part, int n_rows, int n, const float *__restrict__ tail,
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
      int sgi = -1;
      for (int k = 0; k < sg.n; ++k)
        if (j >= sg.start[k] && j < sg.start[k] + sg.len[k]) sgi = k;
      if (sgi >= 0) {
        const OptScalars o{DRX_OPT_ADAM, 0.f, 0.f, b1, b2, eps, sg.alpha[sgi]};
        float pp = p[j], mm = m[j], vv = v[j];
        opt_update1(o, fmaf(sg.l2_coef[sgi], pp, t), pp, mm, vv);
        p[j] = pp; m[j] = mm; v[j] = vv;
      }
    }
  }
}

// ---- bf16 MFMA all-pairs cosine scorer: out[u, n] = max(1e-6, ru[u] . ri[n]) for l2-normalised fp32 rows of width 32 -------
// v_mfma_f32_32x32x16_bf16 per 32x32 quarter and 16 k (K = 32: two of them).  A = users (row r = lane & 31, k = 8*(lane >> 5) + j),
// B = items (col r, same k); C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ short f2bf(float x) {
  union { float f; uint32_t u; } c; c.f = x;
  return (short)((c.u + 0x7FFFu + ((c.u >> 16) & 1u)) >> 16);      // round-to-nearest-even (inputs are finite, |x| <= 1)
}

// Workgroup = 4 waves = a 64 x 64 tile of scores (each wave a 32 x 32 quarter).  The two operand tiles are converted to bf16 ONCE,
// by all 256 threads with coalesced 32-byte reads, and staged in LDS (rows padded by 8 bf16: 16-byte fragments, no two of a wave's rows
// on the same bank group); every wave then takes its A / B fragments as one 16-byte LDS read per k-step.
__global__ __launch_bounds__(256) void k_score_pairs_bf16(const float *__restrict__ ru, int n_u, const float *__restrict__ ri, int n_i,
                                                          int ld, int kdim, const float *__restrict__ scale, float *__restrict__ out,
                                                          int out_ld) {
  constexpr int T = 64, KMAX = 64, LDSK = KMAX + 8;
  __shared__ __align__(16) short As[T * LDSK], Bs[T * LDSK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int u0 = blockIdx.y * T, i0 = blockIdx.x * T;
  const float wsc = scale ? *scale : 1.0f;
  // stage: thread t converts 16 consecutive k of row (t >> 2) [quarter t & 3 of a 64-wide row], for A and for B
  {
    const int row = tid >> 2, k0 = (tid & 3) * 16;
    if (k0 < kdim) {
      float4 fa[4], fb[4];                           // 16 consecutive floats of the row as four 16-byte loads (ld % 4 == 0)
      const float4 *pa = reinterpret_cast<const float4 *>(ru + (size_t)(u0 + row) * ld + k0);
      const float4 *pb = reinterpret_cast<const float4 *>(ri + (size_t)(i0 + row) * ld + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fa[j] = (u0 + row < n_u) ? pa[j] : f4_zero();
        fb[j] = (i0 + row < n_i) ? pb[j] : f4_zero();
      }
      bf16x8 a0, a1, b0, b1;
      a0[0] = f2bf(fa[0].x); a0[1] = f2bf(fa[0].y); a0[2] = f2bf(fa[0].z); a0[3] = f2bf(fa[0].w);
      a0[4] = f2bf(fa[1].x); a0[5] = f2bf(fa[1].y); a0[6] = f2bf(fa[1].z); a0[7] = f2bf(fa[1].w);
      a1[0] = f2bf(fa[2].x); a1[1] = f2bf(fa[2].y); a1[2] = f2bf(fa[2].z); a1[3] = f2bf(fa[2].w);
      a1[4] = f2bf(fa[3].x); a1[5] = f2bf(fa[3].y); a1[6] = f2bf(fa[3].z); a1[7] = f2bf(fa[3].w);
      b0[0] = f2bf(fb[0].x); b0[1] = f2bf(fb[0].y); b0[2] = f2bf(fb[0].z); b0[3] = f2bf(fb[0].w);
      b0[4] = f2bf(fb[1].x); b0[5] = f2bf(fb[1].y); b0[6] = f2bf(fb[1].z); b0[7] = f2bf(fb[1].w);
      b1[0] = f2bf(fb[2].x); b1[1] = f2bf(fb[2].y); b1[2] = f2bf(fb[2].z); b1[3] = f2bf(fb[2].w);
      b1[4] = f2bf(fb[3].x); b1[5] = f2bf(fb[3].y); b1[6] = f2bf(fb[3].z); b1[7] = f2bf(fb[3].w);
      *reinterpret_cast<bf16x8 *>(&As[row * LDSK + k0]) = a0;
      *reinterpret_cast<bf16x8 *>(&As[row * LDSK + k0 + 8]) = a1;
      *reinterpret_cast<bf16x8 *>(&Bs[row * LDSK + k0]) = b0;
      *reinterpret_cast<bf16x8 *>(&Bs[row * LDSK + k0 + 8]) = b1;
    }
  }
  __syncthreads();
  const int wu = (w >> 1) * 32, wi = (w & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int ks = 0; ks < kdim / 16; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&As[(wu + r) * LDSK + ks * 16 + 8 * h]);
    const bf16x8 bb = *reinterpret_cast<const bf16x8 *>(&Bs[(wi + r) * LDSK + ks * 16 + 8 * h]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (u0 + wu + row < n_u && i0 + wi + r < n_i) out[(size_t)(u0 + wu + row) * out_ld + i0 + wi + r] = wsc * fmaxf(1e-6f, acc[reg]);
  }
}

static int check_dims(const DrxDmfDims *D) {
  if (!D || D->n_small < 1) return DRX_EINVAL;
  for (int tw = 0; tw < 2; ++tw) {
    if (D->n_layers[tw] < 1 || D->n_layers[tw] > kDmfMaxLayers) return DRX_EINVAL;
    for (int l = 0; l < D->n_layers[tw]; ++l)
      if (D->f[tw][l] < 1 || D->f[tw][l] > 128) return DRX_EINVAL;
    if (D->ld0[tw] < D->f[tw][0] || (D->ld0[tw] & 3) || D->ld0[tw] > 128) return DRX_EINVAL;
  }
  if (D->off_scale < -1 || D->off_scale >= D->n_small) return DRX_EINVAL;
  return (size_t)D->n_small * 4 <= 150 * 1024 ? DRX_OK : DRX_EINVAL;
}

// ---- distinct ids of a batch on the device (DMF.fit(device_sampler=True)) --------------------------------------------------------
// keys: users [0, U), items U + iid; vals: the sample (items: B + sample).  After ONE stable sort the B user pairs come first, every
// id's samples ascending — grows IS the sorted sample column.
static __global__ void k_distinct_keys(const int32_t *__restrict__ uid, const int32_t *__restrict__ iid, int B, int n_users,
                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * B; j += gridDim.x * blockDim.x) {
    keys[j] = j < B ? (uint32_t)uid[j] : (uint32_t)n_users + (uint32_t)iid[j - B];
    vals[j] = (uint32_t)j;
  }
}

// ONE workgroup numbers the runs of the sorted list: threads [0, 512) share the user half, [512, 1024) the item half, a contiguous
// slice each — heads counted, block-scanned, then slots handed out in a second walk.  Also the batch mean of y, summed in a fixed order.
static __global__ __launch_bounds__(1024) void k_distinct_number(const uint32_t *__restrict__ ks, const uint32_t *__restrict__ vs, int B,
                                                                 int n_users, const float *__restrict__ y, int32_t *du, int32_t *di,
                                                                 int32_t *inv_u, int32_t *inv_i, int32_t *gptr_u, int32_t *gptr_i,
                                                                 int32_t *grows_u, int32_t *grows_i, int32_t *nd, float *y_mean) {
  __shared__ int wsum[16];
  __shared__ double red[16];
  const int t = threadIdx.x, half = t >> 9, th = t & 511, lane = t & 63, w = t >> 6;
  const int per = (B + 511) / 512;
  const int lo = half * B + min(B, th * per), hi = half * B + min(B, (th + 1) * per);
  int heads = 0;
  for (int j = lo; j < hi; ++j) heads += (j == half * B || ks[j] != ks[j - 1]) ? 1 : 0;
  int inc = heads;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int before = inc - heads, n_half[2] = {0, 0};
  for (int ww = 0; ww < 16; ++ww) {
    if ((ww >> 3) == half && ww < w) before += wsum[ww];
    n_half[ww >> 3] += wsum[ww];
  }
  int32_t *const dd = half ? di : du, *const inv = half ? inv_i : inv_u, *const gp = half ? gptr_i : gptr_u, *const gr = half ? grows_i : grows_u;
  int slot = before - 1;
  for (int j = lo; j < hi; ++j) {
    const uint32_t k = ks[j];
    const int jl = j - half * B, b = (int)vs[j] - half * B;
    if (j == half * B || k != ks[j - 1]) { ++slot; dd[slot] = (int32_t)(k - (half ? (uint32_t)n_users : 0u)); gp[slot] = jl; }
    inv[b] = slot;
    gr[jl] = b;
  }
  if (th == 0) { gp[n_half[half]] = B; nd[half] = n_half[half]; }
  double a = 0.0;                                  // (fp64: the host path hands the kernels float(mean in fp64))
  for (int b = t; b < B; b += 1024) a += (double)y[b];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
  if (lane == 0) red[w] = a;
  __syncthreads();
  if (t == 0) { double s = 0.0; for (int ww = 0; ww < 16; ++ww) s += red[ww]; y_mean[0] = (float)(s / (double)B); }
}

struct DistinctLayout { uint32_t *keys, *vals, *ks, *vs; void *sort_temp; size_t sort_bytes; int bits; };
static DistinctLayout distinct_layout(Carver &cv, int B, int n_users, int n_items) {
  DistinctLayout L{};
  L.bits = bits_for((uint64_t)n_users + (uint64_t)n_items + 1);
  L.keys = cv.take<uint32_t>((size_t)2 * B); L.vals = cv.take<uint32_t>((size_t)2 * B);
  L.ks = cv.take<uint32_t>((size_t)2 * B); L.vs = cv.take<uint32_t>((size_t)2 * B);
  L.sort_bytes = sort_pairs_temp_bytes((size_t)2 * B, L.bits);
  L.sort_temp = cv.take<char>(L.sort_bytes);
  return L;
}

}  // namespace drx

using namespace drx;

extern "C" {

size_t drx_dmf_distinct_scratch_bytes(int32_t B, int32_t n_users, int32_t n_items) {
  if (B < 1 || n_users < 1 || n_items < 1) return 0;
  Carver cv(nullptr, 0);
  (void)distinct_layout(cv, B, n_users, n_items);
  return align_up(cv.off, 256) + 256;
}

int drx_dmf_batch_distinct_device(const int32_t *uid, const int32_t *iid, const float *y, int32_t B, int32_t n_users, int32_t n_items,
                                  int32_t *du, int32_t *di, int32_t *inv_u, int32_t *inv_i, int32_t *gptr_u, int32_t *gptr_i,
                                  int32_t *grows_u, int32_t *grows_i, int32_t *nd, float *y_mean, void *scratch, size_t scratch_bytes,
                                  void *stream) {
  if (!uid || !iid || !y || !du || !di || !inv_u || !inv_i || !gptr_u || !gptr_i || !grows_u || !grows_i || !nd || !y_mean || !scratch ||
      B < 1 || B > (1 << 20) || n_users < 1 || n_items < 1 || (int64_t)n_users + n_items >= 0x7FFFFFFFll)
    return DRX_EINVAL;
  Carver cv(scratch, scratch_bytes);
  const DistinctLayout L = distinct_layout(cv, B, n_users, n_items);
  if (!cv.ok()) return DRX_ESCRATCH;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_distinct_keys, dim3((2 * B + 255) / 256 < 1024 ? (2 * B + 255) / 256 : 1024), dim3(256), 0, st, uid, iid, B, n_users,
                     L.keys, L.vals);
  const int rc = sort_pairs(L.sort_temp, L.sort_bytes, L.keys, L.ks, L.vals, L.vs, (size_t)2 * B, L.bits, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_distinct_number, dim3(1), dim3(1024), 0, st, L.ks, L.vs, B, n_users, y, du, di, inv_u, inv_i, gptr_u, gptr_i,
                     grows_u, grows_i, nd, y_mean);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}


// drx_dmf_work_order on the device: ONE workgroup; a distinct id's entries (its segments) are consecutive, the classes — bit length of
// the degree, descending — are filled through LDS counters (the order inside a class does not matter: a work item writes its own row)
static __global__ __launch_bounds__(1024) void k_dmf_work_order(const int64_t *__restrict__ u_indptr, const int64_t *__restrict__ i_indptr,
                                                                const int32_t *__restrict__ du, const int32_t *__restrict__ di,
                                                                const int32_t *__restrict__ nd, int seg_len, int32_t *__restrict__ order,
                                                                int order_cap, int32_t *__restrict__ zseg, int32_t *__restrict__ out2) {
  __shared__ int count[33], start[33], parts, entries;
  if (threadIdx.x < 33) count[threadIdx.x] = 0;
  if (threadIdx.x == 0) { parts = 0; entries = 0; }
  __syncthreads();
  const int n_u = nd[0], n = n_u + nd[1];
  auto degree = [&](int i) {
    const int id = i < n_u ? du[i] : di[i - n_u];
    const int64_t *ip = i < n_u ? u_indptr : i_indptr;
    return (int)(ip[id + 1] - ip[id]);
  };
  auto cls = [](int d) { return d <= 0 ? 0 : 32 - __clz(d); };
  auto segs = [seg_len](int d) { const int ns = seg_len > 0 && d > seg_len ? (d + seg_len - 1) / seg_len : 1; return ns > 255 ? 255 : ns; };
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int d = degree(i), ns = segs(d);
    atomicAdd(&count[cls(d)], ns);
    atomicAdd(&entries, ns);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int c = 32; c >= 0; --c) { start[c] = run; run += count[c]; }
  }
  __syncthreads();
  const bool fits = entries <= order_cap;               // (the caller's capacity rule guarantees it; if not: uncut, one entry per id)
  if (!fits) {
    __syncthreads();
    if (threadIdx.x < 33) count[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&count[cls(degree(i))], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int c = 32; c >= 0; --c) { start[c] = run; run += count[c]; }
      entries = n;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int d = degree(i), ns = fits ? segs(d) : 1;
    const int at = atomicAdd(&start[cls(d)], ns);
    for (int g = 0; g < ns; ++g) order[at + g] = i | (g << 24);
    int z = 0;
    if (ns > 1) z = (atomicAdd(&parts, ns - 1) << 8) | (ns - 1);
    zseg[i] = z;
  }
  __syncthreads();
  if (threadIdx.x == 0) { out2[0] = entries; out2[1] = parts; }
}

int drx_dmf_work_order_device(const int64_t *u_indptr, const int64_t *i_indptr, const int32_t *du, const int32_t *di, const int32_t *nd_dev,
                              int32_t seg_len, int32_t *order, int32_t order_cap, int32_t *zseg, int32_t *out2, void *stream) {
  if (!u_indptr || !i_indptr || !du || !di || !nd_dev || !order || !zseg || !out2 || seg_len < 0 || order_cap < 1) return DRX_EINVAL;
  hipLaunchKernelGGL(k_dmf_work_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, u_indptr, i_indptr, du, di, nd_dev, seg_len, order,
                     order_cap, zseg, out2);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

// number of batch chunks of k_dmf_wgrad = rows of gsw_part / entries of loss_part the caller provides
// (16 samples per chunk up to 256 chunks: a thread of k_dmf_wgrad walks its chunk serially, two strided loads per sample — 64-sample
// chunks made that walk 25 us of a 130 us step)
int drx_dmf_grid(int32_t B) { return B <= 16 ? 1 : (B + 15) / 16 < 256 ? (B + 15) / 16 : 256; }

size_t drx_dmf_work_bytes(int32_t B) {
  if (B < 1) return 0;
  return ((size_t)2 * B * 128 + (size_t)4 * B * kDmfMaxLayers * 128 + (size_t)2 * B) * 4 + 256;      // (rows of 128: the widest towers)
}

static int dmf_fwd_bwd_impl(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                            const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream);

int drx_dmf_fwd_bwd(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, void *stream) {
  return dmf_fwd_bwd_impl(D, A, gsw_out, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, stream);
}

// drx_dmf_fwd_bwd with the small weights' Keras Adam in the launch that sums the chunks' partial gradients (drx_adam_segments' work; the
// first-layer update — drx_dmf_k0_update — reads none of `sw` and may follow)
int drx_dmf_step_small(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                       const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream) {
  if (!sw || !sw_m || !sw_v || !sg || sg->n < 1 || sg->n > DRX_MAX_SEGMENTS || (A && sw != A->sw)) return DRX_EINVAL;
  return dmf_fwd_bwd_impl(D, A, gsw_out, sw, sw_m, sw_v, sg, beta1, beta2, eps, stream);
}

static int dmf_fwd_bwd_impl(const DrxDmfDims *D, const DrxDmfArgs *A, float *gsw_out, float *sw, float *sw_m, float *sw_v,
                            const DrxAdamSegments *sg, float beta1, float beta2, float eps, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->K0u || !A->K0i || !A->sw || !A->u_indptr || !A->u_indices || !A->u_values || !A->i_indptr || !A->i_indices ||
      !A->i_values || !A->uid || !A->iid || !A->y || !A->dz0u || !A->dz0i || !A->gsw_part || !A->loss_part || !gsw_out ||
      A->B < 1 || (A->target_mode != 0 && A->target_mode != 1))
    return DRX_EINVAL;
  // the first-layer gradient leaves either as touches for drx_scatter_rows or through the id maps for drx_dmf_k0_update
  const bool touches = A->tkeys_u || A->tkeys_i;
  if (touches && (!A->off_u || !A->off_i || !A->tkeys_u || !A->tsrc_u || !A->tcoef_u || !A->tkeys_i || !A->tsrc_i || !A->tcoef_i))
    return DRX_EINVAL;
  const bool maps = A->map_u || A->map_i;
  if (maps && (!A->map_u || !A->map_i || !A->rho_u || !A->rho_i || A->stamp == 0)) return DRX_EINVAL;
  if ((A->rho_u != nullptr) != (A->rho_i != nullptr)) return DRX_EINVAL;
  if (!touches && !maps) return DRX_EINVAL;
  if (!A->work) return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = drx_dmf_grid(A->B);
  const int chunk = (A->B + chunks - 1) / chunks;
  if (!A->inv_u || !A->inv_i || !A->gptr_u || !A->gptr_i || !A->grows_u || !A->grows_i || A->n_du < 1 || A->n_di < 1 || A->n_du > A->B ||
      A->n_di > A->B)
    return DRX_EINVAL;
  if (A->target_mode == 1 && A->nd_dev && !A->y_mean_dev) return DRX_EINVAL;      // a device-prepared batch knows its mean only there
  const int wv = dmf_waves(A->n_du + A->n_di);
  const int items = A->n_du + A->n_di;
  const bool listed = A->work_order && (!A->nd_dev || A->n_work_dev);
  if (listed && (A->n_work < items || A->seg_len < 0 || (A->seg_len > 0 && (!A->zseg || !A->zpart)))) return DRX_EINVAL;
  const int gitems = listed ? A->n_work : items;            // (long rows / columns cut into segments: more work items than ids)
  const int ggrid = gitems < 8192 ? gitems : 8192;
  const int nu = dmf_units(*D), W = 64 * nu;
#define DRX_DMF_GATHER(WVN, NUN) hipLaunchKernelGGL((k_dmf_gather<WVN, NUN>), dim3(ggrid), dim3(64 * WVN), 0, st, *D, *A)
  if (nu == 1) { if (wv == 16) DRX_DMF_GATHER(16, 1); else if (wv == 8) DRX_DMF_GATHER(8, 1); else DRX_DMF_GATHER(4, 1); }
  else { if (wv == 16) DRX_DMF_GATHER(16, 2); else if (wv == 8) DRX_DMF_GATHER(8, 2); else DRX_DMF_GATHER(4, 2); }
#undef DRX_DMF_GATHER
  const size_t lds = (size_t)D->n_small * 4;
  const int dgrid = (A->B + 3) / 4;
  // (DRX_DMF_DENSE_WAVE: the wave-per-sample kernel for every shape — the A / B build of scripts/build_variant.sh)
#ifndef DRX_DMF_DENSE_WAVE
#define DRX_DMF_DENSE_WAVE 0
#endif
  const size_t lds_t = ((size_t)((D->n_small + 3) & ~3) + (size_t)(2 * kDmfMaxLayers + 2) * kDmfTile + 64) * 4;
  if (nu == 1 && !DRX_DMF_DENSE_WAVE && lds_t <= 160 * 1024) {            // tiles of 16 samples on the matrix cores, four waves per tile
    if (lds_t > 48 * 1024)
      DRX_HIP(hipFuncSetAttribute((const void *)k_dmf_dense_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    const int tiles = (A->B + 15) / 16;
    hipLaunchKernelGGL(k_dmf_dense_tile, dim3(tiles < 2048 ? tiles : 2048), dim3(256), lds_t, st, *D, *A);
  } else if (nu == 1) {
    if (lds > 48 * 1024) DRX_HIP(hipFuncSetAttribute((const void *)k_dmf_dense<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_dmf_dense<1>, dim3(dgrid < 2048 ? dgrid : 2048), dim3(256), lds, st, *D, *A);
  } else {
    if (lds > 48 * 1024) DRX_HIP(hipFuncSetAttribute((const void *)k_dmf_dense<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_dmf_dense<2>, dim3(dgrid < 2048 ? dgrid : 2048), dim3(256), lds, st, *D, *A);
  }
  hipLaunchKernelGGL(k_dmf_dzsum, dim3((items + 3) / 4 < 2048 ? (items + 3) / 4 : 2048), dim3(256), 0, st, *D, *A, W);
  hipLaunchKernelGGL(k_dmf_wgrad, dim3((D->n_small + 1 + 255) / 256, chunks), dim3(256), 0, st, *D, *A, chunk, W);
  if (sg)
    hipLaunchKernelGGL(k_sum_partials2_adam, dim3((D->n_small + 64) / 64), dim3(1024), 0, st, A->gsw_part, chunks, D->n_small,
                       A->loss_part, gsw_out, sw, sw_m, sw_v, *sg, beta1, beta2, eps);
  else
    hipLaunchKernelGGL(k_sum_partials2, dim3((D->n_small + 64) / 64), dim3(1024), 0, st, A->gsw_part, chunks, D->n_small,
                       A->loss_part, gsw_out);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_dmf_norms(const DrxDmfDims *D, const int64_t *indptr, const float *values, int32_t n, float *out, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!indptr || !values || !out || n < 1) return DRX_EINVAL;
  const int grid = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(k_dmf_norms, dim3(grid), dim3(256), 0, (hipStream_t)stream, *D, indptr, values, n, out);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_dmf_k0_update(const DrxDmfDims *D, const DrxDmfArgs *A, const DrxDmfK0Update *up, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !up || !A->u_indptr || !A->u_indices || !A->u_values || !A->i_indptr || !A->i_indices || !A->i_values || !A->dz0u ||
      !A->dz0i || !A->map_u || !A->map_i || !A->rho_u || !A->rho_i || A->stamp == 0 || up->n_items < 1 || up->n_users < 1)
    return DRX_EINVAL;
  if (!up->K0u || !up->K0i || !up->m_u || !up->m_i || !up->v_u || !up->v_i) return DRX_EINVAL;
  if (D->ld0[0] > 64 || D->ld0[1] > 64) return DRX_EINVAL;      // (a quarter-wave per dz0 row; wider first layers: touches + drx_scatter_rows)
  if (((uintptr_t)up->K0u | (uintptr_t)up->K0i | (uintptr_t)up->m_u | (uintptr_t)up->m_i | (uintptr_t)up->v_u | (uintptr_t)up->v_i |
       (uintptr_t)A->dz0u | (uintptr_t)A->dz0i) & 15)
    return DRX_EINVAL;
  K0Tables U{{up->K0u, up->K0i}, {up->m_u, up->m_i}, {up->v_u, up->v_i}, {up->n_items, up->n_users}, {up->alpha_u, up->alpha_i},
             up->l2_coef, up->beta1, up->beta2, up->eps, up->row_order, 0};
  const int rows = up->n_items + up->n_users;
  int grid = rows;
  if (up->row_order) {                               // the order's first n_long rows: a workgroup each; the others a wave each
    if (up->n_long < 0 || up->n_long > rows) return DRX_EINVAL;
    U.n_long = up->n_long;
    grid = up->n_long + (rows - up->n_long + 3) / 4;
  }
  hipLaunchKernelGGL(k_dmf_k0_update, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *D, *A, U);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_dmf_predict(const DrxDmfDims *D, const DrxDmfArgs *A, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->K0u || !A->K0i || !A->sw || !A->u_indptr || !A->u_indices || !A->u_values || !A->i_indptr || !A->i_indices ||
      !A->i_values || !A->uid || !A->iid || A->B < 1 || (!A->pred_out && !A->rep_u_out && !A->rep_i_out))
    return DRX_EINVAL;
  const int wv = dmf_waves(A->B);
  const dim3 grid(A->B < 4096 ? A->B : 4096);
  const int nu = dmf_units(*D);
#define DRX_DMF_PREDICT(WVN, NUN) hipLaunchKernelGGL((k_dmf_predict<WVN, NUN>), grid, dim3(64 * WVN), (size_t)2 * WVN * 64 * NUN * 4, (hipStream_t)stream, *D, *A)
  if (nu == 1) { if (wv == 16) DRX_DMF_PREDICT(16, 1); else if (wv == 8) DRX_DMF_PREDICT(8, 1); else DRX_DMF_PREDICT(4, 1); }
  else { if (wv == 16) DRX_DMF_PREDICT(16, 2); else if (wv == 8) DRX_DMF_PREDICT(8, 2); else DRX_DMF_PREDICT(4, 2); }
#undef DRX_DMF_PREDICT
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

int drx_score_pairs_bf16(const float *ru, int32_t n_u, const float *ri, int32_t n_i, int32_t ld, int32_t kdim, const float *scale,
                         float *out, int32_t out_ld, void *stream) {
  if (!ru || !ri || !out || n_u < 1 || n_i < 1 || kdim < 16 || (kdim & 15) || kdim > 64 || ld < kdim || (ld & 3) || out_ld < n_i ||
      (((uintptr_t)ru | (uintptr_t)ri) & 15))
    return DRX_EINVAL;
  hipLaunchKernelGGL(k_score_pairs_bf16, dim3((n_i + 63) / 64, (n_u + 63) / 64), dim3(256), 0, (hipStream_t)stream, ru, n_u, ri, n_i,
                     ld, kdim, scale, out, out_ld);
  DRX_LAUNCH_CHECK();
  return DRX_OK;
}

}  // extern "C"
