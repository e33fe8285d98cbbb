// Caser (DRecPy/Recommender/caser.py) forward / backward on gfx950.
//
// One wavefront per sample, lane c = embedding channel (d <= 64): the L item rows and the user row are single
// coalesced row reads; the vertical conv (caser.py:53,103 — kernel [L,d,n_v], it sums over d), the L horizontal convs
// with relu + max over time (caser.py:55-58,106-108), dense_0 (caser.py:63,114) and the T' target dots
// (caser.py:115-120) are channel-parallel FMAs followed by wave reductions.  A workgroup keeps a copy of the small weights
// in LDS (channel-fastest, [s][f][c]: every lane reads its own column); their gradients are summed per workgroup in
// registers, unit by unit, in sample order (no atomics) and reduced over workgroups in a second, ordered pass; the
// gradients of the embedding lookups leave as one row per lookup for drx_rows_csr_adam.
#include "drx_caser_tile.hpp"


namespace drx {

struct CaserLds {
  float *E;       // [L][64] item rows of the current sample
  float *x;       // [nx] concat(out_v, out_h) before dropout
  float *xd;      // [nx] after dropout
  float *pre;     // [nx] pre-activation at the arg-max step (horizontal convs)
  float *dx;      // [nx]
  int *arg;       // [nx]
  float *dz0s;    // [64] dense_0's pre-activation gradient of the sample (read by the accumulating waves)
};

// The NT tap rows of one horizontal (height, filter) pair: sum over the round's samples t of E_t[arg_t + s][c] * dc_t, samples in
// order; lane t holds dc_t / arg_t (dcl / tal).  g = this lane's slot of tap 0 in the workgroup's partial sums, taps `tap` floats apart.
template <int NT>
__device__ __forceinline__ void pair_rows(float *g, int tap, bool first, bool wr, const float *sc /* scratch0 + c */, int per_wave, int nT,
                                          float dcl, int tal) {
  float acc[NT];
#pragma unroll
  for (int s2 = 0; s2 < NT; ++s2) acc[s2] = (first || !wr) ? 0.f : g[s2 * tap];
#pragma unroll 4
  for (int t = 0; t < nT; ++t) {
    const float dc = lane_f(dcl, t);
    const float *const Et = sc + (size_t)t * per_wave + __builtin_amdgcn_readlane(tal, t) * 64;
#pragma unroll
    for (int s2 = 0; s2 < NT; ++s2) acc[s2] = fmaf(Et[s2 * 64], dc, acc[s2]);
  }
  if (wr) {
#pragma unroll
    for (int s2 = 0; s2 < NT; ++s2) g[s2 * tap] = acc[s2];
  }
}

// Horizontal convs backward, the filters of height I + 1 (and, recursively, the taller ones): lane l works out filter l's gradient at
// its arg-max step and holds its position (64 filters at a time); the walk over the filters reads both with v_readlane.  The taps of a
// filter touch the item rows t0 .. t0 + I: t0 is uniform, so each possible t0 is a branch of its own with COMPILE-TIME row indices —
// I + 1 weight reads and FMAs per filter.  (r03 asked every window position for a weight — a clamped one times zero where the filter
// has no tap — so that no index depended on data: 5 reads + 5 FMAs + 15 selects per filter at L = 5, 20 us of a sample's 96.)
// Same FMAs on the same values in the same order as that form: bit for bit.
template <int I>
__device__ __forceinline__ void hconv_backward(const DrxCaserDims &D, const CaserLds &S, const float *wl, int c, int nx, int ld,
                                               float (&dEr)[kCaserMaxL], float &dcl, int &tl, int &pq) {
  if constexpr (I < kCaserMaxL) {
    if (I >= D.L) return;
    const int tap = D.n_h * ld;
    const float *const khi = wl + D.off_kh[I] + c;
    for (int f = 0; f < D.n_h; ++f, ++pq) {
      if ((pq & 63) == 0) {
        const int jl = D.n_v + pq + c;
        dcl = jl < nx ? S.dx[jl] * act_df(D.act_h, S.pre[jl]) : 0.f;
        tl = jl < nx ? S.arg[jl] : 0;
      }
      const float dc = lane_f(dcl, pq & 63);
      if (dc == 0.f) continue;
      const int t0 = __builtin_amdgcn_readlane(tl, pq & 63);
      const float *const kh = khi + f * ld;
#pragma unroll
      for (int T = 0; T + I < kCaserMaxL; ++T) {
        if (t0 == T) {
          float w[I + 1];
#pragma unroll
          for (int s2 = 0; s2 <= I; ++s2) w[s2] = kh[s2 * tap];
#pragma unroll
          for (int s2 = 0; s2 <= I; ++s2) dEr[T + s2] = fmaf(dc, w[s2], dEr[T + s2]);
        }
      }
    }
    hconv_backward<I + 1>(D, S, wl, c, nx, ld, dEr, dcl, tl, pq);
  }
}

// A workgroup is W waves, one sample per wave and round.  r03 (phase stamps, scripts/stamps_caser.py: a sample lived 163 us, 44 of them
// in the horizontal convolutions' 560 weight loads from global memory, 46 in turn-taking over shared LDS gradient accumulators, 24 in
// the chain of LDS read-modify-writes of the horizontal backward): the workgroup now keeps a COPY OF THE SMALL WEIGHTS in LDS where it
// kept their gradient accumulators (same size), every weight read of the forward and backward passes is an LDS read of the lane's own
// column, the item rows' gradients live in registers, and the small-weight gradients are summed by UNIT — a dense_0 row, a vertical
// filter row, a horizontal (height, filter) pair with its taps, 64 bias entries — each unit by one wave (unit % W), in registers, over
// the W samples of the round in sample order, out of the samples' scratch: no atomics, no shared accumulator, every sum in the order
// of the batch (bit for bit the sums of the turn-taking version).  Partial sums of a workgroup go straight to gsw_part[block].
template <bool TRAIN>
__global__ __launch_bounds__(1024) void k_caser(DrxCaserDims D, DrxCaserArgs A CASER_STAMP_ARG) {
  extern __shared__ __align__(16) float lds[];
  const int c = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), W = blockDim.x >> 6;
  const int L = D.L, d = D.d, ld = D.ld, nx = D.n_v + D.L * D.n_h;
  const int per_wave = L * 64 + 5 * nx + 64;
  float *const wl = lds;                                              // [n_small] small weights, then 64 zeros (a row read runs 64 wide)
  float *const scratch0 = lds + D.n_small + 64;
  CaserLds S;
  float *q = scratch0 + (size_t)w * per_wave;
  S.E = q; q += L * 64;
  S.x = q; q += nx;
  S.xd = q; q += nx;
  S.pre = q; q += nx;
  S.dx = q; q += nx;
  S.arg = reinterpret_cast<int *>(q); q += nx;
  S.dz0s = q;
  __shared__ float wloss[16];
  for (int i = threadIdx.x * 4; i < D.n_small; i += blockDim.x * 4)  // (every segment of sw is a multiple of 4 floats long)
    *reinterpret_cast<float4 *>(wl + i) = *reinterpret_cast<const float4 *>(A.sw + i);
  if (threadIdx.x < 64) wl[D.n_small + threadIdx.x] = 0.f;
  __syncthreads();
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
    CSTAMP(0);
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
    CSTAMP(1);
    // ---- 2. vertical conv (dead lanes hold E = 0: whatever they read beside a weight row adds nothing) -------------------------
    for (int f = 0; f < D.n_v; ++f) {
      float part = 0.f;
      for (int t = 0; t < L; ++t) part = fmaf(S.E[t * 64 + c], wl[D.off_kv + (t * D.n_v + f) * ld + c], part);
      const float v = wave_sum(part) + wl[D.off_bv + f];
      if (c == 0) { S.x[f] = v; S.pre[f] = v; S.arg[f] = 0; }
    }
    CSTAMP(2);
    // ---- 3. horizontal convs + act_h + max over time -------------------------------------------------------------------
    // 16 filters at a time: the 16 channel sums leave through one reduce16, and the lanes that end up holding filter f track its
    // maximum over the window positions.
    for (int i = 0; i < L; ++i) {
      const float *const kh = wl + D.off_kh[i] + c;
      for (int f0 = 0; f0 < D.n_h; f0 += 16) {
        const int fm = slot16(c);                               // the filter (of this block) whose total this lane receives
        float best = -3.0e38f, bpre = 0.f;
        int bt = 0;
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
          const float tot = reduce16(part, c);
          const float v = tot + bias;
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
    CSTAMP(3);
    // ---- 4. dropout (mask injected by the host; caser.py:61,114) -------------------------------------------------------
    for (int j = c; j < nx; j += 64) {
      float v = S.x[j];
      if (TRAIN && A.keep) v = A.keep[(size_t)b * nx + j] ? v * inv_keep : 0.f;
      else if (hashed) v = hash_u32(A.mask_seed, (uint32_t)b, (uint32_t)j) >= rthr ? v * inv_keep : 0.f;
      S.xd[j] = v;
    }
    wave_lds_sync();
    CSTAMP(4);
    // ---- 5. dense_0 (act_mlp) ---------------------------------------------------------------------------------------------
    float z0 = wl[D.off_bd + c];
    {
      const float *const wd = wl + D.off_wd + c;
#pragma unroll 4
      for (int j = 0; j < nx; ++j) z0 = fmaf(S.xd[j], wd[j * ld], z0);
    }
    if (!live) z0 = 0.f;
    const float z = act_f(D.act_mlp, z0);
    if (!TRAIN) {
      if (live) { A.cat_out[(size_t)b * D.ld2 + c] = z; A.cat_out[(size_t)b * D.ld2 + d + c] = pu; }
    } else {
    CSTAMP(5);
    // ---- 6. targets: score, sigmoid, Keras BCE, backward to the lookups -------------------------------------------------------
    // eight targets at a time: their sixteen row reads are in flight together, the eight dot products leave through ONE reduce8 (10
    // shuffles instead of 48), the lanes that hold target j's score do its sigmoid / loss / gradient, every lane then fetches the eight
    // gradients with v_readlane
    float dz = 0.f, dpu = 0.f;
    for (int j0 = 0; j0 < D.Tp; j0 += 8) {
      int n[8];
      float wa[8], wb[8];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) n[qq] = j0 + qq < D.Tp ? A.after[b * D.Tp + j0 + qq] : 0;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const bool on = live && j0 + qq < D.Tp;
        wa[qq] = on ? A.W1[(size_t)n[qq] * D.ld2 + c] : 0.f;
        wb[qq] = on ? A.W1[(size_t)n[qq] * D.ld2 + d + c] : 0.f;
      }
      const int jm = j0 + slot8(c);                           // the target whose score this lane receives
      const float bm = jm < D.Tp ? A.b1[A.after[b * D.Tp + jm]] : 0.f;
      float prod[8];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) prod[qq] = fmaf(z, wa[qq], pu * wb[qq]);
      const float sc = reduce8(prod, c) + bm;
      const float p = sigmoidf_(sc);
      const float y = jm < D.T ? 1.f : 0.f;
      const float dsm = jm < D.Tp ? bce_grad(y, p) * inv_bt * p * (1.f - p) : 0.f;
      if ((c & 7) == 0 && jm < D.Tp) {
        loss_acc += bce_elem(y, p);
        A.db1[(size_t)b * D.Tp + jm] = dsm;
      }
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        if (j0 + qq < D.Tp) {
          const float ds = lane_f(dsm, lane8(qq));
          const size_t row = (size_t)b * D.Tp + j0 + qq;
          if (live) { A.dW1[row * D.ld2 + c] = ds * z; A.dW1[row * D.ld2 + d + c] = ds * pu; }
          dz = fmaf(ds, wa[qq], dz);
          dpu = fmaf(ds, wb[qq], dpu);
        }
      }
    }
    if (live) A.dPu[(size_t)b * ld + c] = dpu;
    dz0 = dz * act_df(D.act_mlp, z0);
    CSTAMP(6);
    // ---- 7. dense_0 backward: dx[j] = sum_c dz0[c] * Wd[j][c], 16 rows of Wd per reduce16 (dz0 = 0 in the dead lanes) -----------
    for (int j0 = 0; j0 < nx; j0 += 16) {
      float prod[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        prod[jj] = dz0 * (j < nx ? wl[D.off_wd + j * ld + c] : 0.f);
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
    CSTAMP(7);
    // ---- 8. vertical conv backward (to the item rows; their gradients stay in registers) -----------------------------------------
    float dEr[kCaserMaxL];
#pragma unroll
    for (int t = 0; t < kCaserMaxL; ++t) dEr[t] = 0.f;
    for (int f = 0; f < D.n_v; ++f) {
      const float dv = S.dx[f];
#pragma unroll
      for (int t = 0; t < kCaserMaxL; ++t)
        if (t < L) dEr[t] = fmaf(dv, wl[D.off_kv + (t * D.n_v + f) * ld + c], dEr[t]);
    }
    CSTAMP(8);
    // ---- 9. horizontal convs backward (through act_h at the arg-max step): hconv_backward above ------------------------------
    {
      float dcl = 0.f;
      int tl = 0, pq = 0;
      hconv_backward<0>(D, S, wl, c, nx, ld, dEr, dcl, tl, pq);
    }
    CSTAMP(9);
    // ---- 10. gradient rows of the item lookups ------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < kCaserMaxL; ++t)
      if (t < L && live) A.dE[((size_t)b * L + t) * ld + c] = dEr[t];
    CSTAMP(10);
    }   // TRAIN
    }   // has
    // ---- 11. small-weight gradients, unit by unit ---------------------------------------------------------------------------------
    if (TRAIN) {
      if (has) S.dz0s[c] = dz0;
      __syncthreads();
      const int nT = min(W, A.B - b0);
      const bool first = b0 == (int)blockIdx.x * W;
      float *const gp = A.gsw_part + (size_t)blockIdx.x * D.n_small;
      const int u_kv = nx, u_kh = u_kv + L * D.n_v, u_bd = u_kh + L * D.n_h, u_bv = u_bd + 1, nbv = (D.n_v + 63) / 64, nbh = (D.n_h + 63) / 64;
      const int u_bh = u_bv + nbv, n_units = u_bh + L * nbh;
      // a sample's scratch: E | x | xd | pre | dx | arg | dz0
      const int o_xd = L * 64 + nx, o_pre = o_xd + nx, o_dx = o_pre + nx, o_arg = o_dx + nx, o_dz0 = o_arg + nx;
      for (int un = w; un < n_units; un += W) {
        // lane t of the wave first fetches sample t's scalar of this unit; the walk over the samples then reads it with v_readlane, so
        // that the vector reads of the walk do not wait for one another
        const float *const Tl = scratch0 + (size_t)(c < nT ? c : 0) * per_wave;
        if (un < u_kv) {                                       // dense_0 row j: sum_t xd_t[j] * dz0_t[c]
          const int off = D.off_wd + un * ld + c;
          float acc = first ? 0.f : (c < ld ? gp[off] : 0.f);
          const float xl = Tl[o_xd + un];
#pragma unroll 4
          for (int t = 0; t < nT; ++t) acc = fmaf(lane_f(xl, t), scratch0[(size_t)t * per_wave + o_dz0 + c], acc);
          if (c < ld) gp[off] = acc;
        } else if (un < u_kh) {                                // vertical filter row (tt, f): sum_t E_t[tt][c] * dx_t[f]
          const int qv = un - u_kv, tt = qv / D.n_v, f = qv - tt * D.n_v;
          const int off = D.off_kv + qv * ld + c;
          float acc = first ? 0.f : (c < ld ? gp[off] : 0.f);
          const float dl = Tl[o_dx + f];
#pragma unroll 4
          for (int t = 0; t < nT; ++t) acc = fmaf(scratch0[(size_t)t * per_wave + tt * 64 + c], lane_f(dl, t), acc);
          if (c < ld) gp[off] = acc;
        } else if (un < u_bd) {                                // horizontal pair (i, f): its i + 1 tap rows
          const int pq = un - u_kh, i = pq / D.n_h, f = pq - i * D.n_h, j = D.n_v + pq;
          const int off = D.off_kh[i] + f * ld + c;
          const float dcl = Tl[o_dx + j] * act_df(D.act_h, Tl[o_pre + j]);
          const int tal = reinterpret_cast<const int *>(Tl + o_arg)[j];
          const int tap = D.n_h * ld;
          const bool wr = c < ld;
          switch (i) {                                         // (static tap counts: the reads of a sample's window are in flight together)
            case 0: pair_rows<1>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 1: pair_rows<2>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 2: pair_rows<3>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 3: pair_rows<4>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 4: pair_rows<5>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 5: pair_rows<6>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            case 6: pair_rows<7>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
            default: pair_rows<8>(gp + off, tap, first, wr, scratch0 + c, per_wave, nT, dcl, tal); break;
          }
        } else if (un < u_bv) {                                // dense_0 bias
          float acc = first ? 0.f : (c < ld ? gp[D.off_bd + c] : 0.f);
          for (int t = 0; t < nT; ++t) acc += scratch0[(size_t)t * per_wave + o_dz0 + c];
          if (c < ld) gp[D.off_bd + c] = acc;
        } else if (un < u_bh) {                                // vertical biases, 64 per unit
          const int f = (un - u_bv) * 64 + c, pad = (D.n_v + 3) & ~3;
          float acc = (first || f >= D.n_v) ? 0.f : gp[D.off_bv + f];
          if (f < D.n_v)
            for (int t = 0; t < nT; ++t) acc += scratch0[(size_t)t * per_wave + o_dx + f];
          if (f < pad) gp[D.off_bv + f] = f < D.n_v ? acc : 0.f;
        } else {                                               // horizontal biases of height i, 64 per unit
          const int ub = un - u_bh, i = ub / nbh, f = (ub - i * nbh) * 64 + c, pad = (D.n_h + 3) & ~3;
          const int j = D.n_v + i * D.n_h + f;
          float acc = (first || f >= D.n_h) ? 0.f : gp[D.off_bh[i] + f];
          if (f < D.n_h)
            for (int t = 0; t < nT; ++t) {
              const float *T0 = scratch0 + (size_t)t * per_wave;
              acc += T0[o_dx + j] * act_df(D.act_h, T0[o_pre + j]);
            }
          if (f < pad) gp[D.off_bh[i] + f] = f < D.n_h ? acc : 0.f;
        }
      }
      __syncthreads();
      if (has) CSTAMP(11);
    }
  }
  if (TRAIN) {
    loss_acc = wave_sum(loss_acc);                       // (the lanes that held a target's score carry its loss term)
    if (c == 0) wloss[w] = loss_acc;
    __syncthreads();
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

// the small weights (+ 64 zeros) and a scratch per wave: E | x | xd | pre | dx | arg | dz0
constexpr size_t kCaserLdsLimit = 160 * 1024 - 256;    // dynamic LDS a workgroup may ask for (beside the kernel's static words)

static size_t caser_lds_bytes(const DrxCaserDims &D, bool, int waves) {
  const int nx = D.n_v + D.L * D.n_h;
  return ((size_t)D.n_small + 64 + (size_t)waves * ((size_t)D.L * 64 + 5 * (size_t)nx + 64)) * 4 + 64;
}

// waves per workgroup: as many (16, 8, 4, 2, 1) as fit the CU's LDS beside the shared accumulators, and no more than the batch needs
static int caser_waves(const DrxCaserDims &D, bool train, int B) {
  int w = 16;
  while (w > 1 && (caser_lds_bytes(D, train, w) > 150 * 1024 || (B + w - 1) / w < 128)) w >>= 1;
  return w;
}

static int check_dims(const DrxCaserDims *D) {
  if (!D || D->L < 1 || D->L > kCaserMaxL || D->d < 1 || D->d > 64 || D->ld < D->d || (D->ld & 3) || D->ld2 < 2 * D->d ||
      (D->ld2 & 3) || D->n_v < 1 || D->n_h < 1 || D->T < 1 || D->Tp < D->T || D->n_small < 1 || (D->n_small & 3) || D->act_h < 0 || D->act_h > 3 ||
      D->act_mlp < 0 || D->act_mlp > 3)
    return DRX_EINVAL;
  return caser_lds_bytes(*D, true, 1) <= 150 * 1024 ? DRX_OK : DRX_EINVAL;
}

}  // namespace drx

using namespace drx;

extern "C" {

// the training kernel's variant for these dimensions: 1 = convolution weights in LDS, 0 = read from global memory, -1 = the tile's own
// scratch does not fit a workgroup's LDS (nx too large)
static int caser_tile_variant(const DrxCaserDims &D) {
  if ((size_t)caser_tile_geom(D, true).floats * 4 <= kCaserLdsLimit) return 1;
  if ((size_t)caser_tile_geom(D, false).floats * 4 <= kCaserLdsLimit) return 0;
  return -1;
}

int drx_caser_grid(const DrxCaserDims *D, int32_t B) {
  if (!D || B < 1 || check_dims(D) != DRX_OK) return 0;
#ifdef DRX_CASER_WAVE
  const int w = caser_waves(*D, true, B);
  const int g = (B + w - 1) / w;
  return g < 512 ? g : 512;
#else
  if (caser_tile_variant(*D) < 0) return 0;
  const int g = (B + kTileSamples - 1) / kTileSamples;       // one workgroup per CU (its LDS), tiles of 16 samples in turn
  return g < 256 ? g : 256;
#endif
}

int drx_caser_fwd_bwd(const DrxCaserDims *D, const DrxCaserArgs *A, float *gsw_out, void *stream) {
  int rc = check_dims(D);
  if (rc) return rc;
  if (!A || !A->item_emb || !A->user_emb || !A->W1 || !A->b1 || !A->sw || !A->uid || !A->before || !A->after || !A->dE ||
      !A->dW1 || !A->db1 || !A->dPu || !A->gsw_part || !A->loss_part || !gsw_out || A->B < 1 || A->rate < 0.f || A->rate >= 1.f)
    return DRX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = drx_caser_grid(D, A->B);
  if (grid < 1) return DRX_EINVAL;
#ifdef DRX_CASER_WAVE
  const int waves = caser_waves(*D, true, A->B);
  const size_t lds = caser_lds_bytes(*D, true, waves);
  DRX_HIP(hipFuncSetAttribute((const void *)k_caser<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_caser<true>, dim3(grid), dim3(64 * waves), lds, st, *D, *A CASER_STAMP_PASS);
#else
  const int var = caser_tile_variant(*D);
  const size_t lds = (size_t)caser_tile_geom(*D, var == 1).floats * 4;
  if (var == 1) {
    DRX_HIP(hipFuncSetAttribute((const void *)k_caser_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_caser_tile<true>, dim3(grid), dim3(64 * kTileWaves), lds, st, *D, *A CASER_STAMP_PASS);
  } else {
    DRX_HIP(hipFuncSetAttribute((const void *)k_caser_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_caser_tile<false>, dim3(grid), dim3(64 * kTileWaves), lds, st, *D, *A CASER_STAMP_PASS);
  }
#endif
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
  const int waves = caser_waves(*D, false, A->B);       // (every workgroup copies the small weights into its LDS: few, large workgroups)
  const size_t lds = caser_lds_bytes(*D, false, waves);
  const int g = (A->B + waves - 1) / waves;
  DRX_HIP(hipFuncSetAttribute((const void *)k_caser<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_caser<false>, dim3(g < 512 ? g : 512), dim3(64 * waves), lds, st, *D, *A CASER_STAMP_PASS);
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
