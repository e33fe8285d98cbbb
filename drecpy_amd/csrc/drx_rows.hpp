// Row-group device helpers shared by the libdrx.so kernels (gfx950): coalesced float4 row access for a group of G
// lanes, the hidden-layer activation, Keras optimizer updates, deterministic block reduction.
#pragma once
#include "drx_common.hpp"

namespace drx {

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

// h = sigmoid(scale*bag + V[u] + b), zero in the padding columns (v, bb: the rows V[u] and b).
template <int G, int J>
__device__ __forceinline__ void hidden_act_rows(const DrxCdaeParams &P, float scale, int lane, const float4 (&acc)[J], const float4 (&v)[J],
                                                const float4 (&bb)[J], float4 (&h)[J]) {
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int col = 4 * (lane + j * G);
    h[j].x = colmask(col + 0, P.k, sigmoidf_(fmaf(scale, acc[j].x, v[j].x + bb[j].x)));
    h[j].y = colmask(col + 1, P.k, sigmoidf_(fmaf(scale, acc[j].y, v[j].y + bb[j].y)));
    h[j].z = colmask(col + 2, P.k, sigmoidf_(fmaf(scale, acc[j].z, v[j].z + bb[j].z)));
    h[j].w = colmask(col + 3, P.k, sigmoidf_(fmaf(scale, acc[j].w, v[j].w + bb[j].w)));
  }
}
template <int G, int J>
__device__ __forceinline__ void hidden_act(const DrxCdaeParams &P, int u, float scale, int lane,
                                           const float4 (&acc)[J], float4 (&h)[J]) {
  float4 v[J], bb[J];
  load_row<G, J>(P.V, (size_t)u, P.ld, lane, v);
  load_row<G, J>(P.b, 0, P.ld, lane, bb);
  hidden_act_rows<G, J>(P, scale, lane, acc, v, bb, h);
}

// ------------------------------------------------------------------------------------------------
// optimizer row updates (fp32, Keras formulas)
// ------------------------------------------------------------------------------------------------
struct OptScalars {
  int kind;
  float lr, rb;              // rb = reg_rate / B
  float b1, b2, eps, alpha;  // alpha = Keras-Adam lr_t of the variable being updated
  float inv_k = 0.f;         // 1 / K (row-wise Adagrad: mean of the squared gradient over the K columns)
};

// KIND >= 0: the optimizer is known at compile time (the sparse step's Adagrad instantiation: the segmented reduction then needs 64
// instead of 90 VGPRs — 8 waves per SIMD instead of 5); KIND < 0: o.kind decides at run time.
template <int KIND = -1>
__device__ __forceinline__ void opt_update1(const OptScalars &o, float g, float &p, float &s1, float &s2) {
  const int kind = KIND >= 0 ? KIND : o.kind;
  if (kind == DRX_OPT_ADAM) {
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
template <int G, int J, int KIND = -1>
__device__ __forceinline__ float row_update(const OptScalars &o, float *tab, float *s1, float *s2, size_t row, int ld,
                                            int lane, const float4 (&w)[J], const float4 (&g)[J]) {
  float sq = 0.f;
  float4 *pr = reinterpret_cast<float4 *>(tab + row * (size_t)ld);
  const int kind = KIND >= 0 ? KIND : o.kind;
  if (kind == DRX_OPT_ROWWISE_ADAGRAD) {
    // one accumulator per row (first float of the row's slot): the optimizer state costs 4 bytes of traffic per row, not 4K
    float4 gg[J];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float4 p = w[j];
      gg[j].x = fmaf(o.rb, p.x, g[j].x); gg[j].y = fmaf(o.rb, p.y, g[j].y);
      gg[j].z = fmaf(o.rb, p.z, g[j].z); gg[j].w = fmaf(o.rb, p.w, g[j].w);
      q += f4_dot(gg[j], gg[j]);
      sq += f4_dot(p, p);
    }
    q = group_sum<G>(q) * o.inv_k;
    const float acc = s1[row * (size_t)ld] + q;
    const float step = o.lr / (sqrtf(acc) + o.eps);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = lane + j * G;
      if (4 * c < ld) {
        float4 p = w[j];
        p.x -= step * gg[j].x; p.y -= step * gg[j].y; p.z -= step * gg[j].z; p.w -= step * gg[j].w;
        pr[c] = p;
      }
    }
    if (lane == 0) s1[row * (size_t)ld] = acc;
    return sq;
  }
  float4 *a1 = reinterpret_cast<float4 *>(s1 + row * (size_t)ld);
  float4 *a2 = (kind == DRX_OPT_ADAM) ? reinterpret_cast<float4 *>(s2 + row * (size_t)ld) : nullptr;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = lane + j * G;
    if (4 * c < ld) {
      float4 p = w[j];
      float4 m = a1[c];
      float4 v = a2 ? a2[c] : f4_zero();
      sq += f4_dot(p, p);
      opt_update1<KIND>(o, fmaf(o.rb, p.x, g[j].x), p.x, m.x, v.x);
      opt_update1<KIND>(o, fmaf(o.rb, p.y, g[j].y), p.y, m.y, v.y);
      opt_update1<KIND>(o, fmaf(o.rb, p.z, g[j].z), p.z, m.z, v.z);
      opt_update1<KIND>(o, fmaf(o.rb, p.w, g[j].w), p.w, m.w, v.w);
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
__device__ __forceinline__ float block_sum(float v, float *red /* [blockDim.x/64] in LDS */) {
  v = group_sum<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

}  // namespace drx
