// Micro-benchmark (not product code): would the reduction's gathers leave the fabric if every XCD gathered from a WINDOW of the gradient
// buffer that fits its own L2?  mb_rows.hip's kernels (gather-sum of T random 512-byte rows of a 33.5 MB buffer; read-modify-write of R
// random distinct rows of two 512 MB tables; both in one launch) with the gather indices of workgroup b drawn from window b % 8 of W
// rows (workgroups go to the XCDs round-robin) instead of the whole buffer; the combined launch timed behind a 1 GB memset (cold HBM
// rows, as in the step).  DESIGN.md section 9.2.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_l2band scripts/mb/mb_l2band.hip && /tmp/mb_l2band
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int LD = 128;

template <int U>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ rows, const uint32_t *__restrict__ idx, int T, int per, float4 *out) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int t0 = g * per, t1 = min(T, t0 + per);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int t = t0; t < t1; t += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int tt = min(t + u, t1 - 1); v[u] = rows[(size_t)idx[tt] * 32 + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.f) out[g * 32 + lane] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_rmw(float4 *__restrict__ p, float4 *__restrict__ a, const uint32_t *__restrict__ idx, int R, int per) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int t0 = g * per, t1 = min(R, t0 + per);
  for (int t = t0; t < t1; t += U) {
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (t + u < t1) { const size_t r = (size_t)idx[t + u] * 32 + lane; v[u] = p[r]; w[u] = a[r]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (t + u < t1) {
      const size_t r = (size_t)idx[t + u] * 32 + lane;
      w[u].x += v[u].x * v[u].x; w[u].y += v[u].y * v[u].y; w[u].z += v[u].z * v[u].z; w[u].w += v[u].w * v[u].w;
      v[u].x -= 1e-3f * w[u].x; v[u].y -= 1e-3f * w[u].y; v[u].z -= 1e-3f * w[u].z; v[u].w -= 1e-3f * w[u].w;
      p[r] = v[u]; a[r] = w[u];
    }
  }
}

// both: group g gathers `per` touches, then RMWs `rper` rows — the reduction's mix, without its dependencies between the two
template <int U>
__global__ __launch_bounds__(256) void k_both(const float4 *__restrict__ rows, const uint32_t *__restrict__ idx, int T, int per,
                                              float4 *__restrict__ p, float4 *__restrict__ a, const uint32_t *__restrict__ ridx, int R, int rper) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int t0 = g * per, t1 = min(T, t0 + per);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int t = t0; t < t1; t += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int tt = min(t + u, t1 - 1); v[u] = rows[(size_t)idx[tt] * 32 + lane]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  const int r0 = g * rper, r1 = min(R, r0 + rper);
  for (int t = r0; t < r1; t += U) {
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (t + u < r1) { const size_t r = (size_t)ridx[t + u] * 32 + lane; v[u] = p[r]; w[u] = a[r]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (t + u < r1) {
      const size_t r = (size_t)ridx[t + u] * 32 + lane;
      w[u].x += acc.x * acc.x; w[u].y += acc.y * acc.y; w[u].z += acc.z * acc.z; w[u].w += acc.w * acc.w;
      v[u].x -= 1e-3f * w[u].x; v[u].y -= 1e-3f * w[u].y; v[u].z -= 1e-3f * w[u].z; v[u].w -= 1e-3f * w[u].w;
      p[r] = v[u]; a[r] = w[u];
    }
  }
}

template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const int B = 65536, T = 1057680, R = 200000, N = 1000000, per = 64;
  const int groups = (T + per - 1) / per, grid = (groups + 7) / 8, rper = (R + groups - 1) / groups;
  std::mt19937 rng(1);
  std::vector<uint32_t> ridx(R);
  { std::vector<uint32_t> all(N); std::iota(all.begin(), all.end(), 0u); std::shuffle(all.begin(), all.end(), rng); std::copy(all.begin(), all.begin() + R, ridx.begin()); }
  float4 *rows, *p, *a, *out, *junk; uint32_t *d_idx, *d_ridx;
  CK(hipMalloc(&rows, (size_t)B * 512)); CK(hipMalloc(&p, (size_t)N * 512)); CK(hipMalloc(&a, (size_t)N * 512)); CK(hipMalloc(&out, (size_t)1 << 24));
  CK(hipMalloc(&junk, (size_t)1 << 30));
  CK(hipMalloc(&d_idx, T * 4)); CK(hipMalloc(&d_ridx, R * 4));
  CK(hipMemcpy(d_ridx, ridx.data(), R * 4, hipMemcpyHostToDevice));
  CK(hipMemset(rows, 0, (size_t)B * 512)); CK(hipMemset(p, 0, (size_t)N * 512)); CK(hipMemset(a, 0, (size_t)N * 512));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto cold = [&](auto launch) {                          // mean of 5 launches, each behind a 1 GB memset
    float tot = 0;
    for (int i = 0; i < 5; ++i) {
      CK(hipMemsetAsync(junk, i, (size_t)1 << 30));
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    return tot * 1000.f / 5;
  };
  const float rmw_cold = cold([&] { hipLaunchKernelGGL(k_rmw<2>, dim3((R / 2 + 7) / 8), dim3(256), 0, 0, p, a, d_ridx, R, 2); });
  printf("read-modify-write alone, cold: %d x 2 KB (410 MB) %.1f us (%.2f TB/s)\n", R, rmw_cold, R * 2048.0 / rmw_cold / 1e6);
  for (int W : {65536, 16384, 8192, 4096, 2048}) {        // window rows per XCD (65536: the whole buffer, no banding)
    std::vector<uint32_t> idx(T);
    for (int t = 0; t < T; ++t) {
      const int g = t / per, b = g / 8, x = b % 8;        // the gather group, its workgroup, the XCD the workgroup lands on
      idx[t] = W >= B ? rng() % B : (uint32_t)((x * (B / 8) + rng() % W) % B);
    }
    CK(hipMemcpy(d_idx, idx.data(), T * 4, hipMemcpyHostToDevice));
    const float g4 = timeit([&] { hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, rows, d_idx, T, per, out); });
    const float b4 = cold([&] { hipLaunchKernelGGL(k_both<4>, dim3(grid), dim3(256), 0, 0, rows, d_idx, T, per, p, a, d_ridx, R, rper); });
    printf("window %5d rows (%4.1f MB per XCD): gather alone %.1f us (%.2f TB/s); gather + cold read-modify-write in one launch %.1f us (sum of the parts %.1f)\n",
           W, W * 512.0 / 1e6, g4, T * 512.0 / g4 / 1e6, b4, g4 + rmw_cold);
  }
  return 0;
}
