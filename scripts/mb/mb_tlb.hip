// Micro-benchmark (not product code): does the FOOTPRINT of a table change what a random 512-byte-row read-modify-write costs?
// R random distinct rows of two tables of N rows each (parameter + slot), N = 1 M (0.5 GB each) .. 10 M (5.1 GB each: the V table of the
// headline workload).  Same kernel as mb_rows.hip's k_rmw.
//   hipcc --offload-arch=gfx950 -O3 -o drecpy_amd/csrc/build/mb_tlb scripts/mb/mb_tlb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

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
__global__ void k_read(const float4 *__restrict__ p, const uint32_t *__restrict__ idx, int R, float4 *out) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (g >= R) return;
  const float4 v = p[(size_t)idx[g] * 32 + lane];
  if (v.x == 12345.f) out[0] = v;
}

int main() {
  std::mt19937_64 rng(1);
  float4 *p, *a, *junk; uint32_t *d_idx;
  const size_t NMAX = 10000000;
  CK(hipMalloc(&p, NMAX * 512)); CK(hipMalloc(&a, NMAX * 512)); CK(hipMalloc(&junk, (size_t)1 << 30));
  CK(hipMemset(p, 0, NMAX * 512)); CK(hipMemset(a, 0, NMAX * 512));
  CK(hipMalloc(&d_idx, 400000 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t N : {(size_t)1000000, (size_t)2500000, (size_t)5000000, (size_t)10000000}) {
    for (int R : {65536, 200000}) {
      float tot = 0, totr = 0;
      const int reps = 6;
      for (int it = 0; it < reps; ++it) {
        std::vector<uint32_t> idx(R);
        for (auto &x : idx) x = (uint32_t)(rng() % N);      // (fresh rows every launch: nothing warm; duplicates are rare and harmless here)
        CK(hipMemcpy(d_idx, idx.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(junk, it, (size_t)1 << 30));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rmw<1>, dim3((R + 7) / 8), dim3(256), 0, 0, p, a, d_idx, R, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) tot += ms;
        for (auto &x : idx) x = (uint32_t)(rng() % N);
        CK(hipMemcpy(d_idx, idx.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(junk, it, (size_t)1 << 30));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_read, dim3((R + 7) / 8), dim3(256), 0, 0, p, d_idx, R, junk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (it) totr += ms;
      }
      const double us = tot * 1000.0 / (reps - 1), usr = totr * 1000.0 / (reps - 1);
      printf("N = %8zu rows (%.1f GB per table), R = %6d cold random rows: rmw %.1f us (%.2f TB/s of 2 KB per row); read only %.1f us (%.2f TB/s of 512 B per row)\n",
             N, N * 512.0 / 1e9, R, us, R * 2048.0 / us / 1e6, usr, R * 512.0 / usr / 1e6);
    }
  }
  return 0;
}
