// Micro-benchmark (not product code): would keeping a parameter row and its optimizer slot row ADJACENT (one 1 KiB block per table row)
// make the cold read-modify-write of random rows cheaper than two 512-byte rows in two tables?  R random rows of N = 1 M, fresh rows
// every launch, 1 GB written in between.
//   hipcc --offload-arch=gfx950 -O3 -o drecpy_amd/csrc/build/mb_interleave scripts/mb/mb_interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// two tables: p[N][32 float4], a[N][32 float4]; a group of 32 lanes per row
__global__ __launch_bounds__(256) void k_two(float4 *__restrict__ p, float4 *__restrict__ a, const uint32_t *__restrict__ idx, int R) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (g >= R) return;
  const size_t r = (size_t)idx[g] * 32 + lane;
  float4 v = p[r], w = a[r];
  w.x += v.x * v.x; w.y += v.y * v.y; w.z += v.z * v.z; w.w += v.w * v.w;
  v.x -= 1e-3f * w.x; v.y -= 1e-3f * w.y; v.z -= 1e-3f * w.z; v.w -= 1e-3f * w.w;
  p[r] = v; a[r] = w;
}
// one table: t[N][64 float4] = [parameter row | slot row]; the same group of 32 lanes per row
__global__ __launch_bounds__(256) void k_one(float4 *__restrict__ t, const uint32_t *__restrict__ idx, int R) {
  const int lane = threadIdx.x & 31, g = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (g >= R) return;
  const size_t r = (size_t)idx[g] * 64 + lane;
  float4 v = t[r], w = t[r + 32];
  w.x += v.x * v.x; w.y += v.y * v.y; w.z += v.z * v.z; w.w += v.w * v.w;
  v.x -= 1e-3f * w.x; v.y -= 1e-3f * w.y; v.z -= 1e-3f * w.z; v.w -= 1e-3f * w.w;
  t[r] = v; t[r + 32] = w;
}
// ... and with a whole wave per row (64 lanes x 16 B = the 1 KiB block in one instruction each way)
__global__ __launch_bounds__(256) void k_one_wave(float4 *__restrict__ t, const uint32_t *__restrict__ idx, int R) {
  const int lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= R) return;
  const size_t r = (size_t)idx[g] * 64 + lane;
  float4 x = t[r];
  // lanes 0..31 hold the parameter, 32..63 the slot: exchange halves
  float4 o;
  o.x = __shfl_xor(x.x, 32); o.y = __shfl_xor(x.y, 32); o.z = __shfl_xor(x.z, 32); o.w = __shfl_xor(x.w, 32);
  float4 v = lane < 32 ? x : o, w = lane < 32 ? o : x;
  w.x += v.x * v.x; w.y += v.y * v.y; w.z += v.z * v.z; w.w += v.w * v.w;
  v.x -= 1e-3f * w.x; v.y -= 1e-3f * w.y; v.z -= 1e-3f * w.z; v.w -= 1e-3f * w.w;
  t[r] = lane < 32 ? v : w;
}

int main() {
  const size_t N = 1000000;
  std::mt19937_64 rng(1);
  float4 *p, *a, *t, *junk; uint32_t *d_idx;
  CK(hipMalloc(&p, N * 512)); CK(hipMalloc(&a, N * 512)); CK(hipMalloc(&t, N * 1024)); CK(hipMalloc(&junk, (size_t)1 << 30));
  CK(hipMemset(p, 0, N * 512)); CK(hipMemset(a, 0, N * 512)); CK(hipMemset(t, 0, N * 1024));
  CK(hipMalloc(&d_idx, 400000 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int R : {65536, 200000}) {
    float tot[3] = {0, 0, 0};
    const int reps = 7;
    for (int it = 0; it < reps; ++it)
      for (int which = 0; which < 3; ++which) {
        std::vector<uint32_t> idx(R);
        for (auto &x : idx) x = (uint32_t)(rng() % N);
        CK(hipMemcpy(d_idx, idx.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(junk, it, (size_t)1 << 30));
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k_two, dim3((R + 7) / 8), dim3(256), 0, 0, p, a, d_idx, R);
        else if (which == 1) hipLaunchKernelGGL(k_one, dim3((R + 7) / 8), dim3(256), 0, 0, t, d_idx, R);
        else hipLaunchKernelGGL(k_one_wave, dim3((R + 3) / 4), dim3(256), 0, 0, t, d_idx, R);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it) tot[which] += ms;
      }
    const char *names[3] = {"two tables, 2 x 512 B per row", "one table, [param | slot] 1 KiB blocks, 32 lanes", "one table, 1 KiB blocks, a wave per row"};
    for (int which = 0; which < 3; ++which) {
      const double us = tot[which] * 1000.0 / (reps - 1);
      printf("R = %6d cold random rows, %-52s: %.1f us (%.2f TB/s of 2 KB per row)\n", R, names[which], us, R * 2048.0 / us / 1e6);
    }
  }
  return 0;
}
