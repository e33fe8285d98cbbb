// What does a v_mfma_f32_16x16x4_f32 cost in the patterns k_caser_tile uses?  One workgroup per CU, W waves, each running REPS rounds of
// 8 k-steps (16 LDS reads + 8 MFMAs on two accumulators).  Prints ns per MFMA per wave (100 MHz wall clock) for:
//   0  MFMAs only (operands in registers)            1  16 ds_read_b32, then 8 MFMAs (sched_barrier between)
//   2  loads of round r+1 issued before the MFMAs of round r (software pipeline)
//   hipcc --offload-arch=gfx950 -O3 -o mb_mfma32 mb_mfma32.hip && ./mb_mfma32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v mfma4(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *t, int reps) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1e-3f * (i & 63);
  __syncthreads();
  const int lane = threadIdx.x & 63, m16 = lane & 15, q4 = lane >> 4;
  const float *ap = lds + m16 * 260 + q4, *bp = lds + 4096 + m16 * 52 + q4;
  f4v a0 = {0, 0, 0, 0}, a1 = a0;
  const unsigned long long t0 = wall_clock64();
  if (MODE == 0) {
    float av[8], bv[8];
    for (int n = 0; n < 8; ++n) { av[n] = ap[4 * n]; bv[n] = bp[4 * n]; }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int n = 0; n < 8; ++n) { if (n & 1) a1 = mfma4(av[n], bv[n], a1); else a0 = mfma4(av[n], bv[n], a0); }
    }
  } else if (MODE == 1) {
    for (int r = 0; r < reps; ++r) {
      float av[8], bv[8];
      const int o = (r & 7) * 32;
#pragma unroll
      for (int n = 0; n < 8; ++n) { av[n] = ap[o + 4 * n]; bv[n] = bp[o + 4 * n]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 8; ++n) { if (n & 1) a1 = mfma4(av[n], bv[n], a1); else a0 = mfma4(av[n], bv[n], a0); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    float av[8], bv[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) { av[n] = ap[4 * n]; bv[n] = bp[4 * n]; }
    for (int r = 0; r < reps; ++r) {
      float an[8], bn[8];
      const int o = ((r + 1) & 7) * 32;
#pragma unroll
      for (int n = 0; n < 8; ++n) { an[n] = ap[o + 4 * n]; bn[n] = bp[o + 4 * n]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 8; ++n) { if (n & 1) a1 = mfma4(av[n], bv[n], a1); else a0 = mfma4(av[n], bv[n], a0); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 8; ++n) { av[n] = an[n]; bv[n] = bn[n]; }
    }
  }
  const unsigned long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1];
  if (lane == 0) t[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(int waves, int reps) {
  float *out; unsigned long long *t;
  const int grid = 256;
  hipMalloc(&out, grid * 1024 * 4); hipMalloc(&t, grid * 16 * 8);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), 140 * 1024, 0, out, t, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * waves);
  hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("mode %d  waves/CU %2d: %.1f ns per MFMA per wave\n", MODE, waves, s / h.size() * 10.0 / (reps * 8.0));
  hipFree(out); hipFree(t);
}
int main() {
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  for (int w : {4, 8, 16}) { run<0>(w, 2000); run<1>(w, 2000); run<2>(w, 2000); }
  return 0;
}
