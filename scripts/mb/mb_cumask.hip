// Probe (not product code): which CUs does bit i of a hipExtStreamCreateWithCUMask mask name?  Launches a census kernel on streams with
// different masks and prints, per XCD, how many distinct CUs ran workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o drecpy_amd/csrc/build/mb_cumask scripts/mb/mb_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_census(uint32_t *out, int spin) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the CU busy for a moment so that the dispatcher spreads the grid over every CU the mask allows
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) { }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static void census(const char *name, hipStream_t st, uint32_t *d_out, int grid) {
  std::vector<uint32_t> h(2 * grid);
  hipLaunchKernelGGL(k_census, dim3(grid), dim3(64), 0, st, d_out, 2000);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  std::map<int, std::set<uint32_t>> cus;
  for (int i = 0; i < grid; ++i) {
    const uint32_t xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
    // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[(int)xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%-28s:", name);
  for (auto &kv : cus) { printf(" xcd%d=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total %d CUs\n", total);
}

int main() {
  const int grid = 8192;
  uint32_t *d_out; CK(hipMalloc(&d_out, 2 * grid * 4));
  census("default stream", 0, d_out, grid);
  struct M { const char *name; std::vector<uint32_t> m; };
  std::vector<M> masks;
  masks.push_back({"bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0..7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0..63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 32..63", {0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"every 8th bit (32 bits)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}});
  masks.push_back({"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu}});
  masks.push_back({"all but bits 0..31", {0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}});
  for (auto &mk : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mk.m.size(), mk.m.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask: %s\n", mk.name, hipGetErrorString(e)); continue; }
    census(mk.name, st, d_out, grid);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
