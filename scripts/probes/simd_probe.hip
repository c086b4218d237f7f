// Where do the wavefronts of a 512-thread workgroup land?  Prints (xcc, se, cu, simd) per wave for a few workgroups and
// the histogram of "waves w and w+4 share a SIMD" — the persistent chain kernel relies on wave i → SIMD i mod 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(uint32_t* out, uint32_t spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);   // keep every workgroup resident while the rest is placed
  if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = hw; out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = xcc; }
}
int main() {
  for (uint32_t G : {5u, 256u, 512u}) {
    uint32_t* d; hipMalloc(&d, G * 8 * 8);
    hipLaunchKernelGGL(k, G, 512, 0, 0, d, 200000u);   // 2 ms
    std::vector<uint32_t> h(G * 16); hipMemcpy(h.data(), d, G * 64, hipMemcpyDeviceToHost);
    uint32_t rr = 0, distinct_cu = 0; std::vector<int> seen(1 << 16, 0);
    for (uint32_t b = 0; b < G; ++b) {
      bool ok = true;
      for (uint32_t w = 0; w < 8; ++w) { uint32_t simd = (h[2 * (b * 8 + w)] >> 4) & 3; if (simd != ((h[2 * (b * 8)] >> 4) + w) % 4) ok = false; }
      rr += ok;
      uint32_t hw = h[2 * b * 8], key = ((h[2 * b * 8 + 1] & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
      if (!seen[key]++) ++distinct_cu;
    }
    printf("G=%u: %u workgroups with wave w on SIMD (s0+w)%%4; %u distinct CUs\n", G, rr, distinct_cu);
    for (uint32_t b = 0; b < 3; ++b) { printf("  wg %u:", b); for (uint32_t w = 0; w < 8; ++w) { uint32_t hw = h[2 * (b * 8 + w)]; printf(" [x%u se%u cu%u simd%u slot%u]", h[2 * (b * 8 + w) + 1] & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15); } printf("\n"); }
    hipFree(d);
  }
}
