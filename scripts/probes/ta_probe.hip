// Cost of a wavefront-wide global load as a function of its address pattern (L2-resident footprint, full occupancy):
// how many core clocks of the CU's texture-address / L1 path does one wave instruction occupy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

constexpr int ITERS = 256;
template <int WORDS>
__global__ __launch_bounds__(256) void k_load(const uint32_t* __restrict__ data, const uint32_t* __restrict__ idx, uint32_t mask, uint32_t* out) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, j = idx[tid & mask];
  for (int it = 0; it < ITERS; ++it) {
    const uint32_t* p = data + (size_t)j * WORDS;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) acc += p[w];
    j = (j + 64 * 97) & mask;   // same pattern shifted: stays L2 resident
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint32_t n = 1u << 18;   // elements (≤ 4 MB at 16 B)
  std::vector<uint32_t> h(n);
  uint32_t *data, *idx, *out;
  hipMalloc(&data, (size_t)n * 16 + 64); hipMemset(data, 1, (size_t)n * 16 + 64);
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::mt19937 rng(1);
  struct Pat { const char* name; int kind; } pats[] = {{"coalesced (lane = consecutive element)", 0}, {"consecutive with a jump every 8 lanes", 1},
                                                      {"stride 4 elements", 2}, {"stride 16 elements", 3}, {"random", 4}};
  const int blocks = 256 * 8;
  for (auto& p : pats) {
    for (uint32_t i = 0; i < n; ++i) {
      switch (p.kind) {
        case 0: h[i] = i; break;
        case 1: h[i] = ((i / 8) * 1009u * 8u + (i % 8)) & (n - 1); break;
        case 2: h[i] = (i * 4u) & (n - 1); break;
        case 3: h[i] = (i * 16u) & (n - 1); break;
        default: h[i] = rng() & (n - 1); break;
      }
    }
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    float ms[4] = {0, 0, 0, 0};
    for (int wi = 0; wi < 4; ++wi) {
      float best = 1e9f;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        switch (wi) {
          case 0: hipLaunchKernelGGL(k_load<1>, blocks, 256, 0, 0, data, idx, n - 1, out); break;
          case 1: hipLaunchKernelGGL(k_load<2>, blocks, 256, 0, 0, data, idx, n - 1, out); break;
          case 2: hipLaunchKernelGGL(k_load<3>, blocks, 256, 0, 0, data, idx, n - 1, out); break;
          default: hipLaunchKernelGGL(k_load<4>, blocks, 256, 0, 0, data, idx, n - 1, out); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (t < best) best = t;
      }
      ms[wi] = best;
    }
    // wave instructions per CU = blocks*4*ITERS/256; clocks per wave instruction at 2.4 GHz
    const double winst = (double)blocks * 4 * ITERS / 256.0;
    printf("%-44s clk/wave-load: x1 %6.1f  x2 %6.1f  x3 %6.1f  x4 %6.1f\n", p.name, ms[0] * 2.4e6 / winst, ms[1] * 2.4e6 / winst, ms[2] * 2.4e6 / winst, ms[3] * 2.4e6 / winst);
  }
  return 0;
}
