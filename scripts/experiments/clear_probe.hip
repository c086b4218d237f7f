// clear_probe.hip — round 6: the by-value multi-range fill (k_clear_ranges) on the table stage's 20 + 20 + 30 MB, alone and beside a device-to-host copy
//   hipcc -O3 --offload-arch=gfx950 -o clear_probe clear_probe.hip && ./clear_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstring>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct ClearRanges { void* p[24]; uint64_t bytes[24]; uint8_t value[24]; uint32_t count; };
__global__ __launch_bounds__(256) void k_dword(const ClearRanges r) {
  uint32_t* __restrict__ d = static_cast<uint32_t*>(r.p[blockIdx.y]);
  const uint64_t bytes = r.bytes[blockIdx.y], n = bytes >> 2;
  const uint32_t b = r.value[blockIdx.y], w = b * 0x01010101u;
  for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (uint64_t)gridDim.x * 256) d[v] = w;
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 3u)) reinterpret_cast<uint8_t*>(d)[(n << 2) + threadIdx.x] = (uint8_t)b;
}
// 16-byte stores over the 16-byte aligned middle of the range, bytes at its two ends
__global__ __launch_bounds__(256) void k_x4(const ClearRanges r) {
  uint8_t* __restrict__ p = static_cast<uint8_t*>(r.p[blockIdx.y]);
  const uint64_t bytes = r.bytes[blockIdx.y];
  const uint32_t b = r.value[blockIdx.y], w = b * 0x01010101u;
  const uint64_t head = std::min<uint64_t>(bytes, (16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u), n16 = (bytes - head) >> 4, tail = bytes - head - (n16 << 4);
  uint4* __restrict__ d = reinterpret_cast<uint4*>(p + head);
  const uint4 w4 = make_uint4(w, w, w, w);
  for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < n16; v += (uint64_t)gridDim.x * 256) d[v] = w4;
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) p[threadIdx.x] = (uint8_t)b;
    if (threadIdx.x < tail) p[head + (n16 << 4) + threadIdx.x] = (uint8_t)b;
  }
}
int main() {
  const size_t A = 20u << 20, B = 20u << 20, Cc = 30u << 20;
  uint8_t *a, *b, *c, *big, *host;
  CK(hipMalloc(&a, A + 64)); CK(hipMalloc(&b, B + 64)); CK(hipMalloc(&c, Cc + 64)); CK(hipMalloc(&big, 120u << 20)); CK(hipHostMalloc(&host, 120u << 20));
  uint8_t* reg = nullptr; if (posix_memalign(reinterpret_cast<void**>(&reg), 2u << 20, 120u << 20)) return 1; madvise(reg, 120u << 20, MADV_HUGEPAGE); memset(reg, 1, 120u << 20); CK(hipHostRegister(reg, 120u << 20, hipHostRegisterDefault));
  ClearRanges r{}; r.p[0] = a; r.bytes[0] = 4; r.p[1] = b + 4; r.bytes[1] = 4; r.p[2] = a + 256; r.bytes[2] = A - 256; r.p[3] = b + 256; r.bytes[3] = B - 256; r.value[3] = 0xFF; r.p[4] = c; r.bytes[4] = Cc; r.count = 5;
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto fn, int with_copy) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      if (with_copy) CK(hipMemcpyAsync(with_copy == 2 ? reg : host, big, 120u << 20, hipMemcpyDeviceToHost, s2));
      CK(hipEventRecord(e0, s1)); fn(); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
      CK(hipDeviceSynchronize());
    }
    printf("%-56s %8.1f us%s\n", name, best * 1e3f, with_copy == 2 ? "  (beside a 120 MB device-to-host copy into REGISTERED memory)" : with_copy ? "  (beside a 120 MB device-to-host copy into hipHostMalloc memory)" : "");
  };
  for (int wc = 0; wc < 3; ++wc) {
    time("k_dword grid 512 x 5", [&] { hipLaunchKernelGGL(k_dword, dim3(512, 5), 256, 0, s1, r); }, wc);
    time("k_dword grid 4096 x 5", [&] { hipLaunchKernelGGL(k_dword, dim3(4096, 5), 256, 0, s1, r); }, wc);
    time("k_x4 grid 512 x 5", [&] { hipLaunchKernelGGL(k_x4, dim3(512, 5), 256, 0, s1, r); }, wc);
    time("k_x4 grid 2048 x 5", [&] { hipLaunchKernelGGL(k_x4, dim3(2048, 5), 256, 0, s1, r); }, wc);
    time("three hipMemsetAsync + two small", [&] { CK(hipMemsetAsync(a, 0, 4, s1)); CK(hipMemsetAsync(b + 4, 0, 4, s1)); CK(hipMemsetAsync(a + 256, 0, A - 256, s1)); CK(hipMemsetAsync(b + 256, 0xFF, B - 256, s1)); CK(hipMemsetAsync(c, 0, Cc, s1)); }, wc);
  }
  for (int k = 0; k < 2; ++k) {
    float ms;
    CK(hipEventRecord(e0, s2)); CK(hipMemcpyAsync(k ? reg : host, big, 120u << 20, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(e1, s2)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("120 MB device-to-host into %s memory: %.1f us\n", k ? "registered" : "hipHostMalloc", ms * 1e3f);
  }
  return 0;
}
