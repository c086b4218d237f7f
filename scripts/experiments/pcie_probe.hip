// pcie_probe.hip — what a zero-copy accessor ingestion can count on (round 5): cost of hipHostRegister on pageable memory, bandwidth of a kernel
// reading registered / pinned host memory directly, one large copy against thousands of accessor-sized ones, up + down at once.
// hipcc --offload-arch=gfx950 -O2 -o pcie_probe.out pcie_probe.hip && ./pcie_probe.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// de-stride: rows of 12 bytes out of 32-byte records
__global__ void k_destride(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t rows) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows * 3; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[(i / 3) * 8 + i % 3];
}
int main() {
  const size_t B = (size_t)1 << 30;
  uint8_t* d = nullptr; CK(hipMalloc(&d, B)); uint8_t* d2 = nullptr; CK(hipMalloc(&d2, B));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  // pageable
  uint8_t* pg = (uint8_t*)std::aligned_alloc(4096, B); std::memset(pg, 1, B);
  double t = now(); CK(hipMemcpy(d, pg, B, hipMemcpyHostToDevice)); std::printf("pageable hipMemcpy H2D 1 GiB: %.1f ms\n", now() - t);
  t = now(); CK(hipMemcpy(d, pg, B, hipMemcpyHostToDevice)); std::printf("pageable hipMemcpy H2D 1 GiB (again): %.1f ms\n", now() - t);
  for (int rep = 0; rep < 2; ++rep) {
    t = now(); CK(hipHostRegister(pg, B, hipHostRegisterDefault)); std::printf("hipHostRegister 1 GiB: %.1f ms\n", now() - t);
    t = now(); CK(hipMemcpyAsync(d, pg, B, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); std::printf("  registered H2D 1 GiB: %.1f ms\n", now() - t);
    void* dp = nullptr; CK(hipHostGetDevicePointer(&dp, pg, 0));
    for (int k = 0; k < 2; ++k) { t = now(); hipLaunchKernelGGL(k_copy, 4096, 256, 0, s, (const uint4*)dp, (uint4*)d, B / 16); CK(hipStreamSynchronize(s)); std::printf("  kernel reads registered 1 GiB: %.1f ms\n", now() - t); }
    t = now(); CK(hipHostUnregister(pg)); std::printf("hipHostUnregister: %.1f ms\n", now() - t);
  }
  // registration in pieces of 1 MiB (one per file)
  t = now(); for (size_t o = 0; o < B; o += (size_t)1 << 20) CK(hipHostRegister(pg + o, (size_t)1 << 20, hipHostRegisterDefault)); std::printf("hipHostRegister 1024 x 1 MiB: %.1f ms\n", now() - t);
  t = now(); for (size_t o = 0; o < B; o += (size_t)1 << 20) CK(hipHostUnregister(pg + o)); std::printf("hipHostUnregister 1024 x 1 MiB: %.1f ms\n", now() - t);
  // pinned
  uint8_t* pin = nullptr; t = now(); CK(hipHostMalloc(&pin, B, hipHostMallocDefault)); std::printf("hipHostMalloc 1 GiB: %.1f ms\n", now() - t);
  std::memset(pin, 2, B);
  uint8_t* pin2 = nullptr; CK(hipHostMalloc(&pin2, B, hipHostMallocDefault)); std::memset(pin2, 3, B);
  for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpyAsync(d, pin, B, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); std::printf("pinned H2D 1 GiB: %.1f ms\n", now() - t); }
  for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpyAsync(pin2, d2, B, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); std::printf("pinned D2H 1 GiB: %.1f ms\n", now() - t); }
  for (int k = 0; k < 2; ++k) { t = now(); CK(hipMemcpyAsync(d, pin, B, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(pin2, d2, B, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); std::printf("pinned H2D + D2H 1 GiB each, two streams: %.1f ms\n", now() - t); }
  for (int k = 0; k < 2; ++k) { t = now(); hipLaunchKernelGGL(k_copy, 4096, 256, 0, s, (const uint4*)pin, (uint4*)d, B / 16); CK(hipStreamSynchronize(s)); std::printf("kernel reads pinned 1 GiB: %.1f ms\n", now() - t); }
  for (int g : {256, 1024, 16384}) { t = now(); hipLaunchKernelGGL(k_copy, g, 256, 0, s, (const uint4*)pin, (uint4*)d, B / 16); CK(hipStreamSynchronize(s)); std::printf("kernel reads pinned 1 GiB, grid %d: %.1f ms\n", g, now() - t); }
  for (int k = 0; k < 2; ++k) { t = now(); hipLaunchKernelGGL(k_copy, 4096, 256, 0, s, (const uint4*)d2, (uint4*)pin2, B / 16); CK(hipStreamSynchronize(s)); std::printf("kernel writes pinned 1 GiB: %.1f ms\n", now() - t); }
  { t = now(); hipLaunchKernelGGL(k_destride, 4096, 256, 0, s, (const uint32_t*)pin, (uint32_t*)d, B / 32); CK(hipStreamSynchronize(s)); std::printf("kernel de-strides 12 of 32 bytes out of pinned 1 GiB: %.1f ms\n", now() - t); }
  // 4096 accessor-sized copies
  for (int k = 0; k < 2; ++k) { t = now(); const size_t piece = B / 4096; for (size_t o = 0; o < B; o += piece) CK(hipMemcpyAsync(d + o, pin + o, piece, hipMemcpyHostToDevice, s)); const double ti = now() - t; CK(hipStreamSynchronize(s)); std::printf("4096 x 256 KiB pinned H2D: issued in %.1f ms, done in %.1f ms\n", ti, now() - t); }
  for (int k = 0; k < 2; ++k) { t = now(); const size_t piece = B / 1024; for (size_t o = 0; o < B; o += piece) CK(hipMemcpyAsync(d + o, pin + o, piece, hipMemcpyHostToDevice, s)); const double ti = now() - t; CK(hipStreamSynchronize(s)); std::printf("1024 x 1 MiB pinned H2D: issued in %.1f ms, done in %.1f ms\n", ti, now() - t); }
  // host memcpy pageable -> pinned (the pack) on 1, 4 threads
  t = now(); std::memcpy(pin, pg, B); std::printf("host memcpy pageable -> pinned 1 GiB, 1 thread: %.1f ms\n", now() - t);
  return 0;
}
