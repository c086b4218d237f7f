// sdma_probe.hip — round 6: which hipMemcpyAsync calls the runtime serves with a blit KERNEL (__amd_rocclr_copyBuffer in a kernel trace: it blocks every
// other queue's kernels until it ends — clear_probe.hip) instead of an SDMA engine.  Each case is bracketed by a marker kernel with a distinct grid size;
// run under rocprofv3 --kernel-trace and read the order:  m<k> [copyBuffer?] m<k>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_marker(uint32_t* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_touch(uint32_t* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)i; }
int main() {
  const size_t B = 64u << 20;
  uint8_t *dev, *dev2, *pin, *reg = nullptr; uint32_t* flag;
  CK(hipMalloc(&dev, B)); CK(hipMalloc(&dev2, B)); CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256)); CK(hipHostMalloc(&pin, B));
  if (posix_memalign(reinterpret_cast<void**>(&reg), 2u << 20, B)) return 1;
  madvise(reg, B, MADV_HUGEPAGE); memset(reg, 1, B); CK(hipHostRegister(reg, B, hipHostRegisterDefault));
  uint8_t* pageable = static_cast<uint8_t*>(malloc(B)); memset(pageable, 2, B);
  hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreate(&s3));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  int id = 0;
  auto mark = [&](hipStream_t s) { ++id; hipLaunchKernelGGL(k_marker, id, 64, 0, s, flag); CK(hipStreamSynchronize(s)); };
  auto run = [&](const char* what, auto fn) { mark(s1); printf("case %2d: %s\n", id, what); fn(); CK(hipDeviceSynchronize()); };
  run("D2H into hipHostMalloc memory, idle stream", [&] { CK(hipMemcpyAsync(pin, dev, B, hipMemcpyDeviceToHost, s2)); });
  run("D2H into registered memory, idle stream", [&] { CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, s2)); });
  run("kernel, then D2H into registered memory on the same stream", [&] { hipLaunchKernelGGL(k_touch, 1024, 256, 0, s2, reinterpret_cast<uint32_t*>(dev), B / 4); CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, s2)); });
  run("kernel on s1, event, s2 waits, D2H into registered memory on s2", [&] { hipLaunchKernelGGL(k_touch, 1024, 256, 0, s1, reinterpret_cast<uint32_t*>(dev), B / 4); CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s2, ev, 0)); CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, s2)); });
  run("D2H into registered memory at an odd offset (+ 4 bytes, length - 8)", [&] { CK(hipMemcpyAsync(reg + 4, dev + 4, B - 8, hipMemcpyDeviceToHost, s2)); });
  run("D2H 4 KB into registered memory", [&] { CK(hipMemcpyAsync(reg, dev, 4096, hipMemcpyDeviceToHost, s2)); });
  run("D2H 1 MB into registered memory", [&] { CK(hipMemcpyAsync(reg, dev, 1u << 20, hipMemcpyDeviceToHost, s2)); });
  run("H2D from registered memory", [&] { CK(hipMemcpyAsync(dev, reg, B, hipMemcpyHostToDevice, s2)); });
  run("H2D from pageable memory", [&] { CK(hipMemcpyAsync(dev, pageable, B, hipMemcpyHostToDevice, s2)); });
  run("D2H into pageable memory", [&] { CK(hipMemcpyAsync(pageable, dev, B, hipMemcpyDeviceToHost, s2)); });
  run("D2H into registered memory on a blocking (default-flag) stream", [&] { CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, s3)); });
  run("D2H into registered memory on the null stream", [&] { CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, 0)); });
  run("two D2H at once on two streams (registered, hipHostMalloc)", [&] { CK(hipMemcpyAsync(reg, dev, B, hipMemcpyDeviceToHost, s2)); CK(hipMemcpyAsync(pin, dev2, B, hipMemcpyDeviceToHost, s1)); });
  run("D2D", [&] { CK(hipMemcpyAsync(dev2, dev, B, hipMemcpyDeviceToDevice, s2)); });
  mark(s1);
  return 0;
}
