// How expensive are same-address device-scope atomics issued once per wavefront / per block by a full-chip grid?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_per_wave(int* target, int naddr, const int* in) {
  int v = in[blockIdx.x * 256 + threadIdx.x];
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) for (int a = 0; a < naddr; ++a) atomicMin(&target[a * 64], v);
}
__global__ __launch_bounds__(256) void k_per_block(int* target, int naddr, const int* in) {
  __shared__ int s[4];
  int v = in[blockIdx.x * 256 + threadIdx.x];
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { v = min(min(s[0], s[1]), min(s[2], s[3])); for (int a = 0; a < naddr; ++a) atomicMin(&target[a * 64], v); }
}
__global__ __launch_bounds__(256) void k_partials(int* target, int naddr, const int* in) {
  __shared__ int s[4];
  int v = in[blockIdx.x * 256 + threadIdx.x];
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { v = min(min(s[0], s[1]), min(s[2], s[3])); for (int a = 0; a < naddr; ++a) target[a * 4096 + blockIdx.x] = v; }
}
__global__ __launch_bounds__(256) void k_none(int* target, int naddr, const int* in) {
  int v = in[blockIdx.x * 256 + threadIdx.x];
  if (v == 123456789) target[0] = v;
}

int main() {
  const int blocks = 2048;
  int *in, *target;
  hipMalloc(&in, blocks * 256 * 4); hipMemset(in, 1, blocks * 256 * 4);
  hipMalloc(&target, 1 << 20); hipMemset(target, 0x7f, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* name; void (*k)(int*, int, const int*); } ks[] = {{"no atomics", k_none}, {"atomicMin per wave", k_per_wave}, {"atomicMin per block", k_per_block}, {"partials per block", k_partials}};
  for (auto& e : ks)
    for (int naddr : {1, 6}) {
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, blocks, 256, 0, 0, target, naddr, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("%-22s addrs=%d  %8.1f us\n", e.name, naddr, best * 1000.0f);
    }
  return 0;
}
