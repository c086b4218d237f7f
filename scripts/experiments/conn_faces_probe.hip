// conn_faces_probe.hip — round 6: where k_conn_faces' 667 µs per 10M faces go, and whether ranking half-edges inside their buckets through a
// per-block LDS window (one global atomic per bucket a tile touches instead of one per half-edge) pays.  Standalone: single mesh, no position map.
//   hipcc -O3 --offload-arch=gfx950 -o conn_faces_probe conn_faces_probe.hip && ./conn_faces_probe [n=2236] [shuffle_vertices=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kBlock = 256;

struct Args { const uint32_t* faces; uint32_t F, V; uint32_t *opp, *ecount, *first, *vmax, *flags; };

// ---- V0: the production kernel's single-mesh form; FIRST = with the corner-per-vertex stores; ATOM = with the counting atomics ----
template <bool FIRST, bool ATOM>
__global__ __launch_bounds__(kBlock) void k_v0(const Args a) {
  uint32_t cur_max = 0;
  for (uint32_t f = blockIdx.x * kBlock + threadIdx.x; f < a.F; f += gridDim.x * kBlock) {
    uint32_t v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = a.faces[3ull * f + k];
    if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) { a.flags[0] = 1; continue; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (ATOM) a.opp[3ull * f + (k + 2) % 3] = atomicAdd(&a.ecount[min(v[k], v[(k + 1) % 3])], 1u);
      else a.opp[3ull * f + (k + 2) % 3] = min(v[k], v[(k + 1) % 3]);
      if (FIRST) a.first[v[k]] = 3u * f + k;
    }
    cur_max = max(cur_max, max(v[0], max(v[1], v[2])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cur_max = max(cur_max, (uint32_t)__shfl_down(cur_max, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(a.vmax, cur_max);
}

// ---- V2: a tile of kBlock·FPT faces per block; buckets within [smallest bucket of the tile, + W) are counted in LDS ----
template <int FPT, int W, bool FIRST_LDS>
__global__ __launch_bounds__(kBlock) void k_v2(const Args a) {
  __shared__ uint32_t cnt[W];
  __shared__ uint32_t fst[FIRST_LDS ? W : 1];
  __shared__ uint32_t smin;
  const uint32_t tile0 = blockIdx.x * (kBlock * FPT);
  uint32_t v[FPT][3], rank[FPT][3];
  bool ok[FPT];
  uint32_t mymin = kNone, cur_max = 0;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    const uint32_t f = tile0 + j * kBlock + threadIdx.x;
    ok[j] = f < a.F;
    if (ok[j]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[j][k] = a.faces[3ull * f + k];
      if (v[j][0] == v[j][1] || v[j][1] == v[j][2] || v[j][0] == v[j][2]) { a.flags[0] = 1; ok[j] = false; }
    }
    if (ok[j]) { mymin = min(mymin, min(v[j][0], min(v[j][1], v[j][2]))); cur_max = max(cur_max, max(v[j][0], max(v[j][1], v[j][2]))); }
  }
  if (threadIdx.x == 0) smin = kNone;
  for (int i = threadIdx.x; i < W; i += kBlock) { cnt[i] = 0; if (FIRST_LDS) fst[i] = kNone; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mymin = min(mymin, (uint32_t)__shfl_down(mymin, off, 64)); cur_max = max(cur_max, (uint32_t)__shfl_down(cur_max, off, 64)); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { atomicMin(&smin, mymin); atomicMax(a.vmax, cur_max); }
  __syncthreads();
  const uint32_t base = smin;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + j * kBlock + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      rank[j][k] = d < (uint32_t)W ? atomicAdd(&cnt[d], 1u) : atomicAdd(&a.ecount[low], 1u);
      if (FIRST_LDS) { const uint32_t e = v[j][k] - base; if (e < (uint32_t)W) fst[e] = 3u * f + k; else a.first[v[j][k]] = 3u * f + k; }
      else a.first[v[j][k]] = 3u * f + k;
    }
  }
  __syncthreads();
  constexpr int kIt = W / kBlock;
  static_assert(W % (kBlock * 8) == 0, "W");
  for (int it = 0; it < kIt; it += 8) {
    uint32_t c[8], r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = cnt[(it + u) * kBlock + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = c[u] ? atomicAdd(&a.ecount[base + (it + u) * kBlock + threadIdx.x], c[u]) : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) cnt[(it + u) * kBlock + threadIdx.x] = r[u];
    if (FIRST_LDS) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const uint32_t x = fst[(it + u) * kBlock + threadIdx.x]; if (x != kNone) a.first[base + (it + u) * kBlock + threadIdx.x] = x; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + j * kBlock + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      a.opp[3ull * f + (k + 2) % 3] = rank[j][k] + (d < (uint32_t)W ? cnt[d] : 0u);
    }
  }
}

// ---- V4: as V2, a thread takes FPT CONSECUTIVE faces (its 12·FPT bytes are one run; neighbours in a strip share buckets) ----
template <int FPT, int W>
__global__ __launch_bounds__(kBlock) void k_v4(const Args a) {
  __shared__ uint32_t cnt[W];
  __shared__ uint32_t fst[W];
  __shared__ uint32_t smin;
  const uint32_t tile0 = blockIdx.x * (kBlock * FPT);
  uint32_t v[FPT][3], rank[FPT][3];
  bool ok[FPT];
  uint32_t mymin = kNone, cur_max = 0;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    const uint32_t f = tile0 + threadIdx.x * FPT + j;
    ok[j] = f < a.F;
    if (ok[j]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[j][k] = a.faces[3ull * f + k];
      if (v[j][0] == v[j][1] || v[j][1] == v[j][2] || v[j][0] == v[j][2]) { a.flags[0] = 1; ok[j] = false; }
    }
    if (ok[j]) { mymin = min(mymin, min(v[j][0], min(v[j][1], v[j][2]))); cur_max = max(cur_max, max(v[j][0], max(v[j][1], v[j][2]))); }
  }
  if (threadIdx.x == 0) smin = kNone;
  for (int i = threadIdx.x; i < W; i += kBlock) { cnt[i] = 0; fst[i] = kNone; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mymin = min(mymin, (uint32_t)__shfl_down(mymin, off, 64)); cur_max = max(cur_max, (uint32_t)__shfl_down(cur_max, off, 64)); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { atomicMin(&smin, mymin); atomicMax(a.vmax, cur_max); }
  __syncthreads();
  const uint32_t base = smin;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + threadIdx.x * FPT + j;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      rank[j][k] = d < (uint32_t)W ? atomicAdd(&cnt[d], 1u) : atomicAdd(&a.ecount[low], 1u);
      const uint32_t e = v[j][k] - base;
      if (e < (uint32_t)W) fst[e] = 3u * f + k; else a.first[v[j][k]] = 3u * f + k;
    }
  }
  __syncthreads();
  constexpr int kIt = W / kBlock;
  for (int it = 0; it < kIt; it += 8) {
    uint32_t c[8], r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = cnt[(it + u) * kBlock + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = c[u] ? atomicAdd(&a.ecount[base + (it + u) * kBlock + threadIdx.x], c[u]) : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) cnt[(it + u) * kBlock + threadIdx.x] = r[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const uint32_t x = fst[(it + u) * kBlock + threadIdx.x]; if (x != kNone) a.first[base + (it + u) * kBlock + threadIdx.x] = x; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + threadIdx.x * FPT + j;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      a.opp[3ull * f + (k + 2) % 3] = rank[j][k] + (d < (uint32_t)W ? cnt[d] : 0u);
    }
  }
}


// ---- V5: block size as a parameter, the corner per vertex kept as a 16-bit tile-relative id (LDS: 6 bytes per window slot) ----
template <int BS, int FPT, int W>
__global__ __launch_bounds__(BS) void k_v5(const Args a) {
  __shared__ uint32_t cnt[W];
  __shared__ uint16_t fst[W];
  __shared__ uint32_t smin;
  static_assert(3 * BS * FPT < 65535, "tile-relative corner ids are 16 bits");
  const uint32_t tile0 = blockIdx.x * (BS * FPT);
  uint32_t v[FPT][3], rank[FPT][3];
  bool ok[FPT];
  uint32_t mymin = kNone, cur_max = 0;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    const uint32_t f = tile0 + j * BS + threadIdx.x;
    ok[j] = f < a.F;
    if (ok[j]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[j][k] = a.faces[3ull * f + k];
      if (v[j][0] == v[j][1] || v[j][1] == v[j][2] || v[j][0] == v[j][2]) { a.flags[0] = 1; ok[j] = false; }
    }
    if (ok[j]) { mymin = min(mymin, min(v[j][0], min(v[j][1], v[j][2]))); cur_max = max(cur_max, max(v[j][0], max(v[j][1], v[j][2]))); }
  }
  if (threadIdx.x == 0) smin = kNone;
  for (int i = threadIdx.x; i < W; i += BS) { cnt[i] = 0; fst[i] = 0xFFFFu; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mymin = min(mymin, (uint32_t)__shfl_down(mymin, off, 64)); cur_max = max(cur_max, (uint32_t)__shfl_down(cur_max, off, 64)); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { atomicMin(&smin, mymin); atomicMax(a.vmax, cur_max); }
  __syncthreads();
  const uint32_t base = smin;
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t tf = j * BS + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      rank[j][k] = d < (uint32_t)W ? atomicAdd(&cnt[d], 1u) : atomicAdd(&a.ecount[low], 1u);
      const uint32_t e = v[j][k] - base;
      if (e < (uint32_t)W) fst[e] = (uint16_t)(3u * tf + k); else a.first[v[j][k]] = 3u * (tile0 + tf) + k;
    }
  }
  __syncthreads();
  constexpr int kIt = W / BS, kU = kIt < 8 ? kIt : 8;
  static_assert(W % (BS * kU) == 0, "W");
  for (int it = 0; it < kIt; it += kU) {
    uint32_t c[kU], r[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) c[u] = cnt[(it + u) * BS + threadIdx.x];
#pragma unroll
    for (int u = 0; u < kU; ++u) r[u] = c[u] ? atomicAdd(&a.ecount[base + (it + u) * BS + threadIdx.x], c[u]) : 0u;
#pragma unroll
    for (int u = 0; u < kU; ++u) cnt[(it + u) * BS + threadIdx.x] = r[u];
#pragma unroll
    for (int u = 0; u < kU; ++u) { const uint32_t x = fst[(it + u) * BS + threadIdx.x]; if (x != 0xFFFFu) a.first[base + (it + u) * BS + threadIdx.x] = 3u * tile0 + x; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < FPT; ++j) {
    if (!ok[j]) continue;
    const uint32_t f = tile0 + j * BS + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t low = min(v[j][k], v[j][(k + 1) % 3]), d = low - base;
      a.opp[3ull * f + (k + 2) % 3] = rank[j][k] + (d < (uint32_t)W ? cnt[d] : 0u);
    }
  }
}

static std::vector<uint32_t> torus_faces(uint32_t n) {
  std::vector<uint32_t> f((size_t)6 * n * n);
  size_t o = 0;
  for (uint32_t a = 0; a < n; ++a) for (uint32_t b = 0; b < n; ++b) {
    const uint32_t a1 = (a + 1) % n, b1 = (b + 1) % n, i00 = a * n + b, i10 = a1 * n + b, i01 = a * n + b1, i11 = a1 * n + b1;
    f[o++] = i00; f[o++] = i10; f[o++] = i11; f[o++] = i00; f[o++] = i11; f[o++] = i01;
  }
  return f;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2236u;
  const int shuffle = argc > 2 ? atoi(argv[2]) : 0;   // 1: vertex labels permuted, 2: face order permuted, 3: both
  std::vector<uint32_t> faces = torus_faces(n);
  const uint32_t F = 2 * n * n, V = n * n;
  std::mt19937 rng(7);
  if (shuffle & 1) { std::vector<uint32_t> p(V); std::iota(p.begin(), p.end(), 0u); std::shuffle(p.begin(), p.end(), rng); for (auto& x : faces) x = p[x]; }
  if (shuffle & 2) {
    std::vector<uint32_t> p(F); std::iota(p.begin(), p.end(), 0u); std::shuffle(p.begin(), p.end(), rng);
    std::vector<uint32_t> g(faces.size());
    for (uint32_t f = 0; f < F; ++f) for (int k = 0; k < 3; ++k) g[3ull * f + k] = faces[3ull * p[f] + k];
    faces.swap(g);
  }
  std::vector<uint32_t> want(V + 1, 0);
  for (uint32_t f = 0; f < F; ++f) for (int k = 0; k < 3; ++k) ++want[std::min(faces[3ull * f + k], faces[3ull * f + (k + 1) % 3])];
  std::vector<uint64_t> start(V + 1, 0);
  for (uint32_t i = 0; i < V; ++i) start[i + 1] = start[i] + want[i];

  uint32_t *d_faces, *d_opp, *d_ecount, *d_first, *d_vmax, *d_flags;
  CK(hipMalloc(&d_faces, faces.size() * 4)); CK(hipMalloc(&d_opp, faces.size() * 4)); CK(hipMalloc(&d_ecount, (V + 1) * 4ull)); CK(hipMalloc(&d_first, (V + 1) * 4ull));
  CK(hipMalloc(&d_vmax, 4)); CK(hipMalloc(&d_flags, 4));
  CK(hipMemcpy(d_faces, faces.data(), faces.size() * 4, hipMemcpyHostToDevice));
  Args a{d_faces, F, V, d_opp, d_ecount, d_first, d_vmax, d_flags};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<uint32_t> h_opp(faces.size()), h_cnt(V + 1), h_first(V + 1);
  std::vector<uint8_t> seen(faces.size());

  auto run = [&](const char* name, auto launch, bool check) {
    float best = 1e9f, sum = 0;
    const int reps = 7;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(d_ecount, 0, (V + 1) * 4ull)); CK(hipMemsetAsync(d_first, 0xFF, (V + 1) * 4ull)); CK(hipMemsetAsync(d_vmax, 0, 4)); CK(hipMemsetAsync(d_flags, 0, 4));
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); if (r) sum += ms;
    }
    CK(hipGetLastError());
    const char* verdict = "unchecked";
    if (check) {
      CK(hipMemcpy(h_opp.data(), d_opp, faces.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_cnt.data(), d_ecount, (V + 1) * 4ull, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h_first.data(), d_first, (V + 1) * 4ull, hipMemcpyDeviceToHost));
      uint32_t vmax; CK(hipMemcpy(&vmax, d_vmax, 4, hipMemcpyDeviceToHost));
      bool good = vmax == V - 1;
      for (uint32_t i = 0; i < V && good; ++i) good = h_cnt[i] == want[i];
      std::fill(seen.begin(), seen.end(), 0);
      for (uint32_t f = 0; f < F && good; ++f) for (int k = 0; k < 3; ++k) {
        const uint32_t low = std::min(faces[3ull * f + k], faces[3ull * f + (k + 1) % 3]), rk = h_opp[3ull * f + (k + 2) % 3];
        if (rk >= want[low] || seen[start[low] + rk]) { good = false; break; }
        seen[start[low] + rk] = 1;
      }
      for (uint32_t i = 0; i < V && good; ++i) { const uint32_t c = h_first[i]; good = c != kNone && c < 3ull * F && faces[c] == i; }
      verdict = good ? "ok" : "WRONG";
    }
    printf("%-44s min %8.1f us  avg %8.1f us  %s\n", name, best * 1e3f, sum / (reps - 1) * 1e3f, verdict);
    fflush(stdout);
  };
  printf("torus n=%u: F=%u V=%u shuffle=%d\n", n, F, V, shuffle);
  const uint32_t g0 = std::min((F + kBlock - 1) / kBlock, 2048u);
  run("v0 production (atomics + first stores)", [&] { hipLaunchKernelGGL((k_v0<true, true>), g0, kBlock, 0, 0, a); }, true);
  run("v0 without first stores", [&] { hipLaunchKernelGGL((k_v0<false, true>), g0, kBlock, 0, 0, a); }, false);
  run("v0 without atomics (plain stores to opp)", [&] { hipLaunchKernelGGL((k_v0<true, false>), g0, kBlock, 0, 0, a); }, false);
  run("v0 without either", [&] { hipLaunchKernelGGL((k_v0<false, false>), g0, kBlock, 0, 0, a); }, false);
#define V2(FPT, W, FL) run("v2 FPT=" #FPT " W=" #W " first_lds=" #FL, [&] { hipLaunchKernelGGL((k_v2<FPT, W, FL>), (F + kBlock * FPT - 1) / (kBlock * FPT), kBlock, 0, 0, a); }, true)
  V2(8, 8192, true); V2(16, 8192, true);
#define V4(FPT, W) run("v4 (consecutive faces) FPT=" #FPT " W=" #W, [&] { hipLaunchKernelGGL((k_v4<FPT, W>), (F + kBlock * FPT - 1) / (kBlock * FPT), kBlock, 0, 0, a); }, true)
  V4(8, 8192);
#define V5(BS, FPT, W) run("v5 BS=" #BS " FPT=" #FPT " W=" #W, [&] { hipLaunchKernelGGL((k_v5<BS, FPT, W>), (F + BS * FPT - 1) / (BS * FPT), BS, 0, 0, a); }, true)
  V5(256, 16, 8192); V5(256, 32, 8192); V5(512, 8, 8192); V5(512, 16, 8192); V5(512, 32, 8192); V5(1024, 8, 8192); V5(1024, 16, 8192); V5(1024, 8, 4096); V5(512, 16, 4096); V5(256, 16, 4096);
  return 0;
}
