// dmi_hostmem.cpp — caller buffers the device reads WHERE THEY LIE (round 5; include/draco_mi.h "host memory the device may read in place").
// Reference seam: io/gltf/transcoder.rs:134-151 reads a file into a Vec<u8> and io/gltf/decode.rs:2277-2309 copies every accessor out of it
// into attribute buffers before MeshBuilder sees them; here the file's bytes are page-locked once (dmi_host_register, or read into
// dmi_host_alloc memory to begin with) and dmi_meshes_build's ingest kernel (dmi_build.hip k_mb_ingest) gathers the accessors' rows and the
// index arrays straight out of them over PCIe — no host pack, no staging copy, no separate upload.  Measured on the pool's MI355X boxes
// (scripts/experiments/pcie_probe.hip): a kernel reads page-locked host memory at 55 GB/s (19.4 ms per GiB, any grid), the same as one
// large DMA; 4096 accessor-sized DMAs take 3 × that; page-locking a GiB of fresh pageable memory costs 39 ms once.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include "dmi_host.hpp"

namespace dmi {
namespace {
struct Range { uintptr_t hi = 0; uint32_t refs = 0; bool owned = false; uintptr_t dev = 0 /* device-visible address of the first byte */; };
std::mutex g_mutex;
std::map<uintptr_t, Range> g_ranges;   // by first byte
}  // namespace

// the device-visible address of p when [p, p + bytes) lies inside page-locked memory this library knows of, else null (a registered range is mapped
// as one piece: its view is looked up once, when it is registered — hipHostGetDevicePointer per accessor cost 5 ms per 1000 arrays)
const void* host_device_view(const void* p, size_t bytes) {
  if (!p) return nullptr;
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_ranges.upper_bound(lo);
  if (it == g_ranges.begin()) return nullptr;
  --it;
  if (lo < it->first || lo + bytes > it->second.hi || !it->second.dev) return nullptr;
  return reinterpret_cast<const void*>(it->second.dev + (lo - it->first));
}
static uintptr_t device_view_of(void* p) {
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return reinterpret_cast<uintptr_t>(d);
}
}  // namespace dmi

using namespace dmi;

extern "C" {

int dmi_host_register(const void* p, size_t bytes) {
  if (!p || !bytes) return host_fail(DMI_ERR_INVALID_ARGUMENT, "dmi_host_register: null / empty");
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_ranges.find(lo);
  if (it != g_ranges.end() && it->second.hi == lo + bytes) { ++it->second.refs; return DMI_OK; }   // the same buffer again (two transcoders over one file list)
  // a range that overlaps a known one in any other way is refused: the caller's accessors in it take the packed path
  auto nx = g_ranges.lower_bound(lo);
  if (nx != g_ranges.end() && nx->first < lo + bytes) return host_fail(DMI_ERR_INVALID_ARGUMENT, "dmi_host_register: overlaps a registered range");
  if (nx != g_ranges.begin()) { auto pv = std::prev(nx); if (pv->second.hi > lo) return host_fail(DMI_ERR_INVALID_ARGUMENT, "dmi_host_register: overlaps a registered range"); }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return host_fail(DMI_ERR_NO_DEVICE, "no HIP device visible"); }
  const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e != hipSuccess) { (void)hipGetLastError(); return host_fail(DMI_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
  g_ranges[lo] = Range{lo + bytes, 1u, false, device_view_of(const_cast<void*>(p))};
  return DMI_OK;
}

int dmi_host_unregister(const void* p) {
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_ranges.find(lo);
  if (it == g_ranges.end() || it->second.owned) return host_fail(DMI_ERR_INVALID_ARGUMENT, "dmi_host_unregister: not a registered buffer");
  if (--it->second.refs) return DMI_OK;
  g_ranges.erase(it);
  if (hipHostUnregister(const_cast<void*>(p)) != hipSuccess) { (void)hipGetLastError(); return host_fail(DMI_ERR_HIP, "hipHostUnregister"); }
  return DMI_OK;
}

void* dmi_host_alloc(size_t bytes) {
  if (!bytes) return nullptr;
  void* q = nullptr;
  const size_t want = (bytes + 4095) & ~(size_t)4095;
  if (posix_memalign(&q, bytes >= ((size_t)2 << 20) ? (size_t)2 << 20 : 4096, want) != 0 || !q) return nullptr;
  if (bytes >= ((size_t)2 << 20)) advise_huge_pages(q, want);
  if (hipHostRegister(q, want, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); std::free(q); return nullptr; }
  std::lock_guard<std::mutex> lock(g_mutex);
  g_ranges[reinterpret_cast<uintptr_t>(q)] = Range{reinterpret_cast<uintptr_t>(q) + want, 1u, true, device_view_of(q)};
  return q;
}

void dmi_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_ranges.find(reinterpret_cast<uintptr_t>(p));
    if (it == g_ranges.end() || !it->second.owned) return;
    g_ranges.erase(it);
  }
  (void)hipHostUnregister(p);
  std::free(p);
}

int dmi_host_is_registered(const void* p, size_t bytes) { return host_device_view(p, bytes) != nullptr; }

}  // extern "C"
