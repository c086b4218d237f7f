// dmi_hostmem.cpp — host memory the copy engines read WHERE IT LIES (round 5; include/draco_mi.h "host memory the device may read in place").
// Reference seam: io/gltf/transcoder.rs:134-151 reads a file into a Vec<u8> and io/gltf/decode.rs:2277-2309 copies every accessor out of it
// into attribute buffers before MeshBuilder sees them.  Here the importer reads the file INTO memory it got from dmi_host_alloc — page-locked,
// on huge pages, registered with the runtime once — and dmi_meshes_build copies the accessors' bytes up straight out of it (the arrays' ranges
// merged into spans, about one DMA per file; rows keep their stride, indices are gathered on the device): no host pack, no staging copy.
//
// Why only memory the LIBRARY allocated.  Round 5 first page-locked the caller's own buffers for the duration of a call (hipHostRegister on
// whatever the caller passed: Python bytes objects on the process heap in the tests).  With the copies issued as hipMemcpyAsync out of such a
// registration, one run in five of the GPU test suite died LATER, in an unrelated call, of "Memory access fault by GPU … on address <process
// heap>": bringing the device to rest before every hipHostUnregister and refusing registrations that share a page did not cure it (a kernel
// reading the same registrations through their device view never faulted — it cost 29 ms of compute-unit time per 1024-file transcode
// instead).  The library's own staging buffers — malloc'd on 2 MiB boundaries, registered once, never unregistered — have carried every
// upload since round 2 without one fault: dmi_host_alloc hands out exactly that kind of block, and dmi_host_free parks it in a pool instead of
// unregistering it.
// Measured on the pool's MI355X boxes (scripts/experiments/pcie_probe.hip): one large DMA 57 GB/s up, 51 down, both at once 22–29 ms per GiB
// each way; 1024 copies of 1 MiB 29 ms per GiB, 4096 of 256 KiB 60 ms; page-locking a GiB of fresh memory 39 ms.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <vector>

#include "dmi_host.hpp"

namespace dmi {
namespace {
struct Block { size_t cap = 0; bool in_use = false; };
std::mutex g_mutex;
std::map<uintptr_t, Block> g_blocks;   // every block ever registered and not yet released, by first byte
size_t g_parked = 0;                   // bytes of the blocks not in use

void release_block(uintptr_t lo) {     // (the device at rest: no command names the block)
  void* p = reinterpret_cast<void*>(lo);
  if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
  std::free(p);
}
}  // namespace

// does [p, p + bytes) lie inside a block dmi_host_alloc handed out?
bool host_in_place(const void* p, size_t bytes) {
  if (!p) return false;
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_blocks.upper_bound(lo);
  if (it == g_blocks.begin()) return false;
  --it;
  return it->second.in_use && lo >= it->first && lo + bytes <= it->first + it->second.cap;
}

// parked blocks back to the system (dmi_release_cached_memory)
void host_blocks_drop_parked() {
  std::vector<uintptr_t> drop;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto it = g_blocks.begin(); it != g_blocks.end();) {
      if (!it->second.in_use) { g_parked -= it->second.cap; drop.push_back(it->first); it = g_blocks.erase(it); } else ++it;
    }
  }
  if (drop.empty()) return;
  (void)hipDeviceSynchronize();
  for (uintptr_t lo : drop) release_block(lo);
}
}  // namespace dmi

using namespace dmi;

extern "C" {

void* dmi_host_alloc(size_t bytes) {
  if (!bytes) return nullptr;
  const size_t want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    uintptr_t best = 0;
    size_t best_cap = 0;
    for (auto& kv : g_blocks) if (!kv.second.in_use && kv.second.cap >= want && kv.second.cap <= 2 * want && (!best || kv.second.cap < best_cap)) { best = kv.first; best_cap = kv.second.cap; }
    if (best) { g_blocks[best].in_use = true; g_parked -= best_cap; return reinterpret_cast<void*>(best); }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return nullptr; }
  void* q = nullptr;
  if (posix_memalign(&q, (size_t)2 << 20, want) != 0 || !q) return nullptr;
  advise_huge_pages(q, want);
  if (hipHostRegister(q, want, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); std::free(q); return nullptr; }
  std::lock_guard<std::mutex> lock(g_mutex);
  g_blocks[reinterpret_cast<uintptr_t>(q)] = Block{want, true};
  return q;
}

void dmi_host_free(void* p) {
  if (!p) return;
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_blocks.find(lo);
    if (it == g_blocks.end() || !it->second.in_use) return;
    it->second.in_use = false;
    if (g_parked + it->second.cap <= host_pool_limit()) { g_parked += it->second.cap; return; }   // parked: still registered, handed out again by the next dmi_host_alloc
    g_blocks.erase(it);
  }
  (void)hipDeviceSynchronize();
  release_block(lo);
}

int dmi_host_is_registered(const void* p, size_t bytes) { return host_in_place(p, bytes) ? 1 : 0; }

}  // extern "C"
