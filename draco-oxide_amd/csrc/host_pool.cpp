// host_pool.cpp — where the library's large host arrays live: operator new on 2 MiB boundaries with transparent huge pages (local to the library), the
// process-wide pool of recycled index / flag arrays (dmi_host.hpp VecPool) and the budget of running serial walks.  Split out of host_conn.cpp in round 5.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <sys/mman.h>

#include "dmi_host.hpp"

namespace dmi {
WalkSlots& walk_slots() { static WalkSlots w; return w; }
}  // namespace dmi

// Large heap arrays of THIS library (every std::vector of index / flag arrays: hidden visibility — no other module's allocations come here) start
// on a 2 MiB boundary and end on one, and ask for transparent huge pages as a whole.  malloc hands a 5 MB flag array out 16 bytes into its
// mapping: the 2 MiB-aligned interior that advise_huge_pages can flag leaves its first and last megabytes on 4 KiB pages, and the serial walks
// (one flag byte per step, a mesh row apart: a new page every step) then miss the TLB on 20–40 % of their flag accesses — the 10M-face
// traversal on the GPU box's EPYC: 64 ms against 48 ms with every array on huge pages.  Memory comes from posix_memalign: released by the
// default operator delete (free).  DMI_NO_THP=1: plain malloc.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define DMI_NO_OPERATOR_NEW 1      // (the sanitizer build keeps the runtime's allocator: it pairs operator new with operator delete)
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define DMI_NO_OPERATOR_NEW 1
#endif
#ifndef DMI_NO_OPERATOR_NEW
void* operator new(std::size_t n) {   // (local to the library: libdraco_mi.map)
  constexpr std::size_t kHuge = (std::size_t)2 << 20;
  static const bool off = std::getenv("DMI_NO_THP") != nullptr;
  if (n >= kHuge && !off) {
    const std::size_t want = (n + kHuge - 1) & ~(kHuge - 1);
    void* p = nullptr;
    if (want >= n && posix_memalign(&p, kHuge, want) == 0 && p) { (void)madvise(p, want, MADV_HUGEPAGE); return p; }
  }
  if (void* p = std::malloc(n ? n : 1)) return p;
  throw std::bad_alloc();
}
void* operator new[](std::size_t n) { return ::operator new(n); }
#endif

namespace dmi {
void advise_huge_pages(void* p, size_t bytes) {
  static const bool off = std::getenv("DMI_NO_THP") != nullptr;
  if (off || !p) return;
  constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
  const uintptr_t lo = ((uintptr_t)p + kHuge - 1) & ~(kHuge - 1), hi = ((uintptr_t)p + bytes) & ~(kHuge - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
}

size_t host_pool_limit() {
  static const size_t limit = [] {
    const char* e = std::getenv("DMI_HOST_CACHE_MB");
    return (size_t)(e ? std::max(0l, std::atol(e)) : 4096l) << 20;
  }();
  return limit;
}
std::atomic<size_t>& host_pool_bytes() { static std::atomic<size_t> b{0}; return b; }
void host_pool_drop_all() { VecPool<uint8_t>::get().drop_all(); VecPool<uint32_t>::get().drop_all(); VecPool<uint64_t>::get().drop_all(); }

}  // namespace dmi
