// host_pool.cpp — where the library's large host arrays live: operator new on 2 MiB boundaries with transparent huge pages (local to the library), the
// process-wide pool of recycled index / flag arrays (dmi_host.hpp VecPool) and the budget of running serial walks.  Split out of host_conn.cpp in round 5.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <sys/mman.h>

#include "dmi_host.hpp"

namespace dmi {
WalkSlots& walk_slots() { static WalkSlots w; return w; }

// ---- the switches of the call a thread is working for, the process's defaults and options (round 6: typed, no environment reads) ----
namespace {
dmi_debug g_default_debug{};                        // dmi_set_default_debug
std::atomic<uint32_t> g_process_flags{0};           // dmi_configure_process
std::atomic<size_t> g_host_cache_mb{4096}, g_device_cache_mb{0}, g_decode_budget_mb{16384};
thread_local const dmi_debug* tl_debug = nullptr;
}  // namespace
const dmi_debug* dbg_ptr() { return tl_debug ? tl_debug : &g_default_debug; }
DebugScope::DebugScope(const dmi_debug* d) : prev(tl_debug) { if (d) tl_debug = d; }
DebugScope::~DebugScope() { tl_debug = prev; }
uint32_t process_flags() { return g_process_flags.load(std::memory_order_relaxed); }
size_t decode_budget_bytes() { return g_decode_budget_mb.load() << 20; }
size_t device_cache_limit_mb() { return g_device_cache_mb.load(); }
}  // namespace dmi

extern "C" {
void dmi_set_default_debug(const dmi_debug* d) { dmi::g_default_debug = d ? *d : dmi_debug{}; }
int dmi_configure_process(const dmi_process_options* o) {
  if (!o) return dmi::host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  dmi::g_process_flags.store(o->flags);
  dmi::g_host_cache_mb.store(o->host_cache_mb ? o->host_cache_mb : 4096);
  dmi::g_device_cache_mb.store(o->device_cache_mb);
  dmi::g_decode_budget_mb.store(o->decode_budget_mb ? o->decode_budget_mb : 16384);
  return DMI_OK;
}
}

// Large heap arrays of THIS library (every std::vector of index / flag arrays: hidden visibility — no other module's allocations come here) start
// on a 2 MiB boundary and end on one, and ask for transparent huge pages as a whole.  malloc hands a 5 MB flag array out 16 bytes into its
// mapping: the 2 MiB-aligned interior that advise_huge_pages can flag leaves its first and last megabytes on 4 KiB pages, and the serial walks
// (one flag byte per step, a mesh row apart: a new page every step) then miss the TLB on 20–40 % of their flag accesses — the 10M-face
// traversal on the GPU box's EPYC: 64 ms against 48 ms with every array on huge pages.  Memory comes from posix_memalign: released by the
// default operator delete (free).  OPT-IN since round 6 (dmi_configure_process, DMI_PROCESS_HUGE_PAGE_NEW): without it this operator new is the
// standard one (malloc or std::bad_alloc) — a drop-in library does not change how its host process allocates unless asked.
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
  // from 4 MiB: the arrays of a 10M-face mesh are 5–120 MB; the per-mesh copies of a batch's seam tables (2–2.4 MB at 200k faces, made and freed a thousand times per
  // transcode) took this path at 2 MiB — an mmap, a madvise, 2 MiB page faults and a munmap each: madvise + munmap were 12 % of a seam transcode's CPU samples
  const uint32_t pf = dmi::process_flags();
  if (n >= 2 * kHuge && (pf & DMI_PROCESS_HUGE_PAGE_NEW) && !(pf & DMI_PROCESS_NO_THP)) {
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
  if ((process_flags() & DMI_PROCESS_NO_THP) || !p) return;
  constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
  const uintptr_t lo = ((uintptr_t)p + kHuge - 1) & ~(kHuge - 1), hi = ((uintptr_t)p + bytes) & ~(kHuge - 1);
  if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
}

size_t host_pool_limit() { return g_host_cache_mb.load() << 20; }
std::atomic<size_t>& host_pool_bytes() { static std::atomic<size_t> b{0}; return b; }
void host_pool_drop_all() { VecPool<uint8_t>::get().drop_all(); VecPool<uint32_t>::get().drop_all(); VecPool<uint64_t>::get().drop_all(); }

}  // namespace dmi

// ---- the library's worker threads (dmi_host.hpp: run_threads) ----
namespace dmi {
namespace {
struct Worker {
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> task;
  bool has = false;
};
struct WorkerPool {
  std::mutex m;
  std::vector<Worker*> idle;
  static WorkerPool& get() {   // (never destroyed: its threads are parked for the life of the process)
    static WorkerPool* p = [] {
      WorkerPool* q = new WorkerPool();
      g_pool = q;
      // a forked child has none of the parent's threads: it starts with an empty pool (the parked workers it inherited on paper are forgotten, not joined)
      (void)pthread_atfork(nullptr, nullptr, [] { if (g_pool) { new (&g_pool->m) std::mutex(); new (&g_pool->idle) std::vector<Worker*>(); } });
      return q;
    }();
    return *p;
  }
  static WorkerPool* g_pool;
  static void loop(WorkerPool* pool, Worker* w) {
    for (;;) {
      std::function<void()> t;
      {
        std::unique_lock<std::mutex> lock(w->m);
        w->cv.wait(lock, [&] { return w->has; });
        t = std::move(w->task);
        w->task = nullptr;
        w->has = false;
      }
      t();
      t = nullptr;
      std::lock_guard<std::mutex> lock(pool->m);
      pool->idle.push_back(w);
    }
  }
  void submit(std::function<void()> fn) {
    Worker* w = nullptr;
    { std::lock_guard<std::mutex> lock(m); if (!idle.empty()) { w = idle.back(); idle.pop_back(); } }
    if (!w) {
      w = new Worker();
      Thread th(&WorkerPool::loop, this, w);   // (throws std::system_error when the process is out of threads: the caller's problem, nothing is queued)
      th.detach();
    }
    { std::lock_guard<std::mutex> lock(w->m); w->task = std::move(fn); w->has = true; }
    w->cv.notify_one();
  }
};
WorkerPool* WorkerPool::g_pool = nullptr;
}  // namespace

void pool_submit(std::function<void()> fn) { WorkerPool::get().submit(std::move(fn)); }
}  // namespace dmi
