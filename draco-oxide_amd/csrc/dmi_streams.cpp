// dmi_streams.cpp — library streams that outlive their threads, and the NUMA placement of a whole-mesh call (split out of dmi_prepare.cpp in round 5).
#include "dmi_prepare.hpp"

using namespace dmi;

#include <sched.h>
#include <sys/prctl.h>

namespace {

// The serial walks of the connectivity stage are bound by memory latency; on a two-socket host a thread that wanders to the other socket
// pays ≈ 100 ns more per miss (10M-triangle traversal: 72 ms with its tables on the local node, 120–130 ms across the sockets).  For the
// duration of a whole-mesh call the calling thread — and the library threads it starts, which inherit its mask — therefore stay on the
// CPUs of ONE memory node: the one the call's GPU hangs off (its staging DMA is local there too; the ranks of a multi-GPU job spread over
// the sockets the way their GPUs do).  The previous mask is restored when the call returns.  DMI_NO_NUMA_PIN=1 turns it off.
int device_numa_node(int device) {
  static std::mutex m;
  static std::vector<std::pair<int, int>> known;
  std::lock_guard<std::mutex> lock(m);
  for (auto& e : known) if (e.first == device) return e.second;
  int node = -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof bus, device) == hipSuccess) {
    for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (std::FILE* f = std::fopen(path.c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
  } else (void)hipGetLastError();
  known.push_back({device, node});
  return node;
}
struct NumaPin {
  cpu_set_t old;
  bool active = false;
  explicit NumaPin(int device) {
    if (!(process_flags() & DMI_PROCESS_NUMA_PIN)) return;   // (opt-in since round 6: dmi_configure_process)
    const int node = device_numa_node(device);
    if (node < 0) return;
    // the node's CPU list, read once per node (a batch pipeline opens a scope per stage step)
    static std::mutex m;
    static std::vector<std::pair<int, cpu_set_t>> lists;
    cpu_set_t want;
    CPU_ZERO(&want);
    {
      std::lock_guard<std::mutex> lock(m);
      bool known = false;
      for (auto& e : lists) if (e.first == node) { want = e.second; known = true; }
      if (!known) {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        char buf[1024] = {0};
        if (std::FILE* f = std::fopen(path.c_str(), "r")) {
          if (!std::fgets(buf, sizeof buf, f)) buf[0] = 0;
          std::fclose(f);
        }
        for (char* p = buf; *p && *p != '\n';) {   // "0-63,128-191"
          char* end = nullptr;
          const long lo = std::strtol(p, &end, 10);
          if (end == p) break;
          long hi = lo;
          p = end;
          if (*p == '-') { hi = std::strtol(p + 1, &end, 10); p = end; }
          for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &want);
          if (*p == ',') ++p;
        }
        lists.push_back({node, want});
      }
    }
    if (CPU_COUNT(&want) == 0) return;
    if (sched_getaffinity(0, sizeof old, &old) != 0) return;
    cpu_set_t both;
    CPU_AND(&both, &old, &want);
    if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, &old)) return;
    if (sched_setaffinity(0, sizeof both, &both) == 0) active = true;
  }
  ~NumaPin() { if (active) (void)sched_setaffinity(0, sizeof old, &old); }
  NumaPin(const NumaPin&) = delete;
  NumaPin& operator=(const NumaPin&) = delete;
};

// Library streams of a host thread outlive it: a thread takes them from a process-wide pool on first use and its exit hands them back (they are not
// destroyed).  Work recorded on them can be waited for by ANOTHER thread later — a transcode pipeline's build thread records a group's table
// event and may be gone when the prepare thread waits for it: with the stream destroyed the runtime's hipEventSynchronize answered "event
// last recorded in a capturing stream" once in twenty calls — and hipStreamCreate (≈ 1 ms, serialised across threads) is paid once per stream,
// not once per pipeline thread.  kind 0: the thread's stream (connectivity stage of a whole-mesh call, adopted by the job it creates);
// kinds 1 / 2: the two group streams (non-blocking).
namespace {
struct StreamPool {
  std::mutex m;
  std::vector<std::pair<int, std::shared_ptr<StreamHolder>>> idle[3];
  static StreamPool& get() { static StreamPool* p = new StreamPool(); return *p; }   // (never destroyed: threads may exit after static destruction began)
};
struct ThreadStreams {
  std::vector<std::pair<int, std::shared_ptr<StreamHolder>>> mine[3];
  ~ThreadStreams() {
    StreamPool& pool = StreamPool::get();
    std::lock_guard<std::mutex> lock(pool.m);
    for (int k = 0; k < 3; ++k) for (auto& e : mine[k]) pool.idle[k].push_back(std::move(e));
  }
  std::shared_ptr<StreamHolder> get(int kind, int device) {
    hip_used().store(true, std::memory_order_relaxed);
    for (auto& e : mine[kind]) if (e.first == device) return e.second;
    std::shared_ptr<StreamHolder> h;
    {
      StreamPool& pool = StreamPool::get();
      std::lock_guard<std::mutex> lock(pool.m);
      auto& idle = pool.idle[kind];
      for (size_t i = 0; i < idle.size(); ++i) if (idle[i].first == device) { h = std::move(idle[i].second); idle.erase(idle.begin() + (long)i); break; }
    }
    if (!h) {
      h = std::make_shared<StreamHolder>();
      const hipError_t e = hipSetDevice(device) != hipSuccess ? hipErrorInvalidDevice : (kind == 0 ? hipStreamCreate(&h->s) : hipStreamCreateWithFlags(&h->s, hipStreamNonBlocking));
      if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    mine[kind].push_back({device, h});
    return h;
  }
};
ThreadStreams& thread_streams() { static thread_local ThreadStreams t; return t; }
}  // namespace
}  // namespace

// Two more library streams per (host thread, device): consecutive groups of a slice alternate between them, so the read-back of one
// group's tables overlaps the upload of the next group's faces (the two directions of the link run side by side).
namespace dmi {
std::shared_ptr<StreamHolder> library_thread_stream(int device) { return thread_streams().get(0, device); }
unsigned long_wait_flags() { return dbg_on(DMI_DBG_SPIN_WAITS) ? hipEventDisableTiming : (hipEventDisableTiming | hipEventBlockingSync); }
hipError_t long_wait_event(hipEvent_t e) {
  if (dbg_on(DMI_DBG_SPIN_WAITS)) return hipEventSynchronize(e);
  // back-to-back queries for the first 60 µs (a wait that short is on somebody's critical path), then 20 µs naps — with the thread's timer slack at 1 µs
  // instead of the default 50 for the duration of the wait (the thread may be the caller's: its own value is put back)
  struct Slack {
    long old = -1;
    void tighten() { if (old < 0) { old = prctl(PR_GET_TIMERSLACK, 0ul, 0ul, 0ul, 0ul); if (old >= 0) (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul); } }
    ~Slack() { if (old >= 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old, 0ul, 0ul, 0ul); }
  } slack;
  const auto t0 = std::chrono::steady_clock::now();
  for (bool napping = false;;) {
    const hipError_t r = hipEventQuery(e);
    if (r == hipSuccess) return hipEventSynchronize(e);   // (done: returns at once; keeps the runtime's own completion bookkeeping in one place)
    if (r != hipErrorNotReady) return r;
    (void)hipGetLastError();                              // (hipErrorNotReady is sticky otherwise)
    if (!napping) {
      if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(60)) continue;
      napping = true;
      slack.tighten();
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}
hipError_t long_wait_stream(hipStream_t s) {
  if (dbg_on(DMI_DBG_SPIN_WAITS)) return hipStreamSynchronize(s);
  {   // nothing pending (a job's stream at its release, usually): no event, no wait
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) return q;
    (void)hipGetLastError();
  }
  // one blocking event per thread and device (an event belongs to the device that was current when it was created)
  struct Ev { int device = -1; hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } };
  thread_local Ev ev[2];
  int device = 0;
  if (hipError_t r = hipGetDevice(&device)) return r;
  Ev* slot = ev[0].device == device ? &ev[0] : (ev[1].device == device ? &ev[1] : nullptr);
  if (!slot) {
    slot = ev[0].e ? &ev[1] : &ev[0];
    if (slot->e) { (void)hipEventDestroy(slot->e); slot->e = nullptr; }
    if (hipError_t r = hipEventCreateWithFlags(&slot->e, hipEventDisableTiming | hipEventBlockingSync)) { slot->device = -1; return r; }
    slot->device = device;
  }
  if (hipError_t r = hipEventRecord(slot->e, s)) return r;
  return long_wait_event(slot->e);
}
hipStream_t library_group_stream(int device, int which) {
  const std::shared_ptr<StreamHolder> h = thread_streams().get(1 + (which & 1), device);
  return h ? h->s : nullptr;
}
NumaScope::NumaScope(int device) : impl(new NumaPin(device)) {}
NumaScope::~NumaScope() { delete static_cast<NumaPin*>(impl); }
}  // namespace dmi

namespace dmi {
void run_threads(uint32_t n, const std::function<void(uint32_t)>& work) {
  if (n == 0) return;
  if (n == 1) { work(0); return; }
  struct Join { std::mutex m; std::condition_variable cv; uint32_t left; } join;
  join.left = 0;
  const dmi_debug* cur = dbg_ptr();
  cpu_set_t mask;
  const bool have_mask = sched_getaffinity(0, sizeof mask, &mask) == 0;
  int device = -1;
  if (hip_used().load(std::memory_order_relaxed) && hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = -1; }
  auto body = [&](uint32_t t) {
    {
      DebugScope scope(cur);
      if (have_mask) (void)sched_setaffinity(0, sizeof mask, &mask);
      if (device >= 0) (void)hipSetDevice(device);
      work(t);
    }
    std::lock_guard<std::mutex> lock(join.m);
    if (--join.left == 0) join.cv.notify_all();
  };
  struct Wait { Join& j; ~Wait() { std::unique_lock<std::mutex> lock(j.m); j.cv.wait(lock, [&] { return j.left == 0; }); } } wait{join};   // (also when submit throws: the tasks already out hold references to this frame)
  for (uint32_t t = 1; t < n; ++t) {
    { std::lock_guard<std::mutex> lock(join.m); ++join.left; }
    try { pool_submit([&body, t] { body(t); }); }
    catch (...) { { std::lock_guard<std::mutex> lock(join.m); --join.left; } throw; }
  }
  { DebugScope scope(cur); work(0); }
}
}  // namespace dmi
