// dmi_host.hpp — host-side (CPU) stages of libdraco_mi: the serial graph walks that feed the GPU
// attribute pass and the byte-level container around it.  Flat uint32 arrays throughout (the
// reference uses BTreeMap / Vec::remove).
//
// Reference behaviour each piece reproduces (paths relative to draco-oxide/src/):
//   CornerTables::build_universal   core/corner_table/mod.rs:84-118,252-416
//   CornerTables::build_attribute   core/corner_table/attribute_corner_table.rs:16-137
//   run_edgebreaker                 encode/connectivity/edgebreaker.rs:128-530,575-656
//   attribute_sequence              shared/attribute/sequence.rs:48-151
//   FreqTable                       encode/entropy/rans.rs:146-239, shared/entropy/mod.rs:41-64
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <memory>
#include <system_error>
#include <type_traits>
#include <pthread.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <vector>

#include "../../include/draco_mi.h"
#include "dmi_debug.hpp"

namespace dmi {

// Library threads: std::thread's interface on pthread_create with a 1 MiB stack.  A stage of the batch path starts dozens of short-lived threads (16 walkers per
// prepare, the build's packers, the assemblers: ≈ 300 per 1024-file transcode) and glibc keeps at most 40 MB of finished threads' stacks for re-use — five of the
// default 8 MiB ones: the rest is an mmap, an mprotect, first-touch faults and a munmap each (pthread_create + mprotect + munmap: 7 % of the call's CPU samples,
// scripts/experiments/transcode_sigprof.py).  At 1 MiB the cache holds them all.  Nothing in the library recurses deeply (the walks keep explicit stacks, the JSON
// parser stops at 200 levels).
class Thread {
  pthread_t h_{};
  bool joinable_ = false;
  static void* tramp(void* p) { std::unique_ptr<std::function<void()>> f(static_cast<std::function<void()>*>(p)); (*f)(); return nullptr; }
 public:
  Thread() noexcept = default;
  template <class F, class... A, class = typename std::enable_if<!std::is_same<typename std::decay<F>::type, Thread>::value>::type>
  explicit Thread(F&& f, A&&... a) {
    auto* fn = new std::function<void()>(std::bind(std::forward<F>(f), std::forward<A>(a)...));
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, (size_t)1 << 20);
    const int rc = pthread_create(&h_, &at, tramp, fn);
    pthread_attr_destroy(&at);
    if (rc) { delete fn; throw std::system_error(rc, std::generic_category(), "pthread_create"); }
    joinable_ = true;
  }
  Thread(const Thread&) = delete;
  Thread& operator=(const Thread&) = delete;
  Thread(Thread&& o) noexcept : h_(o.h_), joinable_(o.joinable_) { o.joinable_ = false; }
  Thread& operator=(Thread&& o) noexcept { if (joinable_) std::terminate(); h_ = o.h_; joinable_ = o.joinable_; o.joinable_ = false; return *this; }
  ~Thread() { if (joinable_) std::terminate(); }
  bool joinable() const noexcept { return joinable_; }
  void join() { if (!joinable_) throw std::system_error(EINVAL, std::generic_category(), "join"); pthread_join(h_, nullptr); joinable_ = false; }
  void detach() { if (joinable_) { pthread_detach(h_); joinable_ = false; } }
};
// work(t) for t in [0, n): t = 0 on the calling thread, the others on threads of a process-wide pool that outlive the call (they are parked, not joined: a
// stage of the batch path used to create and join its sixteen walkers, its packers, its coders — pthread_create and thread start-up were 7 % of a transcode's
// CPU samples and a millisecond of serial spawning per stage).  A pool thread runs under the caller's switches (DebugScope), CPU mask and current HIP device;
// a task that calls run_threads itself gets further threads (the pool grows to the deepest demand seen, it never waits for a free thread).  Returns when all
// n calls have returned.  `work` must not throw.
void run_threads(uint32_t n, const std::function<void(uint32_t)>& work);
void pool_submit(std::function<void()> fn);   // fn on a parked (or new) pool thread; returns at once
// set once the library has used the device in this process (a stream, staging, device memory): run_threads hands the caller's current device to its workers only then —
// a process that only ever calls the host stages (dmi_mesh_build, dmi_encode_connectivity, the decoders' host halves) never initialises the HIP runtime through it
inline std::atomic<bool>& hip_used() { static std::atomic<bool> f{false}; return f; }

// memcpy for large blocks that go INTO staging memory (read next by a DMA engine, not by this core): non-temporal 16-byte stores — no read-for-ownership of the
// destination, the caches keep what the walks use.  The pack of a 1024-file transcode's 1.09 GB of accessors was 13 % of the call's CPU samples as plain memcpy.
inline void stream_copy(void* dst, const void* src, size_t n) {
#if defined(__SSE2__)
  if (n < ((size_t)128 << 10)) { std::memcpy(dst, src, n); return; }
  uint8_t* d = static_cast<uint8_t*>(dst);
  const uint8_t* s = static_cast<const uint8_t*>(src);
  const size_t head = (16u - (size_t)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u;
  if (head) { std::memcpy(d, s, head); d += head; s += head; n -= head; }
  size_t k = n / 64;
  for (; k; --k, d += 64, s += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 32)), e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(d), a); _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c); _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
  }
  _mm_sfence();
  if (n & 63) std::memcpy(d, s, n & 63);
#else
  std::memcpy(dst, src, n);
#endif
}

constexpr uint32_t kNone = 0xFFFFFFFFu;
int host_fail(int code, const std::string& msg);   // sets dmi_last_error() for the calling thread, returns code


// Host threads the library may use at once in one call: the machine's hardware threads, no more than the CPU quota of the process's
// cgroup (a container on a 256-thread host may be allowed 16 CPUs' worth of time: threads beyond the quota only add throttling —
// measured on the GPU box: 1024-mesh prepare 0.36 s on 16 threads, 0.47–0.68 s on 128), capped by dmi_debug::host_threads (per call: one
// process per GPU on a shared host sets it to its share, cores / world size).
inline unsigned cgroup_cpu_quota() {   // 0 = unlimited / unknown
  static const unsigned quota = [] {
    unsigned q = 0;
    if (std::FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {           // cgroup v2: "<quota|max> <period>"
      char a[32] = {0};
      unsigned long period = 0;
      if (std::fscanf(f, "%31s %lu", a, &period) == 2 && period && std::strcmp(a, "max") != 0) q = (unsigned)((std::strtoul(a, nullptr, 10) + period - 1) / period);
      std::fclose(f);
    } else if (std::FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      long quota_us = -1, period_us = 0;
      if (std::fscanf(g, "%ld", &quota_us) != 1) quota_us = -1;
      std::fclose(g);
      if (std::FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%ld", &period_us) != 1) period_us = 0; std::fclose(h); }
      if (quota_us > 0 && period_us > 0) q = (unsigned)((quota_us + period_us - 1) / period_us);
    }
    return q;
  }();
  return quota;
}
// a cap for the calls the CALLING thread makes (dmi_thread_host_threads): a pipeline that runs several library calls side by side gives each
// stage its share, so that together they stay inside the CPU quota (threads beyond it get the whole cgroup throttled)
inline thread_local unsigned g_thread_host_cap = 0;
inline unsigned process_host_threads() {   // what the whole process may keep busy
  static const unsigned machine = std::thread::hardware_concurrency();   // (once: glibc opens and reads /sys/devices/system/cpu/online on every call — 2.6 % of a seam transcode's CPU samples, from WalkSlots::acquire)
  unsigned hw = machine;
  if (!hw) hw = 4;
  if (const unsigned q = cgroup_cpu_quota()) hw = std::min(hw, std::max(1u, q));   // (a 1024-file transcode with 1.5 × / 2 × / 3 × the quota in threads: 86–97 ms either way)
  if (const uint32_t v = dbg().host_threads) hw = std::min<unsigned>(hw, v);
  return hw;
}
inline unsigned host_threads() {
  unsigned hw = process_host_threads();
  if (g_thread_host_cap) hw = std::min(hw, g_thread_host_cap);
  return hw;
}
// Serial per-mesh walks running at once in the whole PROCESS: two batch prepares side by side (a transcoder's stage k+1 starts its walks while stage
// k's coordinator waits for the device: relabelling, fan rows) share one budget of process_host_threads() instead of oversubscribing the CPU quota.
struct WalkSlots {
  std::mutex m;
  std::condition_variable cv;
  unsigned in_use = 0;
  void acquire() { std::unique_lock<std::mutex> lock(m); const unsigned cap = process_host_threads(); cv.wait(lock, [&] { return in_use < cap; }); ++in_use; }
  void release() { { std::lock_guard<std::mutex> lock(m); --in_use; } cv.notify_one(); }
};
WalkSlots& walk_slots();
// [0, n) in contiguous slices on up to 32 host threads (large, embarrassingly parallel index loops); fn(lo, hi)
template <class Fn>
inline void parallel_for(size_t n, Fn&& fn) {
  const unsigned hw = host_threads();
  const size_t n_threads = n < (1u << 20) ? 1 : std::max<size_t>(1, std::min<size_t>({(size_t)hw, (size_t)32, n >> 18}));
  if (n_threads == 1) { fn((size_t)0, n); return; }
  run_threads((uint32_t)n_threads, [&](uint32_t t) { fn(n * t / n_threads, n * (t + 1) / n_threads); });
}

// ---- recycled host arrays ----
// The connectivity stage of a large mesh allocates ≈ 190 bytes per face of index arrays (tables, walk state, sequences) and hands them
// back at the end: ≈ 1.9 GB of first-touch page faults and ≈ 100 ms of munmap per 10M triangles, paid again by the next mesh.  Vectors of
// at least kPoolMinBytes are therefore taken from / given back to a process-wide pool (capacity kept, contents not): dmi_process_options::host_cache_mb
// caps what the pool holds (default 4096), dmi_release_cached_memory() empties it.
constexpr size_t kPoolMinBytes = (size_t)4 << 20;
size_t host_pool_limit();                       // bytes
std::atomic<size_t>& host_pool_bytes();         // bytes held by the pools of every element type
// Large index arrays are walked with strides of a mesh row (tens of KB): with 4 KiB pages nearly every step of the serial walks is a
// TLB miss.  Freshly reserved pool storage asks for transparent huge pages before its first touch (no-op where THP is off).
void advise_huge_pages(void* p, size_t bytes);
template <class T>
struct VecPool {
  std::mutex m;
  std::vector<std::vector<T>> free_;
  static VecPool& get() { static VecPool p; return p; }
  // an empty vector whose capacity is at least n (recycled when one of a fitting size is held)
  std::vector<T> take(size_t n) {
    if (n * sizeof(T) >= kPoolMinBytes) {
      std::lock_guard<std::mutex> lock(m);
      size_t best = free_.size();
      for (size_t k = 0; k < free_.size(); ++k)
        if (free_[k].capacity() >= n && free_[k].capacity() <= 2 * n + 1024 && (best == free_.size() || free_[k].capacity() < free_[best].capacity())) best = k;
      if (best != free_.size()) {
        std::vector<T> v = std::move(free_[best]);
        free_.erase(free_.begin() + (long)best);
        host_pool_bytes() -= v.capacity() * sizeof(T);
        return v;
      }
    }
    std::vector<T> v;
    if (n * sizeof(T) >= kPoolMinBytes) { v.reserve(n); advise_huge_pages(v.data(), v.capacity() * sizeof(T)); }
    return v;
  }
  void give(std::vector<T>& v) {
    const size_t bytes = v.capacity() * sizeof(T);
    if (bytes >= kPoolMinBytes) {
      v.clear();
      std::lock_guard<std::mutex> lock(m);
      if (host_pool_bytes() + bytes <= host_pool_limit()) { host_pool_bytes() += bytes; free_.push_back(std::move(v)); }
    }
    std::vector<T>().swap(v);
  }
  void drop_all() {
    std::lock_guard<std::mutex> lock(m);
    for (auto& v : free_) host_pool_bytes() -= v.capacity() * sizeof(T);
    free_.clear();
    free_.shrink_to_fit();
  }
};
template <class T> inline void pool_give(std::vector<T>& v) { VecPool<T>::get().give(v); }
// make `v` an empty vector with room for n elements, recycling storage when v's own does not fit
template <class T> inline void pool_fit(std::vector<T>& v, size_t n) {
  if (v.capacity() >= n) { v.clear(); return; }
  pool_give(v);
  v = VecPool<T>::get().take(n);
}
// a local array that returns to the pool at the end of its scope
template <class T>
struct Pooled {
  std::vector<T> v;
  Pooled() = default;
  explicit Pooled(size_t n) : v(VecPool<T>::get().take(n)) {}
  Pooled(size_t n, T fill) : v(VecPool<T>::get().take(n)) { v.assign(n, fill); }
  Pooled(const Pooled&) = delete;
  Pooled& operator=(const Pooled&) = delete;
  ~Pooled() { pool_give(v); }
};
void host_pool_drop_all();

struct ByteSink {
  std::vector<uint8_t> b;
  void u8(uint8_t v) { b.push_back(v); }
  void u16(uint32_t v) { u8((uint8_t)v); u8((uint8_t)(v >> 8)); }
  void u32(uint32_t v) { u16(v & 0xFFFF); u16(v >> 16); }
  void f32(float f) { uint32_t x; std::memcpy(&x, &f, 4); u32(x); }
  void leb128(uint64_t v) { do { uint8_t x = v & 0x7F; v >>= 7; u8(v ? (x | 0x80) : x); } while (v); }   // utils/bit_coder.rs:20-33
  void bytes(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
  void bytes(const std::vector<uint8_t>& v) { b.insert(b.end(), v.begin(), v.end()); }
};

// LSB-first bit packer (core/bit_coder.rs:113-188 with LsbFirst): first value lands in the low bits.
struct BitPackerLsb {
  std::vector<uint8_t>& out;
  uint64_t acc = 0;
  unsigned nbits = 0;
  explicit BitPackerLsb(std::vector<uint8_t>& o) : out(o) {}
  void put(unsigned size, uint32_t value) {
    acc |= (uint64_t)value << nbits;
    nbits += size;
    while (nbits >= 8) { out.push_back((uint8_t)acc); acc >>= 8; nbits -= 8; }
  }
  void flush() { if (nbits) { out.push_back((uint8_t)acc); acc = 0; nbits = 0; } }
};

inline uint32_t corner_next(uint32_t c) { return (c % 3 == 2) ? c - 2 : c + 1; }
inline uint32_t corner_prev(uint32_t c) { return (c % 3 == 0) ? c + 2 : c - 1; }

struct AttTable {
  std::vector<uint32_t> c2v, opp, lmc;   // empty when the attribute has no interior seam: its table IS the universal one (same ids, same order)
  std::vector<uint8_t> seam_edge;
  uint32_t num_vertices = 0;
  bool interior_seams = false;           // some edge with two faces is a seam of this attribute
  int alias_of = -1;                     // index (in CornerTables::att) of the earlier attribute this one was copied from
};

struct CornerTables {
  uint32_t F = 0, V = 0;
  // What every reader uses.  The arrays live in the *_own vectors when the host builds the tables (build_universal), in the caller's
  // face array (c2p, and c2v of a mesh without a position map) or in the pinned read-back of the device-built tables (dmi_prepare.cpp).
  const uint32_t *c2p = nullptr, *c2v = nullptr, *opp = nullptr, *lmc = nullptr;
  std::vector<uint32_t> c2p_own, c2v_own, opp_own, lmc_own;
  bool quad = false;          // the VALUES of `opp` are corner ids 4·face + k (kNone stays kNone; the array itself stays dense): only the device stage writes them so, for
                              // meshes none of whose attributes needs a corner table of its own (host_conn.cpp Enc4: the two serial walks then shift where they divided)
  bool no_boundary = false;   // known: every corner has an opposite (the device pass reports it) — the Edgebreaker skips its boundary labelling scan
  std::vector<AttTable> att;   // one per non-position attribute, in attribute order

  // pos_p2v: point→position value index (NULL = identity).  Returns a dmi_status.  copy_faces = false: c2p views `faces` itself.
  int build_universal(const uint32_t* faces, uint32_t num_faces, const uint32_t* pos_p2v, std::string& err, bool copy_faces = true);
  // att_p2v: point→value index of the attribute (NULL = identity).
  void build_attribute(const uint32_t* att_p2v);
  // thread-safe w.r.t. other attribute tables.  same_as_position: att_p2v is the Position attribute's own map — no edge can be a seam
  void build_attribute_into(AttTable& a, const uint32_t* att_p2v, bool same_as_position = false) const;
  void copy_attribute_into(AttTable& a, const AttTable& from) const;   // an attribute with the map of an earlier one
  void attribute_from_seams(AttTable& a) const;   // decoder side: a.seam_edge given (both corners of every seam edge, every boundary corner)
  void finish_attribute(AttTable& a, const std::vector<uint8_t>& vseam) const;   // a.opp / c2v / lmc from a.seam_edge (vseam: vertices on a seam)
};

struct EdgebreakerResult {
  // corners_of_edgebreaker (edgebreaker.rs:523-529) = reverse(init_face_connectivity_corners) ++ processed_connectivity_corners, kept as its two
  // parts: the attribute sequencers read them where the traversal left them (no 4-bytes-per-face copy on the critical path)
  std::vector<uint32_t> init_rev, processed;
  bool processed_quad = false;          // `processed` holds 4·face + k ids (the traversal ran over a CornerTables::quad table)
  std::vector<uint32_t> seeds;          // the concatenation, only when a caller asks for it (materialize_seeds: dmi_encode_connectivity's public view)
  std::vector<uint8_t> connectivity;    // bytes written by encode_connectivity (without the 11-byte header)
  void materialize_seeds() {
    if (!seeds.empty() || (init_rev.empty() && processed.empty())) return;
    pool_fit(seeds, init_rev.size() + processed.size());
    seeds.assign(init_rev.begin(), init_rev.end());
    if (processed_quad) for (uint32_t c : processed) seeds.push_back(c - (c >> 2));
    else seeds.insert(seeds.end(), processed.begin(), processed.end());
  }
};
// Optional call-outs of run_edgebreaker, so that a large mesh's other serial walks overlap it: `seeds_ready` fires when the traversal
// is over and out.seeds is final (the attribute sequencers only need those), `before_seams` right before the attribute seam flags are
// read (ct.att[] — built by another thread meanwhile — must be complete when it returns).
struct EdgebreakerHooks { std::function<void()> seeds_ready, before_seams; };
int run_edgebreaker(const CornerTables& ct, EdgebreakerResult& out, std::string& err, const EdgebreakerHooks* hooks = nullptr);
extern std::atomic<uint64_t> g_eb_ns[6];   // (trace)

// Attribute sequencer over a flat table view.
struct TableRef {
  uint32_t F, V;
  const uint32_t *c2v, *opp, *lmc;
  bool quad = false;   // see CornerTables::quad
  bool closed = false; // known: every corner has an opposite (CornerTables::no_boundary)
};
// on_boundary (optional): one byte per vertex, != 0 ⇔ the vertex lies on a boundary of `t` (vertex_boundary_flags) — spares the walk two
// dependent loads per vertex
void attribute_sequence(const TableRef& t, const uint32_t* seeds, uint32_t n_seeds, std::vector<uint32_t>& seq, const uint8_t* on_boundary = nullptr);
// the seeds in two parts (first ++ second), e.g. EdgebreakerResult::init_rev ++ processed
// progress (optional): where the walk publishes its output while it runs — `host` once the output array is in place (it does not move afterwards), `written`
// every 2^16 entries (release: the entries below it are readable) — so that another thread can ship the sequence onwards before the walk is over
struct SeqProgress { std::atomic<const uint32_t*> host{nullptr}; std::atomic<uint32_t> written{0}; };
void attribute_sequence(const TableRef& t, const uint32_t* first, uint32_t n_first, const uint32_t* second, uint32_t n_second, std::vector<uint32_t>& seq, const uint8_t* on_boundary = nullptr,
                        bool second_quad = false /* the second part holds 4·face + k ids */, SeqProgress* progress = nullptr);
inline void attribute_sequence(const TableRef& t, const EdgebreakerResult& eb, std::vector<uint32_t>& seq, const uint8_t* on_boundary = nullptr, SeqProgress* progress = nullptr) {
  attribute_sequence(t, eb.init_rev.data(), (uint32_t)eb.init_rev.size(), eb.processed.data(), (uint32_t)eb.processed.size(), seq, on_boundary, eb.processed_quad, progress);
}
void vertex_boundary_flags(const TableRef& t, std::vector<uint8_t>& on_boundary);   // parallel slices for a large table

// rABS bit coder (encode/entropy/rans.rs:71-128), host version for the small connectivity streams.
struct RabsHost {
  uint32_t state = 4096, p0;
  std::vector<uint8_t> out;
  explicit RabsHost(uint32_t zero_prob) : p0(zero_prob) {}
  void put(unsigned bit) {
    const uint32_t f1 = 256 - p0, f = bit ? f1 : p0;
    if (state >= ((16u * f) << 8)) { out.push_back((uint8_t)state); state >>= 8; }
    state = ((state / f) << 8) + state % f + (bit ? 0 : f1);
  }
  bool finish();   // appends the tagged state; false on StateTooLarge
};
uint8_t zero_probability(uint64_t count_zero, float denominator);
// the rABS stream of n flags fed first to last, on the multiply-high host coder (host_chains.cpp); false on StateTooLarge
bool host_rabs_constant(uint8_t zero_prob, uint32_t bit, uint64_t n, std::vector<uint8_t>& bytes);   // n copies of one bit, without the n steps (host_chains.cpp)
bool host_rabs_bytes(uint8_t zero_prob, const uint8_t* fed, uint64_t n, std::vector<uint8_t>& bytes);   // the f32 "(c0/len)*256+0.5 → clamp(1,255)" idiom

// Normalised frequency table + its serialisation.
struct FreqTable {
  uint32_t precision = 12;       // rANS precision bits
  uint32_t bit_length = 1;       // value written before the table (symbol_coding.rs:118-119)
  std::vector<uint32_t> freq;    // normalised, sums to 2^precision
  std::vector<uint32_t> cum;
  std::vector<uint8_t> header;   // u8(1) method, u8(bit_length), leb128(num_symbols), table bytes
  // hist[s] = count of symbol s; returns dmi_status
  int build(const uint32_t* hist, size_t bins, std::string& err);
};

bool append_tagged_state(uint32_t state_minus_base, std::vector<uint8_t>& out);   // rans.rs:48-68

// fn(i) for i in [0, n) on at most `max_threads` host threads (a fixed pool, not a thread per item: n may come out of an untrusted file).
// An exception in a worker — std::bad_alloc on a size a damaged file asked for, std::system_error when the process is out of threads — is
// caught there (a throw out of a dmi::Thread body is std::terminate) and reported: returns 0, 1 = out of memory, 2 = another exception.
template <class Fn>
inline int guarded_pool(size_t n, unsigned max_threads, Fn&& fn) {
  std::atomic<size_t> next{0};
  std::atomic<int> status{0};
  const dmi_debug* cur = dbg_ptr();
  auto work = [&] {
    DebugScope scope(cur);
    try {
      for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
    } catch (const std::bad_alloc&) { status.store(1); next.store(n); }
    catch (...) { int z = 0; status.compare_exchange_strong(z, 2); next.store(n); }
  };
  const size_t nt = std::max<size_t>(1, std::min<size_t>(n, max_threads ? max_threads : 1));
  try {
    run_threads((uint32_t)nt, [&](uint32_t) { work(); });
  } catch (...) { int z = 0; status.compare_exchange_strong(z, 2); }   // (the process is out of threads)
  return status.load();
}

// What dmi_built_mesh::owner points to: the host builder's arrays (host_mesh.cpp) or a member of a device-built group (dmi_build.cpp).
struct BuiltBase {
  virtual ~BuiltBase() = default;
};

}  // namespace dmi
